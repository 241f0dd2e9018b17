// Small kernels of the student (distilled VQ-VAE) step, SURVEY.md section 8 row A23:
//   * softmax cross-entropy rows with hard targets (utils.categorical_crossentropy, utils.py:24-49) or soft targets
//     (utils.distilled_categorical_crossentropy, utils.py:131-159): loss and d loss / d logits in one pass;
//   * AuxiliaryDecoderRelative.upscale (auxiliary_decoder_relative.py:116-130): repeat_interleave + learned offsets.
// All of them are bandwidth-trivial (rows = batch x masked events, a few KB); one wavefront per row, no atomics.
#include "common.h"

namespace vq {

// loss[r] = -sum_v t[v] * log_softmax(x[r])[v];  grad[r][v] = softmax(x[r])[v] - t[v]
//   hard: t = onehot(target[r])        soft: t = softmax(tl[r])
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const int64_t* __restrict__ target,
                                                         const float* __restrict__ tl, int64_t ldt,
                                                         float* __restrict__ loss, float* __restrict__ grad, int64_t R,
                                                         int V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= R) return;
    const float* xr = x + r * ldx;
    float m = -3.0e38f;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, xr[v]);
    m = wave_max(m);
    float s = 0.0f;
    for (int v = lane; v < V; v += 64) s += expf(xr[v] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    float acc = 0.0f;
    if (tl) {
        const float* tr = tl + r * ldt;
        float tm = -3.0e38f;
        for (int v = lane; v < V; v += 64) tm = fmaxf(tm, tr[v]);
        tm = wave_max(tm);
        float ts = 0.0f;
        for (int v = lane; v < V; v += 64) ts += expf(tr[v] - tm);
        ts = wave_sum(ts);
        const float inv = 1.0f / ts;
        for (int v = lane; v < V; v += 64) {
            const float t = expf(tr[v] - tm) * inv;
            acc -= t * (xr[v] - lse);
            grad[r * V + v] = expf(xr[v] - lse) - t;
        }
    } else {
        const int tgt = (int)target[r];
        for (int v = lane; v < V; v += 64) {
            const float t = v == tgt ? 1.0f : 0.0f;
            acc -= t * (xr[v] - lse);
            grad[r * V + v] = expf(xr[v] - lse) - t;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) loss[r] = acc;
}

// out[r][v] = g[r] * in[r][v]
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ in, const float* __restrict__ g,
                                                         float* __restrict__ out, int64_t R, int V) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < R * V) out[e] = in[e] * g[e / V];
}

// out[(r*f + u)][c] = x[r][c] + emb[u][c]
__global__ __launch_bounds__(256) void upscale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ emb,
                                                          float* __restrict__ out, int64_t rows, int f, int d4) {
    const int64_t total = rows * f * d4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % d4);
        const int64_t ro = e / d4;
        const int u = (int)(ro % f);
        const int64_t r = ro / f;
        const float4 a = reinterpret_cast<const float4*>(x)[r * d4 + c];
        const float4 b = reinterpret_cast<const float4*>(emb)[u * d4 + c];
        reinterpret_cast<float4*>(out)[e] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// dx[r][c] = sum_u g[r*f+u][c];  ws[chunk][u][c] = sum over the chunk's rows of g[r*f+u][c]
// input rows per workgroup: a thread walks them one after the other (dependent on nothing, but one load round trip each), so
// few rows per workgroup when there are few rows at all -- the student step's 192 rows took 70 us in 6 workgroups of 32 rows
static int up_rows_per_wg(int64_t rows) { return (int)std::min<int64_t>(32, std::max<int64_t>(1, rows / 1024)); }
constexpr int kUpMaxF = 8;

__global__ __launch_bounds__(256) void upscale_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx,
                                                          float* __restrict__ ws, int64_t rows, int f, int d, int rows_per_wg) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, rows);
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float acc[kUpMaxF];
#pragma unroll
        for (int u = 0; u < kUpMaxF; ++u) acc[u] = 0.0f;
        for (int64_t r = r0; r < r1; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int u = 0; u < kUpMaxF; ++u) {
                if (u < f) {
                    const float v = g[(r * f + u) * d + c];
                    acc[u] += v;
                    s += v;
                }
            }
            dx[r * d + c] = s;
        }
#pragma unroll
        for (int u = 0; u < kUpMaxF; ++u)
            if (u < f) ws[((int64_t)blockIdx.x * f + u) * d + c] = acc[u];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 'same_sequence' negatives (bach_cpc_dataloader.py:110-181 _build_negatives_sameSeq):
// for target block k of `second`, the negatives are every block of `first` followed by the blocks of `second` except k.
//   first (B, Ka*ev, V), second (B, Kb*ev, V) int64 tokens (ticks x voices)  ->  out (B, Ka+Kb-1, Kb, ev, V)
__global__ __launch_bounds__(256) void same_seq_negatives_kernel(const int64_t* __restrict__ first,
                                                                 const int64_t* __restrict__ second,
                                                                 int64_t* __restrict__ out, int64_t B, int Ka, int Kb,
                                                                 int blk) {
    const int N = Ka + Kb - 1;
    const int64_t total = B * N * Kb * blk;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e % blk);                       // token inside the block (tick-major, voice fastest)
        int64_t r = e / blk;
        const int k = (int)(r % Kb);
        r /= Kb;
        const int n = (int)(r % N);
        const int64_t b = r / N;
        int64_t v;
        if (n < Ka) {
            v = first[(b * Ka + n) * blk + t];
        } else {
            const int j = n - Ka;
            v = second[(b * Kb + (j < k ? j : j + 1)) * blk + t];
        }
        out[e] = v;
    }
}

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_softmax_ce(const float* logits, int64_t ld, const int64_t* target, const float* target_logits, int64_t ldt,
                     float* loss, float* grad, int64_t R, int V, void* stream) {
    if (R == 0) return VQCPC_OK;
    VQ_REQUIRE(logits && loss && grad && ((target != nullptr) != (target_logits != nullptr)),
               "softmax_ce: need logits, loss, grad and exactly one of target / target_logits");
    VQ_REQUIRE(R >= 0 && V >= 1 && ld >= V && (!target_logits || ldt >= V), "softmax_ce: bad shape");
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream, logits, ld,
                       target, target_logits, ldt, loss, grad, R, V);
    VQ_CHECK_LAUNCH("softmax_ce");
    return VQCPC_OK;
}

int vqcpc_same_sequence_negatives(const int64_t* first, const int64_t* second, int64_t* out, int64_t B, int blocks_first,
                                  int blocks_second, int tokens_per_block, void* stream) {
    if (B == 0) return VQCPC_OK;
    VQ_REQUIRE(first && second && out && B >= 0 && blocks_first >= 0 && blocks_second >= 1 && tokens_per_block >= 1 &&
                   blocks_first + blocks_second >= 2,
               "same_sequence_negatives: bad arguments");
    const int64_t total = B * (blocks_first + blocks_second - 1) * blocks_second * tokens_per_block;
    const int grid = (int)std::min<int64_t>(ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(same_seq_negatives_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, first, second, out, B,
                       blocks_first, blocks_second, tokens_per_block);
    VQ_CHECK_LAUNCH("same_sequence_negatives");
    return VQCPC_OK;
}

int vqcpc_scale_rows(const float* in, const float* g, float* out, int64_t R, int V, void* stream) {
    if (R == 0) return VQCPC_OK;
    VQ_REQUIRE(in && g && out && R >= 0 && V >= 1, "scale_rows: bad arguments");
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)ceil_div(R * V, 256)), dim3(256), 0, (hipStream_t)stream, in, g,
                       out, R, V);
    VQ_CHECK_LAUNCH("scale_rows");
    return VQCPC_OK;
}

int vqcpc_upscale_fwd(const float* x, const float* emb, float* out, int64_t rows, int f, int d, void* stream) {
    if (rows == 0) return VQCPC_OK;
    VQ_REQUIRE(x && emb && out && rows >= 0 && f >= 1 && d >= 4 && d % 4 == 0, "upscale_fwd: bad arguments");
    VQ_REQUIRE(aligned16(x) && aligned16(emb) && aligned16(out), "upscale_fwd: buffers must be 16-byte aligned");
    const int64_t total = rows * f * (d / 4);
    const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(upscale_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, emb, out, rows, f, d / 4);
    VQ_CHECK_LAUNCH("upscale_fwd");
    return VQCPC_OK;
}

int64_t vqcpc_upscale_bwd_workspace(int64_t rows, int f, int d) {
    return ceil_div(std::max<int64_t>(rows, 1), up_rows_per_wg(std::max<int64_t>(rows, 1))) * f * d * (int64_t)sizeof(float);
}

int vqcpc_upscale_bwd(const float* g, float* dx, float* d_emb, int64_t rows, int f, int d, void* workspace,
                      int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(g && dx && d_emb && workspace && rows >= 1 && f >= 1 && f <= kUpMaxF && d >= 1,
               "upscale_bwd: bad arguments (upscale factor <= %d)", kUpMaxF);
    if (workspace_bytes < vqcpc_upscale_bwd_workspace(rows, f, d)) {
        set_error("upscale_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const int rpw = up_rows_per_wg(rows);
    const int chunks = (int)ceil_div(rows, rpw);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(upscale_bwd_kernel, dim3(chunks), dim3(256), 0, s, g, dx, (float*)workspace, rows, f, d, rpw);
    VQ_CHECK_LAUNCH("upscale_bwd");
    return launch_reduce_splits((const float*)workspace, (int64_t)f * d, chunks, d_emb, (int64_t)f * d, 0, s);
}

}  // extern "C"
