// Fused bilinear CPC scores + InfoNCE + accuracy (one workgroup per window b).
// Reference: FksModule.forward (vqcpc_helper.py:86-98), negatives reshuffle (vqcpc_encoder_trainer.py:245-263),
// nce_loss (vqcpc_helper.py:5-29), score matrix (vqcpc_encoder_trainer.py:269).
#include "common.h"

namespace vq {

constexpr int kNceThreads = 256;

// LDS layout: Wc [K][zdim] | f [K][N+1] (negatives first, positive last, as torch.cat([negatives, positive]))
__global__ __launch_bounds__(kNceThreads) void nce_fwd_kernel(const float* __restrict__ c, const float* __restrict__ W,
                                                              const float* __restrict__ z_pos,
                                                              const float* __restrict__ z_neg, int B, int K, int N,
                                                              int zdim, int cdim, float* __restrict__ f_pos,
                                                              float* __restrict__ f_neg, float* __restrict__ loss_b,
                                                              float* __restrict__ hits) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wc = lds;                   // [K][zdim]
    float* f = wc + K * zdim;          // [K][N+1]
    float* cs = f + K * (N + 1);       // [cdim]
    float* lk = cs + cdim;             // [K] per-k loss terms
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < cdim; i += kNceThreads) cs[i] = c[(int64_t)b * cdim + i];
    __syncthreads();
    // Wc[k][z] = sum_c W[z][c][k] * c[b][c]        (torch.matmul(c_t, W).permute(1, 2, 0))
    for (int o = threadIdx.x; o < K * zdim; o += kNceThreads) {
        const int k = o % K, zz = o / K;
        float acc = 0.0f;
        for (int cc = 0; cc < cdim; ++cc) acc += cs[cc] * W[((int64_t)zz * cdim + cc) * K + k];
        wc[k * zdim + zz] = acc;
    }
    __syncthreads();
    // scores: slot n < N negative n, slot N positive
    for (int o = threadIdx.x; o < K * (N + 1); o += kNceThreads) {
        const int k = o / (N + 1), n = o % (N + 1);
        const float* zv = (n == N) ? z_pos + ((int64_t)b * K + k) * zdim
                                   : z_neg + (((int64_t)b * N + n) * K + k) * zdim;
        float acc = 0.0f;
        for (int zz = 0; zz < zdim; ++zz) acc += wc[k * zdim + zz] * zv[zz];
        f[k * (N + 1) + n] = acc;
        if (n == N) f_pos[(int64_t)b * K + k] = acc;
        else f_neg[((int64_t)b * K + k) * N + n] = acc;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        const float* fk = f + k * (N + 1);
        float m = fk[0];
        for (int n = 1; n <= N; ++n) m = fmaxf(m, fk[n]);
        float s = 0.0f;
        for (int n = 0; n <= N; ++n) s += expf(fk[n] - m);
        const float lse = m + logf(s);
        lk[k] = fk[N] - lse;
        float mneg = N > 0 ? fk[0] : -INFINITY;
        for (int n = 1; n < N; ++n) mneg = fmaxf(mneg, fk[n]);
        hits[(int64_t)b * K + k] = fk[N] > mneg ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s += lk[k];
        loss_b[b] = -s;
    }
}

// backward, stage 1 (per window): softmax from the saved scores, d_z_pos, d_z_neg, d_c, and dWc[b][k][z] -> workspace
__global__ __launch_bounds__(kNceThreads) void nce_bwd_kernel(const float* __restrict__ c, const float* __restrict__ W,
                                                              const float* __restrict__ z_pos,
                                                              const float* __restrict__ z_neg,
                                                              const float* __restrict__ f_pos,
                                                              const float* __restrict__ f_neg, const float* __restrict__ g,
                                                              int B, int K, int N, int zdim, int cdim,
                                                              float* __restrict__ d_c, float* __restrict__ d_z_pos,
                                                              float* __restrict__ d_z_neg, float* __restrict__ dwc_ws) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wc = lds;                   // [K][zdim]
    float* df = wc + K * zdim;         // [K][N+1]  d loss_b / d score
    float* dwc = df + K * (N + 1);     // [K][zdim]
    float* cs = dwc + K * zdim;        // [cdim]
    const int b = blockIdx.x;
    const float gb = g[b];
    for (int i = threadIdx.x; i < cdim; i += kNceThreads) cs[i] = c[(int64_t)b * cdim + i];
    __syncthreads();
    for (int o = threadIdx.x; o < K * zdim; o += kNceThreads) {
        const int k = o % K, zz = o / K;
        float acc = 0.0f;
        for (int cc = 0; cc < cdim; ++cc) acc += cs[cc] * W[((int64_t)zz * cdim + cc) * K + k];
        wc[k * zdim + zz] = acc;
    }
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        const float fp = f_pos[(int64_t)b * K + k];
        const float* fn = f_neg + ((int64_t)b * K + k) * N;
        float m = fp;
        for (int n = 0; n < N; ++n) m = fmaxf(m, fn[n]);
        float s = 0.0f;
        for (int n = 0; n < N; ++n) s += expf(fn[n] - m);
        s += expf(fp - m);
        const float inv = 1.0f / s;
        // loss_b = -sum_k (pos - lse):  d/dpos = -(1 - p_pos),  d/dneg_n = p_n
        for (int n = 0; n < N; ++n) df[k * (N + 1) + n] = gb * expf(fn[n] - m) * inv;
        df[k * (N + 1) + N] = -gb * (1.0f - expf(fp - m) * inv);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < K * zdim; o += kNceThreads) {
        const int k = o / zdim, zz = o % zdim;
        const float w = wc[k * zdim + zz];
        const float dp = df[k * (N + 1) + N];
        const float zp = z_pos[((int64_t)b * K + k) * zdim + zz];
        d_z_pos[((int64_t)b * K + k) * zdim + zz] = dp * w;
        float acc = dp * zp;
        for (int n = 0; n < N; ++n) {
            const float dn = df[k * (N + 1) + n];
            const int64_t off = (((int64_t)b * N + n) * K + k) * zdim + zz;
            acc += dn * z_neg[off];
            d_z_neg[off] = dn * w;
        }
        dwc[k * zdim + zz] = acc;
        dwc_ws[((int64_t)b * K + k) * zdim + zz] = acc;
    }
    __syncthreads();
    // d_c[b][cc] = sum_{k,z} dWc[k][z] * W[z][cc][k]
    for (int cc = threadIdx.x; cc < cdim; cc += kNceThreads) {
        float acc = 0.0f;
        for (int zz = 0; zz < zdim; ++zz)
            for (int k = 0; k < K; ++k) acc += dwc[k * zdim + zz] * W[((int64_t)zz * cdim + cc) * K + k];
        d_c[(int64_t)b * cdim + cc] = acc;
    }
}

// backward, stage 2: d_W[z][cc][k] = sum_b dWc[b][k][z] * c[b][cc] -- for every k a (zdim x B) . (B x cdim) contraction over
// the batch, on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: an exact fmaf chain per output element).  One workgroup per
// (k, 32 x 32 output tile); its 4 waves take the batch quarters [w B/4, (w+1) B/4) -- every operand of a wave is requested
// before its first MFMA -- and their partial tiles are summed through LDS in wave order: deterministic.  The scalar loop this
// replaces walked the batch with one dependent load pair per element (69 us at B = 256; this: ~8 us).
//   MFMA layouts (l = lane, i = l & 31, h = l >> 5):  A[row i][k h]   B[k h][col i]   D[row (r & 3) + 8 (r >> 2) + 4 h][col i]
typedef float nce_f16 __attribute__((ext_vector_type(16)));
constexpr int kDwWaves = 4, kDwMaxPairs = 32;     // batch pairs per wave and pass held in registers (one pass up to B = 256)

__global__ __launch_bounds__(kDwWaves * 64) void nce_dw_mfma_kernel(const float* __restrict__ dwc_ws,
                                                                    const float* __restrict__ c, int B, int K, int zdim,
                                                                    int cdim, float* __restrict__ d_W) {
    __shared__ float part[kDwWaves][32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int ctiles = (cdim + 31) / 32, ztiles = (zdim + 31) / 32;
    const int k = blockIdx.x / (ztiles * ctiles), zt = (blockIdx.x / ctiles) % ztiles, ct = blockIdx.x % ctiles;
    const int zz = zt * 32 + i, cc = ct * 32 + i;
    const bool zok = zz < zdim, cok = cc < cdim;
    nce_f16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int per = (B + kDwWaves - 1) / kDwWaves;                 // batch rows of this wave: [b_lo, b_hi)
    const int b_lo = wave * per, b_hi = min(B, b_lo + per);
    for (int b0 = b_lo; b0 < b_hi; b0 += 2 * kDwMaxPairs) {
        float av[kDwMaxPairs], bv[kDwMaxPairs];
#pragma unroll
        for (int j = 0; j < kDwMaxPairs; ++j) {
            const int b = b0 + 2 * j + h;
            const bool ok = b < b_hi;
            av[j] = (ok && zok) ? dwc_ws[((int64_t)b * K + k) * zdim + zz] : 0.0f;
            bv[j] = (ok && cok) ? c[(int64_t)b * cdim + cc] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < kDwMaxPairs; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + i] = acc[r];
    __syncthreads();
    for (int o = threadIdx.x; o < 32 * 32; o += kDwWaves * 64) {
        const int row = o >> 5, col = o & 31;
        const int z2 = zt * 32 + row, c2 = ct * 32 + col;
        if (z2 < zdim && c2 < cdim) {
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < kDwWaves; ++w) tot += part[w][row * 33 + col];
            d_W[((int64_t)z2 * cdim + c2) * K + k] = tot;
        }
    }
}

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_nce_fwd(const float* c, const float* W, const float* z_pos, const float* z_neg, int B, int K, int N, int zdim,
                  int cdim, float* f_pos, float* f_neg, float* loss_b, float* hits, void* stream) {
    VQ_REQUIRE(c && W && z_pos && z_neg && f_pos && f_neg && loss_b && hits, "nce_fwd: null pointer");
    VQ_REQUIRE(B >= 1 && K >= 1 && K <= kNceThreads && N >= 1 && zdim >= 1 && cdim >= 1, "nce_fwd: bad shape");
    const size_t lds = ((size_t)K * zdim + (size_t)K * (N + 1) + cdim + K) * sizeof(float);
    VQ_REQUIRE(lds <= 64 * 1024, "nce_fwd: problem does not fit the LDS");
    hipLaunchKernelGGL(nce_fwd_kernel, dim3(B), dim3(kNceThreads), lds, (hipStream_t)stream, c, W, z_pos, z_neg, B, K, N,
                       zdim, cdim, f_pos, f_neg, loss_b, hits);
    VQ_CHECK_LAUNCH("nce_fwd");
    return VQCPC_OK;
}

int64_t vqcpc_nce_bwd_workspace(int B, int K, int N, int zdim, int cdim) {
    (void)N;
    (void)cdim;
    return (int64_t)B * K * zdim * (int64_t)sizeof(float);
}

int vqcpc_nce_bwd(const float* c, const float* W, const float* z_pos, const float* z_neg, const float* f_pos,
                  const float* f_neg, const float* g, int B, int K, int N, int zdim, int cdim, float* d_c, float* d_W,
                  float* d_z_pos, float* d_z_neg, void* workspace, int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(c && W && z_pos && z_neg && f_pos && f_neg && g && d_c && d_W && d_z_pos && d_z_neg && workspace,
               "nce_bwd: null pointer");
    VQ_REQUIRE(B >= 1 && K >= 1 && K <= kNceThreads && N >= 1 && zdim >= 1 && cdim >= 1, "nce_bwd: bad shape");
    if (workspace_bytes < vqcpc_nce_bwd_workspace(B, K, N, zdim, cdim)) {
        set_error("nce_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const size_t lds = (2 * (size_t)K * zdim + (size_t)K * (N + 1) + cdim) * sizeof(float);
    VQ_REQUIRE(lds <= 64 * 1024, "nce_bwd: problem does not fit the LDS");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nce_bwd_kernel, dim3(B), dim3(kNceThreads), lds, s, c, W, z_pos, z_neg, f_pos, f_neg, g, B, K, N,
                       zdim, cdim, d_c, d_z_pos, d_z_neg, (float*)workspace);
    VQ_CHECK_LAUNCH("nce_bwd");
    const int tiles = ceil_div(zdim, 32) * ceil_div(cdim, 32);
    hipLaunchKernelGGL(nce_dw_mfma_kernel, dim3(K * tiles), dim3(kDwWaves * 64), 0, s, (const float*)workspace, c, B, K, zdim,
                       cdim, d_W);
    VQ_CHECK_LAUNCH("nce_dw");
    return VQCPC_OK;
}

}  // extern "C"
