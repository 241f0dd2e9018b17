// GRU cell pointwise kernels for the CPC context network (CModule, vqcpc_helper.py:54-76: nn.GRU, gate order r | z | n,
// h0 = 0, dropout on the outputs of every layer but the last).  The two matrix products of a step,
// gi = x W_ih^T + b_ih (all steps at once) and gh = h_{t-1} W_hh^T + b_hh, are vqcpc_gemm_nt launches; these kernels do
// the gate arithmetic and its backward, so the whole context network runs on this library (no MIOpen RNN).
//   r = sigmoid(gi_r + gh_r)   u = sigmoid(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   h = (1 - u) n + u h_prev
// One float4 of hidden units per lane; bandwidth-trivial (B x H = 256 x 512 per step at C1).
#include "common.h"

namespace vq {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev, float* __restrict__ h_out,
                                                           float* __restrict__ y_out, int64_t B, int H, uint32_t thr,
                                                           float inv_keep, uint64_t seed, uint64_t idx_base) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * H) return;
    const int64_t b = e / H;
    const int c = (int)(e - b * H);
    const float* gib = gi + b * 3 * H;
    const float* ghb = gh + b * 3 * H;
    const float r = sigmoidf_(gib[c] + ghb[c]);
    const float u = sigmoidf_(gib[H + c] + ghb[H + c]);
    const float n = tanhf(gib[2 * H + c] + r * ghb[2 * H + c]);
    const float hp = h_prev ? h_prev[e] : 0.0f;
    const float h = (1.0f - u) * n + u * hp;
    h_out[e] = h;
    if (y_out) y_out[e] = h * drop_scale(seed, idx_base + (uint64_t)e, thr, inv_keep);
}

// dh = d_y * mask + d_h ;  outputs d_gi, d_gh [B][3H] and the direct part of d h_prev (= dh * u)
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev, const float* __restrict__ d_y,
                                                           const float* __restrict__ d_h, float* __restrict__ d_gi,
                                                           float* __restrict__ d_gh, float* __restrict__ d_hprev, int64_t B,
                                                           int H, uint32_t thr, float inv_keep, uint64_t seed,
                                                           uint64_t idx_base) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * H) return;
    const int64_t b = e / H;
    const int c = (int)(e - b * H);
    const int64_t o = b * 3 * H + c;
    const float ghn = gh[o + 2 * H];
    const float r = sigmoidf_(gi[o] + gh[o]);
    const float u = sigmoidf_(gi[o + H] + gh[o + H]);
    const float n = tanhf(gi[o + 2 * H] + r * ghn);
    const float hp = h_prev ? h_prev[e] : 0.0f;
    float dh = d_h ? d_h[e] : 0.0f;
    if (d_y) dh += d_y[e] * drop_scale(seed, idx_base + (uint64_t)e, thr, inv_keep);
    const float da_n = dh * (1.0f - u) * (1.0f - n * n);
    const float da_u = dh * (hp - n) * u * (1.0f - u);
    const float da_r = da_n * ghn * r * (1.0f - r);
    d_gi[o] = da_r;
    d_gi[o + H] = da_u;
    d_gi[o + 2 * H] = da_n;
    d_gh[o] = da_r;
    d_gh[o + H] = da_u;
    d_gh[o + 2 * H] = da_n * r;
    d_hprev[e] = dh * u;
}

// =====================================================================================================================
// One launch per time step: recurrent product + gate arithmetic fused.  The step chain of the context network is 2 layers x
// 8 dependent steps forward and backward; as "GEMM launch + gate launch (+ split-K reduction)" per step it was 78 launches
// of 5-17 us, ~0.7 ms of pure launch-to-launch latency at C1.  Here a workgroup owns a 32 (batch rows) x 32 (hidden units)
// tile of the step: its NW waves split the contraction (each lane half takes a contiguous range: the MFMA k index is a
// summation index, so operands are float4 row loads straight from global / L2 -- the form of gemm_nt_skinny_kernel), meet
// in LDS in wave order (deterministic), and the same workgroup applies the gate arithmetic to its tile.
//   fp32 matrix cores (v_mfma_f32_32x32x2_f32 = exact fmaf chains), layouts: A[row l & 31][k l >> 5], B[k l >> 5][col l & 31],
//   D[row (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col l & 31].
// =====================================================================================================================
typedef float gru_f16 __attribute__((ext_vector_type(16)));
constexpr int kGruNW = 8;

#define GRU_MFMA4(ACC, AV, BV)                                                     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.x, BV.x, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.y, BV.y, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.z, BV.z, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.w, BV.w, ACC, 0, 0, 0);

// Epilogue ownership: wave w finishes accumulator registers r = 2w, 2w + 1 of the tile (rows (r & 3) + 8 (r >> 2) + 4 g);
// its element-wise operands are requested BEFORE the product loop, so their latency hides under the MFMAs (an epilogue that
// loads gi / h_prev per register after the reduction is a chain of 16 dependent global round trips: 2x slower than the
// unfused launches).
// forward step: gh = h_prev W_hh^T + b_hh (three 32 x 32 tiles: gates r | z | n of the tile's hidden units), then the cell.
__global__ __launch_bounds__(kGruNW * 64) void gru_step_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                                   const float* __restrict__ b_hh,
                                                                   const float* __restrict__ h_prev, float* __restrict__ gh,
                                                                   float* __restrict__ h_out, float* __restrict__ y_out,
                                                                   int B, int H, uint32_t thr, float inv_keep, uint64_t seed,
                                                                   uint64_t idx_base) {
    extern __shared__ float gru_red[];                          // [kGruNW][3][16][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, l31 = lane & 31;
    const int i = blockIdx.y * 32 + l31, j = blockIdx.x * 32 + l31;
    // this wave's two epilogue elements per lane
    float e_gi[2][3], e_hp[2];
    int e_row[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 2 * wave + q;
        e_row[q] = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const int rr = min(e_row[q], B - 1);
        const int64_t o = (int64_t)rr * 3 * H + j;
        e_gi[q][0] = gi[o];
        e_gi[q][1] = gi[o + H];
        e_gi[q][2] = gi[o + 2 * H];
        e_hp[q] = h_prev ? h_prev[(int64_t)rr * H + j] : 0.0f;
    }
    const float br = b_hh[j], bu = b_hh[H + j], bn = b_hh[2 * H + j];
    gru_f16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    if (h_prev != nullptr) {
        const int len = H / (2 * kGruNW);                       // floats per lane half (multiple of 4)
        const int k0 = (wave * 2 + g) * len;
        const float* ap = h_prev + (int64_t)min(i, B - 1) * H + k0;
        const float* bp = w_hh + (int64_t)j * H + k0;
        int kk = 0;
        for (; kk + 16 <= len; kk += 16) {                      // 16 x 16-byte loads in flight per lane
            float4 a[4], b[3][4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                a[v] = *reinterpret_cast<const float4*>(ap + kk + 4 * v);
#pragma unroll
                for (int q = 0; q < 3; ++q) b[q][v] = *reinterpret_cast<const float4*>(bp + (int64_t)q * H * H + kk + 4 * v);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                GRU_MFMA4(acc[0], a[v], b[0][v])
                GRU_MFMA4(acc[1], a[v], b[1][v])
                GRU_MFMA4(acc[2], a[v], b[2][v])
            }
        }
        for (; kk < len; kk += 4) {                             // len % 16 != 0 (H = 64, 128, 192, ...)
            const float4 a = *reinterpret_cast<const float4*>(ap + kk);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4 b = *reinterpret_cast<const float4*>(bp + (int64_t)q * H * H + kk);
                GRU_MFMA4(acc[q], a, b)
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) gru_red[((wave * 3 + q) * 16 + r) * 64 + lane] = acc[q][r];
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 2 * wave + q;
        float v3[3] = {0.0f, 0.0f, 0.0f};
        if (h_prev != nullptr) {
#pragma unroll
            for (int w = 0; w < kGruNW; ++w)                     // fixed wave order: deterministic
#pragma unroll
                for (int t = 0; t < 3; ++t) v3[t] += gru_red[((w * 3 + t) * 16 + r) * 64 + lane];
        }
        if (e_row[q] < B) {
            const float vr = v3[0] + br, vu = v3[1] + bu, vn = v3[2] + bn;
            const int64_t o = (int64_t)e_row[q] * 3 * H + j, e = (int64_t)e_row[q] * H + j;
            gh[o] = vr; gh[o + H] = vu; gh[o + 2 * H] = vn;
            const float rr = sigmoidf_(e_gi[q][0] + vr);
            const float uu = sigmoidf_(e_gi[q][1] + vu);
            const float nn = tanhf(e_gi[q][2] + rr * vn);
            const float hv = (1.0f - uu) * nn + uu * e_hp[q];
            h_out[e] = hv;
            if (y_out) y_out[e] = hv * drop_scale(seed, idx_base + (uint64_t)e, thr, inv_keep);
        }
    }
}

// backward step t -> t-1:  dh_{t-1} = dgh_t W_hh + dhp (direct term dh_t * u_t, written by the previous launch), then the cell
// backward of step t-1 on the same tile: d_gi / d_gh of step t-1 and the new direct term dhp = dh_{t-1} * u_{t-1} (in place).
__global__ __launch_bounds__(kGruNW * 64) void gru_step_bwd_kernel(const float* __restrict__ dgh_next,
                                                                   const float* __restrict__ whh_t, float* __restrict__ dhp,
                                                                   const float* __restrict__ gi, const float* __restrict__ gh,
                                                                   const float* __restrict__ h_prev,
                                                                   const float* __restrict__ d_y, float* __restrict__ d_gi,
                                                                   float* __restrict__ d_gh, int B, int H, uint32_t thr,
                                                                   float inv_keep, uint64_t seed, uint64_t idx_base) {
    extern __shared__ float gru_red[];                          // [kGruNW][16][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, l31 = lane & 31;
    const int i = blockIdx.y * 32 + l31, j = blockIdx.x * 32 + l31;
    const int K3 = 3 * H;
    float e_gi[2][3], e_gh[2][3], e_hp[2], e_dh[2];
    int e_row[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 2 * wave + q;
        e_row[q] = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const int rr = min(e_row[q], B - 1);
        const int64_t o = (int64_t)rr * 3 * H + j, e = (int64_t)rr * H + j;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            e_gi[q][t] = gi[o + (int64_t)t * H];
            e_gh[q][t] = gh[o + (int64_t)t * H];
        }
        e_hp[q] = h_prev ? h_prev[e] : 0.0f;
        e_dh[q] = dhp[e];
        if (d_y) e_dh[q] += d_y[e] * drop_scale(seed, idx_base + (uint64_t)e, thr, inv_keep);
    }
    const int len = K3 / (2 * kGruNW);
    const int k0 = (wave * 2 + g) * len;
    const float* ap = dgh_next + (int64_t)min(i, B - 1) * K3 + k0;
    const float* bp = whh_t + (int64_t)j * K3 + k0;
    gru_f16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    int kk = 0;
    for (; kk + 32 <= len; kk += 32) {                          // 16 x 16-byte loads in flight per lane
        float4 a[8], b[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            a[v] = *reinterpret_cast<const float4*>(ap + kk + 4 * v);
            b[v] = *reinterpret_cast<const float4*>(bp + kk + 4 * v);
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) { GRU_MFMA4(acc, a[v], b[v]) }
    }
    for (; kk < len; kk += 4) {
        const float4 a = *reinterpret_cast<const float4*>(ap + kk);
        const float4 b = *reinterpret_cast<const float4*>(bp + kk);
        GRU_MFMA4(acc, a, b)
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) gru_red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 2 * wave + q;
        float dh = 0.0f;
#pragma unroll
        for (int w = 0; w < kGruNW; ++w) dh += gru_red[(w * 16 + r) * 64 + lane];       // fixed wave order: deterministic
        if (e_row[q] < B) {
            const int64_t o = (int64_t)e_row[q] * 3 * H + j, e = (int64_t)e_row[q] * H + j;
            dh += e_dh[q];
            const float ghn = e_gh[q][2];
            const float rr = sigmoidf_(e_gi[q][0] + e_gh[q][0]);
            const float uu = sigmoidf_(e_gi[q][1] + e_gh[q][1]);
            const float nn = tanhf(e_gi[q][2] + rr * ghn);
            const float hp = e_hp[q];
            const float da_n = dh * (1.0f - uu) * (1.0f - nn * nn);
            const float da_u = dh * (hp - nn) * uu * (1.0f - uu);
            const float da_r = da_n * ghn * rr * (1.0f - rr);
            d_gi[o] = da_r; d_gi[o + H] = da_u; d_gi[o + 2 * H] = da_n;
            d_gh[o] = da_r; d_gh[o + H] = da_u; d_gh[o + 2 * H] = da_n * rr;
            dhp[e] = dh * uu;
        }
    }
}
#undef GRU_MFMA4

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_gru_cell_fwd(const float* gi, const float* gh, const float* h_prev, float* h_out, float* y_out, int64_t B, int H,
                       float drop_p, uint64_t seed, uint64_t idx_base, void* stream) {
    if (B == 0) return VQCPC_OK;
    VQ_REQUIRE(gi && gh && h_out && B >= 0 && H >= 1, "gru_cell_fwd: bad arguments");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gru_cell_fwd: bad dropout probability");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((unsigned)ceil_div(B * H, 256)), dim3(256), 0, (hipStream_t)stream, gi, gh,
                       h_prev, h_out, y_out, B, H, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, idx_base);
    VQ_CHECK_LAUNCH("gru_cell_fwd");
    return VQCPC_OK;
}

int vqcpc_gru_cell_bwd(const float* gi, const float* gh, const float* h_prev, const float* d_y, const float* d_h, float* d_gi,
                       float* d_gh, float* d_hprev, int64_t B, int H, float drop_p, uint64_t seed, uint64_t idx_base,
                       void* stream) {
    if (B == 0) return VQCPC_OK;
    VQ_REQUIRE(gi && gh && d_gi && d_gh && d_hprev && (d_y || d_h) && B >= 0 && H >= 1, "gru_cell_bwd: bad arguments");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gru_cell_bwd: bad dropout probability");
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((unsigned)ceil_div(B * H, 256)), dim3(256), 0, (hipStream_t)stream, gi, gh,
                       h_prev, d_y, d_h, d_gi, d_gh, d_hprev, B, H, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed,
                       idx_base);
    VQ_CHECK_LAUNCH("gru_cell_bwd");
    return VQCPC_OK;
}

int vqcpc_gru_step_supported(int64_t B, int H) { return (B >= 1 && H >= 64 && H % 64 == 0 && H <= 4096) ? 1 : 0; }

int vqcpc_gru_step_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h_prev, float* gh, float* h_out,
                       float* y_out, int64_t B, int H, float drop_p, uint64_t seed, uint64_t idx_base, void* stream) {
    if (B == 0) return VQCPC_OK;
    VQ_REQUIRE(gi && w_hh && b_hh && gh && h_out, "gru_step_fwd: null pointer");
    VQ_REQUIRE(vqcpc_gru_step_supported(B, H) && B < (1 << 30), "gru_step_fwd: needs H %% 64 == 0 (B=%lld H=%d)", (long long)B, H);
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gru_step_fwd: bad dropout probability");
    VQ_REQUIRE(aligned16(w_hh) && (!h_prev || aligned16(h_prev)), "gru_step_fwd: operands must be 16-byte aligned");
    const size_t lds = (size_t)kGruNW * 3 * 16 * 64 * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gru_step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(H / 32, (unsigned)ceil_div(B, 32)), dim3(kGruNW * 64), lds, (hipStream_t)stream,
                       gi, w_hh, b_hh, h_prev, gh, h_out, y_out, (int)B, H, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed,
                       idx_base);
    VQ_CHECK_LAUNCH("gru_step_fwd");
    return VQCPC_OK;
}

int vqcpc_gru_step_bwd(const float* dgh_next, const float* whh_t, float* dhp, const float* gi, const float* gh,
                       const float* h_prev, const float* d_y, float* d_gi, float* d_gh, int64_t B, int H, float drop_p,
                       uint64_t seed, uint64_t idx_base, void* stream) {
    if (B == 0) return VQCPC_OK;
    VQ_REQUIRE(dgh_next && whh_t && dhp && gi && gh && d_gi && d_gh, "gru_step_bwd: null pointer");
    VQ_REQUIRE(vqcpc_gru_step_supported(B, H) && B < (1 << 30), "gru_step_bwd: needs H %% 64 == 0 (B=%lld H=%d)", (long long)B, H);
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gru_step_bwd: bad dropout probability");
    VQ_REQUIRE(aligned16(dgh_next) && aligned16(whh_t), "gru_step_bwd: operands must be 16-byte aligned");
    const size_t lds = (size_t)kGruNW * 16 * 64 * sizeof(float);
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(H / 32, (unsigned)ceil_div(B, 32)), dim3(kGruNW * 64), lds, (hipStream_t)stream,
                       dgh_next, whh_t, dhp, gi, gh, h_prev, d_y, d_gi, d_gh, (int)B, H, drop_threshold(drop_p),
                       1.0f / (1.0f - drop_p), seed, idx_base);
    VQ_CHECK_LAUNCH("gru_step_bwd");
    return VQCPC_OK;
}

}  // extern "C"
