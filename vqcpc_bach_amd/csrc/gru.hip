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

}  // extern "C"
