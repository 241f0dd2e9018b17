// Error plumbing, partial-sum reduction, dropout mask probe, transposes, upscaler activation, flat-buffer optimiser.
#include <stdarg.h>

#include <algorithm>

#include "common.h"

namespace vq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------------------------------
// 64 columns x 16 split-groups per workgroup: group g sums splits g, g+16, ... (4 independent loads in flight), the 16
// group partials are combined in a fixed order -> deterministic, and no lane walks more than nsplit/16 dependent loads.
constexpr int kRedCols = 64, kRedGroups = 16;

// Two column segments per launch (a weight gradient and its bias gradient share one launch): columns [0, count) come
// from (ws, stride) and go to `out`, columns [count, count + count2) from (ws2, stride2) to `out2`.
__global__ __launch_bounds__(kRedCols * kRedGroups) void reduce_splits_kernel(const float* __restrict__ ws,
                                                                              int64_t stride, int nsplit,
                                                                              float* __restrict__ out, int64_t count,
                                                                              const float* __restrict__ ws2, int64_t stride2,
                                                                              float* __restrict__ out2, int64_t count2,
                                                                              int accumulate) {
    __shared__ float part[kRedGroups][kRedCols];
    const int tx = threadIdx.x % kRedCols, ty = threadIdx.x / kRedCols;
    int64_t col = (int64_t)blockIdx.x * kRedCols + tx;
    const int64_t first = (count + kRedCols - 1) / kRedCols * kRedCols;      // segment 2 starts on a workgroup boundary
    if (col >= first) {
        col -= first;
        ws = ws2;
        stride = stride2;
        out = out2;
        count = count2;
    }
    float acc = 0.0f;
    if (col < count) {
        int s = ty;
        for (; s + 3 * kRedGroups < nsplit; s += 4 * kRedGroups) {
            const float a = ws[(int64_t)s * stride + col];
            const float b = ws[(int64_t)(s + kRedGroups) * stride + col];
            const float c = ws[(int64_t)(s + 2 * kRedGroups) * stride + col];
            const float d = ws[(int64_t)(s + 3 * kRedGroups) * stride + col];
            acc = ((acc + a) + b) + (c + d);
        }
        for (; s < nsplit; s += kRedGroups) acc += ws[(int64_t)s * stride + col];
    }
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && col < count) {
        float tot = accumulate ? out[col] : 0.0f;
#pragma unroll
        for (int g = 0; g < kRedGroups; ++g) tot += part[g][tx];
        out[col] = tot;
    }
}

// Large segments, few splits (the weight-gradient partial sums of a GEMM: 10^5..10^6 columns x 4..32 splits): one float4
// column group per lane, all splits walked by that lane in a fixed order with eight 16-byte loads in flight.  The kernel
// above spends a 1024-lane workgroup on 64 columns, which is right for "many splits of a short vector" only: at the student
// step's 2048 x 512 gradients (8 splits) it ran at 1 TB/s, this one streams them at the L2 / MALL rate.
constexpr int kRedVecThreads = 256;

__global__ __launch_bounds__(kRedVecThreads) void reduce_splits_vec_kernel(const float* __restrict__ ws, int64_t stride,
                                                                          int nsplit, float* __restrict__ out, int64_t count,
                                                                          const float* __restrict__ ws2, int64_t stride2,
                                                                          float* __restrict__ out2, int64_t count2,
                                                                          int accumulate) {
    int64_t q = (int64_t)blockIdx.x * kRedVecThreads + threadIdx.x;                     // float4 index
    const int64_t first = (count / 4 + kRedVecThreads - 1) / kRedVecThreads * kRedVecThreads;  // segment 2: workgroup boundary
    if (q >= first) {
        q -= first;
        ws = ws2;
        stride = stride2;
        out = out2;
        count = count2;
    }
    if (q * 4 >= count) return;
    const float4* p = reinterpret_cast<const float4*>(ws) + q;
    const int64_t st4 = stride / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define RV_ADD(X, Y) make_float4(X.x + Y.x, X.y + Y.y, X.z + Y.z, X.w + Y.w)
    int s = 0;
    for (; s + 7 < nsplit; s += 8) {
        const float4 v0 = p[(s + 0) * st4], v1 = p[(s + 1) * st4], v2 = p[(s + 2) * st4], v3 = p[(s + 3) * st4];
        const float4 v4 = p[(s + 4) * st4], v5 = p[(s + 5) * st4], v6 = p[(s + 6) * st4], v7 = p[(s + 7) * st4];
        const float4 a = RV_ADD(v0, v1), b = RV_ADD(v2, v3), c = RV_ADD(v4, v5), d = RV_ADD(v6, v7);
        const float4 ab = RV_ADD(a, b), cd = RV_ADD(c, d);
        const float4 t = RV_ADD(ab, cd);
        acc = RV_ADD(acc, t);
    }
    for (; s < nsplit; ++s) {
        const float4 v = p[s * st4];
        acc = RV_ADD(acc, v);
    }
    float4* o = reinterpret_cast<float4*>(out) + q;
    if (accumulate) {
        const float4 prev = *o;
        acc = RV_ADD(prev, acc);
    }
#undef RV_ADD
    *o = acc;
}

static bool reduce_vec_ok(const float* ws, int64_t stride, const float* out, int64_t count) {
    return count <= 0 || (count % 4 == 0 && stride % 4 == 0 && aligned16(ws) && aligned16(out));
}

int launch_reduce_splits2(const float* ws, int64_t stride, int nsplit, float* out, int64_t count, const float* ws2,
                          int64_t stride2, float* out2, int64_t count2, int accumulate, hipStream_t stream) {
    if (count <= 0 && count2 <= 0) return VQCPC_OK;
    if (count >= (1 << 16) && nsplit <= 64 && reduce_vec_ok(ws, stride, out, count) &&
        reduce_vec_ok(ws2, stride2, out2, count2)) {
        const int64_t blocks = ceil_div(count / 4, (int64_t)kRedVecThreads) +
                               ceil_div(std::max<int64_t>(count2, 0) / 4, (int64_t)kRedVecThreads);
        hipLaunchKernelGGL(reduce_splits_vec_kernel, dim3((unsigned)blocks), dim3(kRedVecThreads), 0, stream, ws, stride,
                           nsplit, out, count, ws2, stride2, out2, std::max<int64_t>(count2, 0), accumulate);
        VQ_CHECK_LAUNCH("reduce_splits_vec");
        return VQCPC_OK;
    }
    const int64_t blocks = ceil_div(std::max<int64_t>(count, 0), kRedCols) + ceil_div(std::max<int64_t>(count2, 0), kRedCols);
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)blocks), dim3(kRedCols * kRedGroups), 0, stream, ws, stride,
                       nsplit, out, std::max<int64_t>(count, 0), ws2, stride2, out2, std::max<int64_t>(count2, 0), accumulate);
    VQ_CHECK_LAUNCH("reduce_splits");
    return VQCPC_OK;
}

// n independent "sum nsplit partial rows" reductions in one launch (32 per launch): out_i[c] (+)= sum_s ws_i[s * stride_i + c].
// The table travels by value in the kernel arguments.  Same summation order per column as reduce_splits_kernel.
constexpr int kRedMany = 32;
struct RedManyArgs {
    const float* ws[kRedMany];
    float* out[kRedMany];
    int stride[kRedMany], nsplit[kRedMany], count[kRedMany];
    int blk_begin[kRedMany + 1];
    int n, accumulate;
};

__global__ __launch_bounds__(kRedCols * kRedGroups) void reduce_many_kernel(const RedManyArgs g) {
    __shared__ float part[kRedGroups][kRedCols];
    const int b = (int)blockIdx.x;
    int p = 0;
    for (int i = 1; i < g.n; ++i) p += (b >= g.blk_begin[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    const float* __restrict__ ws = g.ws[p];
    const int64_t stride = g.stride[p];
    const int nsplit = g.nsplit[p], count = g.count[p];
    const int tx = threadIdx.x % kRedCols, ty = threadIdx.x / kRedCols;
    const int col = (b - g.blk_begin[p]) * kRedCols + tx;
    float acc = 0.0f;
    if (col < count) {
        int s = ty;
        for (; s + 3 * kRedGroups < nsplit; s += 4 * kRedGroups) {
            const float a = ws[(int64_t)s * stride + col];
            const float b2 = ws[(int64_t)(s + kRedGroups) * stride + col];
            const float c = ws[(int64_t)(s + 2 * kRedGroups) * stride + col];
            const float d = ws[(int64_t)(s + 3 * kRedGroups) * stride + col];
            acc = ((acc + a) + b2) + (c + d);
        }
        for (; s < nsplit; s += kRedGroups) acc += ws[(int64_t)s * stride + col];
    }
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && col < count) {
        float* out = g.out[p];
        float tot = g.accumulate ? out[col] : 0.0f;
#pragma unroll
        for (int q = 0; q < kRedGroups; ++q) tot += part[q][tx];
        out[col] = tot;
    }
}

// The float4 form for many LARGE segments in one launch (the weight-gradient partial sums of a whole backward pass, deferred to
// its end): per column group the arithmetic of reduce_splits_vec_kernel, so a deferred reduction gives the bits of the
// immediate one.  Segment p owns workgroups [blk_begin[p], blk_begin[p + 1]).
__global__ __launch_bounds__(kRedVecThreads) void reduce_many_vec_kernel(const RedManyArgs g) {
    const int b = (int)blockIdx.x;
    int p = 0;
    for (int i = 1; i < g.n; ++i) p += (b >= g.blk_begin[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    const int64_t q = (int64_t)(b - g.blk_begin[p]) * kRedVecThreads + threadIdx.x;      // float4 index inside the segment
    if (q * 4 >= g.count[p]) return;
    const float4* ptr = reinterpret_cast<const float4*>(g.ws[p]) + q;
    const int64_t st4 = g.stride[p] / 4;
    const int nsplit = g.nsplit[p];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define RV_ADD(X, Y) make_float4(X.x + Y.x, X.y + Y.y, X.z + Y.z, X.w + Y.w)
    int s = 0;
    for (; s + 7 < nsplit; s += 8) {
        const float4 v0 = ptr[(s + 0) * st4], v1 = ptr[(s + 1) * st4], v2 = ptr[(s + 2) * st4], v3 = ptr[(s + 3) * st4];
        const float4 v4 = ptr[(s + 4) * st4], v5 = ptr[(s + 5) * st4], v6 = ptr[(s + 6) * st4], v7 = ptr[(s + 7) * st4];
        const float4 a = RV_ADD(v0, v1), bb = RV_ADD(v2, v3), c = RV_ADD(v4, v5), d = RV_ADD(v6, v7);
        const float4 ab = RV_ADD(a, bb), cd = RV_ADD(c, d);
        const float4 t = RV_ADD(ab, cd);
        acc = RV_ADD(acc, t);
    }
    for (; s < nsplit; ++s) {
        const float4 v = ptr[s * st4];
        acc = RV_ADD(acc, v);
    }
    float4* o = reinterpret_cast<float4*>(g.out[p]) + q;
    if (g.accumulate) {
        const float4 prev = *o;
        acc = RV_ADD(prev, acc);
    }
#undef RV_ADD
    *o = acc;
}

int launch_reduce_splits(const float* ws, int64_t stride, int nsplit, float* out, int64_t count, int accumulate,
                         hipStream_t stream) {
    return launch_reduce_splits2(ws, stride, nsplit, out, count, nullptr, 0, nullptr, 0, accumulate, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* mask, int64_t n, uint32_t thr, uint64_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) mask[i] = (thr == 0 || rng_u24(seed, (uint64_t)i) >= thr) ? 1.0f : 0.0f;
}

// 32x32 LDS-tiled transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int r = r0 + ty + k * 8, c = c0 + tx;
        if (r < R && c < C) tile[ty + k * 8][tx] = in[(int64_t)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = c0 + ty + k * 8, r = r0 + tx;
        if (r < R && c < C) out[(int64_t)c * R + r] = tile[tx][ty + k * 8];
    }
}

// All the W^T operands of a backward pass in ONE launch: matrix i lives at base_in + desc[4 i] (rows desc[4 i + 1], cols
// desc[4 i + 2]) and its transpose goes to the same offset of base_out; desc[4 i + 3] = index of its first 32 x 32 tile
// in the launch (ascending), so a workgroup finds its matrix by bisection.  (A training step transposed 25-80 weights
// with one 5 us launch each.)
__global__ __launch_bounds__(256) void transpose_many_kernel(const float* __restrict__ base_in, float* __restrict__ base_out,
                                                             const int64_t* __restrict__ desc, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;
    const int64_t t = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[4 * mid + 3] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t off = desc[4 * lo];
    const int R = (int)desc[4 * lo + 1], C = (int)desc[4 * lo + 2];
    const int local = (int)(t - desc[4 * lo + 3]), tiles_c = (C + 31) / 32;
    const float* in = base_in + off;
    float* out = base_out + off;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = (local % tiles_c) * 32, r0 = (local / tiles_c) * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + k * 8, c = c0 + tx;
        if (r < R && c < C) tile[ty + k * 8][tx] = in[(int64_t)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + k * 8, r = r0 + tx;
        if (r < R && c < C) out[(int64_t)c * R + r] = tile[tx][ty + k * 8];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float selu_f(float x) {
    const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
    return scale * (x > 0.0f ? x : alpha * (__expf(x) - 1.0f));
}
__device__ __forceinline__ float selu_grad(float x) {
    const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
    return scale * (x > 0.0f ? 1.0f : alpha * __expf(x));
}

__global__ __launch_bounds__(256) void dropout_selu_fwd_kernel(const float* __restrict__ h, float* __restrict__ out,
                                                               int64_t n, uint32_t thr, float inv_keep, uint64_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) out[i] = selu_f(h[i] * drop_scale(seed, (uint64_t)i, thr, inv_keep));
}

__global__ __launch_bounds__(256) void dropout_selu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ g,
                                                               float* __restrict__ gh, int64_t n, uint32_t thr,
                                                               float inv_keep, uint64_t seed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const float s = drop_scale(seed, (uint64_t)i, thr, inv_keep);
        gh[i] = g[i] * selu_grad(h[i] * s) * s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// sum of squares, two deterministic stages with double accumulation
constexpr int kSumsqBlocks = 1024;

__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ g, int64_t n, float scale,
                                                    double* __restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < n4; i += step) {
        float4 v = g4[i];
        float s = (v.x * scale) * (v.x * scale) + (v.y * scale) * (v.y * scale) + (v.z * scale) * (v.z * scale) +
                  (v.w * scale) * (v.w * scale);
        acc += (double)s;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        float v = g[(n4 << 2) + threadIdx.x] * scale;
        acc += (double)(v * v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sumsq_stage2(const double* __restrict__ partial, int nparts,
                                                    double* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr_over_bc1, float beta1,
                                                   float beta2, float eps, float inv_sqrt_bc2, float grad_scale,
                                                   float max_norm, const double* __restrict__ sumsq) {
    float coef = grad_scale;
    if (sumsq != nullptr) {
        const float total = (float)sqrt(sumsq[0]);
        coef *= fminf(1.0f, max_norm / (total + 1e-6f));
    }
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        g[i] = gi;   // gradients are left clipped in place, as clip_grad_norm_ does
        p[i] -= lr_over_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Step salt registry (see common.h): addresses of every translation unit's g_rng_salt, resolved lazily (the HIP runtime
// must be up), and the kernels that write them.
static SaltAddrFn g_salt_fns[32];
static int g_salt_nfn = 0;
void register_rng_salt(SaltAddrFn fn) {
    if (g_salt_nfn < 32) g_salt_fns[g_salt_nfn++] = fn;
}
struct SaltAddrs {
    uint64_t* p[32];
    int n;
};
static SaltAddrs salt_addrs() {
    static SaltAddrs a = [] {
        SaltAddrs r{};
        for (int i = 0; i < g_salt_nfn; ++i) {
            uint64_t* q = g_salt_fns[i]();
            if (q) r.p[r.n++] = q;
        }
        return r;
    }();
    return a;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void rng_salt_set_kernel(SaltAddrs a, uint64_t value) {
    if (threadIdx.x < a.n) *a.p[threadIdx.x] = value;
}
// counter[0] += 1; salt = splitmix64(base ^ counter[0]) (never 0): ONE thread decides, then every copy is written
__global__ void rng_salt_advance_kernel(SaltAddrs a, uint64_t* counter, uint64_t base) {
    __shared__ uint64_t salt;
    if (threadIdx.x == 0) {
        const uint64_t c = counter[0] + 1;
        counter[0] = c;
        const uint64_t v = splitmix64(base ^ c);
        salt = v ? v : 1;
    }
    __syncthreads();
    if (threadIdx.x < a.n) *a.p[threadIdx.x] = salt;
}
// the same salt again, WITHOUT advancing the counter: a later captured stage of the same step (graphs.py: bucketed all-reduces)
__global__ void rng_salt_from_counter_kernel(SaltAddrs a, const uint64_t* counter, uint64_t base) {
    const uint64_t v = splitmix64(base ^ counter[0]);
    if (threadIdx.x < a.n) *a.p[threadIdx.x] = v ? v : 1;
}

// Adam with its per-step scalars on the device: lr from lr_dev[0], step count t from step_dev[0] (graph replay)
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, const float* __restrict__ lr_dev,
                                                       float beta1, float beta2, float eps,
                                                       const uint64_t* __restrict__ step_dev, float grad_scale,
                                                       float max_norm, const double* __restrict__ sumsq) {
    __shared__ float sh[2];
    if (threadIdx.x == 0) {
        const double t = (double)step_dev[0];
        const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
        sh[0] = (float)((double)lr_dev[0] / bc1);
        sh[1] = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();
    const float lr_over_bc1 = sh[0], inv_sqrt_bc2 = sh[1];
    float coef = grad_scale;
    if (sumsq != nullptr) {
        const float total = (float)sqrt(sumsq[0]);
        coef *= fminf(1.0f, max_norm / (total + 1e-6f));
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        g[i] = gi;   // gradients are left clipped in place, as clip_grad_norm_ does
        p[i] -= lr_over_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Token range check.  nn.Embedding raises on an id outside its table; the kernels of this library index tables (and an LDS
// accumulator in the embedding backward) directly, so every token tensor passes through here once: the copy that the
// kernels consume is clamped into [0, limit[voice]) and `flag` records that a clamp happened (the host raises on it at
// its next synchronisation point).  voice = flat index % n_voices ((tick, voice) order, voices fastest).
struct TokenLimits {
    int v[16];
};
__global__ __launch_bounds__(256) void check_tokens_kernel(const int64_t* __restrict__ in, int64_t n, int nv,
                                                           TokenLimits lim, int64_t* __restrict__ out,
                                                           int* __restrict__ flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t t = in[i];
        const int64_t hi = lim.v[(int)(i % nv)] - 1;
        const int64_t c = t < 0 ? 0 : (t > hi ? hi : t);
        bad |= c != t;
        out[i] = c;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

struct Acc8 {
    float* dst[8];
    const float* src[8];
    int n[8];
};
__global__ __launch_bounds__(256) void accumulate8_kernel(Acc8 a) {
    const int t = blockIdx.y;
    float* d = a.dst[t];
    const float* s = a.src[t];
    const int n = a.n[t];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] += s[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Number of distinct (merged) product codes among the rows of up to two index tensors: the per-step `num_codewords`
// metrics of VQCPCEncoderTrainer.epoch (vqcpc_encoder_trainer.py:320-331: len(torch.unique(merged codes))).  One
// workgroup, a bit per possible code in LDS (K^ncb <= 2^20 bits = 128 KB), set with LDS atomics and counted: a set has no
// order, so the result is deterministic.  Replaces cat + sort + compare + sum (~12 launches per metric per step).
constexpr int kCountThreads = 1024;
constexpr int64_t kCountMaxSpace = 1 << 20;

__global__ __launch_bounds__(kCountThreads) void count_distinct_codes_kernel(const int64_t* __restrict__ idx_a, int64_t rows_a,
                                                                            const int64_t* __restrict__ idx_b, int64_t rows_b,
                                                                            int ncb, int K, int words, float* __restrict__ out) {
    extern __shared__ uint32_t code_bits[];
    __shared__ int wave_tot[kCountThreads / 64];
    for (int i = threadIdx.x; i < words; i += kCountThreads) code_bits[i] = 0u;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < rows_a + rows_b; i += kCountThreads) {
        const int64_t* row = i < rows_a ? idx_a + i * ncb : idx_b + (i - rows_a) * ncb;
        int64_t code = 0, mul = 1;
        for (int c = 0; c < ncb; ++c) {
            code += row[c] * mul;
            mul *= K;
        }
        if (code >= 0 && code < (int64_t)words * 32) atomicOr(&code_bits[code >> 5], 1u << (code & 31));
    }
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < words; i += kCountThreads) cnt += __popc(code_bits[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < kCountThreads / 64; ++w) tot += wave_tot[w];
        out[0] = (float)tot;
    }
}

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_abi_version(void) { return VQCPC_ABI_VERSION; }
const char* vqcpc_last_error(void) { return vq::g_err; }
int vqcpc_clear_runtime_error(void) { return (int)hipGetLastError(); }

int vqcpc_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(mask != nullptr && n >= 0 && p >= 0.f && p < 1.f, "dropout_mask: bad arguments");
    int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, n, drop_threshold(p),
                       seed);
    VQ_CHECK_LAUNCH("dropout_mask");
    return VQCPC_OK;
}

int vqcpc_transpose(const float* in, float* out, int R, int C, void* stream) {
    VQ_REQUIRE(in && out && R > 0 && C > 0, "transpose: bad arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0, (hipStream_t)stream, in,
                       out, R, C);
    VQ_CHECK_LAUNCH("transpose");
    return VQCPC_OK;
}

int vqcpc_transpose_many(const float* base_in, float* base_out, const int64_t* desc, int n, int64_t total_tiles,
                         void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(base_in && base_out && desc && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31) && base_in != base_out,
               "transpose_many: bad arguments");
    hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, base_in, base_out,
                       desc, n);
    VQ_CHECK_LAUNCH("transpose_many");
    return VQCPC_OK;
}

int vqcpc_dropout_selu_fwd(const float* h, float* out, int64_t n, float drop_p, uint64_t seed, void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(h && out && n >= 0 && drop_p >= 0.f && drop_p < 1.f, "dropout_selu_fwd: bad arguments");
    int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(dropout_selu_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h, out, n,
                       drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed);
    VQ_CHECK_LAUNCH("dropout_selu_fwd");
    return VQCPC_OK;
}

int vqcpc_dropout_selu_bwd(const float* h, const float* g_out, float* g_h, int64_t n, float drop_p, uint64_t seed,
                           void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(h && g_out && g_h && n >= 0 && drop_p >= 0.f && drop_p < 1.f, "dropout_selu_bwd: bad arguments");
    int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(dropout_selu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h, g_out, g_h, n,
                       drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed);
    VQ_CHECK_LAUNCH("dropout_selu_bwd");
    return VQCPC_OK;
}

int64_t vqcpc_sumsq_workspace(int64_t n) {
    (void)n;
    return (int64_t)kSumsqBlocks * sizeof(double);
}

int vqcpc_sumsq(const float* g, int64_t n, float grad_scale, double* out, void* workspace, int64_t workspace_bytes,
                void* stream) {
    VQ_REQUIRE(g && out && workspace && n > 0, "sumsq: bad arguments");
    VQ_REQUIRE(aligned16(g), "sumsq: g must be 16-byte aligned");
    if (workspace_bytes < vqcpc_sumsq_workspace(n)) {
        set_error("sumsq: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    int blocks = (int)std::min<int64_t>(ceil_div(n / 4 + 1, 256), kSumsqBlocks);
    hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, grad_scale,
                       (double*)workspace);
    VQ_CHECK_LAUNCH("sumsq_stage1");
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, blocks, out);
    VQ_CHECK_LAUNCH("sumsq_stage2");
    return VQCPC_OK;
}

int vqcpc_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step, float grad_scale, float max_norm, const double* sumsq, void* stream) {
    VQ_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(lr / bc1),
                       beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale, max_norm, sumsq);
    VQ_CHECK_LAUNCH("adam_step");
    return VQCPC_OK;
}

int vqcpc_check_tokens(const int64_t* tokens, int64_t n, int n_voices, const int32_t* limits, int64_t* clamped, int32_t* flag,
                       void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(tokens && limits && clamped && flag && n > 0 && n_voices >= 1 && n_voices <= 16,
               "check_tokens: bad arguments (n_voices <= 16)");
    TokenLimits lim;
    for (int i = 0; i < 16; ++i) lim.v[i] = i < n_voices ? limits[i] : 1;
    for (int i = 0; i < n_voices; ++i) VQ_REQUIRE(lim.v[i] >= 1, "check_tokens: empty vocabulary for voice %d", i);
    int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 2048);
    hipLaunchKernelGGL(check_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tokens, n, n_voices, lim,
                       clamped, flag);
    VQ_CHECK_LAUNCH("check_tokens");
    return VQCPC_OK;
}

int vqcpc_rng_salt_set(uint64_t value, void* stream) {
    const SaltAddrs a = salt_addrs();
    hipLaunchKernelGGL(rng_salt_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, value);
    VQ_CHECK_LAUNCH("rng_salt_set");
    return VQCPC_OK;
}

int vqcpc_rng_salt_advance(uint64_t* counter, uint64_t base, void* stream) {
    VQ_REQUIRE(counter != nullptr, "rng_salt_advance: null counter");
    const SaltAddrs a = salt_addrs();
    hipLaunchKernelGGL(rng_salt_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, counter, base);
    VQ_CHECK_LAUNCH("rng_salt_advance");
    return VQCPC_OK;
}

int vqcpc_rng_salt_from_counter(const uint64_t* counter, uint64_t base, void* stream) {
    VQ_REQUIRE(counter != nullptr, "rng_salt_from_counter: null counter");
    const SaltAddrs a = salt_addrs();
    hipLaunchKernelGGL(rng_salt_from_counter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, counter, base);
    VQ_CHECK_LAUNCH("rng_salt_from_counter");
    return VQCPC_OK;
}

int vqcpc_adam_step_dev(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                        float eps, const uint64_t* step_dev, float grad_scale, float max_norm, const double* sumsq,
                        void* stream) {
    VQ_REQUIRE(p && g && m && v && lr_dev && step_dev && n > 0, "adam_step_dev: bad arguments");
    int blocks = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_dev, beta1, beta2,
                       eps, step_dev, grad_scale, max_norm, sumsq);
    VQ_CHECK_LAUNCH("adam_step_dev");
    return VQCPC_OK;
}


// dst[i][:] += src[i][:] for up to 8 small tensors in ONE launch: the gradients of the small per-layer parameters
// (LayerNorm gamma / beta, the relative-position tables) that the layer's backward would otherwise hand to autograd, which
// adds each into the flat gradient buffer with a kernel of its own (43 launches per C1 step).
int vqcpc_accumulate8(float* const* dst, const float* const* src, const int* counts, int n_tensors, void* stream) {
    VQ_REQUIRE(dst && src && counts && n_tensors >= 0 && n_tensors <= 8, "accumulate8: at most 8 tensors");
    if (n_tensors == 0) return VQCPC_OK;
    Acc8 a{};
    int maxn = 0;
    for (int i = 0; i < n_tensors; ++i) {
        VQ_REQUIRE(dst[i] && src[i] && counts[i] >= 0, "accumulate8: null pointer");
        a.dst[i] = dst[i];
        a.src[i] = src[i];
        a.n[i] = counts[i];
        maxn = std::max(maxn, counts[i]);
    }
    if (maxn == 0) return VQCPC_OK;
    const int blocks_x = std::min((maxn + 255) / 256, 64);
    hipLaunchKernelGGL(accumulate8_kernel, dim3(blocks_x, n_tensors), dim3(256), 0, (hipStream_t)stream, a);
    VQ_CHECK_LAUNCH("accumulate8");
    return VQCPC_OK;
}

int vqcpc_count_distinct_codes_supported(int num_codebooks, int codebook_size) {
    if (num_codebooks < 1 || codebook_size < 1) return 0;
    int64_t space = 1;
    for (int c = 0; c < num_codebooks; ++c) {
        space *= codebook_size;
        if (space > kCountMaxSpace) return 0;
    }
    return 1;
}

int vqcpc_count_distinct_codes(const int64_t* idx_a, int64_t rows_a, const int64_t* idx_b, int64_t rows_b, int num_codebooks,
                               int codebook_size, float* out, void* stream) {
    VQ_REQUIRE(out && rows_a >= 0 && rows_b >= 0 && (idx_a || rows_a == 0) && (idx_b || rows_b == 0),
               "count_distinct_codes: bad arguments");
    VQ_REQUIRE(vqcpc_count_distinct_codes_supported(num_codebooks, codebook_size),
               "count_distinct_codes: %d codebooks of %d codes exceed the 2^20-code LDS bitmap (count with a sort instead)",
               num_codebooks, codebook_size);
    int64_t space = 1;
    for (int c = 0; c < num_codebooks; ++c) space *= codebook_size;
    const int words = (int)((space + 31) / 32);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)count_distinct_codes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(kCountMaxSpace / 8));
        attr_done = true;
    }
    hipLaunchKernelGGL(count_distinct_codes_kernel, dim3(1), dim3(kCountThreads), (size_t)words * 4, (hipStream_t)stream, idx_a,
                       rows_a, idx_b, rows_b, num_codebooks, codebook_size, words, out);
    VQ_CHECK_LAUNCH("count_distinct_codes");
    return VQCPC_OK;
}

int vqcpc_reduce_grouped_vec(int n, const void* const* ws, const int64_t* stride, const int* nsplit, void* const* out,
                             const int64_t* count, int accumulate, void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(n > 0 && ws && stride && nsplit && out && count, "reduce_grouped_vec: null pointer");
    for (int i = 0; i < n; ++i)
        VQ_REQUIRE(ws[i] && out[i] && nsplit[i] >= 1 && count[i] >= 4 && count[i] % 4 == 0 && stride[i] % 4 == 0 &&
                       stride[i] >= count[i] && stride[i] < (1ll << 31) && count[i] < (1ll << 31) && aligned16(ws[i]) &&
                       aligned16(out[i]),
                   "reduce_grouped_vec: bad segment %d (counts and strides multiples of 4, 16-byte aligned)", i);
    hipStream_t s = (hipStream_t)stream;
    int i = 0;
    while (i < n) {
        RedManyArgs g;
        int c = 0, rb = 0;
        while (i < n && c < kRedMany) {
            bool dup = false;                      // a repeated output waits for the next launch (stream-ordered accumulation)
            for (int j = 0; j < c; ++j) dup = dup || g.out[j] == (float*)out[i];
            if (dup) break;
            g.ws[c] = (const float*)ws[i];
            g.out[c] = (float*)out[i];
            g.stride[c] = (int)stride[i];
            g.nsplit[c] = nsplit[i];
            g.count[c] = (int)count[i];
            g.blk_begin[c] = rb;
            rb += (int)ceil_div(count[i] / 4, (int64_t)kRedVecThreads);
            ++c;
            ++i;
        }
        for (int j = c; j <= kRedMany; ++j) g.blk_begin[j] = rb;
        for (int j = c; j < kRedMany; ++j) {
            g.ws[j] = nullptr; g.out[j] = nullptr; g.stride[j] = g.nsplit[j] = g.count[j] = 0;
        }
        g.n = c;
        g.accumulate = accumulate;
        hipLaunchKernelGGL(reduce_many_vec_kernel, dim3((unsigned)rb), dim3(kRedVecThreads), 0, s, g);
        VQ_CHECK_LAUNCH("reduce_grouped_vec");
    }
    return VQCPC_OK;
}

int vqcpc_reduce_grouped(int n, const void* const* ws, const int64_t* stride, const int* nsplit, void* const* out,
                         const int64_t* count, int accumulate, void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(n > 0 && ws && stride && nsplit && out && count, "reduce_grouped: null pointer");
    for (int i = 0; i < n; ++i)
        VQ_REQUIRE(ws[i] && out[i] && nsplit[i] >= 1 && count[i] >= 1 && stride[i] >= count[i] && stride[i] < (1ll << 31) &&
                       count[i] < (1ll << 31),
                   "reduce_grouped: bad segment %d", i);
    hipStream_t s = (hipStream_t)stream;
    int i = 0;
    while (i < n) {
        RedManyArgs g;
        int c = 0, rb = 0;
        while (i < n && c < kRedMany) {
            bool dup = false;                      // a repeated output waits for the next launch (stream-ordered accumulation)
            for (int j = 0; j < c; ++j) dup = dup || g.out[j] == (float*)out[i];
            if (dup) break;
            g.ws[c] = (const float*)ws[i];
            g.out[c] = (float*)out[i];
            g.stride[c] = (int)stride[i];
            g.nsplit[c] = nsplit[i];
            g.count[c] = (int)count[i];
            g.blk_begin[c] = rb;
            rb += (int)ceil_div(count[i], kRedCols);
            ++c;
            ++i;
        }
        for (int j = c; j <= kRedMany; ++j) g.blk_begin[j] = rb;
        for (int j = c; j < kRedMany; ++j) {
            g.ws[j] = nullptr; g.out[j] = nullptr; g.stride[j] = g.nsplit[j] = g.count[j] = 0;
        }
        g.n = c;
        g.accumulate = accumulate;
        hipLaunchKernelGGL(reduce_many_kernel, dim3((unsigned)rb), dim3(kRedCols * kRedGroups), 0, s, g);
        VQ_CHECK_LAUNCH("reduce_grouped");
    }
    return VQCPC_OK;
}

}  // extern "C"
