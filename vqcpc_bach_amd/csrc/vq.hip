// Product vector quantiser: LDS-staged codebook, one row per lane, canonical (non-FMA, t-ascending, k-ascending) distances.
// Compiled with -ffp-contract=off; the distance uses explicitly rounded __fsub_rn/__fmul_rn/__fadd_rn so that the
// argmin is bit-identical to oracle/vqcpc_oracle.py:vq_distances_canonical (reference: vector_quantizer.py:105-116).
#include "common.h"

namespace vq {

constexpr int kVqThreads = 256;
constexpr int kVqLdsFloats = 36 * 1024;  // 144 KiB of codebook per workgroup at most

// ---------------------------------------------------------------------------------------------------------------------
// forward.  Each lane owns one row; the workgroup walks the codebooks, staging codebook c in LDS.
// DSUB > 0: sub-vector kept in registers; DSUB == 0: runtime dsub, sub-vector re-read from global (L1) per code.
template <int DSUB>
__global__ __launch_bounds__(kVqThreads) void vq_fwd_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                            int64_t R, int ncb, int K, int dsub_rt, float beta,
                                                            int squared, int assign, int64_t* __restrict__ idx_out,
                                                            float* __restrict__ zq_out, float* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int dsub = DSUB > 0 ? DSUB : dsub_rt;
    const int D = ncb * dsub;
    const int64_t r = (int64_t)blockIdx.x * kVqThreads + threadIdx.x;
    const bool live = r < R;
    const float* zr = z + r * D;
    float lsum = 0.0f;   // sum over the whole row of (q - z)^2   or  ((q - z) + eps)^2

    for (int c = 0; c < ncb; ++c) {
        __syncthreads();
        const float* src = cb + (int64_t)c * K * dsub;
        for (int i = threadIdx.x; i < K * dsub; i += kVqThreads) lds[i] = src[i];
        __syncthreads();
        if (!live) continue;

        float x[DSUB > 0 ? DSUB : 1];
        if (DSUB > 0) {
#pragma unroll
            for (int t = 0; t < DSUB; ++t) x[t] = zr[c * DSUB + t];
        }
        float best = 0.0f;
        int bi = assign ? 0 : (int)idx_out[r * ncb + c];   // assign == 0: indices are given (label corruption path)
        int kstart = 0;
        if (DSUB > 0 && assign) {
            // four codes at a time: every distance is still the canonical chain (t ascending, separately rounded sub / mul /
            // add), but the four chains are independent, so their latencies overlap (one row per lane leaves a single wave
            // per SIMD with nothing else to issue); candidates are compared in code order with the same strict '<'
            constexpr int KU = 4;
            for (; kstart + KU <= K; kstart += KU) {
                float dd[KU];
#pragma unroll
                for (int u = 0; u < KU; ++u) dd[u] = 0.0f;
#pragma unroll
                for (int t = 0; t < DSUB; ++t) {
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        const float df = __fsub_rn(x[t], lds[(kstart + u) * DSUB + t]);
                        dd[u] = __fadd_rn(dd[u], __fmul_rn(df, df));
                    }
                }
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    if ((kstart + u) == 0 || dd[u] < best) {
                        best = dd[u];
                        bi = kstart + u;
                    }
                }
            }
        }
        for (int k = kstart; k < (assign ? K : 0); ++k) {
            const float* e = lds + k * dsub;
            float d = 0.0f;
            if (DSUB > 0) {
#pragma unroll
                for (int t = 0; t < DSUB; ++t) {
                    const float df = __fsub_rn(x[t], e[t]);
                    d = __fadd_rn(d, __fmul_rn(df, df));
                }
            } else {
                for (int t = 0; t < dsub; ++t) {
                    const float df = __fsub_rn(zr[c * dsub + t], e[t]);
                    d = __fadd_rn(d, __fmul_rn(df, df));
                }
            }
            if (k == 0 || d < best) {   // strict '<': the first index wins ties; NaN never replaces
                best = d;
                bi = k;
            }
        }
        if (assign) idx_out[r * ncb + c] = (int64_t)bi;
        if (!zq_out) continue;                                      // index-only (inference) mode
        const float* q = lds + bi * dsub;
        for (int t = 0; t < dsub; ++t) {
            const float xv = DSUB > 0 ? zr[c * dsub + t] : zr[c * dsub + t];
            const float diff = __fsub_rn(q[t], xv);                 // (quantized - inputs)
            zq_out[r * D + c * dsub + t] = __fadd_rn(xv, diff);     // inputs + (quantized - inputs).detach()
            const float v = squared ? diff : __fadd_rn(diff, 1e-5f);
            lsum = __fadd_rn(lsum, __fmul_rn(v, v));
        }
    }
    if (live && loss_out) {
        const float l = squared ? lsum : sqrtf(lsum);
        loss_out[r] = __fadd_rn(l, __fmul_rn(beta, l));             // q_latent + commitment_cost * e_latent
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  Workgroup = chunk of 256 rows.  Phase A (lane = row): d_z and the per-row codebook contribution
// val = g_loss * dloss/dq, kept in LDS.  Phase B per codebook: cell (k, t) of an LDS accumulator is owned by exactly
// one lane, rows are applied in ascending order -> deterministic segment sum, no atomics.  Partials go to workspace.
__global__ __launch_bounds__(kVqThreads) void vq_bwd_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                            const int64_t* __restrict__ idx, const float* __restrict__ g_zq,
                                                            const float* __restrict__ g_loss, int64_t R, int ncb, int K,
                                                            int dsub, float beta, int squared, float* __restrict__ d_z,
                                                            float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = ncb * dsub;
    float* acc = lds;                               // [K][dsub]
    float* val = lds + K * dsub;                    // [256][dsub]  (current codebook)
    int* sidx = reinterpret_cast<int*>(val + kVqThreads * dsub);   // [256]
    const int64_t r = (int64_t)blockIdx.x * kVqThreads + threadIdx.x;
    const bool live = r < R;
    const int rows_here = (int)min((int64_t)kVqThreads, R - (int64_t)blockIdx.x * kVqThreads);

    // row norm for the non-squared variant: n = || (q - z) + eps || over the full D
    float inv_n = 0.0f;
    if (live && !squared) {
        float s = 0.0f;
        for (int c = 0; c < ncb; ++c) {
            const float* q = cb + ((int64_t)c * K + idx[r * ncb + c]) * dsub;
            for (int t = 0; t < dsub; ++t) {
                const float v = (q[t] - z[r * D + c * dsub + t]) + 1e-5f;
                s += v * v;
            }
        }
        inv_n = 1.0f / sqrtf(s);
    }
    const float gl = live ? g_loss[r] : 0.0f;

    for (int c = 0; c < ncb; ++c) {
        __syncthreads();
        for (int i = threadIdx.x; i < K * dsub; i += kVqThreads) acc[i] = 0.0f;
        if (live) {
            const int k = (int)idx[r * ncb + c];
            sidx[threadIdx.x] = k;
            const float* q = cb + ((int64_t)c * K + k) * dsub;
            for (int t = 0; t < dsub; ++t) {
                const int col = c * dsub + t;
                const float diff = q[t] - z[r * D + col];
                // dloss/dq_t  (q_latent term) and dloss/dz_t (commitment term)
                const float dq = squared ? 2.0f * diff : (diff + 1e-5f) * inv_n;
                val[threadIdx.x * dsub + t] = gl * dq;
                d_z[r * D + col] = g_zq[r * D + col] - gl * beta * dq;
            }
        }
        __syncthreads();
        // owner of cell (k, t): lane ((k % kgroups) * dsub + t)
        const int kgroups = kVqThreads / dsub;      // dsub <= 256 checked on the host
        const int my_t = threadIdx.x % dsub, my_g = threadIdx.x / dsub;
        if (my_g < kgroups) {
            for (int row = 0; row < rows_here; ++row) {
                const int k = sidx[row];
                if (k % kgroups == my_g) acc[k * dsub + my_t] += val[row * dsub + my_t];
            }
        }
        __syncthreads();
        float* dst = ws + ((int64_t)blockIdx.x * ncb + c) * K * dsub;
        for (int i = threadIdx.x; i < K * dsub; i += kVqThreads) dst[i] = acc[i];
    }
}

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_vq_fwd(const float* z, const float* codebooks, int64_t R, int ncb, int K, int dsub, float beta, int squared,
                 int assign, int64_t* idx, float* zq_sg, float* loss, void* stream) {
    if (R == 0) return VQCPC_OK;
    VQ_REQUIRE(z && codebooks && idx && ((zq_sg != nullptr) == (loss != nullptr)), "vq_fwd: null pointer");
    VQ_REQUIRE(zq_sg || assign, "vq_fwd: index-only mode (zq_sg == loss == NULL) needs assign = 1");
    VQ_REQUIRE(R >= 0 && ncb >= 1 && K >= 1 && dsub >= 1, "vq_fwd: bad shape R=%lld ncb=%d K=%d dsub=%d", (long long)R, ncb,
               K, dsub);
    VQ_REQUIRE((int64_t)K * dsub <= kVqLdsFloats, "vq_fwd: codebook of %d x %d floats does not fit the LDS", K, dsub);
    const dim3 grid((unsigned)ceil_div(R, kVqThreads)), block(kVqThreads);
    const size_t lds = (size_t)K * dsub * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define VQ_LAUNCH(DS)                                                                                                  \
    if (lds > 64 * 1024)                                                                                               \
        (void)hipFuncSetAttribute((const void*)vq_fwd_kernel<DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(vq_fwd_kernel<DS>, grid, block, lds, s, z, codebooks, R, ncb, K, dsub, beta, squared, assign, idx, \
                       zq_sg, loss)
    switch (dsub) {
        case 3: VQ_LAUNCH(3); break;
        case 4: VQ_LAUNCH(4); break;
        case 8: VQ_LAUNCH(8); break;
        case 16: VQ_LAUNCH(16); break;
        case 32: VQ_LAUNCH(32); break;
        case 64: VQ_LAUNCH(64); break;
        default: VQ_LAUNCH(0); break;
    }
#undef VQ_LAUNCH
    VQ_CHECK_LAUNCH("vq_fwd");
    return VQCPC_OK;
}

int64_t vqcpc_vq_bwd_workspace(int64_t R, int ncb, int K, int dsub) {
    return ceil_div(std::max<int64_t>(R, 1), kVqThreads) * ncb * K * dsub * (int64_t)sizeof(float);
}

int vqcpc_vq_bwd(const float* z, const float* codebooks, const int64_t* idx, const float* g_zq, const float* g_loss,
                 int64_t R, int ncb, int K, int dsub, float beta, int squared, float* d_z, float* d_codebooks,
                 void* workspace, int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(z && codebooks && idx && g_zq && g_loss && d_z && d_codebooks && workspace, "vq_bwd: null pointer");
    VQ_REQUIRE(R >= 1 && ncb >= 1 && K >= 1 && dsub >= 1 && dsub <= kVqThreads, "vq_bwd: bad shape");
    const size_t lds = ((size_t)K * dsub + (size_t)kVqThreads * dsub + kVqThreads) * sizeof(float);
    VQ_REQUIRE(lds <= 160 * 1024, "vq_bwd: codebook does not fit the LDS");
    if (workspace_bytes < vqcpc_vq_bwd_workspace(R, ncb, K, dsub)) {
        set_error("vq_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const int nchunks = (int)ceil_div(R, kVqThreads);
    hipStream_t s = (hipStream_t)stream;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)vq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(vq_bwd_kernel, dim3(nchunks), dim3(kVqThreads), lds, s, z, codebooks, idx, g_zq, g_loss, R, ncb, K,
                       dsub, beta, squared, d_z, (float*)workspace);
    VQ_CHECK_LAUNCH("vq_bwd");
    const int64_t count = (int64_t)ncb * K * dsub;
    return launch_reduce_splits((const float*)workspace, count, nchunks, d_codebooks, count, 0, s);
}

}  // extern "C"
