// Product vector quantiser: LDS-staged codebook, four lanes per row, canonical (non-FMA, t-ascending, k-ascending) distances.
// Compiled with -ffp-contract=off; the distance uses explicitly rounded __fsub_rn/__fmul_rn/__fadd_rn so that the
// argmin is bit-identical to oracle/vqcpc_oracle.py:vq_distances_canonical (reference: vector_quantizer.py:105-116).
#include "common.h"

namespace vq {

constexpr int kVqThreads = 256;
constexpr int kVqLdsFloats = 36 * 1024;  // 144 KiB of codebook per workgroup at most

// ---------------------------------------------------------------------------------------------------------------------
// forward.  kVqLpr = 4 lanes share one row (64 rows per workgroup: 544 workgroups at C1's 34 816 rows, 2 waves per SIMD on
// every CU; one row per lane gave 136 workgroups on 256 CUs).  Lane j of a row scans the codes k = j, j + 4, j + 8, ...
// of the LDS-staged codebook -- the four lanes of a row read four consecutive codes (64 bytes apart at dsub = 16:
// distinct banks) and the 16 rows of a wave read the same addresses (broadcast) -- and the four partial results meet in a
// two-step shuffle.  Every distance is the canonical chain (t ascending, separately rounded sub / mul / add); a lane keeps
// the FIRST minimum of its ascending code list (strict '<') and the merge prefers the smaller distance, then the smaller
// index: exactly the sequential k-ascending scan with strict '<' (ties -> first index; a NaN distance never replaces,
// and a NaN at k = 0 stays, as in the sequential scan).  The z rows of a wave are 16 consecutive rows = one contiguous
// 16 * D * 4-byte stretch, read as float4 (the four lanes of a row read the same 64 bytes).
// DSUB > 0: sub-vector kept in registers; DSUB == 0: runtime dsub, sub-vector re-read from global (L1) per code.
constexpr int kVqLpr = 4;                          // lanes per row
constexpr int kVqRows = kVqThreads / kVqLpr;       // rows per workgroup

template <int DSUB>
__global__ __launch_bounds__(kVqThreads) void vq_fwd_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                            int64_t R, int ncb, int K, int dsub_rt, float beta,
                                                            int squared, int assign, int64_t* __restrict__ idx_out,
                                                            float* __restrict__ zq_out, float* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int dsub = DSUB > 0 ? DSUB : dsub_rt;
    const int D = ncb * dsub;
    const int j = threadIdx.x & (kVqLpr - 1);
    const int64_t r = (int64_t)blockIdx.x * kVqRows + (threadIdx.x >> 2);
    const bool live = r < R;
    const float* zr = z + (live ? r : 0) * D;
    float lsum = 0.0f;   // sum over the whole row of (q - z)^2   or  ((q - z) + eps)^2   (lane 0 of the row)
    const float kInf = __builtin_inff();

    for (int c = 0; c < ncb; ++c) {
        __syncthreads();
        const float* src = cb + (int64_t)c * K * dsub;
        if ((K * dsub) % 4 == 0) {
            for (int i = threadIdx.x * 4; i < K * dsub; i += kVqThreads * 4)
                *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(src + i);
        } else {
            for (int i = threadIdx.x; i < K * dsub; i += kVqThreads) lds[i] = src[i];
        }
        __syncthreads();

        float x[DSUB > 0 ? DSUB : 1];
        if (DSUB > 0) {
            if (DSUB % 4 == 0) {
#pragma unroll
                for (int t = 0; t < DSUB; t += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(zr + c * DSUB + t);
                    x[t] = v.x, x[t + 1] = v.y, x[t + 2] = v.z, x[t + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < DSUB; ++t) x[t] = zr[c * DSUB + t];
            }
        }
        // lane-local scan of the codes j, j + 4, ...: lane 0 starts from code 0 unconditionally (the sequential scan's
        // k == 0), the others from "nothing yet" = (+inf, -1): +inf never replaces anything in the sequential scan either
        float best = kInf;
        int bi = -1;
        if (assign) {
            int k = j;
            if (DSUB > 0) {
                // four codes at a time: four independent canonical chains whose latencies overlap; candidates are compared
                // in code order with the same strict '<'
                constexpr int KU = 4;
                for (; k + (KU - 1) * kVqLpr < K; k += KU * kVqLpr) {
                    float dd[KU];
#pragma unroll
                    for (int u = 0; u < KU; ++u) dd[u] = 0.0f;
#pragma unroll
                    for (int t = 0; t < DSUB; ++t) {
#pragma unroll
                        for (int u = 0; u < KU; ++u) {
                            const float df = __fsub_rn(x[t], lds[(k + u * kVqLpr) * DSUB + t]);
                            dd[u] = __fadd_rn(dd[u], __fmul_rn(df, df));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        const int kk = k + u * kVqLpr;
                        if (kk == 0 || dd[u] < best) {
                            best = dd[u];
                            bi = kk;
                        }
                    }
                }
            }
            for (; k < K; k += kVqLpr) {
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
            // merge the four lanes of the row: smaller distance, then smaller index; "nothing yet" never wins; a NaN kept
            // from code 0 (bi == 0 in lane 0) is never replaced because every comparison with it is false
#pragma unroll
            for (int o = 1; o < kVqLpr; o <<= 1) {
                const float od = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                const bool take = oi >= 0 && (bi < 0 || od < best || (od == best && oi < bi));
                // a NaN at code 0 must survive: it sits in the lane that holds bi == 0, and (x < NaN), (x == NaN) are false
                // there; the OTHER lanes see od = NaN, oi = 0 and must adopt it: sequentially nothing ever replaces it
                const bool other_is_stuck_nan = oi == 0 && od != od;
                const bool mine_is_stuck_nan = bi == 0 && best != best;
                if (!mine_is_stuck_nan && (take || other_is_stuck_nan)) {
                    best = od;
                    bi = oi;
                }
            }
        } else {
            bi = live ? (int)idx_out[r * ncb + c] : 0;       // assign == 0: indices are given (label corruption path)
        }
        if (!live || j != 0) continue;                        // one lane per row finishes the row
        if (assign) idx_out[r * ncb + c] = (int64_t)bi;
        if (!zq_out) continue;                                // index-only (inference) mode
        const float* q = lds + bi * dsub;
        for (int t = 0; t < dsub; ++t) {
            const float xv = zr[c * dsub + t];
            const float diff = __fsub_rn(q[t], xv);                 // (quantized - inputs)
            zq_out[r * D + c * dsub + t] = __fadd_rn(xv, diff);     // inputs + (quantized - inputs).detach()
            const float v = squared ? diff : __fadd_rn(diff, 1e-5f);
            lsum = __fadd_rn(lsum, __fmul_rn(v, v));
        }
    }
    if (live && j == 0 && loss_out) {
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
            // k % kgroups per row is a scalar division (8 600 scalar instructions per wave at 256 rows x 2 codebooks:
            // profiles/r03_pmc_kernels_c1.txt); kgroups is a power of two for every dsub that divides 256
            const bool pow2 = (kgroups & (kgroups - 1)) == 0;
            const int kmask = kgroups - 1;
            for (int row = 0; row < rows_here; ++row) {
                const int k = sidx[row];
                const int kg = pow2 ? (k & kmask) : (k % kgroups);
                if (kg == my_g) acc[k * dsub + my_t] += val[row * dsub + my_t];
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
    VQ_REQUIRE(aligned16(z) && aligned16(codebooks), "vq_fwd: z and codebooks must be 16-byte aligned");
    const dim3 grid((unsigned)ceil_div(R, kVqRows)), block(kVqThreads);
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
