// Query-subsampled relative attention for the LAST layer of a downscaler stack.
//
// `output[::f]` (relative_transformer_downscaler.py:125) keeps only positions 0, f, 2f, ... of the last layer's output,
// and everything after the attention of that layer is per-token (out-proj, LayerNorm, FFN).  So only the queries at
// positions i = f*iq are needed, while keys / values still cover the whole block.  Results are identical to computing
// the full layer and discarding rows; the dropped rows receive zero gradient in the reference as well.
//   q   [n_blocks*LQ][ldq]   (projected from x[::f]; UNSCALED),  LQ = L / F
//   kv  [n_blocks*L ][ldkv]  k | v at column offsets 0, d
// One problem = (block, head); 4*LQ lanes per problem; one wavefront per workgroup (PPW = 64 / (4 LQ) problems).
#include "common.h"

namespace vq {

constexpr int kSubPad = 4;

template <int L, int HD, int F>
struct SubCfg {
    static constexpr int LQ = L / F;
    static constexpr int LPP = 4 * LQ;           // lanes per problem
    static constexpr int PPW = 64 / LPP;         // problems per wave (= per workgroup)
    static constexpr int JPL = L / 4;            // scores per lane
    static constexpr int CPL = HD / 4;           // output columns per lane
    static constexpr int RS = HD + kSubPad;
    static constexpr int NE = 2 * L - 1;
    static constexpr int NER = (NE + LQ - 1) / LQ;   // Erel rows per lane in the dE accumulation
    static constexpr int FWD_FLOATS = (LQ + 2 * L + NE) * RS + LQ * (L + 1);
    static constexpr int BWD_FLOATS = (2 * LQ + 2 * L + NE) * RS + 2 * LQ * (L + 1);
};

template <int ROWS, int HD, int LPP>
__device__ __forceinline__ void sub_stage(float* dst, const float* __restrict__ src, int64_t ld, int sl, float mul) {
    constexpr int RS = HD + kSubPad, V = HD / 4;
    for (int e = sl; e < ROWS * V; e += LPP) {
        const int row = e / V, c4 = e % V;
        float4 v = *reinterpret_cast<const float4*>(src + row * ld + c4 * 4);
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(dst + row * RS + c4 * 4) = v;
    }
}

// LDS regions private to one wavefront need no s_barrier: LDS instructions of a wave execute in order, only the compiler
// must keep the program order of the accesses around the exchange
__device__ __forceinline__ void sub_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// register-staged variant of sub_stage for the backward's block loop (see relattn.hip: fetch_rows / commit_rows)
template <int ROWS, int HD, int LPP>
struct SubRegs {
    static constexpr int V = HD / 4, N = (ROWS * V + LPP - 1) / LPP;
    float4 r[N];
};

template <int ROWS, int HD, int LPP>
__device__ __forceinline__ void sub_fetch(SubRegs<ROWS, HD, LPP>& t, const float* __restrict__ src, int64_t ld, int sl) {
    constexpr int V = HD / 4;
#pragma unroll
    for (int k = 0; k < SubRegs<ROWS, HD, LPP>::N; ++k) {
        const int e = min(sl + k * LPP, ROWS * V - 1);                  // clamped: no branch around the load
        t.r[k] = *reinterpret_cast<const float4*>(src + (e / V) * ld + (e % V) * 4);
    }
}

template <int ROWS, int HD, int LPP>
__device__ __forceinline__ void sub_commit(float* dst, const SubRegs<ROWS, HD, LPP>& t, int sl, float mul) {
    constexpr int RS = HD + kSubPad, V = HD / 4;
#pragma unroll
    for (int k = 0; k < SubRegs<ROWS, HD, LPP>::N; ++k) {
        const int e = sl + k * LPP;
        if (e < ROWS * V) {
            float4 v = t.r[k];
            v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
            *reinterpret_cast<float4*>(dst + (e / V) * RS + (e % V) * 4) = v;
        }
    }
}

template <int L, int HD, int LPP>
__device__ __forceinline__ void sub_stage_erel(float* dst, const float* __restrict__ e1, const float* __restrict__ e2,
                                               int h, int sl) {
    constexpr int RS = HD + kSubPad, V = HD / 4, NE = 2 * L - 1;
    for (int e = sl; e < NE * V; e += LPP) {
        const int r = e / V, c4 = e % V;
        const float* src = r < L ? e1 + ((int64_t)h * L + r) * HD : e2 + ((int64_t)h * L + (r - L + 1)) * HD;
        *reinterpret_cast<float4*>(dst + r * RS + c4 * 4) = *reinterpret_cast<const float4*>(src + c4 * 4);
    }
}

// =====================================================================================================================
template <int L, int HD, int F>
__global__ __launch_bounds__(64) void relattn_sub_fwd_kernel(const float* __restrict__ q, int64_t ldq,
                                                             const float* __restrict__ kv, int64_t ldkv,
                                                             const float* __restrict__ e1, const float* __restrict__ e2,
                                                             float* __restrict__ ctx, int64_t ldo,
                                                             float* __restrict__ probs, int64_t n_blocks, int H,
                                                             float scale, uint32_t thr, float inv_keep, uint64_t seed, int o16) {
    // Persistent slots: slot g of the grid keeps head g % H (the host makes the slot count a multiple of H) and walks the
    // blocks g / H, + slots / H, ...: the relative rows are staged once, the q / k / v rows of the next block are fetched
    // into registers while the current one is processed.  A slot's LDS region is private to its 4 LQ lanes of one wave.
    using C = SubCfg<L, HD, F>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int sl = lane % C::LPP, slot = lane / C::LPP;
    float* Qs = lds + slot * C::FWD_FLOATS;
    float* Ks = Qs + C::LQ * C::RS;
    float* Vs = Ks + L * C::RS;
    float* Er = Vs + L * C::RS;
    float* Ps = Er + C::NE * C::RS;                 // [LQ][L+1]
    const int d = H * HD;
    const int64_t slots = (int64_t)gridDim.x * C::PPW, gs = (int64_t)blockIdx.x * C::PPW + slot;
    const int h = (int)(gs % H);
    const int64_t n0 = gs / H, nstep = slots / H;
    const int iq = sl >> 2, jg = sl & 3;
    const int i = iq * F;                           // absolute position of this query inside the block
    sub_stage_erel<L, HD, C::LPP>(Er, e1, e2, h, sl);
    SubRegs<C::LQ, HD, C::LPP> rq;
    SubRegs<L, HD, C::LPP> rk, rv;
    auto prefetch = [&](int64_t nb) {
        const int64_t nc = min(nb, n_blocks - 1);                    // past the end: re-read the last block, never used
        sub_fetch<C::LQ, HD, C::LPP>(rq, q + nc * C::LQ * ldq + h * HD, ldq, sl);
        sub_fetch<L, HD, C::LPP>(rk, kv + nc * L * ldkv + h * HD, ldkv, sl);
        sub_fetch<L, HD, C::LPP>(rv, kv + nc * L * ldkv + d + h * HD, ldkv, sl);
    };
    prefetch(n0);
    const int64_t iters = (n_blocks + nstep - 1) / nstep;            // uniform trip count; slots past the end idle (live)
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t n = n0 + it * nstep;
        const bool live = n < n_blocks;
        const int64_t prob = n * H + h;
        sub_wave_fence();                                            // the previous block's LDS readers are done
        if (live) {
            sub_commit<C::LQ, HD, C::LPP>(Qs, rq, sl, scale);
            sub_commit<L, HD, C::LPP>(Ks, rk, sl, 1.0f);
            sub_commit<L, HD, C::LPP>(Vs, rv, sl, 1.0f);
        }
        sub_wave_fence();
        prefetch(n + nstep);
        if (live) {
            float s[C::JPL];
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) s[jj] = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                const float4 qv = *reinterpret_cast<const float4*>(Qs + iq * C::RS + c4 * 4);
#pragma unroll
                for (int jj = 0; jj < C::JPL; ++jj) {
                    const int j = jj * 4 + jg;
                    const float4 k = *reinterpret_cast<const float4*>(Ks + j * C::RS + c4 * 4);
                    const float4 e = *reinterpret_cast<const float4*>(Er + (j - i + L - 1) * C::RS + c4 * 4);
                    s[jj] += qv.x * (k.x + e.x) + qv.y * (k.y + e.y) + qv.z * (k.z + e.z) + qv.w * (k.w + e.w);
                }
            }
            float m = s[0];
#pragma unroll
            for (int jj = 1; jj < C::JPL; ++jj) m = fmaxf(m, s[jj]);
            m = fmaxf(m, __shfl_xor(m, 1, 64));
            m = fmaxf(m, __shfl_xor(m, 2, 64));
            float sum = 0.0f;
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) {
                s[jj] = __expf(s[jj] - m);
                sum += s[jj];
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            const float inv = 1.0f / sum;
            const int64_t pbase = (prob * C::LQ + iq) * L;
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) {
                const int j = jj * 4 + jg;
                const float p = s[jj] * inv;
                probs[pbase + j] = p;
                Ps[iq * (L + 1) + j] = p * drop_scale(seed, (uint64_t)(pbase + j), thr, inv_keep);
            }
        }
        sub_wave_fence();
        if (live) {
            float o[C::CPL];
#pragma unroll
            for (int c = 0; c < C::CPL; ++c) o[c] = 0.0f;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const float p = Ps[iq * (L + 1) + j];
#pragma unroll
                for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                    const float4 v = *reinterpret_cast<const float4*>(Vs + j * C::RS + jg * C::CPL + c4 * 4);
                    o[c4 * 4 + 0] += p * v.x; o[c4 * 4 + 1] += p * v.y; o[c4 * 4 + 2] += p * v.z; o[c4 * 4 + 3] += p * v.w;
                }
            }
            const int64_t oo = (n * C::LQ + iq) * ldo + h * HD + jg * C::CPL;
#pragma unroll
            for (int c4 = 0; c4 < C::CPL / 4; ++c4)
                store4_out(ctx, oo + c4 * 4, o[c4 * 4], o[c4 * 4 + 1], o[c4 * 4 + 2], o[c4 * 4 + 3], o16);
        }
    }
}

// =====================================================================================================================
// backward: grid = (chunks, head groups); a slot keeps its head for the whole loop (dErel accumulates in registers).
// ws layout: [chunk][nsub][H][2L-1][HD]
template <int L, int HD, int F>
__global__ __launch_bounds__(64) void relattn_sub_bwd_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ q, int64_t ldq, const float* __restrict__ kv,
    int64_t ldkv, const float* __restrict__ probs, const float* __restrict__ e1, const float* __restrict__ e2,
    float* __restrict__ d_q, int64_t ldgq, float* __restrict__ d_kv, int64_t ldgkv, float* __restrict__ ws,
    int64_t n_blocks, int H, int blocks_per_wg, float scale, uint32_t thr, float inv_keep, uint64_t seed, int g16) {
    using C = SubCfg<L, HD, F>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int sl = lane % C::LPP, slot = lane / C::LPP;
    float* Qs = lds + slot * C::BWD_FLOATS;
    float* Os = Qs + C::LQ * C::RS;                 // d_ctx rows [LQ]
    float* Ks = Os + C::LQ * C::RS;
    float* Vs = Ks + L * C::RS;
    float* Er = Vs + L * C::RS;
    float* Ss = Er + C::NE * C::RS;                 // dS [LQ][L+1]
    float* Ps = Ss + C::LQ * (L + 1);               // P after dropout [LQ][L+1]
    const int d = H * HD;
    int h, nsub, NS;
    if (C::PPW >= H) {
        NS = C::PPW / H;
        h = slot % H;
        nsub = slot / H;
    } else {
        NS = 1;
        h = blockIdx.y * C::PPW + slot;
        nsub = 0;
    }
    const int iq = sl >> 2, jg = sl & 3;
    const int i = iq * F;
    sub_stage_erel<L, HD, C::LPP>(Er, e1, e2, h, sl);
    float de[C::NER][C::CPL];
#pragma unroll
    for (int a = 0; a < C::NER; ++a)
#pragma unroll
        for (int c = 0; c < C::CPL; ++c) de[a][c] = 0.0f;

    const int64_t n_begin = (int64_t)blockIdx.x * blocks_per_wg;
    // software pipeline over the blocks (barrier-lockstep loop): next block's operands in flight during this block
    constexpr bool kPipe = HD <= 32 && L == 4;
    SubRegs<C::LQ, HD, C::LPP> rq, ro;
    SubRegs<L, HD, C::LPP> rk, rv;
    auto prefetch = [&](int64_t nb) {
        if constexpr (kPipe) {
            const int64_t nc = min(nb, n_blocks - 1);                // past the end: re-read the last block, never used
            sub_fetch<C::LQ, HD, C::LPP>(rq, q + nc * C::LQ * ldq + h * HD, ldq, sl);
            sub_fetch<C::LQ, HD, C::LPP>(ro, d_ctx + nc * C::LQ * ldo + h * HD, ldo, sl);
            sub_fetch<L, HD, C::LPP>(rk, kv + nc * L * ldkv + h * HD, ldkv, sl);
            sub_fetch<L, HD, C::LPP>(rv, kv + nc * L * ldkv + d + h * HD, ldkv, sl);
        }
    };
    prefetch(n_begin + nsub);
    for (int it = 0; it < blocks_per_wg; it += NS) {
        const int64_t n = n_begin + it + nsub;
        const bool live = n < n_blocks;
        const int64_t prob = n * H + h;
        __syncthreads();
        if (live) {
            if constexpr (kPipe) {
                sub_commit<C::LQ, HD, C::LPP>(Qs, rq, sl, scale);
                sub_commit<C::LQ, HD, C::LPP>(Os, ro, sl, 1.0f);
                sub_commit<L, HD, C::LPP>(Ks, rk, sl, 1.0f);
                sub_commit<L, HD, C::LPP>(Vs, rv, sl, 1.0f);
            } else {
                sub_stage<C::LQ, HD, C::LPP>(Qs, q + n * C::LQ * ldq + h * HD, ldq, sl, scale);
                sub_stage<C::LQ, HD, C::LPP>(Os, d_ctx + n * C::LQ * ldo + h * HD, ldo, sl, 1.0f);
                sub_stage<L, HD, C::LPP>(Ks, kv + n * L * ldkv + h * HD, ldkv, sl, 1.0f);
                sub_stage<L, HD, C::LPP>(Vs, kv + n * L * ldkv + d + h * HD, ldkv, sl, 1.0f);
            }
        }
        __syncthreads();
        prefetch(n + NS);
        if (live) {
            float dp[C::JPL], p[C::JPL];
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) dp[jj] = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                const float4 o = *reinterpret_cast<const float4*>(Os + iq * C::RS + c4 * 4);
#pragma unroll
                for (int jj = 0; jj < C::JPL; ++jj) {
                    const float4 v = *reinterpret_cast<const float4*>(Vs + (jj * 4 + jg) * C::RS + c4 * 4);
                    dp[jj] += o.x * v.x + o.y * v.y + o.z * v.z + o.w * v.w;
                }
            }
            const int64_t pbase = (prob * C::LQ + iq) * L;
            float rowdot = 0.0f;
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) {
                const int j = jj * 4 + jg;
                p[jj] = probs[pbase + j];
                const float mk = drop_scale(seed, (uint64_t)(pbase + j), thr, inv_keep);
                dp[jj] *= mk;
                Ps[iq * (L + 1) + j] = p[jj] * mk;
                rowdot += dp[jj] * p[jj];
            }
            rowdot += __shfl_xor(rowdot, 1, 64);
            rowdot += __shfl_xor(rowdot, 2, 64);
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) Ss[iq * (L + 1) + jj * 4 + jg] = p[jj] * (dp[jj] - rowdot);
        }
        __syncthreads();
        if (live) {
            // dK / dV for key rows j = iq + LQ*u
#pragma unroll
            for (int u = 0; u < F; ++u) {
                const int j = iq + C::LQ * u;
                float dk[C::CPL], dv[C::CPL];
#pragma unroll
                for (int c = 0; c < C::CPL; ++c) dk[c] = dv[c] = 0.0f;
#pragma unroll
                for (int ii = 0; ii < C::LQ; ++ii) {
                    const float pd = Ps[ii * (L + 1) + j], ds = Ss[ii * (L + 1) + j];
#pragma unroll
                    for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                        const float4 o = *reinterpret_cast<const float4*>(Os + ii * C::RS + jg * C::CPL + c4 * 4);
                        const float4 qq = *reinterpret_cast<const float4*>(Qs + ii * C::RS + jg * C::CPL + c4 * 4);
                        dv[c4 * 4 + 0] += pd * o.x; dv[c4 * 4 + 1] += pd * o.y; dv[c4 * 4 + 2] += pd * o.z; dv[c4 * 4 + 3] += pd * o.w;
                        dk[c4 * 4 + 0] += ds * qq.x; dk[c4 * 4 + 1] += ds * qq.y; dk[c4 * 4 + 2] += ds * qq.z; dk[c4 * 4 + 3] += ds * qq.w;
                    }
                }
                const int64_t go = (n * L + j) * ldgkv + h * HD + jg * C::CPL;
#pragma unroll
                for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                    store4_out(d_kv, go + c4 * 4, dk[c4 * 4], dk[c4 * 4 + 1], dk[c4 * 4 + 2], dk[c4 * 4 + 3], g16 & 1);
                    store4_out(d_kv, go + d + c4 * 4, dv[c4 * 4], dv[c4 * 4 + 1], dv[c4 * 4 + 2], dv[c4 * 4 + 3], g16 & 1);
                }
            }
            // dq_iq = scale * sum_j dS[iq][j] (k_j + Erel[j - i + L - 1])
            float dq[C::CPL];
#pragma unroll
            for (int c = 0; c < C::CPL; ++c) dq[c] = 0.0f;
#pragma unroll
            for (int jj = 0; jj < L; ++jj) {
                const float ds = Ss[iq * (L + 1) + jj];
#pragma unroll
                for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                    const float4 k = *reinterpret_cast<const float4*>(Ks + jj * C::RS + jg * C::CPL + c4 * 4);
                    const float4 e = *reinterpret_cast<const float4*>(Er + (jj - i + L - 1) * C::RS + jg * C::CPL + c4 * 4);
                    dq[c4 * 4 + 0] += ds * (k.x + e.x); dq[c4 * 4 + 1] += ds * (k.y + e.y);
                    dq[c4 * 4 + 2] += ds * (k.z + e.z); dq[c4 * 4 + 3] += ds * (k.w + e.w);
                }
            }
            const int64_t gqo = (n * C::LQ + iq) * ldgq + h * HD + jg * C::CPL;
#pragma unroll
            for (int c4 = 0; c4 < C::CPL / 4; ++c4)
                store4_out(d_q, gqo + c4 * 4, dq[c4 * 4] * scale, dq[c4 * 4 + 1] * scale, dq[c4 * 4 + 2] * scale,
                           dq[c4 * 4 + 3] * scale, g16 & 2);
            // dErel[r] += sum_{i'} dS[i'][jx] * qs[i'],  jx = F*i' + r - (L-1);  this lane owns rows r = iq + LQ*a
#pragma unroll
            for (int a = 0; a < C::NER; ++a) {
                const int r = iq + C::LQ * a;
                if (r < C::NE) {
#pragma unroll
                    for (int ii = 0; ii < C::LQ; ++ii) {
                        const int jx = F * ii + r - (L - 1);
                        if (jx >= 0 && jx < L) {
                            const float ds = Ss[ii * (L + 1) + jx];
#pragma unroll
                            for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                                const float4 qq = *reinterpret_cast<const float4*>(Qs + ii * C::RS + jg * C::CPL + c4 * 4);
                                de[a][c4 * 4 + 0] += ds * qq.x; de[a][c4 * 4 + 1] += ds * qq.y;
                                de[a][c4 * 4 + 2] += ds * qq.z; de[a][c4 * 4 + 3] += ds * qq.w;
                            }
                        }
                    }
                }
            }
        }
    }
    float* dst = ws + ((((int64_t)blockIdx.x * NS + nsub) * H + h) * C::NE) * HD;
#pragma unroll
    for (int a = 0; a < C::NER; ++a) {
        const int r = iq + C::LQ * a;
        if (r < C::NE) {
#pragma unroll
            for (int c = 0; c < C::CPL; ++c) dst[r * HD + jg * C::CPL + c] = de[a][c];
        }
    }
}

// =====================================================================================================================
// L = 16, F = 4 specialisation: ONE (block, head) problem per wavefront, 4 waves per workgroup.
// The generic kernel above packs 4 problems into one 64-thread workgroup whose LDS footprint (43 KB) leaves 3 waves per
// CU; here a wave owns ~10 KB, so 12-16 waves per CU are resident and every phase uses all 64 lanes:
//   scores / softmax : lane (iq = lane >> 4, j = lane & 15), one score per lane, 16-lane shuffles
//   P.V, dq          : lane (iq, column group of HD/16)
//   dK, dV           : lane (key j = lane & 15, column group of HD/4)
//   dErel            : lane (row pair rr = lane >> 2 -> rows rr, rr + 16; column group of HD/4), accumulated in registers
// =====================================================================================================================
template <int HD>
struct Sub16 {
    static constexpr int L = 16, F = 4, LQ = 4, NE = 31;
    static constexpr int RS = HD + kSubPad;
    static constexpr int C16 = HD / 16;          // columns per lane when 16 lanes span a row
    static constexpr int C4 = HD / 4;            // columns per lane when 4 lanes span a row
    static constexpr int FWD_FLOATS = (LQ + 2 * L + NE) * RS + LQ * (L + 1);
    static constexpr int BWD_FLOATS = (2 * LQ + 2 * L + NE) * RS + 2 * LQ * (L + 1);
};

template <int ROWS, int HD>
__device__ __forceinline__ void stage64(float* dst, const float* __restrict__ src, int64_t ld, int lane, float mul) {
    constexpr int RS = HD + kSubPad, V = HD / 4;
    for (int e = lane; e < ROWS * V; e += 64) {
        const int row = e / V, c4 = e % V;
        float4 v = *reinterpret_cast<const float4*>(src + row * ld + c4 * 4);
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(dst + row * RS + c4 * 4) = v;
    }
}

// register-staged variant for the block loops: `fetch64` issues the global loads of the NEXT block before the current one
// is processed, `commit64` writes them to LDS after the barrier that retires the current block's readers
template <int ROWS, int HD>
struct Stage64Regs {
    static constexpr int V = HD / 4, N = (ROWS * V + 63) / 64;
    float4 r[N];
};

template <int ROWS, int HD>
__device__ __forceinline__ void fetch64(Stage64Regs<ROWS, HD>& t, const float* __restrict__ src, int64_t ld, int lane) {
    constexpr int V = HD / 4;
#pragma unroll
    for (int k = 0; k < Stage64Regs<ROWS, HD>::N; ++k) {
        const int e = min(lane + 64 * k, ROWS * V - 1);              // clamped: no branch around the load
        t.r[k] = *reinterpret_cast<const float4*>(src + (e / V) * ld + (e % V) * 4);
    }
}

template <int ROWS, int HD>
__device__ __forceinline__ void commit64(float* dst, const Stage64Regs<ROWS, HD>& t, int lane, float mul) {
    constexpr int RS = HD + kSubPad, V = HD / 4;
#pragma unroll
    for (int k = 0; k < Stage64Regs<ROWS, HD>::N; ++k) {
        const int e = lane + 64 * k;
        if (e < ROWS * V) {
            float4 v = t.r[k];
            v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
            *reinterpret_cast<float4*>(dst + (e / V) * RS + (e % V) * 4) = v;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void relattn_sub16_fwd_kernel(const float* __restrict__ q, int64_t ldq,
                                                                const float* __restrict__ kv, int64_t ldkv,
                                                                const float* __restrict__ e1, const float* __restrict__ e2,
                                                                float* __restrict__ ctx, int64_t ldo,
                                                                float* __restrict__ probs, int64_t n_blocks, int H,
                                                                float scale, uint32_t thr, float inv_keep, uint64_t seed, int o16) {
    // Persistent workgroups: wave w of the grid keeps head w % H (the host makes the wave count a multiple of H), so the 31
    // relative rows -- 45 % of the bytes a problem stages -- go to LDS once, and the q / k / v rows of the wave's next block
    // are fetched into registers while the current one is processed.  The LDS regions are private to a wavefront: only
    // wave-level fences, no workgroup barrier (waves may run different trip counts).
    using C = Sub16<HD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Qs = lds + wave * C::FWD_FLOATS;
    float* Ks = Qs + C::LQ * C::RS;
    float* Vs = Ks + C::L * C::RS;
    float* Er = Vs + C::L * C::RS;
    float* Ps = Er + C::NE * C::RS;                // [4][17]
    const int d = H * HD;
    const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    const int h = (int)(w % H);
    const int64_t n0 = w / H, nstep = nw / H;
    const int iq = lane >> 4, j = lane & 15;
    if (n0 >= n_blocks) return;
    for (int e = lane; e < C::NE * (HD / 4); e += 64) {
        const int r = e / (HD / 4), c4 = e % (HD / 4);
        const float* src = r < C::L ? e1 + ((int64_t)h * C::L + r) * HD : e2 + ((int64_t)h * C::L + (r - C::L + 1)) * HD;
        *reinterpret_cast<float4*>(Er + r * C::RS + c4 * 4) = *reinterpret_cast<const float4*>(src + c4 * 4);
    }
    Stage64Regs<C::LQ, HD> rq;
    Stage64Regs<C::L, HD> rk, rv;
    auto prefetch = [&](int64_t nb) {
        const int64_t nc = min(nb, n_blocks - 1);                    // past the end: re-read the last block, never used
        fetch64<C::LQ, HD>(rq, q + nc * C::LQ * ldq + h * HD, ldq, lane);
        fetch64<C::L, HD>(rk, kv + nc * C::L * ldkv + h * HD, ldkv, lane);
        fetch64<C::L, HD>(rv, kv + nc * C::L * ldkv + d + h * HD, ldkv, lane);
    };
    prefetch(n0);
    for (int64_t n = n0; n < n_blocks; n += nstep) {
        const int64_t prob = n * H + h;
        sub_wave_fence();                                            // the previous block's LDS readers are done
        commit64<C::LQ, HD>(Qs, rq, lane, scale);
        commit64<C::L, HD>(Ks, rk, lane, 1.0f);
        commit64<C::L, HD>(Vs, rv, lane, 1.0f);
        sub_wave_fence();
        prefetch(n + nstep);
        {
            const float* er = Er + (j - C::F * iq + C::L - 1) * C::RS;
            float s = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                const float4 qv = *reinterpret_cast<const float4*>(Qs + iq * C::RS + c4 * 4);
                const float4 k = *reinterpret_cast<const float4*>(Ks + j * C::RS + c4 * 4);
                const float4 e = *reinterpret_cast<const float4*>(er + c4 * 4);
                s += qv.x * (k.x + e.x) + qv.y * (k.y + e.y) + qv.z * (k.z + e.z) + qv.w * (k.w + e.w);
            }
            float m = s;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float ex = __expf(s - m);
            float sum = ex;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
            const float p = ex / sum;
            const int64_t pidx = prob * 64 + lane;                       // probs[prob][iq][j]: fully coalesced
            probs[pidx] = p;
            Ps[iq * (C::L + 1) + j] = p * drop_scale(seed, (uint64_t)pidx, thr, inv_keep);
        }
        sub_wave_fence();
        {
            float o[C::C16];
#pragma unroll
            for (int c = 0; c < C::C16; ++c) o[c] = 0.0f;
#pragma unroll
            for (int jj = 0; jj < C::L; ++jj) {
                const float p = Ps[iq * (C::L + 1) + jj];
#pragma unroll
                for (int c = 0; c < C::C16; ++c) o[c] += p * Vs[jj * C::RS + j * C::C16 + c];
            }
            const int64_t oo = (n * C::LQ + iq) * ldo + h * HD + j * C::C16;
            if constexpr (C::C16 % 4 == 0) {
#pragma unroll
                for (int c = 0; c < C::C16; c += 4) store4_out(ctx, oo + c, o[c], o[(c + 1) % C::C16], o[(c + 2) % C::C16], o[(c + 3) % C::C16], o16);
            } else {
#pragma unroll
                for (int c = 0; c < C::C16; ++c) store1_out(ctx, oo + c, o[c], o16);
            }
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void relattn_sub16_bwd_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ q, int64_t ldq, const float* __restrict__ kv,
    int64_t ldkv, const float* __restrict__ probs, const float* __restrict__ e1, const float* __restrict__ e2,
    float* __restrict__ d_q, int64_t ldgq, float* __restrict__ d_kv, int64_t ldgkv, float* __restrict__ ws,
    int64_t n_blocks, int H, int blocks_per_wg, float scale, uint32_t thr, float inv_keep, uint64_t seed, int g16) {
    using C = Sub16<HD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Qs = lds + wave * C::BWD_FLOATS;
    float* Os = Qs + C::LQ * C::RS;
    float* Ks = Os + C::LQ * C::RS;
    float* Vs = Ks + C::L * C::RS;
    float* Er = Vs + C::L * C::RS;
    float* Ss = Er + C::NE * C::RS;                // dS [4][17]
    float* Pd = Ss + C::LQ * (C::L + 1);           // P after dropout [4][17]
    const int d = H * HD;
    int h, nsub, NS;
    if (4 >= H) {
        NS = 4 / H;
        h = wave % H;
        nsub = wave / H;
    } else {
        NS = 1;
        h = blockIdx.y * 4 + wave;
        nsub = 0;
    }
    const int iq = lane >> 4, j = lane & 15;       // score phase
    const int cg4 = lane >> 4;                     // dK/dV phase: key = lane & 15, columns [cg4*C4, +C4)
    const int rr = lane >> 2, cgE = lane & 3;      // dErel phase
    for (int e = lane; e < C::NE * (HD / 4); e += 64) {
        const int r = e / (HD / 4), c4 = e % (HD / 4);
        const float* src = r < C::L ? e1 + ((int64_t)h * C::L + r) * HD : e2 + ((int64_t)h * C::L + (r - C::L + 1)) * HD;
        *reinterpret_cast<float4*>(Er + r * C::RS + c4 * 4) = *reinterpret_cast<const float4*>(src + c4 * 4);
    }
    float de[2][C::C4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < C::C4; ++c) de[a][c] = 0.0f;

    const int64_t n_begin = (int64_t)blockIdx.x * blocks_per_wg;
    // software pipeline over the blocks (790 -> 580 us at C1): the operands of block it + NS are in flight while block it
    // is processed.  Not at head_dim 64: its 24 extra staging registers would halve the occupancy
    constexpr bool kPipe = HD <= 32;
    Stage64Regs<C::LQ, HD> rq, ro;
    Stage64Regs<C::L, HD> rk, rv;
    auto prefetch = [&](int64_t nb) {
        if constexpr (kPipe) {
            const int64_t nc = min(nb, n_blocks - 1);                // past the end: re-read the last block, never used
            fetch64<C::LQ, HD>(rq, q + nc * C::LQ * ldq + h * HD, ldq, lane);
            fetch64<C::LQ, HD>(ro, d_ctx + nc * C::LQ * ldo + h * HD, ldo, lane);
            fetch64<C::L, HD>(rk, kv + nc * C::L * ldkv + h * HD, ldkv, lane);
            fetch64<C::L, HD>(rv, kv + nc * C::L * ldkv + d + h * HD, ldkv, lane);
        }
    };
    prefetch(n_begin + nsub);
    for (int it = 0; it < blocks_per_wg; it += NS) {
        const int64_t n = n_begin + it + nsub;
        const bool live = n < n_blocks;
        const int64_t prob = n * H + h;
        __syncthreads();
        if (live) {
            if constexpr (kPipe) {
                commit64<C::LQ, HD>(Qs, rq, lane, scale);
                commit64<C::LQ, HD>(Os, ro, lane, 1.0f);
                commit64<C::L, HD>(Ks, rk, lane, 1.0f);
                commit64<C::L, HD>(Vs, rv, lane, 1.0f);
            } else {
                stage64<C::LQ, HD>(Qs, q + n * C::LQ * ldq + h * HD, ldq, lane, scale);
                stage64<C::LQ, HD>(Os, d_ctx + n * C::LQ * ldo + h * HD, ldo, lane, 1.0f);
                stage64<C::L, HD>(Ks, kv + n * C::L * ldkv + h * HD, ldkv, lane, 1.0f);
                stage64<C::L, HD>(Vs, kv + n * C::L * ldkv + d + h * HD, ldkv, lane, 1.0f);
            }
        }
        __syncthreads();
        prefetch(n + NS);
        if (live) {
            float dp = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                const float4 o = *reinterpret_cast<const float4*>(Os + iq * C::RS + c4 * 4);
                const float4 v = *reinterpret_cast<const float4*>(Vs + j * C::RS + c4 * 4);
                dp += o.x * v.x + o.y * v.y + o.z * v.z + o.w * v.w;
            }
            const int64_t pidx = prob * 64 + lane;
            const float p = probs[pidx];
            const float mk = drop_scale(seed, (uint64_t)pidx, thr, inv_keep);
            dp *= mk;
            float rowdot = dp * p;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) rowdot += __shfl_xor(rowdot, o, 64);
            Pd[iq * (C::L + 1) + j] = p * mk;
            Ss[iq * (C::L + 1) + j] = p * (dp - rowdot);
        }
        __syncthreads();
        if (live) {
            // dK / dV: key row j, columns [cg4*C4, +C4)
            float dk[C::C4], dv[C::C4];
#pragma unroll
            for (int c = 0; c < C::C4; ++c) dk[c] = dv[c] = 0.0f;
#pragma unroll
            for (int ii = 0; ii < C::LQ; ++ii) {
                const float pd = Pd[ii * (C::L + 1) + j], ds = Ss[ii * (C::L + 1) + j];
#pragma unroll
                for (int c4 = 0; c4 < C::C4 / 4; ++c4) {
                    const float4 o = *reinterpret_cast<const float4*>(Os + ii * C::RS + cg4 * C::C4 + c4 * 4);
                    const float4 qq = *reinterpret_cast<const float4*>(Qs + ii * C::RS + cg4 * C::C4 + c4 * 4);
                    dv[c4 * 4 + 0] += pd * o.x; dv[c4 * 4 + 1] += pd * o.y; dv[c4 * 4 + 2] += pd * o.z; dv[c4 * 4 + 3] += pd * o.w;
                    dk[c4 * 4 + 0] += ds * qq.x; dk[c4 * 4 + 1] += ds * qq.y; dk[c4 * 4 + 2] += ds * qq.z; dk[c4 * 4 + 3] += ds * qq.w;
                }
            }
            const int64_t go = (n * C::L + j) * ldgkv + h * HD + cg4 * C::C4;
#pragma unroll
            for (int c4 = 0; c4 < C::C4 / 4; ++c4) {
                store4_out(d_kv, go + c4 * 4, dk[c4 * 4], dk[c4 * 4 + 1], dk[c4 * 4 + 2], dk[c4 * 4 + 3], g16 & 1);
                store4_out(d_kv, go + d + c4 * 4, dv[c4 * 4], dv[c4 * 4 + 1], dv[c4 * 4 + 2], dv[c4 * 4 + 3], g16 & 1);
            }
            // dq: lane (iq, column group j of C16 columns)
            float dq[C::C16];
#pragma unroll
            for (int c = 0; c < C::C16; ++c) dq[c] = 0.0f;
#pragma unroll
            for (int jj = 0; jj < C::L; ++jj) {
                const float ds = Ss[iq * (C::L + 1) + jj];
                const float* kr = Ks + jj * C::RS + j * C::C16;
                const float* er = Er + (jj - C::F * iq + C::L - 1) * C::RS + j * C::C16;
#pragma unroll
                for (int c = 0; c < C::C16; ++c) dq[c] += ds * (kr[c] + er[c]);
            }
            const int64_t gqo = (n * C::LQ + iq) * ldgq + h * HD + j * C::C16;
            if constexpr (C::C16 % 4 == 0) {
#pragma unroll
                for (int c = 0; c < C::C16; c += 4)
                    store4_out(d_q, gqo + c, dq[c] * scale, dq[(c + 1) % C::C16] * scale, dq[(c + 2) % C::C16] * scale,
                               dq[(c + 3) % C::C16] * scale, g16 & 2);
            } else {
#pragma unroll
                for (int c = 0; c < C::C16; ++c) store1_out(d_q, gqo + c, dq[c] * scale, g16 & 2);
            }
            // dErel rows rr and rr + 16, columns [cgE*C4, +C4):  r = jx - 4 i' + 15
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int r = rr + 16 * a;
                if (r < C::NE) {
#pragma unroll
                    for (int ii = 0; ii < C::LQ; ++ii) {
                        const int jx = r - (C::L - 1) + C::F * ii;
                        if (jx >= 0 && jx < C::L) {
                            const float ds = Ss[ii * (C::L + 1) + jx];
#pragma unroll
                            for (int c4 = 0; c4 < C::C4 / 4; ++c4) {
                                const float4 qq = *reinterpret_cast<const float4*>(Qs + ii * C::RS + cgE * C::C4 + c4 * 4);
                                de[a][c4 * 4 + 0] += ds * qq.x; de[a][c4 * 4 + 1] += ds * qq.y;
                                de[a][c4 * 4 + 2] += ds * qq.z; de[a][c4 * 4 + 3] += ds * qq.w;
                            }
                        }
                    }
                }
            }
        }
    }
    float* dst = ws + ((((int64_t)blockIdx.x * NS + nsub) * H + h) * C::NE) * HD;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int r = rr + 16 * a;
        if (r < C::NE) {
#pragma unroll
            for (int c = 0; c < C::C4; ++c) dst[r * HD + cgE * C::C4 + c] = de[a][c];
        }
    }
}

__global__ __launch_bounds__(256) void relattn_sub_de_split(const float* __restrict__ tot, int H, int L, int HD,
                                                            float* __restrict__ d_e1, float* __restrict__ d_e2) {
    const int NE = 2 * L - 1;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= H * NE * HD) return;
    const int c = o % HD, r = (o / HD) % NE, h = o / (HD * NE);
    const float acc = tot[o];
    if (r < L) d_e1[((int64_t)h * L + r) * HD + c] = acc;
    else d_e2[((int64_t)h * L + (r - L + 1)) * HD + c] = acc;
    if (r == 0) d_e2[((int64_t)h * L) * HD + c] = 0.0f;
}

static int sub_blocks_per_wg(int64_t n_blocks, int ppw, int H) {
    const int ns = std::max(1, ppw / H);
    int64_t b = ceil_div(n_blocks, 4096);
    return (int)round_up(std::max<int64_t>(b, ns), ns);
}

static bool sub_supported(int L, int F, int H, int hd) {
    if (!((L == 16 || L == 4) && F == 4)) return false;
    if (!(hd == 16 || hd == 32 || hd == 64)) return false;
    const int ppw = 64 / (4 * (L / F));
    return H >= 1 && ((ppw % H) == 0 || (H % ppw) == 0);
}

template <int L, int HD, int F>
static int sub_launch_fwd(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const float* e1, const float* e2,
                          float* ctx, int64_t ldo, float* probs, int64_t n_blocks, int H, float drop_p, uint64_t seed,
                          hipStream_t s, int b16) {
    using C = SubCfg<L, HD, F>;
    const size_t lds = (size_t)C::PPW * C::FWD_FLOATS * sizeof(float);
    auto kern = relattn_sub_fwd_kernel<L, HD, F>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // persistent slots (8192 single-wave workgroups at most); the slot count must be a multiple of H
    int64_t grid = std::min<int64_t>(ceil_div(n_blocks * H, C::PPW), 8192);
    int gcd = H, rem = C::PPW;
    while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
    const int64_t unit = H / gcd;
    grid = ceil_div(grid, unit) * unit;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, s, q, ldq, kv, ldkv, e1, e2, ctx, ldo, probs, n_blocks, H,
                       1.0f / sqrtf((float)HD), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, b16);
    VQ_CHECK_LAUNCH("relattn_sub_fwd");
    return VQCPC_OK;
}

template <int L, int HD, int F>
static int sub_launch_bwd(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* kv, int64_t ldkv,
                          const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, float* d_kv,
                          int64_t ldgkv, float* d_e1, float* d_e2, int64_t n_blocks, int H, float drop_p, uint64_t seed,
                          float* ws, hipStream_t s, int b16) {
    using C = SubCfg<L, HD, F>;
    const size_t lds = (size_t)C::PPW * C::BWD_FLOATS * sizeof(float);
    auto kern = relattn_sub_bwd_kernel<L, HD, F>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int bpw = sub_blocks_per_wg(n_blocks, C::PPW, H);
    const int chunks = (int)ceil_div(n_blocks, bpw);
    const int gy = C::PPW >= H ? 1 : H / C::PPW;
    const int NS = C::PPW >= H ? C::PPW / H : 1;
    hipLaunchKernelGGL(kern, dim3(chunks, gy), dim3(64), lds, s, d_ctx, ldo, q, ldq, kv, ldkv, probs, e1, e2, d_q, ldgq,
                       d_kv, ldgkv, ws, n_blocks, H, bpw, 1.0f / sqrtf((float)HD), drop_threshold(drop_p),
                       1.0f / (1.0f - drop_p), seed, b16);
    VQ_CHECK_LAUNCH("relattn_sub_bwd");
    const int total = H * C::NE * HD;
    float* tot = ws + (int64_t)chunks * NS * total;
    int rc = launch_reduce_splits(ws, total, chunks * NS, tot, total, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(relattn_sub_de_split, dim3(ceil_div(total, 256)), dim3(256), 0, s, tot, H, L, HD, d_e1, d_e2);
    VQ_CHECK_LAUNCH("relattn_sub_de_split");
    return VQCPC_OK;
}

static int sub16_blocks_per_wg(int64_t n_blocks, int H) {
    const int ns = std::max(1, 4 / H);
    int64_t b = ceil_div(n_blocks, 4096);
    return (int)round_up(std::max<int64_t>(b, ns), ns);
}

template <int HD>
static int sub16_launch_fwd(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const float* e1, const float* e2,
                            float* ctx, int64_t ldo, float* probs, int64_t n_blocks, int H, float drop_p, uint64_t seed,
                            hipStream_t s, int b16) {
    using C = Sub16<HD>;
    const size_t lds = (size_t)4 * C::FWD_FLOATS * sizeof(float);
    auto kern = relattn_sub16_fwd_kernel<HD>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // persistent waves (about 4 resident workgroups per CU by LDS); the wave count must be a multiple of H
    int64_t grid = std::min<int64_t>(ceil_div(n_blocks * H, 4), 2048);
    int gcd = H, rem = 4;
    while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
    const int64_t unit = H / gcd;
    grid = ceil_div(grid, unit) * unit;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, q, ldq, kv, ldkv, e1, e2, ctx, ldo, probs, n_blocks, H,
                       1.0f / sqrtf((float)HD), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, b16);
    VQ_CHECK_LAUNCH("relattn_sub16_fwd");
    return VQCPC_OK;
}

template <int HD>
static int sub16_launch_bwd(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* kv, int64_t ldkv,
                            const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, float* d_kv,
                            int64_t ldgkv, float* d_e1, float* d_e2, int64_t n_blocks, int H, float drop_p, uint64_t seed,
                            float* ws, hipStream_t s, int b16) {
    using C = Sub16<HD>;
    const size_t lds = (size_t)4 * C::BWD_FLOATS * sizeof(float);
    auto kern = relattn_sub16_bwd_kernel<HD>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int bpw = sub16_blocks_per_wg(n_blocks, H);
    const int chunks = (int)ceil_div(n_blocks, bpw);
    const int gy = 4 >= H ? 1 : H / 4;
    const int NS = 4 >= H ? 4 / H : 1;
    hipLaunchKernelGGL(kern, dim3(chunks, gy), dim3(256), lds, s, d_ctx, ldo, q, ldq, kv, ldkv, probs, e1, e2, d_q, ldgq, d_kv,
                       ldgkv, ws, n_blocks, H, bpw, 1.0f / sqrtf((float)HD), drop_threshold(drop_p), 1.0f / (1.0f - drop_p),
                       seed, b16);
    VQ_CHECK_LAUNCH("relattn_sub16_bwd");
    const int total = H * C::NE * HD;
    float* tot = ws + (int64_t)chunks * NS * total;
    int rc = launch_reduce_splits(ws, total, chunks * NS, tot, total, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(relattn_sub_de_split, dim3(ceil_div(total, 256)), dim3(256), 0, s, tot, H, 16, HD, d_e1, d_e2);
    VQ_CHECK_LAUNCH("relattn_sub_de_split");
    return VQCPC_OK;
}

}  // namespace vq

using namespace vq;

#define VQ_SUB_DISPATCH(CALL)                        \
    if (L == 16 && hd == 16) return CALL(16, 16, 4); \
    if (L == 16 && hd == 32) return CALL(16, 32, 4); \
    if (L == 16 && hd == 64) return CALL(16, 64, 4); \
    if (L == 4 && hd == 16) return CALL(4, 16, 4);   \
    if (L == 4 && hd == 32) return CALL(4, 32, 4);   \
    if (L == 4 && hd == 64) return CALL(4, 64, 4);

extern "C" {

static int sub_fwd_impl(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const float* e1, const float* e2,
                        float* ctx, int64_t ldo, float* probs, int64_t n_blocks, int L, int F, int H, int hd, float drop_p,
                        uint64_t seed, void* stream, int b16) {
    if (n_blocks == 0) return VQCPC_OK;
    VQ_REQUIRE(q && kv && e1 && e2 && ctx && probs, "relattn_sub_fwd: null pointer");
    VQ_REQUIRE(sub_supported(L, F, H, hd), "relattn_sub_fwd: unsupported L=%d F=%d H=%d hd=%d", L, F, H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && ldq >= H * hd && ldkv >= 2 * H * hd && ldo >= H * hd &&
                   n_blocks >= 0,
               "relattn_sub_fwd: bad strides");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn_sub_fwd: bad dropout probability");
    hipStream_t s = (hipStream_t)stream;
    if (L == 16 && (4 % H == 0 || H % 4 == 0)) {
        if (hd == 16) return sub16_launch_fwd<16>(q, ldq, kv, ldkv, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s, b16);
        if (hd == 32) return sub16_launch_fwd<32>(q, ldq, kv, ldkv, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s, b16);
        if (hd == 64) return sub16_launch_fwd<64>(q, ldq, kv, ldkv, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s, b16);
    }
#define CALL(LL, DD, FF) sub_launch_fwd<LL, DD, FF>(q, ldq, kv, ldkv, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s, b16)
    VQ_SUB_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

int64_t vqcpc_relattn_sub_bwd_workspace(int64_t n_blocks, int L, int F, int H, int hd) {
    if (L == 16 && H >= 1 && (4 % H == 0 || H % 4 == 0)) {
        const int bpw16 = sub16_blocks_per_wg(std::max<int64_t>(n_blocks, 1), H);
        const int64_t chunks16 = ceil_div(std::max<int64_t>(n_blocks, 1), bpw16);
        const int NS16 = 4 >= H ? 4 / H : 1;
        return (chunks16 * NS16 + 1) * H * 31 * hd * (int64_t)sizeof(float);
    }
    const int ppw = 64 / (4 * std::max(1, L / std::max(F, 1)));
    const int bpw = sub_blocks_per_wg(std::max<int64_t>(n_blocks, 1), ppw, std::max(H, 1));
    const int64_t chunks = ceil_div(std::max<int64_t>(n_blocks, 1), bpw);
    const int NS = ppw >= H ? ppw / H : 1;
    return (chunks * NS + 1) * H * (2 * L - 1) * hd * (int64_t)sizeof(float);
}

static int sub_bwd_impl(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* kv, int64_t ldkv,
                        const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, float* d_kv,
                        int64_t ldgkv, float* d_e1, float* d_e2, int64_t n_blocks, int L, int F, int H, int hd, float drop_p,
                        uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream, int b16) {
    VQ_REQUIRE(d_ctx && q && kv && probs && e1 && e2 && d_q && d_kv && d_e1 && d_e2 && workspace,
               "relattn_sub_bwd: null pointer");
    VQ_REQUIRE(sub_supported(L, F, H, hd), "relattn_sub_bwd: unsupported L=%d F=%d H=%d hd=%d", L, F, H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && ldgq % 4 == 0 && ldgkv % 4 == 0 && ldq >= H * hd &&
                   ldgq >= H * hd && ldkv >= 2 * H * hd && ldgkv >= 2 * H * hd && ldo >= H * hd && n_blocks >= 1,
               "relattn_sub_bwd: bad strides");
    if (workspace_bytes < vqcpc_relattn_sub_bwd_workspace(n_blocks, L, F, H, hd)) {
        set_error("relattn_sub_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (L == 16 && (4 % H == 0 || H % 4 == 0)) {
#define CALL16(DD)                                                                                                     \
    sub16_launch_bwd<DD>(d_ctx, ldo, q, ldq, kv, ldkv, probs, e1, e2, d_q, ldgq, d_kv, ldgkv, d_e1, d_e2, n_blocks, H, drop_p, \
                         seed, (float*)workspace, s, b16)
        if (hd == 16) return CALL16(16);
        if (hd == 32) return CALL16(32);
        if (hd == 64) return CALL16(64);
#undef CALL16
    }
#define CALL(LL, DD, FF)                                                                                                \
    sub_launch_bwd<LL, DD, FF>(d_ctx, ldo, q, ldq, kv, ldkv, probs, e1, e2, d_q, ldgq, d_kv, ldgkv, d_e1, d_e2, n_blocks, H, \
                               drop_p, seed, (float*)workspace, s, b16)
    VQ_SUB_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

int vqcpc_relattn_sub_fwd(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const float* e1, const float* e2,
                          float* ctx, int64_t ldo, float* probs, int64_t n_blocks, int L, int F, int H, int hd,
                          float drop_p, uint64_t seed, void* stream) {
    return sub_fwd_impl(q, ldq, kv, ldkv, e1, e2, ctx, ldo, probs, n_blocks, L, F, H, hd, drop_p, seed, stream, 0);
}

int vqcpc_relattn_sub_bwd(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* kv, int64_t ldkv,
                          const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, float* d_kv,
                          int64_t ldgkv, float* d_e1, float* d_e2, int64_t n_blocks, int L, int F, int H, int hd,
                          float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream) {
    return sub_bwd_impl(d_ctx, ldo, q, ldq, kv, ldkv, probs, e1, e2, d_q, ldgq, d_kv, ldgkv, d_e1, d_e2, n_blocks, L, F, H, hd,
                        drop_p, seed, workspace, workspace_bytes, stream, 0);
}

/* bf16-output forms (the bf16 training path, configs[4]): ctx_b16 / d_kv_b16 hold bf16 elements (leading dimensions in elements);
 * d_q -- 1/8 of the gradient bytes, and an operand of an fp32 GEMM with two residual inputs -- stays fp32. */
int vqcpc_relattn_sub_b16_supported(int L, int F, int H, int hd) { return sub_supported(L, F, H, hd) ? 1 : 0; }

int vqcpc_relattn_sub_fwd_b16(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const float* e1, const float* e2,
                              void* ctx_b16, int64_t ldo, float* probs, int64_t n_blocks, int L, int F, int H, int hd,
                              float drop_p, uint64_t seed, void* stream) {
    return sub_fwd_impl(q, ldq, kv, ldkv, e1, e2, reinterpret_cast<float*>(ctx_b16), ldo, probs, n_blocks, L, F, H, hd, drop_p,
                        seed, stream, 1);
}

int vqcpc_relattn_sub_bwd_b16(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* kv, int64_t ldkv,
                              const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, void* d_kv_b16,
                              int64_t ldgkv, float* d_e1, float* d_e2, int64_t n_blocks, int L, int F, int H, int hd,
                              float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream) {
    return sub_bwd_impl(d_ctx, ldo, q, ldq, kv, ldkv, probs, e1, e2, d_q, ldgq,
                        reinterpret_cast<float*>(d_kv_b16), ldgkv, d_e1, d_e2, n_blocks, L, F, H, hd, drop_p, seed, workspace,
                        workspace_bytes, stream, 1);
}

}  // extern "C"
