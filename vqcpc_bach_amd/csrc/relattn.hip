// Fused small-L self-attention with the closed-form learned relative bias (L = 16 or 4).
//
// A "problem" is one (block n, head h): q, k, v are L x hd.  4*L lanes work on one problem, so a wavefront holds
// PPW = 64 / (4L) problems (1 for L = 16, 4 for L = 4) and the whole score tile lives in registers:
//   lane (i = sl >> 2, jg = sl & 3) owns scores S[i][jj*4 + jg], jj < L/4 and output columns [jg*hd/4, (jg+1)*hd/4).
// S[i][j] = q_i . (k_j + Erel[j - i + L - 1]),  Erel[r] = e1[h][r] (r < L: j <= i) | e2[h][r - L + 1] (r >= L: j > i)
// which is SubsampledRelativeAttention.forward (subsampled_relative_attention.py:30-122) without its pad / view / mask
// tensors, added to q.k^T as in MultiheadAttentionCustom.forward (multihead_attention_custom.py:314-343).
// f32 MFMA runs at the f32 VALU rate on gfx950, and the tile is 16x16x32, so this kernel is plain VALU + LDS and is
// bound by HBM traffic (q, k, v in; ctx, probs out).
#include "common.h"
#include <stdlib.h>

namespace vq {

constexpr int kAttThreads = 256;
constexpr int kPad = 4;

template <int L, int HD>
struct AttCfg {
    static constexpr int LPP = 4 * L;            // lanes per problem
    static constexpr int PPW = 64 / LPP;         // problems per wave
    static constexpr int SLOTS = 4 * PPW;        // problems per workgroup iteration
    static constexpr int JPL = L / 4;            // scores per lane
    static constexpr int CPL = HD / 4;           // output columns per lane
    static constexpr int RS = HD + kPad;         // LDS row stride
    static constexpr int NE = 2 * L - 1;
    static constexpr int FWD_FLOATS = (3 * L + NE) * RS + L * (L + 1);
    static constexpr int BWD_FLOATS = (4 * L + NE) * RS + 2 * L * (L + 1);
};

// cooperative copy of an L x HD matrix (row stride ld in global) into LDS (row stride RS) by the 4L lanes of a problem.
// tok != nullptr: `src` is a block table [vmax * L][ld] and row r of the block is table row tok[r] * L + r (the first
// layer's q | k | v are a function of (token id, position) only: they are read from the L2-resident table instead of a
// gathered per-token copy in HBM).
template <int L, int HD>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, int64_t ld, int sl, float mul,
                                           const int64_t* __restrict__ tok = nullptr) {
    constexpr int LPP = 4 * L, RS = HD + kPad, V = HD / 4;
#pragma unroll
    for (int e = sl; e < L * V; e += LPP) {
        const int row = e / V, c4 = e % V;
        const int64_t grow = tok ? tok[row] * L + row : row;
        float4 v = *reinterpret_cast<const float4*>(src + grow * ld + c4 * 4);
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(dst + row * RS + c4 * 4) = v;
    }
}

// LDS regions private to the lanes of one wavefront need no s_barrier: LDS instructions of a wave execute in order, only
// the compiler must keep the program order of the accesses around the exchange
__device__ __forceinline__ void att_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// register-staged variant for the backward's block loop: `fetch_rows` issues the global loads of the NEXT block before the
// current one is processed, `commit_rows` writes them to LDS after the barrier that retires the current block's readers
template <int L, int HD>
struct RowRegs {
    static constexpr int N = HD / 16;            // float4 per lane: L * (HD / 4) elements over the 4 L lanes of a problem
    float4 r[N];
};

template <int L, int HD>
__device__ __forceinline__ void fetch_rows(RowRegs<L, HD>& t, const float* __restrict__ src, int64_t ld, int sl,
                                           const int64_t* __restrict__ tok = nullptr) {
    constexpr int LPP = 4 * L, V = HD / 4;
#pragma unroll
    for (int k = 0; k < RowRegs<L, HD>::N; ++k) {
        const int e = sl + k * LPP, row = e / V, c4 = e % V;
        const int64_t grow = tok ? tok[row] * L + row : row;
        t.r[k] = *reinterpret_cast<const float4*>(src + grow * ld + c4 * 4);
    }
}

template <int L, int HD>
__device__ __forceinline__ void commit_rows(float* dst, const RowRegs<L, HD>& t, int sl, float mul) {
    constexpr int LPP = 4 * L, RS = HD + kPad, V = HD / 4;
#pragma unroll
    for (int k = 0; k < RowRegs<L, HD>::N; ++k) {
        const int e = sl + k * LPP, row = e / V, c4 = e % V;
        float4 v = t.r[k];
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(dst + row * RS + c4 * 4) = v;
    }
}

template <int L, int HD>
__device__ __forceinline__ void stage_erel(float* dst, const float* __restrict__ e1, const float* __restrict__ e2, int h,
                                           int sl) {
    constexpr int LPP = 4 * L, RS = HD + kPad, V = HD / 4, NE = 2 * L - 1;
#pragma unroll
    for (int e = sl; e < NE * V; e += LPP) {
        const int r = e / V, c4 = e % V;
        const float* src = r < L ? e1 + ((int64_t)h * L + r) * HD : e2 + ((int64_t)h * L + (r - L + 1)) * HD;
        *reinterpret_cast<float4*>(dst + r * RS + c4 * 4) = *reinterpret_cast<const float4*>(src + c4 * 4);
    }
}

// =====================================================================================================================
template <int L, int HD>
__global__ __launch_bounds__(kAttThreads) void relattn_fwd_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                                  const float* __restrict__ e1,
                                                                  const float* __restrict__ e2, float* __restrict__ ctx,
                                                                  int64_t ldo, float* __restrict__ probs,
                                                                  int64_t n_blocks, int H, float scale, uint32_t thr,
                                                                  float inv_keep, uint64_t seed,
                                                                  const int64_t* __restrict__ tokens, int o16) {
    // Persistent slots: slot g of the grid keeps head g % H (the host makes the slot count a multiple of H) and walks the
    // blocks g / H, + slots / H, ...: the relative rows are staged once, the q | k | v rows of the next block are fetched
    // into registers while the current one is processed.  A slot's LDS region is private to its 4 L lanes of one wave.
    using C = AttCfg<L, HD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane % C::LPP, slot = wave * C::PPW + lane / C::LPP;
    float* base = lds + slot * C::FWD_FLOATS;
    float* Qs = base;
    float* Ks = Qs + L * C::RS;
    float* Vs = Ks + L * C::RS;
    float* Er = Vs + L * C::RS;
    float* Ps = Er + C::NE * C::RS;                 // [L][L+1]
    const int d = H * HD;
    const int64_t slots = (int64_t)gridDim.x * C::SLOTS, gs = (int64_t)blockIdx.x * C::SLOTS + slot;
    const int h = (int)(gs % H);
    const int64_t n0 = gs / H, nstep = slots / H;
    const int i = sl >> 2, jg = sl & 3;
    stage_erel<L, HD>(Er, e1, e2, h, sl);
    RowRegs<L, HD> rq, rk, rv;
    auto prefetch = [&](int64_t nb) {
        const int64_t nc = min(nb, n_blocks - 1);                    // past the end: re-read the last block, never used
        const int64_t* tk = tokens ? tokens + nc * L : nullptr;
        const float* qp = tokens ? qkv + h * HD : qkv + nc * L * ldq + h * HD;
        fetch_rows<L, HD>(rq, qp, ldq, sl, tk);
        fetch_rows<L, HD>(rk, qp + d, ldq, sl, tk);
        fetch_rows<L, HD>(rv, qp + 2 * d, ldq, sl, tk);
    };
    prefetch(n0);
    const int64_t iters = (n_blocks + nstep - 1) / nstep;            // uniform trip count; slots past the end idle (live)
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t n = n0 + it * nstep;
        const bool live = n < n_blocks;
        const int64_t prob = n * H + h;
        att_wave_fence();                                            // the previous block's LDS readers are done
        if (live) {
            commit_rows<L, HD>(Qs, rq, sl, scale);
            commit_rows<L, HD>(Ks, rk, sl, 1.0f);
            commit_rows<L, HD>(Vs, rv, sl, 1.0f);
        }
        att_wave_fence();
        prefetch(n + nstep);
        if (live) {
            float s[C::JPL];
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) s[jj] = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                const float4 q = *reinterpret_cast<const float4*>(Qs + i * C::RS + c4 * 4);
#pragma unroll
                for (int jj = 0; jj < C::JPL; ++jj) {
                    const int j = jj * 4 + jg;
                    const float4 k = *reinterpret_cast<const float4*>(Ks + j * C::RS + c4 * 4);
                    const float4 e = *reinterpret_cast<const float4*>(Er + (j - i + L - 1) * C::RS + c4 * 4);
                    s[jj] += q.x * (k.x + e.x) + q.y * (k.y + e.y) + q.z * (k.z + e.z) + q.w * (k.w + e.w);
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
            float* pg = probs + prob * L * L + i * L;
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) {
                const int j = jj * 4 + jg;
                const float p = s[jj] * inv;
                pg[j] = p;                                                          // saved BEFORE dropout
                Ps[i * (L + 1) + j] = p * drop_scale(seed, (uint64_t)(prob * L + i) * L + j, thr, inv_keep);
            }
        }
        att_wave_fence();
        if (live) {
            float o[C::CPL];
#pragma unroll
            for (int c = 0; c < C::CPL; ++c) o[c] = 0.0f;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const float p = Ps[i * (L + 1) + j];
#pragma unroll
                for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                    const float4 v = *reinterpret_cast<const float4*>(Vs + j * C::RS + jg * C::CPL + c4 * 4);
                    o[c4 * 4 + 0] += p * v.x;
                    o[c4 * 4 + 1] += p * v.y;
                    o[c4 * 4 + 2] += p * v.z;
                    o[c4 * 4 + 3] += p * v.w;
                }
            }
            const int64_t oo = (n * L + i) * ldo + h * HD + jg * C::CPL;
#pragma unroll
            for (int c4 = 0; c4 < C::CPL / 4; ++c4)
                store4_out(ctx, oo + c4 * 4, o[c4 * 4], o[c4 * 4 + 1], o[c4 * 4 + 2], o[c4 * 4 + 3], o16);
        }
    }
}

// =====================================================================================================================
// backward.  grid = (chunks, head groups).  Every slot keeps the same head for the whole loop over its blocks, so the
// relative-embedding gradient accumulates in registers and is written once per slot (deterministic).
// ws layout: [chunk][nsub][H][2L-1][HD]
template <int L, int HD>
__global__ __launch_bounds__(kAttThreads) void relattn_bwd_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ qkv, int64_t ldq,
    const float* __restrict__ probs, const float* __restrict__ e1, const float* __restrict__ e2,
    float* __restrict__ d_qkv, int64_t ldg, float* __restrict__ ws, int64_t n_blocks, int H, int blocks_per_wg,
    float scale, uint32_t thr, float inv_keep, uint64_t seed, const int64_t* __restrict__ tokens, int g16) {
    using C = AttCfg<L, HD>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane % C::LPP, slot = wave * C::PPW + lane / C::LPP;
    float* base = lds + slot * C::BWD_FLOATS;
    float* Qs = base;
    float* Ks = Qs + L * C::RS;
    float* Vs = Ks + L * C::RS;
    float* Os = Vs + L * C::RS;                     // d_ctx rows
    float* Er = Os + L * C::RS;
    float* Ss = Er + C::NE * C::RS;                 // dS   [L][L+1]
    float* Ps = Ss + L * (L + 1);                   // P after dropout [L][L+1]
    const int d = H * HD;
    // slot -> (head, block sub-index)
    int h, nsub, NS;
    if (C::SLOTS >= H) {
        NS = C::SLOTS / H;
        h = slot % H;
        nsub = slot / H;
    } else {
        NS = 1;
        h = blockIdx.y * C::SLOTS + slot;
        nsub = 0;
    }
    const int i = sl >> 2, jg = sl & 3;
    stage_erel<L, HD>(Er, e1, e2, h, sl);
    float de[2][C::CPL];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < C::CPL; ++c) de[a][c] = 0.0f;

    const int64_t n_begin = (int64_t)blockIdx.x * blocks_per_wg;
    // software pipeline over the blocks: the operands of the slot's next block are in flight while this one is processed
    // (the loop is barrier-lockstep, so nothing else hides the global-load round trip).  L = 4 only: at L = 16 (the
    // fallback of the matrix-core kernel) and at head_dim 64 the staging registers cost occupancy
    constexpr bool kPipe = HD <= 32 && L == 4;
    RowRegs<L, HD> rq, rk, rv, ro;
    auto prefetch = [&](int64_t nb) {
        if constexpr (kPipe) {
            const int64_t nc = min(nb, n_blocks - 1);                // past the end: re-read the last block, never used
            const int64_t* tk = tokens ? tokens + nc * L : nullptr;
            const float* qp = tokens ? qkv + h * HD : qkv + nc * L * ldq + h * HD;
            fetch_rows<L, HD>(rq, qp, ldq, sl, tk);
            fetch_rows<L, HD>(rk, qp + d, ldq, sl, tk);
            fetch_rows<L, HD>(rv, qp + 2 * d, ldq, sl, tk);
            fetch_rows<L, HD>(ro, d_ctx + nc * L * ldo + h * HD, ldo, sl);
        }
    };
    prefetch(n_begin + nsub);
    for (int it = 0; it < blocks_per_wg; it += NS) {
        const int64_t n = n_begin + it + nsub;
        const bool live = (it + nsub < blocks_per_wg) && n < n_blocks;
        const int64_t prob = n * H + h;
        __syncthreads();
        if (live) {
            if constexpr (kPipe) {
                commit_rows<L, HD>(Qs, rq, sl, scale);
                commit_rows<L, HD>(Ks, rk, sl, 1.0f);
                commit_rows<L, HD>(Vs, rv, sl, 1.0f);
                commit_rows<L, HD>(Os, ro, sl, 1.0f);
            } else {
                const int64_t* tk = tokens ? tokens + n * L : nullptr;
                const float* qp = tokens ? qkv + h * HD : qkv + n * L * ldq + h * HD;
                stage_rows<L, HD>(Qs, qp, ldq, sl, scale, tk);
                stage_rows<L, HD>(Ks, qp + d, ldq, sl, 1.0f, tk);
                stage_rows<L, HD>(Vs, qp + 2 * d, ldq, sl, 1.0f, tk);
                stage_rows<L, HD>(Os, d_ctx + n * L * ldo + h * HD, ldo, sl, 1.0f);
            }
        }
        __syncthreads();
        prefetch(n + NS);
        if (live) {
            // dPd[i][j] = dO_i . V_j ; softmax backward
            float dp[C::JPL], p[C::JPL];
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) dp[jj] = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                const float4 o = *reinterpret_cast<const float4*>(Os + i * C::RS + c4 * 4);
#pragma unroll
                for (int jj = 0; jj < C::JPL; ++jj) {
                    const float4 v = *reinterpret_cast<const float4*>(Vs + (jj * 4 + jg) * C::RS + c4 * 4);
                    dp[jj] += o.x * v.x + o.y * v.y + o.z * v.z + o.w * v.w;
                }
            }
            const float* pg = probs + prob * L * L + i * L;
            float rowdot = 0.0f;
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) {
                const int j = jj * 4 + jg;
                p[jj] = pg[j];
                const float mk = drop_scale(seed, (uint64_t)(prob * L + i) * L + j, thr, inv_keep);
                dp[jj] *= mk;                                   // dP = dPd * mask / (1-p)
                Ps[i * (L + 1) + j] = p[jj] * mk;
                rowdot += dp[jj] * p[jj];
            }
            rowdot += __shfl_xor(rowdot, 1, 64);
            rowdot += __shfl_xor(rowdot, 2, 64);
#pragma unroll
            for (int jj = 0; jj < C::JPL; ++jj) Ss[i * (L + 1) + jj * 4 + jg] = p[jj] * (dp[jj] - rowdot);
        }
        __syncthreads();
        if (live) {
            // row index `i` doubles as the key index j for dK / dV
            float dk[C::CPL], dv[C::CPL], dq[C::CPL];
#pragma unroll
            for (int c = 0; c < C::CPL; ++c) dk[c] = dv[c] = dq[c] = 0.0f;
            const int j = i;
#pragma unroll
            for (int ii = 0; ii < L; ++ii) {
                const float pd = Ps[ii * (L + 1) + j], ds = Ss[ii * (L + 1) + j];
#pragma unroll
                for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                    const float4 o = *reinterpret_cast<const float4*>(Os + ii * C::RS + jg * C::CPL + c4 * 4);
                    const float4 q = *reinterpret_cast<const float4*>(Qs + ii * C::RS + jg * C::CPL + c4 * 4);
                    dv[c4 * 4 + 0] += pd * o.x; dv[c4 * 4 + 1] += pd * o.y; dv[c4 * 4 + 2] += pd * o.z; dv[c4 * 4 + 3] += pd * o.w;
                    dk[c4 * 4 + 0] += ds * q.x; dk[c4 * 4 + 1] += ds * q.y; dk[c4 * 4 + 2] += ds * q.z; dk[c4 * 4 + 3] += ds * q.w;
                }
            }
            // dq_i = scale * sum_j dS[i][j] (k_j + Erel[j - i + L - 1])
#pragma unroll
            for (int jj = 0; jj < L; ++jj) {
                const float ds = Ss[i * (L + 1) + jj];
#pragma unroll
                for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                    const float4 k = *reinterpret_cast<const float4*>(Ks + jj * C::RS + jg * C::CPL + c4 * 4);
                    const float4 e = *reinterpret_cast<const float4*>(Er + (jj - i + L - 1) * C::RS + jg * C::CPL + c4 * 4);
                    dq[c4 * 4 + 0] += ds * (k.x + e.x); dq[c4 * 4 + 1] += ds * (k.y + e.y);
                    dq[c4 * 4 + 2] += ds * (k.z + e.z); dq[c4 * 4 + 3] += ds * (k.w + e.w);
                }
            }
            const int64_t go = (n * L + i) * ldg + h * HD + jg * C::CPL;
#pragma unroll
            for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                store4_out(d_qkv, go + c4 * 4, dq[c4 * 4] * scale, dq[c4 * 4 + 1] * scale, dq[c4 * 4 + 2] * scale,
                           dq[c4 * 4 + 3] * scale, g16);
                store4_out(d_qkv, go + d + c4 * 4, dk[c4 * 4], dk[c4 * 4 + 1], dk[c4 * 4 + 2], dk[c4 * 4 + 3], g16);
                store4_out(d_qkv, go + 2 * d + c4 * 4, dv[c4 * 4], dv[c4 * 4 + 1], dv[c4 * 4 + 2], dv[c4 * 4 + 3], g16);
            }
            // dErel[r] += sum_{i', j: j - i' + L - 1 = r} dS[i'][j] * qs[i']      rows r = i and r = i + L
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int r = i + a * L;
                if (r < C::NE) {
#pragma unroll
                    for (int ii = 0; ii < L; ++ii) {
                        const int jx = ii + r - (L - 1);
                        if (jx >= 0 && jx < L) {
                            const float ds = Ss[ii * (L + 1) + jx];
#pragma unroll
                            for (int c4 = 0; c4 < C::CPL / 4; ++c4) {
                                const float4 q = *reinterpret_cast<const float4*>(Qs + ii * C::RS + jg * C::CPL + c4 * 4);
                                de[a][c4 * 4 + 0] += ds * q.x; de[a][c4 * 4 + 1] += ds * q.y;
                                de[a][c4 * 4 + 2] += ds * q.z; de[a][c4 * 4 + 3] += ds * q.w;
                            }
                        }
                    }
                }
            }
        }
    }
    // one partial per (workgroup, nsub, head)
    float* dst = ws + ((((int64_t)blockIdx.x * NS + nsub) * H + h) * C::NE) * HD;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int r = i + a * L;
        if (r < C::NE) {
#pragma unroll
            for (int c = 0; c < C::CPL; ++c) dst[r * HD + jg * C::CPL + c] = de[a][c];
        }
    }
}

// split the reduced Erel gradient [H][2L-1][HD] back into e1 / e2 (row 0 of e2 is never used by the closed form: 0)
__global__ __launch_bounds__(256) void relattn_de_split(const float* __restrict__ tot, int H, int L, int HD,
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

static int att_blocks_per_wg(int64_t n_blocks, int slots, int H) {
    // aim at ~2048 workgroups; a workgroup iteration covers max(1, slots / H) blocks
    const int ns = std::max(1, slots / H);
    int64_t b = ceil_div(n_blocks, 2048);
    b = round_up(std::max<int64_t>(b, ns), ns);
    return (int)b;
}

template <int L, int HD>
static int launch_fwd(const float* qkv, int64_t ldq, const float* e1, const float* e2, float* ctx, int64_t ldo,
                      float* probs, int64_t n_blocks, int H, float drop_p, uint64_t seed, hipStream_t s,
                      const int64_t* tokens = nullptr, int o16 = 0) {
    using C = AttCfg<L, HD>;
    const size_t lds = (size_t)C::SLOTS * C::FWD_FLOATS * sizeof(float);
    auto kern = relattn_fwd_kernel<L, HD>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // persistent slots (2048 workgroups at most); the slot count must be a multiple of H
    int64_t grid = std::min<int64_t>(ceil_div(n_blocks * H, C::SLOTS), 2048);
    int gcd = H, rem = C::SLOTS;
    while (rem) { const int t = gcd % rem; gcd = rem; rem = t; }
    const int64_t unit = H / gcd;
    grid = ceil_div(grid, unit) * unit;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kAttThreads), lds, s, qkv, ldq, e1, e2, ctx, ldo, probs, n_blocks,
                       H, 1.0f / sqrtf((float)HD), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, tokens, o16);
    VQ_CHECK_LAUNCH("relattn_fwd");
    return VQCPC_OK;
}

template <int L, int HD>
static int launch_bwd(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const float* probs, const float* e1,
                      const float* e2, float* d_qkv, int64_t ldg, float* d_e1, float* d_e2, int64_t n_blocks, int H,
                      float drop_p, uint64_t seed, float* ws, hipStream_t s, const int64_t* tokens = nullptr, int g16 = 0) {
    using C = AttCfg<L, HD>;
    const size_t lds = (size_t)C::SLOTS * C::BWD_FLOATS * sizeof(float);
    auto kern = relattn_bwd_kernel<L, HD>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int bpw = att_blocks_per_wg(n_blocks, C::SLOTS, H);
    const int chunks = (int)ceil_div(n_blocks, bpw);
    const int gy = C::SLOTS >= H ? 1 : H / C::SLOTS;
    const int NS = C::SLOTS >= H ? C::SLOTS / H : 1;
    hipLaunchKernelGGL(kern, dim3(chunks, gy), dim3(kAttThreads), lds, s, d_ctx, ldo, qkv, ldq, probs, e1, e2, d_qkv, ldg,
                       ws, n_blocks, H, bpw, 1.0f / sqrtf((float)HD), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, tokens, g16);
    VQ_CHECK_LAUNCH("relattn_bwd");
    const int total = H * C::NE * HD;
    float* tot = ws + (int64_t)chunks * NS * total;            // tail of the workspace
    int rc = launch_reduce_splits(ws, total, chunks * NS, tot, total, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(relattn_de_split, dim3(ceil_div(total, 256)), dim3(256), 0, s, tot, H, L, HD, d_e1, d_e2);
    VQ_CHECK_LAUNCH("relattn_de_split");
    return VQCPC_OK;
}

static int g_force_general = 0;   // tests: route L = 16 / 4 through the general-L kernels too
// L = 16 runs on the matrix cores (relattn16.hip); VQCPC_RELATTN16_LDS=1 keeps the LDS-tiled VALU kernels (A/B, tests)
static bool use_mfma16(int L, int H, int hd) {
    static const bool lds_only = lab_env_int("VQCPC_RELATTN16_LDS", 0) != 0;
    return L == 16 && !lds_only && relattn16_supported(H, hd);
}
static int finish_de16(float* ws, int nsplit, int H, int hd, float* d_e1, float* d_e2, hipStream_t s);

static bool att_supported(int L, int H, int hd) {
    if (g_force_general) return false;
    if (!(L == 16 || L == 4)) return false;
    if (!(hd == 16 || hd == 32 || hd == 64)) return false;
    const int slots = 4 * (64 / (4 * L));
    return H >= 1 && ((slots % H) == 0 || (H % slots) == 0);
}

static int finish_de16(float* ws, int nsplit, int H, int hd, float* d_e1, float* d_e2, hipStream_t s) {
    const int total = H * 31 * hd;
    float* tot = ws + (int64_t)nsplit * total;
    int rc = launch_reduce_splits(ws, total, nsplit, tot, total, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(relattn_de_split, dim3(ceil_div(total, 256)), dim3(256), 0, s, tot, H, 16, hd, d_e1, d_e2);
    VQ_CHECK_LAUNCH("relattn_de_split");
    return VQCPC_OK;
}

}  // namespace vq

using namespace vq;

#define VQ_ATT_DISPATCH(CALL)                                              \
    if (L == 16 && hd == 16) return CALL(16, 16);                          \
    if (L == 16 && hd == 32) return CALL(16, 32);                          \
    if (L == 16 && hd == 64) return CALL(16, 64);                          \
    if (L == 4 && hd == 16) return CALL(4, 16);                            \
    if (L == 4 && hd == 32) return CALL(4, 32);                            \
    if (L == 4 && hd == 64) return CALL(4, 64);

static bool strip_supported(int L, int H, int hd) {
    return L >= 1 && L <= 1024 && H >= 1 && (hd == 16 || hd == 32 || hd == 64 || hd == 128);
}

extern "C" {

int vqcpc_relattn_force_general(int on) {
    g_force_general = on ? 1 : 0;
    return VQCPC_OK;
}

int vqcpc_relattn_fwd(const float* qkv, int64_t ldq, const float* e1, const float* e2, float* ctx, int64_t ldo,
                      float* probs, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed, void* stream) {
    if (n_blocks == 0) return VQCPC_OK;
    VQ_REQUIRE(qkv && e1 && e2 && ctx && probs, "relattn_fwd: null pointer");
    const bool small = att_supported(L, H, hd);
    VQ_REQUIRE(small || strip_supported(L, H, hd), "relattn_fwd: unsupported L=%d H=%d hd=%d (L <= 1024, hd in {16,32,64,128})",
               L, H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ldq >= 3 * H * hd && ldo >= H * hd && n_blocks >= 0, "relattn_fwd: bad strides");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn_fwd: bad dropout probability");
    hipStream_t s = (hipStream_t)stream;
    if (!small) {
        // any other L: the strip kernels of relattn_x.hip with Lq = Lk, no mask, q | k | v = column blocks of qkv
        VQ_REQUIRE(aligned16(qkv) && aligned16(e1) && aligned16(e2), "relattn_fwd: qkv / e1 / e2 must be 16-byte aligned");
        VQ_REQUIRE(n_blocks * H * (int64_t)((L + 31) / 32) < (1ll << 31), "relattn_fwd: too many strips");
        const int d = H * hd;
        return vqcpc_relattn_x_fwd(qkv, ldq, qkv + d, ldq, qkv + 2 * d, ldq, e1, e2, ctx, ldo, probs, n_blocks, L, L, H, hd, 0,
                                   drop_p, seed, stream);
    }
    if (use_mfma16(L, H, hd) && aligned16(qkv) && aligned16(e1) && aligned16(e2) && aligned16(ctx))
        return relattn16_fwd(qkv, ldq, nullptr, e1, e2, ctx, ldo, probs, n_blocks, H, hd, drop_p, seed, s);
#define CALL(LL, DD) launch_fwd<LL, DD>(qkv, ldq, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s)
    VQ_ATT_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

int64_t vqcpc_relattn_bwd_workspace(int64_t n_blocks, int L, int H, int hd) {
    if (!att_supported(L, H, hd)) return vqcpc_relattn_x_bwd_workspace(n_blocks, std::max(L, 1), std::max(L, 1), std::max(H, 1), hd);
    const int64_t w16 = use_mfma16(L, H, hd) ? relattn16_bwd_workspace(std::max<int64_t>(n_blocks, 1), H, hd) : 0;
    const int slots = 4 * (64 / (4 * std::max(L, 1)));
    const int bpw = att_blocks_per_wg(std::max<int64_t>(n_blocks, 1), slots, std::max(H, 1));
    const int64_t chunks = ceil_div(std::max<int64_t>(n_blocks, 1), bpw);
    const int NS = slots >= H ? slots / H : 1;
    return std::max<int64_t>(w16, (chunks * NS + 1) * H * (2 * L - 1) * hd * (int64_t)sizeof(float));
}

int vqcpc_relattn_bwd(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const float* probs, const float* e1,
                      const float* e2, float* d_qkv, int64_t ldg, float* d_e1, float* d_e2, int64_t n_blocks, int L, int H,
                      int hd, float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(d_ctx && qkv && probs && e1 && e2 && d_qkv && d_e1 && d_e2 && workspace, "relattn_bwd: null pointer");
    const bool small = att_supported(L, H, hd);
    VQ_REQUIRE(small || strip_supported(L, H, hd), "relattn_bwd: unsupported L=%d H=%d hd=%d", L, H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0 && ldq >= 3 * H * hd && ldg >= 3 * H * hd && ldo >= H * hd &&
                   n_blocks >= 1,
               "relattn_bwd: bad strides");
    if (workspace_bytes < vqcpc_relattn_bwd_workspace(n_blocks, L, H, hd)) {
        set_error("relattn_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (!small) {
        VQ_REQUIRE(aligned16(qkv) && aligned16(d_ctx) && aligned16(workspace), "relattn_bwd: buffers must be 16-byte aligned");
        const int d = H * hd;
        return vqcpc_relattn_x_bwd(d_ctx, ldo, qkv, ldq, qkv + d, ldq, qkv + 2 * d, ldq, probs, e1, e2, d_qkv, ldg, d_qkv + d, ldg,
                                   d_qkv + 2 * d, ldg, d_e1, d_e2, n_blocks, L, L, H, hd, 0, drop_p, seed, workspace,
                                   workspace_bytes, stream);
    }
    if (use_mfma16(L, H, hd) && aligned16(qkv) && aligned16(d_ctx) && aligned16(e1) && aligned16(e2) && aligned16(d_qkv)) {
        int nsplit = 0;
        int rc = relattn16_bwd(d_ctx, ldo, qkv, ldq, nullptr, probs, e1, e2, d_qkv, ldg, (float*)workspace, n_blocks, H, hd,
                               drop_p, seed, s, &nsplit);
        return rc ? rc : finish_de16((float*)workspace, nsplit, H, hd, d_e1, d_e2, s);
    }
#define CALL(LL, DD) \
    launch_bwd<LL, DD>(d_ctx, ldo, qkv, ldq, probs, e1, e2, d_qkv, ldg, d_e1, d_e2, n_blocks, H, drop_p, seed, (float*)workspace, s)
    VQ_ATT_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

/* bf16-output forms of the L = 16 matrix-core attention (the bf16 training path, BASELINE configs[4]): ctx / d_qkv are bf16
 * buffers (leading dimensions in elements); `tokens` non-NULL = qkv is the first layer's block table (vqcpc_relattn_tab_*). */
int vqcpc_relattn16_b16_supported(int L, int H, int hd) { return (!g_force_general && use_mfma16(L, H, hd)) ? 1 : 0; }

int vqcpc_relattn16_fwd_b16(const float* qkv, int64_t ldq, const int64_t* tokens, const float* e1, const float* e2, void* ctx_b16,
                            int64_t ldo, float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* stream) {
    if (n_blocks == 0) return VQCPC_OK;
    VQ_REQUIRE(qkv && e1 && e2 && ctx_b16 && probs, "relattn16_fwd_b16: null pointer");
    VQ_REQUIRE(vqcpc_relattn16_b16_supported(16, H, hd), "relattn16_fwd_b16: unsupported H=%d hd=%d", H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ldq >= 3 * H * hd && ldo >= H * hd && n_blocks >= 0 && aligned16(qkv) &&
                   aligned16(e1) && aligned16(e2) && aligned16(ctx_b16),
               "relattn16_fwd_b16: bad strides / alignment");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn16_fwd_b16: bad dropout probability");
    return relattn16_fwd_b16(qkv, ldq, tokens, e1, e2, ctx_b16, ldo, probs, n_blocks, H, hd, drop_p, seed, (hipStream_t)stream);
}

int vqcpc_relattn16_bwd_b16(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const int64_t* tokens,
                            const float* probs, const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* d_e1,
                            float* d_e2, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* workspace,
                            int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(d_ctx && qkv && probs && e1 && e2 && d_qkv_b16 && d_e1 && d_e2 && workspace, "relattn16_bwd_b16: null pointer");
    VQ_REQUIRE(vqcpc_relattn16_b16_supported(16, H, hd), "relattn16_bwd_b16: unsupported H=%d hd=%d", H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0 && ldq >= 3 * H * hd && ldg >= 3 * H * hd && ldo >= H * hd &&
                   n_blocks >= 1 && aligned16(qkv) && aligned16(d_ctx) && aligned16(e1) && aligned16(e2) && aligned16(d_qkv_b16),
               "relattn16_bwd_b16: bad strides / alignment");
    if (workspace_bytes < vqcpc_relattn_bwd_workspace(n_blocks, 16, H, hd)) {
        set_error("relattn16_bwd_b16: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    int nsplit = 0;
    int rc = relattn16_bwd_b16(d_ctx, ldo, qkv, ldq, tokens, probs, e1, e2, d_qkv_b16, ldg, (float*)workspace, n_blocks, H, hd,
                               drop_p, seed, s, &nsplit);
    return rc ? rc : finish_de16((float*)workspace, nsplit, H, hd, d_e1, d_e2, s);
}

/* bf16-output forms for the other block lengths of the encoder stacks (L = 4; L = 16 is forwarded to the matrix-core kernels). */
int vqcpc_relattn_b16_supported(int L, int H, int hd) { return (!g_force_general && att_supported(L, H, hd)) ? 1 : 0; }

int vqcpc_relattn_fwd_b16(const float* qkv, int64_t ldq, const float* e1, const float* e2, void* ctx_b16, int64_t ldo,
                          float* probs, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed, void* stream) {
    if (n_blocks == 0) return VQCPC_OK;
    VQ_REQUIRE(qkv && e1 && e2 && ctx_b16 && probs, "relattn_fwd_b16: null pointer");
    VQ_REQUIRE(!g_force_general && att_supported(L, H, hd), "relattn_fwd_b16: unsupported L=%d H=%d hd=%d (L in {16,4})", L, H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ldq >= 3 * H * hd && ldo >= H * hd && aligned16(qkv) && aligned16(ctx_b16),
               "relattn_fwd_b16: bad strides / alignment");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn_fwd_b16: bad dropout probability");
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma16(L, H, hd) && aligned16(e1) && aligned16(e2))
        return relattn16_fwd_b16(qkv, ldq, nullptr, e1, e2, ctx_b16, ldo, probs, n_blocks, H, hd, drop_p, seed, s);
    float* ctx = reinterpret_cast<float*>(ctx_b16);
#define CALL(LL, DD) launch_fwd<LL, DD>(qkv, ldq, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s, nullptr, 1)
    VQ_ATT_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

int vqcpc_relattn_bwd_b16(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const float* probs, const float* e1,
                          const float* e2, void* d_qkv_b16, int64_t ldg, float* d_e1, float* d_e2, int64_t n_blocks, int L,
                          int H, int hd, float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(d_ctx && qkv && probs && e1 && e2 && d_qkv_b16 && d_e1 && d_e2 && workspace, "relattn_bwd_b16: null pointer");
    VQ_REQUIRE(!g_force_general && att_supported(L, H, hd), "relattn_bwd_b16: unsupported L=%d H=%d hd=%d (L in {16,4})", L, H, hd);
    VQ_REQUIRE(ldq % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0 && ldq >= 3 * H * hd && ldg >= 3 * H * hd && ldo >= H * hd &&
                   n_blocks >= 1 && aligned16(qkv) && aligned16(d_ctx) && aligned16(d_qkv_b16),
               "relattn_bwd_b16: bad strides / alignment");
    if (workspace_bytes < vqcpc_relattn_bwd_workspace(n_blocks, L, H, hd)) {
        set_error("relattn_bwd_b16: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma16(L, H, hd) && aligned16(e1) && aligned16(e2)) {
        int nsplit = 0;
        int rc = relattn16_bwd_b16(d_ctx, ldo, qkv, ldq, nullptr, probs, e1, e2, d_qkv_b16, ldg, (float*)workspace, n_blocks, H,
                                   hd, drop_p, seed, s, &nsplit);
        return rc ? rc : finish_de16((float*)workspace, nsplit, H, hd, d_e1, d_e2, s);
    }
    float* d_qkv = reinterpret_cast<float*>(d_qkv_b16);
#define CALL(LL, DD)                                                                                                   \
    launch_bwd<LL, DD>(d_ctx, ldo, qkv, ldq, probs, e1, e2, d_qkv, ldg, d_e1, d_e2, n_blocks, H, drop_p, seed, (float*)workspace, s, \
                       nullptr, 1)
    VQ_ATT_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

/* all-bf16 forms: q | k | v (written as bf16 by the in_proj GEMM epilogue) and, in the backward, d ctx (out-proj dgrad epilogue)
 * are bf16 too -- half the bytes of the two kernels' dominant streams.  Leading dimensions in elements, multiples of 8. */
int vqcpc_relattn16_fwd_b16io(const void* qkv_b16, int64_t ldq, const float* e1, const float* e2, void* ctx_b16, int64_t ldo,
                              float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* stream) {
    if (n_blocks == 0) return VQCPC_OK;
    VQ_REQUIRE(qkv_b16 && e1 && e2 && ctx_b16 && probs, "relattn16_fwd_b16io: null pointer");
    VQ_REQUIRE(vqcpc_relattn16_b16_supported(16, H, hd), "relattn16_fwd_b16io: unsupported H=%d hd=%d", H, hd);
    VQ_REQUIRE(ldq % 8 == 0 && ldo % 4 == 0 && ldq >= 3 * H * hd && ldo >= H * hd && n_blocks >= 0 && aligned16(qkv_b16) &&
                   aligned16(e1) && aligned16(e2) && aligned16(ctx_b16),
               "relattn16_fwd_b16io: bad strides / alignment");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn16_fwd_b16io: bad dropout probability");
    return relattn16_fwd_b16io(qkv_b16, ldq, e1, e2, ctx_b16, ldo, probs, n_blocks, H, hd, drop_p, seed, (hipStream_t)stream);
}

int vqcpc_relattn16_bwd_b16io(const void* d_ctx_b16, int64_t ldo, const void* qkv_b16, int64_t ldq, const float* probs,
                              const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* d_e1, float* d_e2,
                              int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(d_ctx_b16 && qkv_b16 && probs && e1 && e2 && d_qkv_b16 && d_e1 && d_e2 && workspace,
               "relattn16_bwd_b16io: null pointer");
    VQ_REQUIRE(vqcpc_relattn16_b16_supported(16, H, hd), "relattn16_bwd_b16io: unsupported H=%d hd=%d", H, hd);
    VQ_REQUIRE(ldq % 8 == 0 && ldo % 8 == 0 && ldg % 4 == 0 && ldq >= 3 * H * hd && ldg >= 3 * H * hd && ldo >= H * hd &&
                   n_blocks >= 1 && aligned16(qkv_b16) && aligned16(d_ctx_b16) && aligned16(e1) && aligned16(e2) &&
                   aligned16(d_qkv_b16),
               "relattn16_bwd_b16io: bad strides / alignment");
    if (workspace_bytes < vqcpc_relattn_bwd_workspace(n_blocks, 16, H, hd)) {
        set_error("relattn16_bwd_b16io: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    int nsplit = 0;
    int rc = relattn16_bwd_b16io(d_ctx_b16, ldo, qkv_b16, ldq, probs, e1, e2, d_qkv_b16, ldg, (float*)workspace, n_blocks, H, hd,
                                 drop_p, seed, s, &nsplit);
    return rc ? rc : finish_de16((float*)workspace, nsplit, H, hd, d_e1, d_e2, s);
}

int vqcpc_relattn_tab_fwd(const float* table, int64_t ldt, const int64_t* tokens, const float* e1, const float* e2, float* ctx,
                          int64_t ldo, float* probs, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed,
                          void* stream) {
    if (n_blocks == 0) return VQCPC_OK;
    VQ_REQUIRE(table && tokens && e1 && e2 && ctx && probs, "relattn_tab_fwd: null pointer");
    VQ_REQUIRE(!g_force_general && att_supported(L, H, hd), "relattn_tab_fwd: unsupported L=%d H=%d hd=%d (L in {16,4})", L, H, hd);
    VQ_REQUIRE(ldt % 4 == 0 && ldo % 4 == 0 && ldt >= 3 * H * hd && ldo >= H * hd, "relattn_tab_fwd: bad strides");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn_tab_fwd: bad dropout probability");
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma16(L, H, hd) && aligned16(table) && aligned16(e1) && aligned16(e2) && aligned16(ctx))
        return relattn16_fwd(table, ldt, tokens, e1, e2, ctx, ldo, probs, n_blocks, H, hd, drop_p, seed, s);
#define CALL(LL, DD) launch_fwd<LL, DD>(table, ldt, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s, tokens)
    VQ_ATT_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

int vqcpc_relattn_tab_bwd(const float* d_ctx, int64_t ldo, const float* table, int64_t ldt, const int64_t* tokens,
                          const float* probs, const float* e1, const float* e2, float* d_qkv, int64_t ldg, float* d_e1,
                          float* d_e2, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(d_ctx && table && tokens && probs && e1 && e2 && d_qkv && d_e1 && d_e2 && workspace,
               "relattn_tab_bwd: null pointer");
    VQ_REQUIRE(!g_force_general && att_supported(L, H, hd), "relattn_tab_bwd: unsupported L=%d H=%d hd=%d", L, H, hd);
    VQ_REQUIRE(ldt % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0 && ldt >= 3 * H * hd && ldg >= 3 * H * hd && ldo >= H * hd &&
                   n_blocks >= 1,
               "relattn_tab_bwd: bad strides");
    if (workspace_bytes < vqcpc_relattn_bwd_workspace(n_blocks, L, H, hd)) {
        set_error("relattn_tab_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (use_mfma16(L, H, hd) && aligned16(table) && aligned16(d_ctx) && aligned16(e1) && aligned16(e2) && aligned16(d_qkv)) {
        int nsplit = 0;
        int rc = relattn16_bwd(d_ctx, ldo, table, ldt, tokens, probs, e1, e2, d_qkv, ldg, (float*)workspace, n_blocks, H, hd,
                               drop_p, seed, s, &nsplit);
        return rc ? rc : finish_de16((float*)workspace, nsplit, H, hd, d_e1, d_e2, s);
    }
#define CALL(LL, DD)                                                                                                   \
    launch_bwd<LL, DD>(d_ctx, ldo, table, ldt, probs, e1, e2, d_qkv, ldg, d_e1, d_e2, n_blocks, H, drop_p, seed,       \
                       (float*)workspace, s, tokens)
    VQ_ATT_DISPATCH(CALL)
#undef CALL
    return VQCPC_EINVAL;
}

}  // extern "C"
