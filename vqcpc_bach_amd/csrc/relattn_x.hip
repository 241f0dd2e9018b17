// Rectangular, masked attention with the learned relative bias: the three attention kinds of the decoder training step
// (decoders/decoder.py:431-543, transformer_custom.py:355-386):
//   * target self-attention      Lq = Lk = T, causal mask                (MultiheadAttentionCustom :171, attn_mask :314-316)
//   * source self-attention      Lq = Lk = S, anticausal (or no) mask
//   * cross-attention            Lq = T = r * S queries over Lk = S memory rows, anticausal (or no) mask, q and k | v from
//                                different tensors (:173-196)
// Relative bias = closed form of SubsampledRelativeAttention.forward (subsampled_relative_attention.py:30-122) for
// seq_len_tgt = r * seq_len_src.  With p(i) = i / r, the memory position query i is aligned with,
//     S[i][j] = qs_i . k_j + qs_i . Erel[j - p(i) + Lk - 1],     qs = q / sqrt(hd),
//     Erel[x] = e1[h][x] (x < Lk: j <= p) | e2[h][x - Lk + 1] (x >= Lk: j > p)
// (its -100 pad values never land on a kept entry; r = 1, p = i is the square form of the encoder path: relattn.hip routes
// every L other than 16 / 4 here -- the teacher's L = 384 and the auxiliary decoder's L = 24 / 96 of the student step).  The additive masks of
// decoder.py:292-308 are index rules in the same p:  causal = keep j <= p,  anticausal = keep j >= p; a masked logit is
// -inf in the reference, so its probability -- and with it every gradient term through it -- is exactly 0.
//
// Mapping: one wavefront (four when the key range has >= 4 tiles) owns a strip of 32 query rows of one (sequence, head)
// problem, walks the key
// tiles with v_mfma_f32_32x32x2_f32 (exact fp32 products), operand fragments are float4 loads straight from global
// memory (each lane half takes a contiguous half of the head dimension: the MFMA k index is a summation index), the
// relative term is one extra GEMM per strip against the band of relative rows the strip can see (rows
// [Lk-1-pmax, 2Lk-2-pmin], at most Lk + 31 of them), and the skew is an LDS read with a per-row offset pmax - p(ii).
// Backward = dq / dkv / de kernels over the same strips, deterministic (no atomics).
#include "common.h"

namespace vq {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kXMaxLk = 1024;
constexpr int kXWaves = 4;      // wavefronts per strip when the key range has at least that many 32-column tiles
constexpr float kNegBigX = -1.0e30f;

__device__ __forceinline__ int xrow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

static inline int x_tiles(int L) { return (L + 31) / 32; }
static inline int x_sw_fwd(int Lk) { return 32 * (x_tiles(Lk) + 1) + 4; }
static inline int x_sw_bwd(int Lk) { return 32 * x_tiles(Lk) + 64 + 4; }

template <int N>
__device__ __forceinline__ void xload_row(float (&dst)[N], const float* __restrict__ p, bool ok, float mul) {
#pragma unroll
    for (int v = 0; v < N / 4; ++v) {
        float4 t = ok ? *reinterpret_cast<const float4*>(p + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        dst[4 * v + 0] = t.x * mul;
        dst[4 * v + 1] = t.y * mul;
        dst[4 * v + 2] = t.z * mul;
        dst[4 * v + 3] = t.w * mul;
    }
}

__device__ __forceinline__ const float* xerel_row(const float* __restrict__ e1, const float* __restrict__ e2, int h, int Lk,
                                                  int HD, int x) {
    x = min(max(x, 0), 2 * Lk - 2);
    return x < Lk ? e1 + ((int64_t)h * Lk + x) * HD : e2 + ((int64_t)h * Lk + (x - Lk + 1)) * HD;
}

__device__ __forceinline__ bool x_keep(int mask, int j, int p) {
    return (mask == 0) | ((mask == 1) & (j <= p)) | ((mask == 2) & (j >= p));       // bitwise: no short-circuit branches
}
// drop_scale without its uniform early-out (thr == 0 keeps everything at inv_keep == 1): straight-line code, so that the
// loads of a whole tile can be issued before the first MFMA (a branch per element serialises load -> wait -> MFMA)
__device__ __forceinline__ float xdrop(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
    return rng_u24(seed, idx) >= thr ? inv_keep : 0.0f;
}

// =====================================================================================================================
// NW wavefronts share one strip (NW = 4 when there are >= 4 key tiles, else 1): band / score / P.V tiles and softmax rows
// are dealt round-robin to the waves, the partial P.V products are summed through LDS in wave order (deterministic).
// With one wave per 54 KB strip (L = 384) only 2 waves fit a CU; four waves per strip keep every SIMD busy.
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) void relattn_x_fwd_kernel(const float* __restrict__ q, int64_t ldq,
                                                                const float* __restrict__ k, int64_t ldk,
                                                                const float* __restrict__ v, int64_t ldv,
                                                                const float* __restrict__ e1, const float* __restrict__ e2,
                                                                float* __restrict__ ctx, int64_t ldo,
                                                                float* __restrict__ probs, int Lq, int Lk, int ratio, int H,
                                                                int mask, float scale, uint32_t thr, float inv_keep,
                                                                uint64_t seed) {
    constexpr int KH = HD / 2, CT = (HD + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 5, l31 = lane & 31;
    const int QT = (Lq + 31) / 32, KT = (Lk + 31) / 32, SW = 32 * (KT + 1) + 4;
    const int64_t prob = blockIdx.x / QT;
    const int i0 = (int)(blockIdx.x % QT) * 32;
    const int64_t n = prob / H;
    const int h = (int)(prob % H);
    const float* qbase = q + n * Lq * ldq + h * HD;
    const float* kbase = k + n * Lk * ldk + h * HD;
    const float* vbase = v + n * Lk * ldv + h * HD;
    const int pmax = (i0 + 31) / ratio;
    // key tiles the strip can see: causal keeps j <= p <= pmax, anticausal keeps j >= p >= pmin; the other tiles have
    // probability 0 and are never touched (half of the work of the causal target self-attention)
    const int jt0 = mask == 2 ? (i0 / ratio) / 32 : 0;
    const int KTe = mask == 1 ? min(KT, pmax / 32 + 1) : KT;
    const int jlo = 32 * jt0, jhi = min(Lk, 32 * KTe);

    float qa[KH];
    {
        const int i = i0 + l31;
        xload_row<KH>(qa, qbase + (int64_t)min(i, Lq - 1) * ldq + g * KH, i < Lq, scale);
    }
    int pofs[16], prow[16];              // per accumulator register: aligned memory position of its row, band offset
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        prow[r] = (i0 + xrow(r, lane)) / ratio;
        pofs[r] = pmax - prow[r];
    }
    // ---- phase 1: QE band  strip[ii][x] = qs_ii . Erel[rlo + x]
    const int rlo = Lk - 1 - pmax;
    for (int t = jt0 + wave; t <= KTe; t += NW) {
        float eb[KH];
        xload_row<KH>(eb, xerel_row(e1, e2, h, Lk, HD, rlo + 32 * t + l31) + g * KH, true, 1.0f);
        floatx16 acc = {0};
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], eb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) strip[xrow(r, lane) * SW + 32 * t + l31] = acc[r];
    }
    __syncthreads();
    // ---- phase 2: scores, in place.  Tile jt reads columns [32jt, 32jt+62] and writes [32jt, 32jt+31]: a round of NW
    // consecutive tiles reads everything before anyone writes; later rounds only read columns no earlier round writes
    for (int jb = jt0; jb < KTe; jb += NW) {
        const int jt = jb + wave;
        const bool act = jt < KTe;
        const int j = 32 * jt + l31;
        float sv[16];
        if (act) {
            float kb[KH];
            xload_row<KH>(kb, kbase + (int64_t)min(j, Lk - 1) * ldk + g * KH, j < Lk, 1.0f);
            floatx16 acc = {0};
#pragma unroll
            for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], kb[s], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = acc[r] + strip[xrow(r, lane) * SW + j + pofs[r]];
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                strip[xrow(r, lane) * SW + j] = (j < Lk && x_keep(mask, j, prow[r])) ? sv[r] : kNegBigX;
        }
    }
    __syncthreads();
    // ---- softmax by whole waves, kSR rows of a wave in flight (independent max / sum chains hide the LDS and
    // cross-lane latency of one row behind the others); probs saved BEFORE dropout
    const int rows = min(32, Lq - i0);
    constexpr int kSR = 4;
    for (int k0 = 0; k0 < 32 / NW; k0 += kSR) {
        float* row[kSR];
        bool live[kSR];
        float m[kSR], sum[kSR];
#pragma unroll
        for (int u = 0; u < kSR; ++u) {
            const int ii = wave + (k0 + u) * NW;
            live[u] = ii < rows;
            row[u] = strip + min(ii, 31) * SW;
            m[u] = kNegBigX;
            sum[u] = 0.0f;
        }
        if (!live[0]) break;                                         // rows are dealt in ascending order
        for (int j = jlo + lane; j < jhi; j += 64) {
#pragma unroll
            for (int u = 0; u < kSR; ++u) m[u] = fmaxf(m[u], row[u][j]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < kSR; ++u) m[u] = fmaxf(m[u], __shfl_xor(m[u], o, 64));
        }
        for (int j = jlo + lane; j < jhi; j += 64) {
#pragma unroll
            for (int u = 0; u < kSR; ++u) {
                const float e = __expf(row[u][j] - m[u]);
                if (live[u]) row[u][j] = e;
                sum[u] += e;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < kSR; ++u) sum[u] += __shfl_xor(sum[u], o, 64);
        }
#pragma unroll
        for (int u = 0; u < kSR; ++u) {
            if (!live[u]) continue;
            const float inv = 1.0f / sum[u];
            const int64_t pbase = (prob * Lq + i0 + wave + (k0 + u) * NW) * Lk;
            for (int j = lane; j < 32 * KT; j += 64) {
                float pd = 0.0f;
                if (j < Lk) {
                    const float p = (j >= jlo && j < jhi) ? row[u][j] * inv : 0.0f;
                    probs[pbase + j] = p;
                    pd = p * drop_scale(seed, (uint64_t)(pbase + j), thr, inv_keep);
                }
                row[u][j] = pd;
            }
        }
    }
    for (int ii = rows + wave; ii < 32; ii += NW)
        for (int j = lane; j < 32 * KT; j += 64) strip[ii * SW + j] = 0.0f;
    __syncthreads();
    // ---- ctx = Pd . V
    floatx16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) o[ct] = floatx16{0};
    for (int jt = jt0 + wave; jt < KTe; jt += NW) {
        float pa[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 t = *reinterpret_cast<const float4*>(strip + l31 * SW + 32 * jt + 16 * g + 4 * u);
            pa[4 * u] = t.x; pa[4 * u + 1] = t.y; pa[4 * u + 2] = t.z; pa[4 * u + 3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int jc = min(32 * jt + 16 * g + s, Lk - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + l31;
                const float vb = c < HD ? vbase[(int64_t)jc * ldv + c] : 0.0f;
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], vb, o[ct], 0, 0, 0);
            }
        }
    }
    if (NW == 1) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = 32 * ct + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + xrow(r, lane);
                if (i < Lq && c < HD) ctx[(n * Lq + i) * ldo + h * HD + c] = o[ct][r];
            }
        }
    } else {
        __syncthreads();                           // the strip is dead: its memory takes the partial products [NW][32][HD]
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = 32 * ct + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (c < HD) strip[(wave * 32 + xrow(r, lane)) * HD + c] = o[ct][r];
        }
        __syncthreads();
        for (int e = tid; e < 32 * HD; e += 64 * NW) {
            const int row = e / HD, c = e % HD;
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += strip[(w * 32 + row) * HD + c];
            if (i0 + row < Lq) ctx[(n * Lq + i0 + row) * ldo + h * HD + c] = sum;
        }
    }
}

// =====================================================================================================================
// dS strip + dq (same wave / tile assignment as the forward).  dS is also written to dSg [n][H][Lq][Lk] for the
// dkv / de kernels.
template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) void relattn_x_bwd_dq_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ k, int64_t ldk, const float* __restrict__ v,
    int64_t ldv, const float* __restrict__ probs, const float* __restrict__ e1, const float* __restrict__ e2,
    float* __restrict__ d_q, int64_t ldgq, float* __restrict__ dSg, int Lq, int Lk, int ratio, int H, int mask, float scale,
    uint32_t thr, float inv_keep, uint64_t seed) {
    constexpr int KH = HD / 2, CT = (HD + 31) / 32, OFF = 32;
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 5, l31 = lane & 31;
    const int QT = (Lq + 31) / 32, KT = (Lk + 31) / 32, SW = 32 * KT + 64 + 4;
    const int64_t prob = blockIdx.x / QT;
    const int i0 = (int)(blockIdx.x % QT) * 32;
    const int64_t n = prob / H;
    const int h = (int)(prob % H);
    const float* kbase = k + n * Lk * ldk + h * HD;
    const float* vbase = v + n * Lk * ldv + h * HD;
    const int pmax = (i0 + 31) / ratio;
    // visible key tiles (see the forward kernel); dSg is only written there, its readers apply the same rule
    const int jt0 = mask == 2 ? (i0 / ratio) / 32 : 0;
    const int KTe = mask == 1 ? min(KT, pmax / 32 + 1) : KT;
    float* rdbuf = strip + 32 * SW;                                  // [NW][32] row-sum partials

    for (int e = tid; e < 32 * SW; e += 64 * NW) strip[e] = 0.0f;
    float doa[KH];
    {
        const int i = i0 + l31;
        xload_row<KH>(doa, d_ctx + (n * Lq + min(i, Lq - 1)) * ldo + h * HD + g * KH, i < Lq, 1.0f);
    }
    __syncthreads();
    float rd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rd[r] = 0.0f;
    // pass 1: dP = (dO . V^T) * dropout mask, row sums of dP * P
    for (int jt = jt0 + wave; jt < KTe; jt += NW) {
        const int j = 32 * jt + l31;
        float vb[KH];
        xload_row<KH>(vb, vbase + (int64_t)min(j, Lk - 1) * ldv + g * KH, j < Lk, 1.0f);
        floatx16 acc = {0};
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(doa[s], vb[s], acc, 0, 0, 0);
        // P of the tile: 16 loads issued back to back (clamped addresses; `* okf` instead of a select keeps the compiler
        // from sinking each load behind its own branch, which serialises load -> wait -> use per element)
        float pl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            pl[r] = probs[(prob * Lq + min(i0 + xrow(r, lane), Lq - 1)) * Lk + min(j, Lk - 1)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = xrow(r, lane), i = i0 + ii;
            const float okf = (i < Lq && j < Lk) ? 1.0f : 0.0f;
            const int64_t idx = (prob * Lq + min(i, Lq - 1)) * Lk + min(j, Lk - 1);
            const float p = pl[r] * okf;
            const float dp = acc[r] * xdrop(seed, (uint64_t)idx, thr, inv_keep) * okf;
            rd[r] += dp * p;
            strip[ii * SW + OFF + j] = dp;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) rd[r] += __shfl_xor(rd[r], o, 64);
    }
    if (NW > 1) {                                                    // row sums across the waves, in wave order
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rdbuf[wave * 32 + xrow(r, lane)] = rd[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += rdbuf[w * 32 + xrow(r, lane)];
            rd[r] = t;
        }
    }
    // pass 2: dS = P (dP - rowsum); every lane revisits exactly the strip entries it wrote
    for (int jt = jt0 + wave; jt < KTe; jt += NW) {
        const int j = 32 * jt + l31;
        float pl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            pl[r] = probs[(prob * Lq + min(i0 + xrow(r, lane), Lq - 1)) * Lk + min(j, Lk - 1)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = xrow(r, lane), i = i0 + ii;
            const bool ok = i < Lq && j < Lk;
            const float ds = pl[r] * (ok ? 1.0f : 0.0f) * (strip[ii * SW + OFF + j] - rd[r]);
            strip[ii * SW + OFF + j] = ds;
            if (ok) dSg[(prob * Lq + i) * Lk + j] = ds;
        }
    }
    __syncthreads();
    floatx16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = floatx16{0};
    // dS . K
    for (int jt = jt0 + wave; jt < KTe; jt += NW) {
        float pa[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 t = *reinterpret_cast<const float4*>(strip + l31 * SW + OFF + 32 * jt + 16 * g + 4 * u);
            pa[4 * u] = t.x; pa[4 * u + 1] = t.y; pa[4 * u + 2] = t.z; pa[4 * u + 3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int jc = min(32 * jt + 16 * g + s, Lk - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + l31;
                const float kb = c < HD ? kbase[(int64_t)jc * ldk + c] : 0.0f;
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], kb, acc[ct], 0, 0, 0);
            }
        }
    }
    // skew(dS) . Erel_band :  skew[ii][x] = dS[ii][x + p(ii) - pmax]  (zero padding on both sides of the strip)
    const int rlo = Lk - 1 - pmax;
    const int shift = (i0 + l31) / ratio - pmax;              // in [-31, 0]
    for (int xt = jt0 + wave; xt <= KTe; xt += NW) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int x = 32 * xt + 16 * g + s;
            const float a = strip[l31 * SW + OFF + x + shift];
            const float* er = xerel_row(e1, e2, h, Lk, HD, rlo + x);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + l31;
                const float eb = c < HD ? er[c] : 0.0f;
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, eb, acc[ct], 0, 0, 0);
            }
        }
    }
    if (NW == 1) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = 32 * ct + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + xrow(r, lane);
                if (i < Lq && c < HD) d_q[(n * Lq + i) * ldgq + h * HD + c] = acc[ct][r] * scale;
            }
        }
    } else {
        __syncthreads();                           // the strip is dead: its memory takes the partial products [NW][32][HD]
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = 32 * ct + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (c < HD) strip[(wave * 32 + xrow(r, lane)) * HD + c] = acc[ct][r];
        }
        __syncthreads();
        for (int e = tid; e < 32 * HD; e += 64 * NW) {
            const int row = e / HD, c = e % HD;
            float sum = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += strip[(w * 32 + row) * HD + c];
            if (i0 + row < Lq) d_q[(n * Lq + i0 + row) * ldgq + h * HD + c] = sum * scale;
        }
    }
}

// =====================================================================================================================
// one wavefront per (problem, key tile): dV = (P * dropout)^T dO,  dK = dS^T qs
template <int HD>
__global__ __launch_bounds__(64) void relattn_x_bwd_dkv_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ q, int64_t ldq, const float* __restrict__ probs,
    const float* __restrict__ dSg, float* __restrict__ d_k, int64_t ldgk, float* __restrict__ d_v, int64_t ldgv, int Lq,
    int Lk, int ratio, int H, int mask, float scale, uint32_t thr, float inv_keep, uint64_t seed) {
    constexpr int CT = (HD + 31) / 32;
    const int lane = threadIdx.x, g = lane >> 5, l31 = lane & 31;
    const int QT = (Lq + 31) / 32, KT = (Lk + 31) / 32;
    const int64_t prob = blockIdx.x / KT;
    const int j0 = (int)(blockIdx.x % KT) * 32;
    const int64_t n = prob / H;
    const int h = (int)(prob % H);
    const int jA = j0 + l31;
    floatx16 dk[CT], dv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dk[ct] = dv[ct] = floatx16{0};
    // query tiles that can see this key tile: causal needs p(i) >= j0, anticausal p(i) <= j0 + 31
    const int it0 = mask == 1 ? (int)(((int64_t)j0 * ratio) / 32) : 0;
    const int itE = mask == 2 ? (int)min((int64_t)QT, (((int64_t)j0 + 32) * ratio - 1) / 32 + 1) : QT;
    const int jc = min(jA, Lk - 1);
    const int cc[2] = {min(l31, HD - 1), min(32 + l31, HD - 1)};      // clamped head columns (HD = 16: the upper lanes)
    for (int it = it0; it < itE; ++it) {
        // all loads of a half tile first (clamped addresses, no branches), selects afterwards, then the MFMAs
#pragma unroll
        for (int hs = 0; hs < 16; hs += 8) {
            float pv[8], dsv[8], dob[8][CT], qb[8][CT];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ic = min(32 * it + 16 * g + hs + u, Lq - 1);
                const int64_t idx = (prob * Lq + ic) * Lk + jc;
                pv[u] = probs[idx];
                dsv[u] = dSg[idx];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    dob[u][ct] = d_ctx[(n * Lq + ic) * ldo + h * HD + cc[ct & 1] + 64 * (ct >> 1)];
                    qb[u][ct] = q[(n * Lq + ic) * ldq + h * HD + cc[ct & 1] + 64 * (ct >> 1)];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = 32 * it + 16 * g + hs + u;
                const bool ok = i < Lq && jA < Lk && x_keep(mask, jA, i / ratio);
                const int64_t idx = (prob * Lq + min(i, Lq - 1)) * Lk + jc;
                const float p = ok ? pv[u] * xdrop(seed, (uint64_t)idx, thr, inv_keep) : 0.0f;
                const float ds = ok ? dsv[u] : 0.0f;              // a select: unwritten (masked) dS entries may hold anything
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bool cok = 32 * ct + l31 < HD;
                    dv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, cok ? dob[u][ct] : 0.0f, dv[ct], 0, 0, 0);
                    dk[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, cok ? qb[u][ct] * scale : 0.0f, dk[ct], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + xrow(r, lane);
            if (j < Lk && c < HD) {
                d_k[(n * Lk + j) * ldgk + h * HD + c] = dk[ct][r];
                d_v[(n * Lk + j) * ldgv + h * HD + c] = dv[ct][r];
            }
        }
    }
}

// =====================================================================================================================
// dErel[x] = sum_n sum_i dS[i][x + p(i) - (Lk-1)] qs_i.   grid = (H * RT, chunks), ws [chunk][H][2Lk-1][HD]
template <int HD>
__global__ __launch_bounds__(64) void relattn_x_bwd_de_kernel(const float* __restrict__ q, int64_t ldq,
                                                              const float* __restrict__ dSg, float* __restrict__ ws,
                                                              int64_t n_seq, int seq_per_chunk, int Lq, int Lk, int ratio,
                                                              int H, int mask, float scale) {
    constexpr int CT = (HD + 31) / 32;
    const int lane = threadIdx.x, g = lane >> 5, l31 = lane & 31;
    const int QT = (Lq + 31) / 32, NE = 2 * Lk - 1, RT = (NE + 31) / 32;
    const int h = blockIdx.x / RT;
    const int r0 = (blockIdx.x % RT) * 32;
    const int rA = r0 + l31;
    floatx16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = floatx16{0};
    const int cc[2] = {min(l31, HD - 1), min(32 + l31, HD - 1)};      // clamped head columns (HD = 16: the upper lanes)
    const int64_t n_begin = (int64_t)blockIdx.y * seq_per_chunk;
    // relative rows x >= Lk (e2) pair keys j > p, rows x < Lk - 1 keys j < p: a causal / anticausal mask leaves them zero
    const bool dead = (mask == 1 && r0 >= Lk) || (mask == 2 && r0 + 31 < Lk - 1);
    const int64_t n_end = dead ? n_begin : min(n_begin + seq_per_chunk, n_seq);
    for (int64_t n = n_begin; n < n_end; ++n) {
        const int64_t prob = n * H + h;
        for (int it = 0; it < QT; ++it) {
            const int jlo = r0 + (32 * it) / ratio - (Lk - 1);              // key range this tile pair can touch
            const int jhi = r0 + 31 + (32 * it + 31) / ratio - (Lk - 1);
            if (jhi < 0 || jlo >= Lk) continue;
            // loads of a half tile first (clamped addresses, no branches), selects afterwards, then the MFMAs
#pragma unroll
            for (int hs = 0; hs < 16; hs += 8) {
                float av[8], qb[8][CT];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ic = min(32 * it + 16 * g + hs + u, Lq - 1);
                    const int jc = min(max(rA + ic / ratio - (Lk - 1), 0), Lk - 1);
                    av[u] = dSg[(prob * Lq + ic) * Lk + jc];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        qb[u][ct] = q[(n * Lq + ic) * ldq + h * HD + cc[ct & 1] + 64 * (ct >> 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = 32 * it + 16 * g + hs + u;
                    const int pi = i / ratio;
                    const int j = rA + pi - (Lk - 1);
                    const bool ok = i < Lq && j >= 0 && j < Lk && rA < NE && x_keep(mask, j, pi);
                    const float a = ok ? av[u] : 0.0f;            // a select: unwritten (masked) dS entries may hold anything
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const bool cok = 32 * ct + l31 < HD;
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cok ? qb[u][ct] * scale : 0.0f, acc[ct], 0, 0, 0);
                    }
                }
            }
        }
    }
    float* dst = ws + ((int64_t)blockIdx.y * H + h) * NE * HD;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + xrow(r, lane);
            if (rr < NE && c < HD) dst[(int64_t)rr * HD + c] = acc[ct][r];
        }
    }
}

__global__ __launch_bounds__(256) void relattn_x_de_split(const float* __restrict__ tot, int H, int L, int HD,
                                                          float* __restrict__ d_e1, float* __restrict__ d_e2) {
    const int NE = 2 * L - 1;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= H * NE * HD) return;
    const int c = o % HD, r = (o / HD) % NE, h = o / (HD * NE);
    const float acc = tot[o];
    if (r < L) d_e1[((int64_t)h * L + r) * HD + c] = acc;
    else d_e2[((int64_t)h * L + (r - L + 1)) * HD + c] = acc;
    if (r == 0) d_e2[((int64_t)h * L) * HD + c] = 0.0f;      // e2 row 0 is never read by the closed form
}

// ---- host side ------------------------------------------------------------------------------------------------------
static int x_chunks(int64_t n_seq, int Lk, int H) {
    const int RT = (2 * Lk - 1 + 31) / 32;
    const int64_t want = std::max<int64_t>(1, 8192 / ((int64_t)H * RT));
    return (int)std::min<int64_t>(n_seq, want);
}

static bool x_supported(int Lq, int Lk, int H, int hd) {
    return Lk >= 1 && Lk <= kXMaxLk && Lq >= Lk && Lq % Lk == 0 && H >= 1 && (hd == 16 || hd == 32 || hd == 64 || hd == 128);
}

template <int HD>
static int x_fwd_t(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* e1,
                   const float* e2, float* ctx, int64_t ldo, float* probs, int64_t n_seq, int Lq, int Lk, int H, int mask,
                   float drop_p, uint64_t seed, hipStream_t s) {
    const int64_t grid = n_seq * H * x_tiles(Lq);
    const float scale = 1.0f / sqrtf((float)HD), inv_keep = 1.0f / (1.0f - drop_p);
    const uint32_t thr = drop_threshold(drop_p);
    if (x_tiles(Lk) >= kXWaves) {
        const size_t lds = sizeof(float) * std::max<size_t>((size_t)32 * x_sw_fwd(Lk), (size_t)kXWaves * 32 * HD);
        auto kern = relattn_x_fwd_kernel<HD, kXWaves>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kXWaves), lds, s, q, ldq, k, ldk, v, ldv, e1, e2, ctx, ldo,
                           probs, Lq, Lk, Lq / Lk, H, mask, scale, thr, inv_keep, seed);
    } else {
        const size_t lds = (size_t)32 * x_sw_fwd(Lk) * sizeof(float);
        hipLaunchKernelGGL((relattn_x_fwd_kernel<HD, 1>), dim3((unsigned)grid), dim3(64), lds, s, q, ldq, k, ldk, v, ldv, e1,
                           e2, ctx, ldo, probs, Lq, Lk, Lq / Lk, H, mask, scale, thr, inv_keep, seed);
    }
    VQ_CHECK_LAUNCH("relattn_x_fwd");
    return VQCPC_OK;
}

template <int HD>
static int x_bwd_t(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                   int64_t ldv, const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, float* d_k,
                   int64_t ldgk, float* d_v, int64_t ldgv, float* d_e1, float* d_e2, int64_t n_seq, int Lq, int Lk, int H,
                   int mask, float drop_p, uint64_t seed, float* ws, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)HD), inv_keep = 1.0f / (1.0f - drop_p);
    const uint32_t thr = drop_threshold(drop_p);
    const int ratio = Lq / Lk;
    float* dSg = ws;
    float* part = ws + round_up(n_seq * H * (int64_t)Lq * Lk, 64);
    if (x_tiles(Lk) >= kXWaves) {
        const size_t lds = sizeof(float) * std::max<size_t>((size_t)32 * x_sw_bwd(Lk) + kXWaves * 32, (size_t)kXWaves * 32 * HD);
        auto kern = relattn_x_bwd_dq_kernel<HD, kXWaves>;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(n_seq * H * x_tiles(Lq))), dim3(64 * kXWaves), lds, s, d_ctx, ldo, k, ldk, v,
                           ldv, probs, e1, e2, d_q, ldgq, dSg, Lq, Lk, ratio, H, mask, scale, thr, inv_keep, seed);
        VQ_CHECK_LAUNCH("relattn_x_bwd_dq");
    } else {
        const size_t lds = ((size_t)32 * x_sw_bwd(Lk) + 32) * sizeof(float);
        hipLaunchKernelGGL((relattn_x_bwd_dq_kernel<HD, 1>), dim3((unsigned)(n_seq * H * x_tiles(Lq))), dim3(64), lds, s, d_ctx,
                           ldo, k, ldk, v, ldv, probs, e1, e2, d_q, ldgq, dSg, Lq, Lk, ratio, H, mask, scale, thr, inv_keep, seed);
        VQ_CHECK_LAUNCH("relattn_x_bwd_dq");
    }
    hipLaunchKernelGGL(relattn_x_bwd_dkv_kernel<HD>, dim3((unsigned)(n_seq * H * x_tiles(Lk))), dim3(64), 0, s, d_ctx, ldo, q,
                       ldq, probs, dSg, d_k, ldgk, d_v, ldgv, Lq, Lk, ratio, H, mask, scale, thr, inv_keep, seed);
    VQ_CHECK_LAUNCH("relattn_x_bwd_dkv");
    const int chunks = x_chunks(n_seq, Lk, H);
    const int spc = (int)ceil_div(n_seq, chunks);
    const int nchunk = (int)ceil_div(n_seq, spc);
    const int RT = (2 * Lk - 1 + 31) / 32;
    hipLaunchKernelGGL(relattn_x_bwd_de_kernel<HD>, dim3(H * RT, nchunk), dim3(64), 0, s, q, ldq, dSg, part, n_seq, spc, Lq,
                       Lk, ratio, H, mask, scale);
    VQ_CHECK_LAUNCH("relattn_x_bwd_de");
    const int total = H * (2 * Lk - 1) * HD;
    float* tot = part + (int64_t)chunks * total;
    int rc = launch_reduce_splits(part, total, nchunk, tot, total, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(relattn_x_de_split, dim3(ceil_div(total, 256)), dim3(256), 0, s, tot, H, Lk, HD, d_e1, d_e2);
    VQ_CHECK_LAUNCH("relattn_x_de_split");
    return VQCPC_OK;
}

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_relattn_x_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                        const float* e1, const float* e2, float* ctx, int64_t ldo, float* probs, int64_t n_seq, int Lq,
                        int Lk, int H, int hd, int mask, float drop_p, uint64_t seed, void* stream) {
    if (n_seq == 0) return VQCPC_OK;
    VQ_REQUIRE(q && k && v && e1 && e2 && ctx && probs, "relattn_x_fwd: null pointer");
    VQ_REQUIRE(x_supported(Lq, Lk, H, hd), "relattn_x_fwd: unsupported Lq=%d Lk=%d H=%d hd=%d", Lq, Lk, H, hd);
    VQ_REQUIRE(mask >= 0 && mask <= 2, "relattn_x_fwd: mask must be 0 (none), 1 (causal) or 2 (anticausal)");
    VQ_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && ldq >= H * hd && ldk >= H * hd &&
                   ldv >= H * hd && ldo >= H * hd && n_seq >= 0 && aligned16(q) && aligned16(k) && aligned16(v),
               "relattn_x_fwd: bad strides / alignment");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn_x_fwd: bad dropout probability");
    hipStream_t s = (hipStream_t)stream;
#define CALL(DD) x_fwd_t<DD>(q, ldq, k, ldk, v, ldv, e1, e2, ctx, ldo, probs, n_seq, Lq, Lk, H, mask, drop_p, seed, s)
    if (hd == 16) return CALL(16);
    if (hd == 32) return CALL(32);
    if (hd == 64) return CALL(64);
    return CALL(128);
#undef CALL
}

int64_t vqcpc_relattn_x_bwd_workspace(int64_t n_seq, int Lq, int Lk, int H, int hd) {
    n_seq = std::max<int64_t>(n_seq, 1);
    const int64_t ds = n_seq * H * (int64_t)Lq * Lk;
    const int64_t part = ((int64_t)x_chunks(n_seq, std::max(Lk, 1), std::max(H, 1)) + 1) * H * (2 * Lk - 1) * hd;
    return (round_up(ds, 64) + part) * (int64_t)sizeof(float);
}

int vqcpc_relattn_x_bwd(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* k, int64_t ldk,
                        const float* v, int64_t ldv, const float* probs, const float* e1, const float* e2, float* d_q,
                        int64_t ldgq, float* d_k, int64_t ldgk, float* d_v, int64_t ldgv, float* d_e1, float* d_e2,
                        int64_t n_seq, int Lq, int Lk, int H, int hd, int mask, float drop_p, uint64_t seed, void* workspace,
                        int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(d_ctx && q && k && v && probs && e1 && e2 && d_q && d_k && d_v && d_e1 && d_e2 && workspace,
               "relattn_x_bwd: null pointer");
    VQ_REQUIRE(x_supported(Lq, Lk, H, hd), "relattn_x_bwd: unsupported Lq=%d Lk=%d H=%d hd=%d", Lq, Lk, H, hd);
    VQ_REQUIRE(mask >= 0 && mask <= 2, "relattn_x_bwd: mask must be 0 (none), 1 (causal) or 2 (anticausal)");
    VQ_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && ldq >= H * hd && ldk >= H * hd &&
                   ldv >= H * hd && ldo >= H * hd && ldgq >= H * hd && ldgk >= H * hd && ldgv >= H * hd && n_seq >= 1 &&
                   aligned16(d_ctx) && aligned16(k) && aligned16(v),
               "relattn_x_bwd: bad strides / alignment");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relattn_x_bwd: bad dropout probability");
    if (workspace_bytes < vqcpc_relattn_x_bwd_workspace(n_seq, Lq, Lk, H, hd)) {
        set_error("relattn_x_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
#define CALL(DD)                                                                                                        \
    x_bwd_t<DD>(d_ctx, ldo, q, ldq, k, ldk, v, ldv, probs, e1, e2, d_q, ldgq, d_k, ldgk, d_v, ldgv, d_e1, d_e2, n_seq, Lq, Lk, \
                H, mask, drop_p, seed, (float*)workspace, s)
    if (hd == 16) return CALL(16);
    if (hd == 32) return CALL(32);
    if (hd == 64) return CALL(64);
    return CALL(128);
#undef CALL
}

}  // extern "C"
