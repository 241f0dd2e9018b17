// L = 16 self-attention with the closed-form relative bias on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
// One wavefront owns one (block, head) problem: the 16 x 16 score tile is ONE MFMA accumulator (4 registers per lane).
// The LDS-tiled VALU kernels of relattn.hip read every operand of every FMA from LDS (~225 KB of LDS traffic per problem
// in the backward: 0.9 ms of pure LDS time at C1); here the operands of the contractions over hd are float4 row loads
// straight from global / L2 (the MFMA k index is a summation index, so lane group g takes the contiguous columns
// [g hd/4, (g+1) hd/4) of its row) and only the two things that need a transposed or skewed view -- P for P.V / dS.K and
// the relative term -- pass through a 16 x 33 float LDS tile per wave.
//   MFMA layouts (l = lane, g = l >> 4, c = l & 15):  A[row c][k g]   B[k g][col c]   D[row 4g + r][col c], r = 0..3
// Head columns are mapped to (column tile ct, lane c) as column = CT*c + ct (CT = hd/16 tiles): the CT values a lane
// feeds to / receives from the CT column tiles are contiguous in memory -> one 8/16-byte access per lane, full 128-byte
// lines per 16-lane group instead of CT half-line dword accesses.
// Same semantics / buffers as relattn.hip (probs saved before dropout, dropout index ((prob*16 + i)*16 + j),
// S[i][j] = qs_i.k_j + qs_i.Erel[j - i + 15], Erel[r] = e1[h][r] (r < 16) | e2[h][r - 15] (r >= 16)), optional token
// indirection into the first layer's block table.
#include "gemm_common.h"

namespace vq {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kA16Waves = 4;
constexpr int kA16RS = 33;              // LDS row stride of the per-wave 16 x 32 tile

__device__ __forceinline__ const float* erel16(const float* __restrict__ e1, const float* __restrict__ e2, int h, int HD,
                                               int x) {
    x = min(x, 30);
    return x < 16 ? e1 + ((int64_t)h * 16 + x) * HD : e2 + ((int64_t)h * 16 + (x - 15)) * HD;
}

// The LDS tile is private to a wavefront and LDS instructions of one wave execute in order, so a cross-lane exchange
// needs no s_barrier: only the compiler must keep the program order of the accesses around this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N>
__device__ __forceinline__ void load_f4(float (&dst)[N], const float* __restrict__ p, float mul) {
#pragma unroll
    for (int v = 0; v < N / 4; ++v) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * v);
        dst[4 * v] = t.x * mul; dst[4 * v + 1] = t.y * mul; dst[4 * v + 2] = t.z * mul; dst[4 * v + 3] = t.w * mul;
    }
}

template <int CT>
__device__ __forceinline__ void load_ct(float (&dst)[CT], const float* __restrict__ p, float mul) {
    if (CT == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        dst[0] = t.x * mul; dst[1] = t.y * mul; dst[2 % CT] = t.z * mul; dst[3 % CT] = t.w * mul;
    } else if (CT == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        dst[0] = t.x * mul; dst[1 % CT] = t.y * mul;
    } else {
        dst[0] = p[0] * mul;
    }
}
template <int CT>
__device__ __forceinline__ void store_ct(float* __restrict__ p, const float (&v)[CT]) {
    if (CT == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % CT], v[2 % CT], v[3 % CT]);
    else if (CT == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1 % CT]);
    else p[0] = v[0];
}

// bf16 outputs of the bf16 training path (BASELINE configs[4]): the attention context / its input gradients only feed GEMMs
// there, which read bf16 from HBM -- the kernels round on the way out instead of writing fp32 for a cast pass to re-read.
// (round 5: the hardware conversion, v_cvt_pk_bf16_f32 -- round to nearest even, NaN stays a quiet NaN; the integer form it replaces
// was ~6 VALU instructions per element, 290 of the 1 580 per (block, head) problem of the head_dim 64 backward)
template <int CT>
__device__ __forceinline__ void store_ct_b16(unsigned short* __restrict__ p, const float (&v)[CT]) {
    if (CT == 4) {
        *reinterpret_cast<uint2*>(p) = make_uint2(bf16x2_rn(v[0], v[1 % CT]), bf16x2_rn(v[2 % CT], v[3 % CT]));
    } else if (CT == 2) {
        *reinterpret_cast<uint32_t*>(p) = bf16x2_rn(v[0], v[1 % CT]);
    } else {
        p[0] = (unsigned short)(bf16x2_rn(v[0], 0.0f) & 0xFFFFu);
    }
}
template <int CT, bool B16>
__device__ __forceinline__ void store_out(float* __restrict__ base, int64_t off, const float (&v)[CT]) {
    if (B16) store_ct_b16<CT>(reinterpret_cast<unsigned short*>(base) + off, v);
    else store_ct<CT>(base + off, v);
}


// bf16 INPUTS of the bf16 training path (IN16): q | k | v and d ctx are bf16 in HBM (written so by the in_proj / out-proj-dgrad
// GEMM epilogues); offsets and leading dimensions are in elements either way.
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
template <int N, bool IN16>
__device__ __forceinline__ void load_vec(float (&dst)[N], const float* __restrict__ base, int64_t off, float mul) {
    if constexpr (!IN16) {
        load_f4<N>(dst, base + off, mul);
    } else {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + off;
        if constexpr (N == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(p);
            dst[0] = bf16_lo(t.x) * mul; dst[1] = bf16_hi(t.x) * mul; dst[2] = bf16_lo(t.y) * mul; dst[3] = bf16_hi(t.y) * mul;
        } else {
#pragma unroll
            for (int v = 0; v < N / 8; ++v) {
                const uint4 t = *reinterpret_cast<const uint4*>(p + 8 * v);
                dst[8 * v] = bf16_lo(t.x) * mul; dst[8 * v + 1] = bf16_hi(t.x) * mul;
                dst[8 * v + 2] = bf16_lo(t.y) * mul; dst[8 * v + 3] = bf16_hi(t.y) * mul;
                dst[8 * v + 4] = bf16_lo(t.z) * mul; dst[8 * v + 5] = bf16_hi(t.z) * mul;
                dst[8 * v + 6] = bf16_lo(t.w) * mul; dst[8 * v + 7] = bf16_hi(t.w) * mul;
            }
        }
    }
}
template <int CT, bool IN16>
__device__ __forceinline__ void load_cols(float (&dst)[CT], const float* __restrict__ base, int64_t off, float mul) {
    if constexpr (!IN16) {
        load_ct<CT>(dst, base + off, mul);
    } else {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + off;
        if constexpr (CT == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(p);
            dst[0] = bf16_lo(t.x) * mul; dst[1] = bf16_hi(t.x) * mul; dst[2 % CT] = bf16_lo(t.y) * mul; dst[3 % CT] = bf16_hi(t.y) * mul;
        } else if constexpr (CT == 2) {
            const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
            dst[0] = bf16_lo(t) * mul; dst[1 % CT] = bf16_hi(t) * mul;
        } else {
            dst[0] = __uint_as_float((uint32_t)p[0] << 16) * mul;
        }
    }
}

// exact 3-way bf16 split of a lane's 8 contraction elements (gemm_common.h: x == h + m + l): the operand form of
// v_mfma_f32_16x16x32_bf16, whose lane (row c, k group g) holds k = 8 g .. 8 g + 7 -- the very columns the fp32 kernels
// assign to lane group g for head_dim 32
__device__ __forceinline__ void split8x3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    uint2 h0, m0, l0, h1, m1, l1;
    split3x4(make_float4(x[0], x[1], x[2], x[3]), h0, m0, l0);
    split3x4(make_float4(x[4], x[5], x[6], x[7]), h1, m1, l1);
    h = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    m = __builtin_bit_cast(bf16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
    l = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}
// a . b over 32 contraction elements as six bf16 MFMAs (hl + lh + mm, hm + mh, hh: smallest terms first), fp32 accumulate:
// fp32-class accuracy (the bf16x6 arithmetic of the GEMMs) at 6 x 16x16x32 instead of 8 dependent 16x16x4 fp32 MFMAs
__device__ __forceinline__ floatx4 dot32_x6(const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8& bh,
                                            const bf16x8& bm, const bf16x8& bl, floatx4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ float grp16_sum(float v) {
    v += __shfl_xor(v, 1, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 8, 16);
    return v;
}
__device__ __forceinline__ float grp16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 16)); v = fmaxf(v, __shfl_xor(v, 2, 16));
    v = fmaxf(v, __shfl_xor(v, 4, 16)); v = fmaxf(v, __shfl_xor(v, 8, 16));
    return v;
}

// =====================================================================================================================
// X6 (head_dim 32 / 64): q . k and q . Erel on the bf16 matrix pipe as (up to) six products of the exact 3-way split
template <int HD, bool B16 = false, bool IN16 = false, bool X6 = false>      // B16: `ctx` points to bf16 elements (ldo in elements); IN16: `qkv` too
__global__ __launch_bounds__(kA16Waves * 64) void relattn16_fwd_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                                       const int64_t* __restrict__ tokens,
                                                                       const float* __restrict__ e1,
                                                                       const float* __restrict__ e2, float* __restrict__ ctx,
                                                                       int64_t ldo, float* __restrict__ probs,
                                                                       int64_t total, int H, float scale, uint32_t thr,
                                                                       float inv_keep, uint64_t seed) {
    constexpr int KH = HD / 4, CT = HD / 16;
    __shared__ float lds[kA16Waves][16 * kA16RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    float* buf = lds[wave];
    const int64_t prob = min((int64_t)blockIdx.x * kA16Waves + wave, total - 1);       // tail waves redo the last problem
    const bool live = (int64_t)blockIdx.x * kA16Waves + wave < total;
    const int64_t n = prob / H;
    const int h = (int)(prob % H);
    const int d = H * HD;
    const int64_t tokv = tokens ? tokens[n * 16 + c] : 0;
    const int64_t row_c = tokens ? tokv * 16 + c : n * 16 + c;                         // qkv / table row of token c
    const int64_t ro = row_c * ldq + h * HD + g * KH;

    float qa[KH], kb[KH];
    load_vec<KH, IN16>(qa, qkv, ro, scale);
    load_vec<KH, IN16>(kb, qkv, ro + d, 1.0f);
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
    if constexpr (X6 && (HD == 32 || HD == 64)) {
        // head_dim 64: a lane holds 16 contraction elements = two chunks of 8 (columns 16 g + 8 ch ..): chunk ch of every lane
        // is one K = 32 MFMA block (the k index is a summation index: any assignment shared by both operands is valid).
        // bf16 INPUTS (IN16) are single pieces: k always, q when the scale 1 / sqrt(head_dim) is a power of two (head_dim 64:
        // q / 8 is a bf16 number): q . k is then ONE exact bf16 MFMA per chunk, q . Erel three (Erel is fp32).
        constexpr int NCH = KH / 8;
        constexpr bool Q1 = IN16 && HD == 64, K1 = IN16;
        bf16x8 qh[NCH], qm[NCH], ql[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            float q8[8], k8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { q8[k] = qa[8 * ch + k]; k8[k] = kb[8 * ch + k]; }
            bf16x8 kh_, km, kl;
            split8x3(q8, qh[ch], qm[ch], ql[ch]);
            split8x3(k8, kh_, km, kl);
            if constexpr (Q1 && K1) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[ch], kh_, s, 0, 0, 0);
            } else if constexpr (K1) {                  // q: three pieces, k: one
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql[ch], kh_, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qm[ch], kh_, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[ch], kh_, s, 0, 0, 0);
            } else {
                s = dot32_x6(qh[ch], qm[ch], ql[ch], kh_, km, kl, s);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float eb[KH];
            load_f4<KH>(eb, erel16(e1, e2, h, HD, 16 * t + c) + g * KH, 1.0f);
            floatx4 qe = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                float e8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e8[k] = eb[8 * ch + k];
                bf16x8 eh, em, el;
                split8x3(e8, eh, em, el);
                if constexpr (Q1) {                     // q: one piece, Erel: three
                    qe = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[ch], el, qe, 0, 0, 0);
                    qe = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[ch], em, qe, 0, 0, 0);
                    qe = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh[ch], eh, qe, 0, 0, 0);
                } else {
                    qe = dot32_x6(qh[ch], qm[ch], ql[ch], eh, em, el, qe);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[(4 * g + r) * kA16RS + 16 * t + c] = qe[r];
        }
    } else {
#pragma unroll
    for (int k = 0; k < KH; ++k) s = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[k], kb[k], s, 0, 0, 0);
    // relative term: QE[i][x] = qs_i . Erel[x], x = 0..30 (two 16-column tiles), skewed into the scores through LDS
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float eb[KH];
        load_f4<KH>(eb, erel16(e1, e2, h, HD, 16 * t + c) + g * KH, 1.0f);
        floatx4 qe = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KH; ++k) qe = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[k], eb[k], qe, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(4 * g + r) * kA16RS + 16 * t + c] = qe[r];
    }
    }
    wave_lds_fence();
    float p[4], pd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        const float v = s[r] + buf[i * kA16RS + c - i + 15];
        const float m = grp16_max(v);
        const float e = __expf(v - m);
        p[r] = e / grp16_sum(e);
        const int64_t idx = (prob * 16 + i) * 16 + c;
        if (live) probs[idx] = p[r];
        pd[r] = p[r] * drop_scale(seed, (uint64_t)idx, thr, inv_keep);
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(4 * g + r) * kA16RS + c] = pd[r];
    wave_lds_fence();
    // ctx = Pd . V :  A[row i = c][k j = 4g + s] = Pd[c][4g + s]   B[k j][col] = V[j][col]
    floatx4 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) o[ct] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        const int j = 4 * g + sidx;
        const float pa = buf[c * kA16RS + j];
        const int64_t tj = __shfl(tokv, j, 16);
        const int64_t row_j = tokens ? tj * 16 + j : n * 16 + j;
        float vb[CT];
        load_cols<CT, IN16>(vb, qkv, row_j * ldq + 2 * d + h * HD + CT * c, 1.0f);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) o[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vb[ct], o[ct], 0, 0, 0);
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) v[ct] = o[ct][r];
            store_out<CT, B16>(ctx, (n * 16 + 4 * g + r) * ldo + h * HD + CT * c, v);
        }
    }
}

// =====================================================================================================================
// grid = (chunks, H).  Wave w of a workgroup walks the blocks chunk*bpc + w, + 4, ... of head blockIdx.y and keeps the
// relative-embedding gradient of its head in registers; partials ws[(chunk*4 + w)][H][31][HD] (deterministic reduce).
template <int HD, bool B16 = false, bool IN16 = false>      // B16: `d_qkv` points to bf16 elements (ldg in elements); IN16: `qkv`, `d_ctx` too
__global__ __launch_bounds__(kA16Waves * 64) void relattn16_bwd_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ qkv, int64_t ldq,
    const int64_t* __restrict__ tokens, const float* __restrict__ probs, const float* __restrict__ e1,
    const float* __restrict__ e2, float* __restrict__ d_qkv, int64_t ldg, float* __restrict__ ws, int64_t n_blocks, int H,
    int blocks_per_chunk, float scale, uint32_t thr, float inv_keep, uint64_t seed) {
    constexpr int KH = HD / 4, CT = HD / 16;
    __shared__ float lds[kA16Waves][16 * kA16RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    float* buf = lds[wave];
    const int h = blockIdx.y;
    const int d = H * HD;
    floatx4 de[2][CT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) de[t][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int64_t b0 = (int64_t)blockIdx.x * blocks_per_chunk;
    const int64_t b1 = min(b0 + blocks_per_chunk, n_blocks);
    // every wave runs the same number of iterations (barriers inside); out-of-range waves redo block b1 - 1 without stores
    const int iters = (int)((b1 - b0 + kA16Waves - 1) / kA16Waves);
    // Software pipeline (head_dim <= 32): every global operand of block it + 1 is requested while block it is processed --
    // a block is otherwise four dependent round trips (token ids -> operand rows -> probs -> second operand set) with two
    // waves per SIMD to hide them.  The token ids run one block further ahead (they address the operand rows).
    constexpr bool kPipe = HD <= 32;
    struct Blk {
        float doa[KH], vb[KH], p[4], dob[4][CT], qb[4][CT], kb[4][CT];
    };
    auto block_of = [&](int it) { return min(b0 + (int64_t)it * kA16Waves + wave, b1 - 1); };
    auto load_tok = [&](int64_t n) -> int64_t { return tokens ? tokens[n * 16 + c] : 0; };
    auto load_blk = [&](Blk& B, int64_t n, int64_t tokv) {
        const int64_t prob = n * H + h;
        const int64_t row_c = tokens ? tokv * 16 + c : n * 16 + c;
        load_vec<KH, IN16>(B.doa, d_ctx, (n * 16 + c) * ldo + h * HD + g * KH, 1.0f);
        load_vec<KH, IN16>(B.vb, qkv, row_c * ldq + h * HD + 2 * d + g * KH, 1.0f);
#pragma unroll
        for (int r = 0; r < 4; ++r) B.p[r] = probs[(prob * 16 + 4 * g + r) * 16 + c];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int j = 4 * g + sidx;
            const int64_t tj = __shfl(tokv, j, 16);
            const int64_t row = tokens ? tj * 16 + j : n * 16 + j;
            load_cols<CT, IN16>(B.dob[sidx], d_ctx, (n * 16 + j) * ldo + h * HD + CT * c, 1.0f);
            load_cols<CT, IN16>(B.qb[sidx], qkv, row * ldq + h * HD + CT * c, scale);
            load_cols<CT, IN16>(B.kb[sidx], qkv, row * ldq + d + h * HD + CT * c, 1.0f);
        }
    };
    Blk cur;
    int64_t tok_n1 = 0;                                   // token ids of block it + 1
    if (kPipe) {
        load_blk(cur, block_of(0), load_tok(block_of(0)));
        tok_n1 = load_tok(block_of(1));
    }
    for (int it = 0; it < iters; ++it) {
        const int64_t want = b0 + (int64_t)it * kA16Waves + wave;
        const bool live = want < b1;
        const int64_t n = live ? want : b1 - 1;
        const int64_t prob = n * H + h;
        Blk nxt;
        if (kPipe) {
            load_blk(nxt, block_of(it + 1), tok_n1);
            tok_n1 = load_tok(block_of(it + 2));
        } else {
            load_blk(cur, n, load_tok(n));
        }
        // dP = dO . V^T, softmax backward in the accumulator layout (rows 4g + r, column c)
        floatx4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KH; ++k) dp = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.doa[k], cur.vb[k], dp, 0, 0, 0);
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t idx = (prob * 16 + 4 * g + r) * 16 + c;
            const float p = cur.p[r];
            const float mk = drop_scale(seed, (uint64_t)idx, thr, inv_keep);
            const float dpm = dp[r] * mk;
            pd[r] = p * mk;
            ds[r] = p * (dpm - grp16_sum(dpm * p));
        }
        // dV = Pd^T dO, dK = dS^T qs :  A[row j = c][k i = 4g + s] = X[4g + s][c] = this lane's register s
        floatx4 dv[CT], dk[CT], dq[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dv[ct] = dk[ct] = dq[ct] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                dv[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(pd[sidx], cur.dob[sidx][ct], dv[ct], 0, 0, 0);
                dk[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[sidx], cur.qb[sidx][ct], dk[ct], 0, 0, 0);
            }
        }
        wave_lds_fence();                                  // previous iteration's LDS readers are done
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(4 * g + r) * kA16RS + c] = ds[r];
        wave_lds_fence();
        // dq = scale * (dS . K + skew(dS) . Erel):  A[row i = c][k] from the LDS tile
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const float a = buf[c * kA16RS + 4 * g + sidx];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                dq[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, cur.kb[sidx][ct], dq[ct], 0, 0, 0);
        }
#pragma unroll
        for (int xt = 0; xt < 2; ++xt)
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int x = 16 * xt + 4 * g + sidx;          // relative row
                const int jj = x + c - 15;                     // key index seen from query c
                const float a = (x <= 30 && jj >= 0 && jj < 16) ? buf[c * kA16RS + jj] : 0.0f;
                float eb2[CT];
                load_ct<CT>(eb2, erel16(e1, e2, h, HD, x) + CT * c, 1.0f);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) dq[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, eb2[ct], dq[ct], 0, 0, 0);
            }
        // dErel[x] += sum_i dS[i][x + i - 15] qs_i :  A[row x = 16 xt + c][k i = 4g + s]
        if (live) {
#pragma unroll
            for (int xt = 0; xt < 2; ++xt)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int i = 4 * g + sidx, x = 16 * xt + c;
                    const int jj = x + i - 15;
                    const float a = (x <= 30 && jj >= 0 && jj < 16) ? buf[i * kA16RS + jj] : 0.0f;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        de[xt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, cur.qb[sidx][ct], de[xt][ct], 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float vq[CT], vk[CT], vv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    vq[ct] = dq[ct][r] * scale;
                    vk[ct] = dk[ct][r];
                    vv[ct] = dv[ct][r];
                }
                const int64_t go = (n * 16 + 4 * g + r) * ldg + h * HD + CT * c;
                store_out<CT, B16>(d_qkv, go, vq);
                store_out<CT, B16>(d_qkv, go + d, vk);
                store_out<CT, B16>(d_qkv, go + 2 * d, vv);
            }
        }
        if (kPipe) cur = nxt;
    }
    float* dst = ws + (((int64_t)blockIdx.x * kA16Waves + wave) * H + h) * 31 * HD;
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int x = 16 * xt + 4 * g + r;
                if (x <= 30) dst[x * HD + CT * c + ct] = de[xt][ct][r];
            }
}

// =====================================================================================================================
// All-bf16 backward (q | k | v, d ctx in, d qkv out: bf16) with EVERY contraction on the bf16 matrix pipe (round 5).
// The fp32 16 x 16 x 4 MFMAs of relattn16_bwd_kernel are 4 096 pipe cycles per (block, head) problem at head_dim 64 (128
// instructions of 32 cycles) and the bf16 inputs are unpacked to fp32 for them (1 577 VALU instructions per problem,
// profiles/r05_pmc_kernels_c4.txt: 3.2 ms per launch at configs[4] against 1.5 ms of HBM time).  Here:
//   * the operands that ARE bf16 in memory (dO, Q, K, V) go to v_mfma_f32_16x16x32_bf16 as loaded -- exact products, fp32 accumulate;
//   * the fp32 factors computed here (Pd = P . mask, dS) and the fp32 parameter Erel are carried as TWO bf16 pieces (x = hi + lo
//     + O(2^-17 x)), both pieces of a 16-token contraction in ONE K = 32 instruction: slots 0-3 of lane group g hold the hi pieces
//     of tokens 4g .. 4g+3, slots 4-7 the lo pieces, the bf16 operand is repeated in both slot halves (the k index of an MFMA is
//     a summation index: any assignment shared by A and B is valid);  dS . Erel = [hi | lo] . [H | H] + [hi | lo] . [L | 0];
//   * a token-contraction B operand (rows 4g .. 4g+3 of dO / Q / K at this lane's CT columns) is a 4 x CT transpose of 16-bit
//     elements: v_perm_b32, no conversion.
// 38 MFMAs of 16 cycles per problem at head_dim 64.  Softmax backward, dropout mask, partial d Erel layout: as relattn16_bwd_kernel.
// Error vs that kernel: summation order + 2^-17 relative in Pd / dS / Erel, i.e. far below the bf16 rounding of the outputs.
// =====================================================================================================================
__device__ __forceinline__ uint32_t perm_lo(uint32_t hi_src, uint32_t lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u); }
__device__ __forceinline__ uint32_t perm_hi(uint32_t hi_src, uint32_t lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }
// four fp32 -> bf16 pieces: h01 / h23 = the rounded values (element 0 / 2 in the low half), l01 / l23 = rn(x - h)
__device__ __forceinline__ void split2x4(const float (&v)[4], uint32_t& h01, uint32_t& h23, uint32_t& l01, uint32_t& l23) {
    h01 = bf16x2_rn(v[0], v[1]);
    h23 = bf16x2_rn(v[2], v[3]);
    l01 = bf16x2_rn(v[0] - bf16_lo(h01), v[1] - bf16_hi(h01));
    l23 = bf16x2_rn(v[2] - bf16_lo(h23), v[3] - bf16_hi(h23));
}
__device__ __forceinline__ bf16x8 frag8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(bf16x8, make_uint4(a, b, c, d));
}
// four LDS dwords (hi << 16 | lo) of tokens e = 0..3 -> the A operand [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]
__device__ __forceinline__ bf16x8 frag_from_packed(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
    return frag8(perm_hi(d1, d0), perm_hi(d3, d2), perm_lo(d1, d0), perm_lo(d3, d2));
}
// R[r][w]: row 4g + r, this lane's CT columns (two per dword) -> T[ct] = {rows 0 1, rows 2 3} of column ct
template <int CT>
__device__ __forceinline__ void tr_rows4(const uint32_t (&R)[4][CT / 2], uint32_t (&T)[CT][2]) {
#pragma unroll
    for (int w = 0; w < CT / 2; ++w) {
        T[2 * w][0] = perm_lo(R[1][w], R[0][w]);
        T[2 * w][1] = perm_lo(R[3][w], R[2][w]);
        T[2 * w + 1][0] = perm_hi(R[1][w], R[0][w]);
        T[2 * w + 1][1] = perm_hi(R[3][w], R[2][w]);
    }
}
template <int CT>
__device__ __forceinline__ void load_row_b16(uint32_t (&dst)[CT / 2], const unsigned short* __restrict__ p) {
    if constexpr (CT == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        dst[0] = t.x; dst[1 % (CT / 2)] = t.y;
    } else {
        dst[0] = *reinterpret_cast<const uint32_t*>(p);
    }
}

template <int HD>
__global__ __launch_bounds__(kA16Waves * 64, 2) void relattn16_bwd_mm16_kernel(
    const unsigned short* __restrict__ d_ctx, int64_t ldo, const unsigned short* __restrict__ qkv, int64_t ldq,
    const float* __restrict__ probs, const float* __restrict__ e1, const float* __restrict__ e2,
    unsigned short* __restrict__ d_qkv, int64_t ldg, float* __restrict__ ws, int64_t n_blocks, int H, int blocks_per_chunk,
    float scale, uint32_t thr, float inv_keep, uint64_t seed) {
    static_assert(HD == 32 || HD == 64, "head_dim 32 / 64");
    constexpr int KH = HD / 4, CT = HD / 16, NCH = HD / 32;
    __shared__ uint32_t lds[kA16Waves][16 * kA16RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    uint32_t* buf = lds[wave];
    const int h = blockIdx.y;
    const int d = H * HD;
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the relative rows of this head as bf16 pieces, once per wave: B[k x = 16 xt + 4g + e][col CT c + ct] = {H01, H23, L01, L23}, parked
    // in LDS (32 registers otherwise: two waves per SIMD would spill).  The four waves of a workgroup work on the same head: they
    // write identical values to the same lane-private slots and each reads its own lane's -- no workgroup barrier needed.
    __shared__ uint4 erel_lds[2 * CT][64];
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
        float ev[4][CT];
#pragma unroll
        for (int e = 0; e < 4; ++e) load_ct<CT>(ev[e], erel16(e1, e2, h, HD, 16 * xt + 4 * g + e) + CT * c, 1.0f);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float v[4] = {ev[0][ct], ev[1][ct], ev[2][ct], ev[3][ct]};
            uint4 pk;
            split2x4(v, pk.x, pk.y, pk.z, pk.w);
            erel_lds[xt * CT + ct][lane] = pk;
        }
    }
    wave_lds_fence();
    floatx4 de[2][CT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) de[t][ct] = zero4;
    const int64_t b0 = (int64_t)blockIdx.x * blocks_per_chunk;
    const int64_t b1 = min(b0 + blocks_per_chunk, n_blocks);
    const int iters = (int)((b1 - b0 + kA16Waves - 1) / kA16Waves);
    struct Blk {
        uint4 doa[NCH], vb[NCH];
        float p[4];
        uint32_t dob[4][CT / 2], qb[4][CT / 2], kb[4][CT / 2];
    };
    auto block_of = [&](int it) { return min(b0 + (int64_t)it * kA16Waves + wave, b1 - 1); };
    auto load_blk = [&](Blk& B, int64_t n) {
        const int64_t prob = n * H + h;
        const unsigned short* dc = d_ctx + (n * 16 + c) * ldo + h * HD + g * KH;
        const unsigned short* vr = qkv + (n * 16 + c) * ldq + 2 * d + h * HD + g * KH;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            B.doa[ch] = *reinterpret_cast<const uint4*>(dc + 8 * ch);
            B.vb[ch] = *reinterpret_cast<const uint4*>(vr + 8 * ch);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) B.p[r] = probs[(prob * 16 + 4 * g + r) * 16 + c];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int64_t row = n * 16 + 4 * g + sidx;
            load_row_b16<CT>(B.dob[sidx], d_ctx + row * ldo + h * HD + CT * c);
            load_row_b16<CT>(B.qb[sidx], qkv + row * ldq + h * HD + CT * c);
            load_row_b16<CT>(B.kb[sidx], qkv + row * ldq + d + h * HD + CT * c);
        }
    };
    Blk cur;
    load_blk(cur, block_of(0));
    for (int it = 0; it < iters; ++it) {
        const int64_t want = b0 + (int64_t)it * kA16Waves + wave;
        const bool live = want < b1;
        const int64_t n = live ? want : b1 - 1;
        const int64_t prob = n * H + h;
        Blk nxt;
        load_blk(nxt, block_of(it + 1));               // every operand of the next block is in flight while this one is processed
        // dP = dO . V^T over head_dim: both operands as loaded (lane group g holds columns g KH .. of its row)
        floatx4 dp = zero4;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.doa[ch]), __builtin_bit_cast(bf16x8, cur.vb[ch]),
                                                         dp, 0, 0, 0);
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t idx = (prob * 16 + 4 * g + r) * 16 + c;
            const float p = cur.p[r];
            const float mk = drop_scale(seed, (uint64_t)idx, thr, inv_keep);
            const float dpm = dp[r] * mk;
            pd[r] = p * mk;
            ds[r] = p * (dpm - grp16_sum(dpm * p));
        }
        uint32_t ph01, ph23, pl01, pl23, sh01, sh23, sl01, sl23;
        split2x4(pd, ph01, ph23, pl01, pl23);
        split2x4(ds, sh01, sh23, sl01, sl23);
        uint32_t td[CT][2], tq[CT][2], tk[CT][2];
        tr_rows4<CT>(cur.dob, td);
        tr_rows4<CT>(cur.qb, tq);
        tr_rows4<CT>(cur.kb, tk);
        // dV = Pd^T dO, dK = dS^T Q (x scale at the store):  A[row j = c][k i = 4g + e] = X[4g + e][c] = this lane's own values
        const bf16x8 a_pd = frag8(ph01, ph23, pl01, pl23), a_ds = frag8(sh01, sh23, sl01, sl23);
        floatx4 dv[CT], dk[CT], dq[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            dv[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_pd, frag8(td[ct][0], td[ct][1], td[ct][0], td[ct][1]), zero4, 0, 0, 0);
            dk[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_ds, frag8(tq[ct][0], tq[ct][1], tq[ct][0], tq[ct][1]), zero4, 0, 0, 0);
        }
        // dS through the wave's LDS tile as (hi << 16 | lo) dwords: the transposed and the skewed views
        wave_lds_fence();                                  // previous iteration's LDS readers are done
        buf[(4 * g + 0) * kA16RS + c] = perm_lo(sh01, sl01);
        buf[(4 * g + 1) * kA16RS + c] = perm_hi(sh01, sl01);
        buf[(4 * g + 2) * kA16RS + c] = perm_lo(sh23, sl23);
        buf[(4 * g + 3) * kA16RS + c] = perm_hi(sh23, sl23);
        wave_lds_fence();
        // dq = scale * (dS . K + skew(dS) . Erel):  A[row i = c][k j = 4g + e]
        {
            const uint32_t* rowp = buf + c * kA16RS + 4 * g;
            const bf16x8 a = frag_from_packed(rowp[0], rowp[1], rowp[2], rowp[3]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                dq[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, frag8(tk[ct][0], tk[ct][1], tk[ct][0], tk[ct][1]), zero4, 0, 0, 0);
        }
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            uint32_t dd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = 16 * xt + 4 * g + e;             // relative row
                const int jj = x + c - 15;                     // key index seen from query c
                const bool ok = x <= 30 && jj >= 0 && jj < 16;
                const uint32_t t = buf[c * kA16RS + (ok ? jj : 0)];
                dd[e] = ok ? t : 0u;
            }
            const bf16x8 a = frag_from_packed(dd[0], dd[1], dd[2], dd[3]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const uint4 ep = erel_lds[xt * CT + ct][lane];
                dq[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, frag8(ep.x, ep.y, ep.x, ep.y), dq[ct], 0, 0, 0);
                dq[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, frag8(ep.z, ep.w, 0u, 0u), dq[ct], 0, 0, 0);
            }
        }
        // dErel[x] += sum_i dS[i][x + i - 15] Q_i (x scale at the end):  A[row x = 16 xt + c][k i = 4g + e]
        if (live) {
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                uint32_t dd[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g + e, x = 16 * xt + c;
                    const int jj = x + i - 15;
                    const bool ok = x <= 30 && jj >= 0 && jj < 16;
                    const uint32_t t = buf[i * kA16RS + (ok ? jj : 0)];
                    dd[e] = ok ? t : 0u;
                }
                const bf16x8 a = frag_from_packed(dd[0], dd[1], dd[2], dd[3]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    de[xt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, frag8(tq[ct][0], tq[ct][1], tq[ct][0], tq[ct][1]),
                                                                        de[xt][ct], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float vq[CT], vk[CT], vv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    vq[ct] = dq[ct][r] * scale;
                    vk[ct] = dk[ct][r] * scale;
                    vv[ct] = dv[ct][r];
                }
                const int64_t go = (n * 16 + 4 * g + r) * ldg + h * HD + CT * c;
                store_ct_b16<CT>(d_qkv + go, vq);
                store_ct_b16<CT>(d_qkv + go + d, vk);
                store_ct_b16<CT>(d_qkv + go + 2 * d, vv);
            }
        }
        cur = nxt;
    }
    float* dst = ws + (((int64_t)blockIdx.x * kA16Waves + wave) * H + h) * 31 * HD;
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int x = 16 * xt + 4 * g + r;
                if (x <= 30) dst[x * HD + CT * c + ct] = de[xt][ct][r] * scale;
            }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static int a16_blocks_per_chunk(int64_t n_blocks, int H) {
    // ~8k wavefronts in flight: chunks * 4 waves * H ~ 8192
    const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_blocks, kA16Waves), 2048 / std::max(H, 1)));
    return (int)ceil_div(n_blocks, chunks);
}

bool relattn16_supported(int H, int hd) { return H >= 1 && (hd == 16 || hd == 32 || hd == 64); }

int64_t relattn16_bwd_workspace(int64_t n_blocks, int H, int hd) {
    const int bpc = a16_blocks_per_chunk(n_blocks, H);
    const int64_t chunks = ceil_div(n_blocks, bpc);
    return (chunks * kA16Waves + 1) * H * 31 * hd * (int64_t)sizeof(float);
}

template <int HD, bool B16 = false, bool IN16 = false>
static int a16_fwd_t(const float* qkv, int64_t ldq, const int64_t* tokens, const float* e1, const float* e2, float* ctx,
                     int64_t ldo, float* probs, int64_t n_blocks, int H, float drop_p, uint64_t seed, hipStream_t s) {
    const int64_t total = n_blocks * H;
    static const int x6 = lab_env_int("VQCPC_RELATTN16_X6", 1);      // lab builds: =0 keeps the fp32-MFMA contractions (A/B)
    // GEMM mode 0 is documented as the exact fp32-MFMA arithmetic (include/vqcpc.h): there the contractions over head_dim
    // stay on v_mfma_f32_16x16x4_f32 too; the six-product bf16 form belongs to the bf16x6 / bf16 modes
    if ((HD == 32 || HD == 64) && x6 && vqcpc_gemm_get_mode() != 0) {
        hipLaunchKernelGGL((relattn16_fwd_kernel<HD, B16, IN16, (HD == 32 || HD == 64)>), dim3((unsigned)ceil_div(total, kA16Waves)), dim3(kA16Waves * 64),
                           0, s, qkv, ldq, tokens, e1, e2, ctx, ldo, probs, total, H, 1.0f / sqrtf((float)HD), drop_threshold(drop_p),
                           1.0f / (1.0f - drop_p), seed);
        VQ_CHECK_LAUNCH("relattn16_fwd (x6)");
        return VQCPC_OK;
    }
    hipLaunchKernelGGL((relattn16_fwd_kernel<HD, B16, IN16>), dim3((unsigned)ceil_div(total, kA16Waves)), dim3(kA16Waves * 64), 0, s, qkv,
                       ldq, tokens, e1, e2, ctx, ldo, probs, total, H, 1.0f / sqrtf((float)HD), drop_threshold(drop_p),
                       1.0f / (1.0f - drop_p), seed);
    VQ_CHECK_LAUNCH("relattn16_fwd");
    return VQCPC_OK;
}

template <int HD, bool B16 = false, bool IN16 = false>
static int a16_bwd_t(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const int64_t* tokens,
                     const float* probs, const float* e1, const float* e2, float* d_qkv, int64_t ldg, float* ws,
                     int64_t n_blocks, int H, float drop_p, uint64_t seed, hipStream_t s, int* nsplit) {
    const int bpc = a16_blocks_per_chunk(n_blocks, H);
    const int chunks = (int)ceil_div(n_blocks, bpc);
    if constexpr (B16 && IN16 && (HD == 32 || HD == 64)) {
        // all-bf16 form outside the exact GEMM mode 0: every contraction on the bf16 matrix pipe (relattn16_bwd_mm16_kernel)
        static const int mm16 = lab_env_int("VQCPC_RELATTN16_MM16", 1);      // lab builds: =0 keeps the fp32-MFMA contractions (A/B)
        if (mm16 && tokens == nullptr && vqcpc_gemm_get_mode() != 0) {
            hipLaunchKernelGGL((relattn16_bwd_mm16_kernel<HD>), dim3(chunks, H), dim3(kA16Waves * 64), 0, s,
                               reinterpret_cast<const unsigned short*>(d_ctx), ldo, reinterpret_cast<const unsigned short*>(qkv), ldq,
                               probs, e1, e2, reinterpret_cast<unsigned short*>(d_qkv), ldg, ws, n_blocks, H, bpc,
                               1.0f / sqrtf((float)HD), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed);
            VQ_CHECK_LAUNCH("relattn16_bwd (mm16)");
            *nsplit = chunks * kA16Waves;
            return VQCPC_OK;
        }
    }
    hipLaunchKernelGGL((relattn16_bwd_kernel<HD, B16, IN16>), dim3(chunks, H), dim3(kA16Waves * 64), 0, s, d_ctx, ldo, qkv, ldq, tokens,
                       probs, e1, e2, d_qkv, ldg, ws, n_blocks, H, bpc, 1.0f / sqrtf((float)HD), drop_threshold(drop_p),
                       1.0f / (1.0f - drop_p), seed);
    VQ_CHECK_LAUNCH("relattn16_bwd");
    *nsplit = chunks * kA16Waves;
    return VQCPC_OK;
}

int relattn16_fwd(const float* qkv, int64_t ldq, const int64_t* tokens, const float* e1, const float* e2, float* ctx,
                  int64_t ldo, float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, hipStream_t s) {
    if (hd == 16) return a16_fwd_t<16>(qkv, ldq, tokens, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s);
    if (hd == 32) return a16_fwd_t<32>(qkv, ldq, tokens, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s);
    return a16_fwd_t<64>(qkv, ldq, tokens, e1, e2, ctx, ldo, probs, n_blocks, H, drop_p, seed, s);
}

// writes per-wave partials to ws and returns their count; the caller reduces them (tail of ws) and splits into e1 / e2
int relattn16_bwd(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const int64_t* tokens, const float* probs,
                  const float* e1, const float* e2, float* d_qkv, int64_t ldg, float* ws, int64_t n_blocks, int H, int hd,
                  float drop_p, uint64_t seed, hipStream_t s, int* nsplit) {
    if (hd == 16)
        return a16_bwd_t<16>(d_ctx, ldo, qkv, ldq, tokens, probs, e1, e2, d_qkv, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
    if (hd == 32)
        return a16_bwd_t<32>(d_ctx, ldo, qkv, ldq, tokens, probs, e1, e2, d_qkv, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
    return a16_bwd_t<64>(d_ctx, ldo, qkv, ldq, tokens, probs, e1, e2, d_qkv, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
}

// bf16-output forms (ctx / d_qkv are bf16 buffers; leading dimensions in elements)
int relattn16_fwd_b16(const float* qkv, int64_t ldq, const int64_t* tokens, const float* e1, const float* e2, void* ctx_b16,
                      int64_t ldo, float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, hipStream_t s) {
    float* c = reinterpret_cast<float*>(ctx_b16);
    if (hd == 16) return a16_fwd_t<16, true>(qkv, ldq, tokens, e1, e2, c, ldo, probs, n_blocks, H, drop_p, seed, s);
    if (hd == 32) return a16_fwd_t<32, true>(qkv, ldq, tokens, e1, e2, c, ldo, probs, n_blocks, H, drop_p, seed, s);
    return a16_fwd_t<64, true>(qkv, ldq, tokens, e1, e2, c, ldo, probs, n_blocks, H, drop_p, seed, s);
}

int relattn16_bwd_b16(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const int64_t* tokens, const float* probs,
                      const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* ws, int64_t n_blocks, int H, int hd,
                      float drop_p, uint64_t seed, hipStream_t s, int* nsplit) {
    float* g = reinterpret_cast<float*>(d_qkv_b16);
    if (hd == 16)
        return a16_bwd_t<16, true>(d_ctx, ldo, qkv, ldq, tokens, probs, e1, e2, g, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
    if (hd == 32)
        return a16_bwd_t<32, true>(d_ctx, ldo, qkv, ldq, tokens, probs, e1, e2, g, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
    return a16_bwd_t<64, true>(d_ctx, ldo, qkv, ldq, tokens, probs, e1, e2, g, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
}

// all-bf16 forms: q | k | v (and d ctx in the backward) are bf16 as well; no token indirection (the first layer's table is fp32)
int relattn16_fwd_b16io(const void* qkv_b16, int64_t ldq, const float* e1, const float* e2, void* ctx_b16, int64_t ldo,
                        float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, hipStream_t s) {
    const float* q = reinterpret_cast<const float*>(qkv_b16);
    float* c = reinterpret_cast<float*>(ctx_b16);
    if (hd == 16) return a16_fwd_t<16, true, true>(q, ldq, nullptr, e1, e2, c, ldo, probs, n_blocks, H, drop_p, seed, s);
    if (hd == 32) return a16_fwd_t<32, true, true>(q, ldq, nullptr, e1, e2, c, ldo, probs, n_blocks, H, drop_p, seed, s);
    return a16_fwd_t<64, true, true>(q, ldq, nullptr, e1, e2, c, ldo, probs, n_blocks, H, drop_p, seed, s);
}

int relattn16_bwd_b16io(const void* d_ctx_b16, int64_t ldo, const void* qkv_b16, int64_t ldq, const float* probs,
                        const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* ws, int64_t n_blocks, int H,
                        int hd, float drop_p, uint64_t seed, hipStream_t s, int* nsplit) {
    const float* dc = reinterpret_cast<const float*>(d_ctx_b16);
    const float* q = reinterpret_cast<const float*>(qkv_b16);
    float* g = reinterpret_cast<float*>(d_qkv_b16);
    if (hd == 16)
        return a16_bwd_t<16, true, true>(dc, ldo, q, ldq, nullptr, probs, e1, e2, g, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
    if (hd == 32)
        return a16_bwd_t<32, true, true>(dc, ldo, q, ldq, nullptr, probs, e1, e2, g, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
    return a16_bwd_t<64, true, true>(dc, ldo, q, ldq, nullptr, probs, e1, e2, g, ldg, ws, n_blocks, H, drop_p, seed, s, nsplit);
}

}  // namespace vq
