// fp32 MFMA GEMMs for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD -> 157 TFLOP/s chip peak).
//
// gemm_nt : C[M,N] = epi(A[M,K] . B[N,K]^T)      128x128x32 tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles.
//           Both operands are K-contiguous, staged to LDS as [row][32 + 4 pad] so a lane reads 4 consecutive k with one
//           ds_read_b128 (row stride 144 B: the 16 rows of a b128 lane group hit 16 distinct 16-B slots, conflict-free).
//           A lane (i = lane & 31, kh = lane >> 5) feeds MFMA step s of a k-chunk of 8 with element kh*4 + s, i.e. the
//           eight k of a chunk are consumed in the order (0,4),(1,5),(2,6),(3,7) -- any order is a valid contraction.
// gemm_tn : dW[N,K] = A[M,N]^T . B[M,K]         (weight gradient; contraction over the huge M dimension)
//           128x128 output tile, 32 rows of M per step, operands staged [m][128]; MFMA operands are ds_read_b32
//           (consecutive lanes -> consecutive banks).  M is split over blockIdx.y; partials are reduced deterministically.
//           The bias gradient (column sums of A) is accumulated from the A operand registers for free.
#include <atomic>
#include <vector>
#include <stdlib.h>

#include <string.h>

#include "gemm_common.h"

namespace vq {

// FULL: M % 128 == 0, N % 128 == 0, K % 32 == 0 -> no bounds checks anywhere (all hot-path shapes).
// Two LDS buffers: tile t+1 is staged (global -> registers -> other buffer) while tile t feeds the MFMAs; one barrier
// per K tile.
//
// MODE 0: v_mfma_f32_32x32x2_f32 on fp32 operands (bit-exact fp32 fmaf chains), two LDS buffers, one barrier per K tile.
// MODE 1 ("bf16x6"): every fp32 operand element is split EXACTLY into three bf16 pieces x = h + m + l (8 + 8 + 8 mantissa
//         bits, by truncation) while it is staged to LDS, and a product a*b is evaluated as
//         h*h + (h*m + m*h) + (h*l + l*h + m*m) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped terms
//         (m*l, l*m, l*l) are <= 2^-23 of the product, i.e. fp32 rounding level.  6 bf16 MFMAs at 16x the fp32 MFMA rate
//         = 2.67x the fp32-MFMA throughput at fp32-class accuracy (verified by the same parity tests).
template <bool FULL, int EPI, int MODE>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                 int K, int tiles_n, EpiParams ep) {
    constexpr int kLdsBytes = MODE == 0 ? 2 * (BM + BN) * LDS_S * 4 : 6 * kX6Plane;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];
    float* const As0 = reinterpret_cast<float*>(smem);                       // MODE 0: As[2][BM*LDS_S] | Bs[2][BN*LDS_S]
    float* const Bs0 = As0 + 2 * BM * LDS_S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(t / tiles_n) * BM;
    const int n0 = (t % tiles_n) * BN;
    if (EPI == 0 && FULL) {
        // split-K launch (gemm_nt_splitk below): workgroup row y contracts k in [y K, (y + 1) K) into partial plane y
        A += (int64_t)blockIdx.y * K;
        B += (int64_t)blockIdx.y * K;
        C += (int64_t)blockIdx.y * ep.split_plane;
    }

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // global -> register staging: 4 float4 of A and 4 of B per thread per k-tile
    const int ld_row = tid >> 3, ld_c4 = (tid & 7) * 4;
    const float* a_src = A + (m0 + ld_row) * lda + ld_c4;
    const float* b_src = B + (int64_t)(n0 + ld_row) * ldb + ld_c4;
    // named registers (arrays captured by lambdas ended up in scratch memory on this compiler)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define NT_LOAD1(RA, RB, I, K0)                                                                                   \
    if (FULL) {                                                                                                   \
        RA = *reinterpret_cast<const float4*>(a_src + (int64_t)(32 * (I)) * lda + (K0));                            \
        RB = *reinterpret_cast<const float4*>(b_src + (int64_t)(32 * (I)) * ldb + (K0));                            \
    } else {                                                                                                      \
        const bool kok_ = (K0) + ld_c4 < K;                                                                       \
        RA = (kok_ && m0 + ld_row + 32 * (I) < M)                                                                 \
                 ? *reinterpret_cast<const float4*>(a_src + (int64_t)(32 * (I)) * lda + (K0)) : zero4;             \
        RB = (kok_ && n0 + ld_row + 32 * (I) < N)                                                                 \
                 ? *reinterpret_cast<const float4*>(b_src + (int64_t)(32 * (I)) * ldb + (K0)) : zero4;             \
    }
#define NT_LOAD(K0) \
    NT_LOAD1(ra0, rb0, 0, K0) NT_LOAD1(ra1, rb1, 1, K0) NT_LOAD1(ra2, rb2, 2, K0) NT_LOAD1(ra3, rb3, 3, K0)
#define NT_STORE1(RA, RB, I, BUF)                                                                      \
    *reinterpret_cast<float4*>(As0 + (BUF) * BM * LDS_S + (ld_row + 32 * (I)) * LDS_S + ld_c4) = RA;   \
    *reinterpret_cast<float4*>(Bs0 + (BUF) * BN * LDS_S + (ld_row + 32 * (I)) * LDS_S + ld_c4) = RB;
#define NT_STORE(BUF) \
    NT_STORE1(ra0, rb0, 0, BUF) NT_STORE1(ra1, rb1, 1, BUF) NT_STORE1(ra2, rb2, 2, BUF) NT_STORE1(ra3, rb3, 3, BUF)
    // MODE 1: split into bf16 planes  A_h | A_m | A_l | B_h | B_m | B_l, each [128][kX6Stride] bf16
#define X6_STORE1(RA, RB, I)                                                                           \
    if (MODE == 2) {                                                                                   \
        const int o_ = ((ld_row + 32 * (I)) * kX6Stride + ld_c4) * 2;                                  \
        *reinterpret_cast<uint2*>(smem + 0 * kX6Plane + o_) = round4_bf16(RA);                         \
        *reinterpret_cast<uint2*>(smem + 3 * kX6Plane + o_) = round4_bf16(RB);                         \
    } else {                                                                                           \
        const int o_ = ((ld_row + 32 * (I)) * kX6Stride + ld_c4) * 2;                                  \
        uint2 h_, m_, l_;                                                                              \
        split3x4(RA, h_, m_, l_);                                                                      \
        *reinterpret_cast<uint2*>(smem + 0 * kX6Plane + o_) = h_;                                      \
        *reinterpret_cast<uint2*>(smem + 1 * kX6Plane + o_) = m_;                                      \
        *reinterpret_cast<uint2*>(smem + 2 * kX6Plane + o_) = l_;                                      \
        split3x4(RB, h_, m_, l_);                                                                      \
        *reinterpret_cast<uint2*>(smem + 3 * kX6Plane + o_) = h_;                                      \
        *reinterpret_cast<uint2*>(smem + 4 * kX6Plane + o_) = m_;                                      \
        *reinterpret_cast<uint2*>(smem + 5 * kX6Plane + o_) = l_;                                      \
    }
#define X6_STORE() X6_STORE1(ra0, rb0, 0) X6_STORE1(ra1, rb1, 1) X6_STORE1(ra2, rb2, 2) X6_STORE1(ra3, rb3, 3)

    NT_LOAD(0)
    if (MODE == 0) { NT_STORE(0) } else { X6_STORE() }
    __syncthreads();
    const int a_off = (wm * 64 + li) * LDS_S + kh * 4;
    const int b_off = (wn * 64 + li) * LDS_S + kh * 4;
    // epilogue operands (gate or residual) are prefetched into registers under the LAST K tile's MFMAs: one global
    // round trip per tile instead of one per element (a load->use->store chain per element serialises on latency)
    constexpr bool RT = (EPI & E_RUNTIME) != 0;
    constexpr bool PREF = !RT && ((EPI & (E_GATE | E_ADD)) != 0) && ((EPI & (E_GATE | E_ADD)) != (E_GATE | E_ADD));
    const int64_t row_base = m0 + wm * 64 + 4 * kh;
    const int col_base = n0 + wn * 64 + li;
    float aux[2][2][16];
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) { NT_LOAD(k0 + BK) }           // HBM/L2 latency hides under the MFMAs below
        if (PREF && !more) {
            const float* src = (EPI & E_GATE) ? ep.gate : ep.add;
            const int64_t lds_ = (EPI & E_GATE) ? ep.ldgate : ep.ldadd;
            if (FULL) {
                // one VGPR of per-lane offset + a scalar offset per element (buffer addressing): no 64-bit address
                // arithmetic per load, so all 64 loads are in flight together
                const __amdgpu_buffer_rsrc_t rs =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(src + m0 * lds_ + n0), 0, 0x7FFFFFFF, 0x00020000);
                const int ldi = (int)lds_;
                const int voff = ((wm * 64 + 4 * kh) * ldi + wn * 64 + li) * 4;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            aux[mt][nt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                rs, voff, ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldi + nt * 32) * 4, 0));
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int64_t row = row_base + mt * 32 + (r & 3) + 8 * (r >> 2);
                            const int col = col_base + nt * 32;
                            aux[mt][nt][r] = (row < M && col < N) ? src[row * lds_ + col] : 0.0f;
                        }
            }
        }
        if (MODE == 0) {
            const float* ap = As0 + cur * BM * LDS_S + a_off;
            const float* bp = Bs0 + cur * BN * LDS_S + b_off;
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                const float4 a0 = *reinterpret_cast<const float4*>(ap + kc * 8);
                const float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * LDS_S + kc * 8);
                const float4 b0 = *reinterpret_cast<const float4*>(bp + kc * 8);
                const float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * LDS_S + kc * 8);
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
                }
            }
            if (more) {
                if (cur) { NT_STORE(0) } else { NT_STORE(1) }   // nobody reads the other buffer any more (previous barrier)
                __syncthreads();
                cur ^= 1;
            }
        } else {
            // lane (i = lane & 31, kg = lane >> 5) holds A[i][kg*8 .. kg*8+7] / B[kg*8 .. +7][j] of a 32x32x16 bf16 MFMA
            const unsigned char* abase = smem + ((wm * 64 + li) * kX6Stride + kh * 8) * 2;
            const unsigned char* bbase = smem + 3 * kX6Plane + ((wn * 64 + li) * kX6Stride + kh * 8) * 2;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                constexpr int NP = MODE == 2 ? 1 : 3;          // planes in use
                bf16x8 a[3][2], b[3][2];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        a[pc][tl] = *reinterpret_cast<const bf16x8*>(abase + pc * kX6Plane + (tl * 32 * kX6Stride + ks * 16) * 2);
                        b[pc][tl] = *reinterpret_cast<const bf16x8*>(bbase + pc * kX6Plane + (tl * 32 * kX6Stride + ks * 16) * 2);
                    }
                // term-major order: consecutive MFMAs hit the 4 different accumulators (an MFMA that depends on the
                // previous one stalls for its full latency); smallest terms first
#define X6_TERM(PA, PB)                                                                                        \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][0], acc[0][0], 0, 0, 0);                \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][1], acc[0][1], 0, 0, 0);                \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][0], acc[1][0], 0, 0, 0);                \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][1], acc[1][1], 0, 0, 0);
                if (MODE != 2) {
                    X6_TERM(2, 0)   // l*h
                    X6_TERM(0, 2)   // h*l
                    X6_TERM(1, 1)   // m*m
                    X6_TERM(1, 0)   // m*h
                    X6_TERM(0, 1)   // h*m
                }
                X6_TERM(0, 0)   // h*h
#undef X6_TERM
            }
            if (more) {
                __syncthreads();                 // every wave is done reading the (single) staging buffer
                X6_STORE()
                __syncthreads();
            }
        }
    }

#undef NT_LOAD
#undef NT_LOAD1
#undef NT_STORE
#undef NT_STORE1
#undef X6_STORE
#undef X6_STORE1
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool has_bias = RT ? ep.bias != nullptr : (EPI & E_BIAS) != 0;
    const bool relu = RT ? ep.act == 1 : (EPI & E_RELU) != 0;
    const bool drop = RT ? ep.thr != 0 : (EPI & E_DROP) != 0;
    const bool has_gate = RT ? ep.gate != nullptr : (EPI & E_GATE) != 0;
    const bool has_add = RT ? ep.add != nullptr : (EPI & E_ADD) != 0;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * ldc + n0), 0, 0x7FFFFFFF, 0x00020000);
    const int ldci = (int)ldc;
    const int voff_c = ((wm * 64 + 4 * kh) * ldci + wn * 64 + li) * 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = col_base + nt * 32;
        if (!FULL && col >= N) continue;
        const float bv = has_bias ? ep.bias[col] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row_base + mt * 32 + (r & 3) + 8 * (r >> 2);
                if (!FULL && row >= M) continue;
                float v = acc[mt][nt][r] + bv;
                if (relu) v = fmaxf(v, 0.0f);
                if (drop) v *= drop_scale(ep.seed, (uint64_t)(row + ep.row0) * N + col, ep.thr, ep.inv_keep);
                if (has_gate) {
                    const float gv = PREF ? aux[mt][nt][r] : ep.gate[row * ep.ldgate + col];
                    v *= (gv > 0.0f ? ep.gate_scale : 0.0f);
                }
                if (has_add) {
                    v += PREF ? aux[mt][nt][r] : ep.add[row * ep.ldadd + col];
                    if ((RT && ep.add2) || (EPI & E_ADD2)) v += ep.add2[row * ep.ldadd2 + col];
                }
                if (FULL)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff_c,
                                                          ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldci + nt * 32) * 4, 0);
                else
                    C[row * ldc + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x6 NT GEMM on 64 x 64 x 32 tiles for UNDER-FILLED launches (student / decoder steps: 3072 x 512 x 512 is 96 tiles of
// 128 x 128 -- one workgroup on a third of the CUs, each walking its 16 K tiles alone with a serial stage / multiply rhythm:
// 38 us for 1.6 GFLOP).  Four times the workgroups, a quarter of the MFMA chain each, two LDS buffers (one barrier per K tile).
// Same operand split, same term order per k16 step and the same K order into ONE accumulator per 32 x 32 MFMA tile as
// gemm_nt_kernel<MODE 1>: results are bit-identical to the 128-tile kernel's.  M, N % 64 == 0, K % 32 == 0.
constexpr int kS64 = 64;
constexpr int kS64Plane = kS64 * kX6Stride * 2;      // 5 120 B
constexpr int kS64Buf = 6 * kS64Plane;               // 30 720 B; two buffers 61 440 B

template <int EPI>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_nt_x6_s64_kernel(const float* __restrict__ A, int64_t lda,
                                                                        const float* __restrict__ B, int64_t ldb,
                                                                        float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                        int K, int tiles_n, EpiParams ep) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kS64Buf];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(t / tiles_n) * kS64;
    const int n0 = (t % tiles_n) * kS64;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // staging: thread -> rows ld_row and ld_row + 32 of both operands, one float4 of k each
    const int ld_row = tid >> 3, ld_c4 = (tid & 7) * 4;
    const float* a_src = A + (m0 + ld_row) * lda + ld_c4;
    const float* b_src = B + (int64_t)(n0 + ld_row) * ldb + ld_c4;
    float4 ra0, ra1, rb0, rb1;
#define S64_LOAD(K0)                                                                 \
    ra0 = *reinterpret_cast<const float4*>(a_src + (K0));                            \
    ra1 = *reinterpret_cast<const float4*>(a_src + (int64_t)32 * lda + (K0));        \
    rb0 = *reinterpret_cast<const float4*>(b_src + (K0));                            \
    rb1 = *reinterpret_cast<const float4*>(b_src + (int64_t)32 * ldb + (K0));
#define S64_ST1(R, PLANE0, ROW, BUFP)                                                \
    {                                                                                \
        uint2 h_, m_, l_;                                                            \
        split3x4(R, h_, m_, l_);                                                     \
        const int o_ = ((ROW) * kX6Stride + ld_c4) * 2;                              \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 0) * kS64Plane + o_) = h_;    \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 1) * kS64Plane + o_) = m_;    \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 2) * kS64Plane + o_) = l_;    \
    }
#define S64_STORE(BUFP) \
    S64_ST1(ra0, 0, ld_row, BUFP) S64_ST1(ra1, 0, ld_row + 32, BUFP) S64_ST1(rb0, 3, ld_row, BUFP) S64_ST1(rb1, 3, ld_row + 32, BUFP)
    S64_LOAD(0)
    S64_STORE(smem)
    __syncthreads();
    const int a_off = ((wm * 32 + li) * kX6Stride + kh * 8) * 2;
    const int b_off = 3 * kS64Plane + ((wn * 32 + li) * kX6Stride + kh * 8) * 2;
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) { S64_LOAD(k0 + BK) }
        const unsigned char* buf = smem + cur * kS64Buf;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                a[pc] = *reinterpret_cast<const bf16x8*>(buf + a_off + pc * kS64Plane + ks * 32);
                b[pc] = *reinterpret_cast<const bf16x8*>(buf + b_off + pc * kS64Plane + ks * 32);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        if (more) {
            S64_STORE(smem + (cur ^ 1) * kS64Buf)      // the other buffer: its last readers passed the previous barrier
            __syncthreads();
            cur ^= 1;
        }
    }
#undef S64_LOAD
#undef S64_ST1
#undef S64_STORE
    // epilogue (order of operations as in gemm_nt_kernel): row = (reg & 3) + 8 (reg >> 2) + 4 kh, col = lane & 31
    const int64_t row_base = m0 + wm * 32 + 4 * kh;
    const int col = n0 + wn * 32 + li;
    const float bv = (EPI & E_BIAS) ? ep.bias[col] : 0.0f;
    float aux[16];
    if (EPI & (E_GATE | E_ADD)) {
        const float* src = (EPI & E_GATE) ? ep.gate : ep.add;
        const int64_t lds_ = (EPI & E_GATE) ? ep.ldgate : ep.ldadd;
#pragma unroll
        for (int r = 0; r < 16; ++r) aux[r] = src[(row_base + (r & 3) + 8 * (r >> 2)) * lds_ + col];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row_base + (r & 3) + 8 * (r >> 2);
        float v = acc[r] + bv;
        if (EPI & E_RELU) v = fmaxf(v, 0.0f);
        if (EPI & E_DROP) v *= drop_scale(ep.seed, (uint64_t)(row + ep.row0) * N + col, ep.thr, ep.inv_keep);
        if (EPI & E_GATE) v *= (aux[r] > 0.0f ? ep.gate_scale : 0.0f);
        if (EPI & E_ADD) v += aux[r];
        C[row * ldc + col] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x6 NT GEMM, 256 x 256 x 16 tile, 8 waves (2 x 4, wave tile 128 x 64), one workgroup per CU, two LDS buffers.
// At 128 x 128 the bf16x6 kernel is bound by operand delivery (32 FLOP per operand byte -> ~5.5 TB/s of L2->CU traffic at
// 175 TFLOP/s); the 256^2 tile doubles the arithmetic intensity.  Shapes must be full tiles (M, N % 256, K % 16).
// Epilogue operands (gate / add) are fetched per 32x32 MFMA tile, one tile ahead of their use.
constexpr int kT2 = 256;                               // tile edge
constexpr int kT2BK = 16;
constexpr int kT2Stride = (kT2BK + 8) * 2;             // 48 B per row: 16 rows of a ds_read_b128 group -> 16 distinct slots
constexpr int kT2Plane = kT2 * kT2Stride;              // 12 288 B
constexpr int kT2Buf = 6 * kT2Plane;                   // A_h A_m A_l B_h B_m B_l = 73 728 B; two buffers = 147 456 B
constexpr int kT2Threads = 512;

template <int EPI>
__global__ __launch_bounds__(kT2Threads, 2) void gemm_nt_x6_256_kernel(const float* __restrict__ A, int64_t lda,
                                                                      const float* __restrict__ B, int64_t ldb,
                                                                      float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                      int K, int tiles_n, int tiles, EpiParams ep) {
    // PERSISTENT: one workgroup per CU walks tiles t = blockIdx.x, + gridDim.x, ...  The first two K tiles of the NEXT
    // output tile are requested before the epilogue of the current one, so the pipeline-fill latency and the workgroup
    // relaunch disappear behind the stores (K = 256 means only 16 K tiles per output tile: fill / drain was ~30 %).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    int t_lin = blockIdx.x;
    int t = xcd_swizzle(t_lin, tiles);
    int64_t m0 = (int64_t)(t / tiles_n) * kT2;
    int n0 = (t % tiles_n) * kT2;

    floatx16 acc[4][2];

    // staging: thread -> (row = tid >> 2 (+128), 4 consecutive k); 2 float4 of A and 2 of B per K tile
    const int ld_row = tid >> 2, ld_c4 = (tid & 3) * 4;
    const float* a_src = A + (m0 + ld_row) * lda + ld_c4;
    const float* b_src = B + (int64_t)(n0 + ld_row) * ldb + ld_c4;
    // two register sets: the loads of K tile t+2 are issued while tile t is computed (one K tile of MFMAs is shorter
    // than the memory latency under load, so a prefetch distance of 1 leaves the loop latency-bound)
    float4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
#define T2_LOAD(S, K0)                                                                  \
    S##a0 = *reinterpret_cast<const float4*>(a_src + (K0));                             \
    S##a1 = *reinterpret_cast<const float4*>(a_src + (int64_t)128 * lda + (K0));        \
    S##b0 = *reinterpret_cast<const float4*>(b_src + (K0));                             \
    S##b1 = *reinterpret_cast<const float4*>(b_src + (int64_t)128 * ldb + (K0));
#define T2_ST1(R, PLANE0, ROW, BUFP)                                                 \
    {                                                                                \
        uint2 h_, m_, l_;                                                            \
        split3x4(R, h_, m_, l_);                                                     \
        const int o_ = (ROW) * kT2Stride + ld_c4 * 2;                                \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 0) * kT2Plane + o_) = h_;     \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 1) * kT2Plane + o_) = m_;     \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 2) * kT2Plane + o_) = l_;     \
    }
#define T2_STORE(S, BUFP) \
    T2_ST1(S##a0, 0, ld_row, BUFP) T2_ST1(S##a1, 0, ld_row + 128, BUFP) T2_ST1(S##b0, 3, ld_row, BUFP) T2_ST1(S##b1, 3, ld_row + 128, BUFP)
#define T2_COMPUTE(BUFP)                                                                                                  \
    {                                                                                                                     \
        const unsigned char* bufp = (BUFP);                                                                               \
        bf16x8 b[3][2];                                                                                                   \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                                  \
            _Pragma("unroll") for (int tl = 0; tl < 2; ++tl)                                                              \
                b[pc][tl] = *reinterpret_cast<const bf16x8*>(bufp + b_off + pc * kT2Plane + tl * 32 * kT2Stride);         \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                                \
            bf16x8 a[3][2];                                                                                               \
            _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                              \
                _Pragma("unroll") for (int tl = 0; tl < 2; ++tl)                                                          \
                    a[pc][tl] = *reinterpret_cast<const bf16x8*>(bufp + a_off + pc * kT2Plane + (hf * 2 + tl) * 32 * kT2Stride); \
            T2_TERM(2, 0) T2_TERM(0, 2) T2_TERM(1, 1) T2_TERM(1, 0) T2_TERM(0, 1) T2_TERM(0, 0)                            \
        }                                                                                                                 \
    }
#define T2_TERM(PA, PB)                                                                                                   \
    acc[hf * 2 + 0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][0], acc[hf * 2 + 0][0], 0, 0, 0);          \
    acc[hf * 2 + 0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][1], acc[hf * 2 + 0][1], 0, 0, 0);          \
    acc[hf * 2 + 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][0], acc[hf * 2 + 1][0], 0, 0, 0);          \
    acc[hf * 2 + 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][1], acc[hf * 2 + 1][1], 0, 0, 0);

    const int a_off = (wm * 128 + li) * kT2Stride + kh * 16;
    const int b_off = 3 * kT2Plane + (wn * 64 + li) * kT2Stride + kh * 16;
    unsigned char* const buf0 = smem2;
    unsigned char* const buf1 = smem2 + kT2Buf;
    // K % 32 == 0 (checked on the host): the steady-state loop has NO conditionals, so the waitcnt pass can keep the
    // youngest 4 loads in flight (a conditional load/store makes it assume the shorter queue and drain everything).
    // sched_barrier pins "issue loads -> MFMAs -> split/store" (hipcc otherwise hoists the split above the MFMAs).
    T2_LOAD(x, 0)
    T2_LOAD(y, kT2BK)
  for (;;) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    T2_STORE(x, buf0)
    __syncthreads();
    for (int k0 = 0; k0 < K - 2 * kT2BK; k0 += 2 * kT2BK) {
        T2_LOAD(x, k0 + 2 * kT2BK)                      // tile t+2 while tile t is computed (y = tile t+1 in flight)
        __builtin_amdgcn_sched_barrier(0);
        T2_COMPUTE(buf0)
        T2_STORE(y, buf1)                               // free to interleave with the MFMAs above (other buffer)
        __syncthreads();
        T2_LOAD(y, k0 + 3 * kT2BK)
        __builtin_amdgcn_sched_barrier(0);
        T2_COMPUTE(buf1)
        T2_STORE(x, buf0)
        __syncthreads();
    }
    T2_COMPUTE(buf0)
    __builtin_amdgcn_sched_barrier(0);
    T2_STORE(y, buf1)
    __syncthreads();
    T2_COMPUTE(buf1)
    // next output tile of this workgroup: request its first two K tiles now, store the current tile meanwhile
    const int64_t m0_cur = m0;
    const int n0_cur = n0;
    t_lin += gridDim.x;
    const bool has_next = t_lin < tiles;
    if (has_next) {
        t = xcd_swizzle(t_lin, tiles);
        m0 = (int64_t)(t / tiles_n) * kT2;
        n0 = (t % tiles_n) * kT2;
        a_src = A + (m0 + ld_row) * lda + ld_c4;
        b_src = B + (int64_t)(n0 + ld_row) * ldb + ld_c4;
        T2_LOAD(x, 0)
        T2_LOAD(y, kT2BK)
    }

    // epilogue (buffer addressing); gate / add operands are fetched one 32x32 tile ahead
    constexpr bool HAS_AUX = (EPI & (E_GATE | E_ADD)) != 0;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0_cur * ldc + n0_cur), 0, 0x7FFFFFFF, 0x00020000);
    const int ldci = (int)ldc;
    const int voff_c = ((wm * 128 + 4 * kh) * ldci + wn * 64 + li) * 4;
    const float* xsrc = (EPI & E_GATE) ? ep.gate : ep.add;
    const int ldxi = (int)((EPI & E_GATE) ? ep.ldgate : ep.ldadd);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_AUX ? xsrc + m0_cur * (int64_t)ldxi + n0_cur : C), 0, 0x7FFFFFFF, 0x00020000);
    const int voff_x = ((wm * 128 + 4 * kh) * ldxi + wn * 64 + li) * 4;
    float aux[2][16];
    if (HAS_AUX) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            aux[0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rx, voff_x, (((r & 3) + 8 * (r >> 2)) * ldxi) * 4, 0));
    }
    const int64_t row_base = m0_cur + wm * 128 + 4 * kh;
    const int col_base = n0_cur + wn * 64 + li;
#pragma unroll
    for (int tile = 0; tile < 8; ++tile) {
        const int mt = tile >> 1, nt = tile & 1;
        if (HAS_AUX && tile + 1 < 8) {
            const int mt2 = (tile + 1) >> 1, nt2 = (tile + 1) & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                aux[(tile + 1) & 1][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rx, voff_x, ((mt2 * 32 + (r & 3) + 8 * (r >> 2)) * ldxi + nt2 * 32) * 4, 0));
        }
        const int col = col_base + nt * 32;
        const float bv = (EPI & E_BIAS) ? ep.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row_base + mt * 32 + (r & 3) + 8 * (r >> 2);
            float v = acc[mt][nt][r] + bv;
            if (EPI & E_RELU) v = fmaxf(v, 0.0f);
            if (EPI & E_DROP) v *= drop_scale(ep.seed, (uint64_t)(row + ep.row0) * N + col, ep.thr, ep.inv_keep);
            if (EPI & E_GATE) v *= (aux[tile & 1][r] > 0.0f ? ep.gate_scale : 0.0f);
            if (EPI & E_ADD) v += aux[tile & 1][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff_c,
                                                  ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldci + nt * 32) * 4, 0);
        }
    }
    if (!has_next) break;
  }
#undef T2_TERM
#undef T2_COMPUTE
#undef T2_LOAD
#undef T2_ST1
#undef T2_STORE
}

// ---------------------------------------------------------------------------------------------------------------------
// PING-PONG variant of the 256 x 256 x 16 bf16x6 NT kernel.  With one workgroup-wide barrier per K tile the two waves
// that share a SIMD run in lockstep: both wait for LDS / global data at the same time and both want the MFMA pipe at
// the same time (measured: MFMA busy 57 %, 32 % of the wave time parked at waitcnt / barrier).  Here the K-tile stream
// of the (persistent) workgroup is cut into a MEMORY phase (request K tile s+2, read the fragments of tile s, split and
// store tile s+1) and an MFMA phase (48 MFMAs), each closed by a barrier, and wave group 1 (waves 4-7, the lower 128
// rows) runs ONE PHASE BEHIND group 0 (one extra barrier up front, one extra for group 0 at the end): while one wave of
// a SIMD issues its MFMAs the other one does its memory phase.  The epilogue of an output tile is part of the memory
// phase that follows its last MFMA phase, so it also runs under the other group's MFMAs.
//   LDS hazards: tile s+1 is written (by both groups) one full phase pair before anyone reads it; the buffer it
//   replaces (tile s-1) was last read two barriers earlier by either group.
// ABL (tools/ablate_pp_gemm.py, env VQCPC_PP_ABL, bias epilogue only) = measurement variants, never used by the library:
//   1: planes written without the split arithmetic (garbage values)          -> +10-12 % (187 -> 211 TFLOP/s, 557056x768x256)
//   2: no global loads after the first K tile                                -> +14-19 %
//   3: both                                                                  -> +25-28 % (234 / 223 / 273 / 241)
//   4: only the B (weight) operand without the split                          -> +5 %
//      CAUTION: variants 1, 3, 4 feed the MFMAs planes made of raw fp32 bit patterns; part of their gain is the higher
//      clock of lower-toggle data (MI355X clocks to its power budget), not removed work.  The REAL thing for variant 4 --
//      weights pre-split once per step into K-tile-major bf16 planes, staged by three 16-byte loads + three ds_write_b128,
//      bit-identical results -- measured 204 vs 204, 190 vs 191, 220 vs 221 TFLOP/s: no gain (row-major bf16 planes:
//      3-5 % SLOWER, 768 cache lines of 32 used bytes per K tile thrash the L1); not kept.
//   8: L2 prefetch touches of the next A cache line two K tiles ahead         -> -3 %
//  16: mid / high A planes read under the MFMAs (56 instead of 72 live fragment registers) and a SECOND raw register
//      set, i.e. every K tile requested two phase pairs before it is split   -> +-0 % (196 vs 197): the latency of the
//      global loads is NOT what the memory phase waits for
//  32: the epilogue without its stores                                        -> +15-17 % at K = 256, +5 % at K = 1024
//      (an LDS-transposed epilogue with 4x fewer, 16-byte stores -- tried on this kernel -- changes nothing, 197 vs 198:
//      it is the M x N x 4 bytes leaving the CU, not the store instructions; de-synchronising the workgroups' tile
//      boundaries with start-up sleeps (across or within XCDs) changes nothing either).  In cycles (tools/pmc_gemm.sh,
//      557056 x 768 x 256): 1.743 M per wave shipped, 1.731 M with the transposed epilogue, 1.524 M without stores -- the
//      output leaves a CU at ~30 bytes per clock whatever the instruction width (tools/micro/store_rate.hip: 36 / 70 B/clk
//      for dword / dwordx4 stores alone), and hiding it would take a second accumulator set
//  64 / 128 / 256: output stores with the nt / sc1 / sc0 cache-policy bits     -> +0.7 % / -1 % / +-0 % (202.0 -> 203.5,
//      190.6 -> 192.0, 205.7 -> 207.3 TFLOP/s with nt): within noise, not adopted (the consumer kernel wants the lines)
// NP = bf16 planes per operand.  3: the exact 3-way truncation split, six products (hh, hm + mh, hl + lh + mm): fp32-class.
//      2: GRADIENT arithmetic (vqcpc_gemm_set_gradient_products(3), opt-in, never the default): each operand is
//         h = rn_bf16(x), m = rn_bf16(x - h) (|x - h - m| <= 2^-18 |x|), products hh + (hm + mh), the mm term (2^-18) dropped:
//         ~2^-17 per product, unbiased -- 4 LDS planes per K tile, 12 fragment reads and 24 MFMAs per phase.
template <int EPI, int ABL = 0, int NP = 3>
__global__ __launch_bounds__(kT2Threads, 2) void gemm_nt_x6_pp_kernel(const float* __restrict__ A, int64_t lda,
                                                                     const float* __restrict__ B, int64_t ldb,
                                                                     float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                     int K, int tiles_n, int tiles, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;           // wm = wave group: 0 leads, 1 runs one phase behind
    const int li = lane & 31, kh = lane >> 5;
    const int T = K / kT2BK;                            // K tiles per output tile (even: K % 32 == 0)
    const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * T;                         // length of this workgroup's K-tile stream

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- load cursor (runs two K tiles ahead of the compute cursor) ----
    // staging is GROUP-LOCAL: group g stages rows [128 g, 128 g + 128) of both operands.  Group 1's MFMA phase still reads
    // A fragments (its own rows 128..255) of tile s-1 while group 0's memory phase already writes tile s+1 into the same
    // buffer -- rows 0..127 only, so the two never touch the same bytes (B fragments are all read in the memory phase).
    const int ld_row = (tid >> 8) * 128 + ((tid & 255) >> 2), ld_c4 = (tid & 3) * 4;
    int ld_tile = blockIdx.x, ld_k = 0;
    const float* a_src;
    const float* b_src;
#define PP_SET_SRC()                                                        \
    {                                                                       \
        const int t_ = xcd_swizzle(min(ld_tile, tiles - 1), tiles);         \
        a_src = A + ((int64_t)(t_ / tiles_n) * kT2 + ld_row) * lda + ld_c4; \
        b_src = B + ((int64_t)(t_ % tiles_n) * kT2 + ld_row) * ldb + ld_c4; \
    }
    PP_SET_SRC()
    float4 xa0, xa1, xb0, xb1;
    float4 ya0, ya1, yb0, yb1;                          // V2: second raw set (prefetch distance 2)
    constexpr bool V3 = (ABL & 1024) != 0 && NP == 3;    // paired loads (below); uses V2's fragment schedule for its registers
    constexpr bool V2 = ((ABL & 16) != 0 || V3) && NP == 3;
#define PP_OPAQUE(V) asm volatile("" : "+v"(V.x), "+v"(V.y), "+v"(V.z), "+v"(V.w));
#define PP_LOAD(S_, PF0, PF1)                                                                    \
    if (!(ABL & 2) || s < 1) {                                                            \
    S_##a0 = *reinterpret_cast<const float4*>(a_src + ld_k);                              \
    S_##a1 = *reinterpret_cast<const float4*>(a_src + (int64_t)64 * lda + ld_k);          \
    S_##b0 = *reinterpret_cast<const float4*>(b_src + ld_k);                              \
    S_##b1 = *reinterpret_cast<const float4*>(b_src + (int64_t)64 * ldb + ld_k);          \
    } else { PP_OPAQUE(S_##a0) PP_OPAQUE(S_##a1) PP_OPAQUE(S_##b0) PP_OPAQUE(S_##b1) }    \
    if (ABL & 8) {                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        const int pk_ = min(ld_k + 2 * kT2BK, K - kT2BK) - ld_c4;                         \
        PF0 = a_src[pk_];                                                                 \
        PF1 = a_src[(int64_t)64 * lda + pk_];                                             \
    }                                                                                     \
    ld_k += kT2BK;                                                                        \
    if (ld_k == K) {                                                                      \
        ld_k = 0;                                                                         \
        ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */ \
        PP_SET_SRC()                                                                      \
    }
#define PP_ST1(R, PLANE0, ROW, BUFP)                                                 \
    {                                                                                \
        uint2 h_, m_, l_;                                                            \
        if (NP == 2) { split2x4(R, h_, m_); l_ = h_; }                               \
        else if (ABL & 1) { h_ = make_uint2(__float_as_uint(R.x), __float_as_uint(R.y)); m_ = make_uint2(__float_as_uint(R.z), __float_as_uint(R.w)); l_ = h_; } \
        else if (ABL & 4) { if (PLANE0 == 0) { split3x4(R, h_, m_, l_); } else { h_ = make_uint2(__float_as_uint(R.x), __float_as_uint(R.y)); m_ = make_uint2(__float_as_uint(R.z), __float_as_uint(R.w)); l_ = h_; } } \
        else split3x4(R, h_, m_, l_);                                                     \
        /* unpadded 32-byte rows, the two 16-byte chunks of a row XOR-swizzled by bit 3 of the row: fragment reads     \
           (16 consecutive rows, one chunk each) and these stores (4 lanes = one row, rows consecutive) are conflict free */ \
        const int o_ = (ROW) * 32 + ((((ld_c4 >> 3) ^ ((ROW) >> 3)) & 1) << 4) + (ld_c4 & 7) * 2; \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 0) * kPPPlane + o_) = h_;     \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 1) * kPPPlane + o_) = m_;     \
        if (NP == 3) *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 2) * kPPPlane + o_) = l_; \
    }
#define PP_STORE(S_, BUFP) \
    PP_ST1(S_##a0, 0, ld_row, BUFP) PP_ST1(S_##a1, 0, ld_row + 64, BUFP) PP_ST1(S_##b0, NP, ld_row, BUFP) PP_ST1(S_##b1, NP, ld_row + 64, BUFP)

    constexpr int kPPPlane = kT2 * 32, kPPBuf = 2 * NP * kPPPlane;  // 8 KB planes, 48 KB (NP = 2: 32 KB) per buffer
    const int swz = ((kh ^ (li >> 3)) & 1) << 4;                   // every fragment row is base + li with base % 16 == 0
    const int a_off = (wm * 128 + li) * 32 + swz;
    const int b_off = NP * kPPPlane + (wn * 64 + li) * 32 + swz;
    unsigned char* const buf0 = smem2;
    unsigned char* const buf1 = smem2 + kPPBuf;
    bf16x8 fb[NP][2], fa[4][NP];                        // all fragments of a K tile: 18 (12) x ds_read_b128 in the memory phase
#define PP_READ_A(BUFP, PC)                                                                                          \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                 \
        fa[mt][PC] = *reinterpret_cast<const bf16x8*>((BUFP) + a_off + (PC) * kPPPlane + mt * 32 * 32);
#define PP_READ_FRAGS(BUFP)                                                                                          \
    _Pragma("unroll") for (int pc = 0; pc < NP; ++pc) {                                                              \
        _Pragma("unroll") for (int tl = 0; tl < 2; ++tl)                                                             \
            fb[pc][tl] = *reinterpret_cast<const bf16x8*>((BUFP) + b_off + pc * kPPPlane + tl * 32 * 32);           \
        if (!V2 || pc == 2) {                                                                                        \
            PP_READ_A(BUFP, pc)                                                                                      \
        }                                                                                                            \
    }
#define PP_TERM(PA, PB)                                                                                              \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                               \
        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][PA], fb[PB][0], acc[mt][0], 0, 0, 0);            \
        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][PA], fb[PB][1], acc[mt][1], 0, 0, 0);            \
    }
#define PP_MFMA_V1() PP_TERM(NP - 1, 0) PP_TERM(0, NP - 1) PP_TERM(1, 1) PP_TERM(1, 0) PP_TERM(0, 1) PP_TERM(0, 0)
    // V2: only the B fragments and the LOW A plane are read in the memory phase; the mid / high A planes of this wave's own
    // rows (group-local staging: nobody writes them meanwhile) are read under the MFMAs that precede their first use, the
    // high plane into the registers of the low one: 56 instead of 72 live fragment registers
#define PP_MFMA_V2(BUFP)                                                          \
    PP_READ_A(BUFP, 1)                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    PP_TERM(2, 0)                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                            \
    PP_READ_A(BUFP, 0)                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    PP_TERM(1, 1) PP_TERM(1, 0)                                                   \
    __builtin_amdgcn_sched_barrier(0);                                            \
    PP_TERM(0, 2) PP_TERM(0, 1) PP_TERM(0, 0)
#define PP_MFMA(BUFP) if constexpr (NP == 2 || (ABL & 512) != 0) { PP_TERM(1, 0) PP_TERM(0, 1) PP_TERM(0, 0) } else if constexpr (V2) { PP_MFMA_V2(BUFP) } else { PP_MFMA_V1() }
#define PP_BARRIER()                          \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue of the output tile with linear index `ep_tile` (buffer addressing, operands one 32x32 tile ahead) ----
    int ep_tile = blockIdx.x;
    constexpr bool HAS_AUX = (EPI & (E_GATE | E_ADD)) != 0;
    const int ldci = (int)ldc;
    const float* xsrc = (EPI & E_GATE) ? ep.gate : ep.add;
    const int ldxi = (int)((EPI & E_GATE) ? ep.ldgate : ep.ldadd);
    float bias_nx0 = 0.0f, bias_nx1 = 0.0f;
    const uint64_t drop_se = rng_seed_eff(ep.seed);
    const uint32_t drop_sh = (uint32_t)(drop_se >> 32), nc1 = (uint32_t)N * kRngMul;
#define PP_BIAS_REQUEST()                                                                                             \
    if (EPI & E_BIAS) {                                                                                                \
        const int tb_ = xcd_swizzle(min(ep_tile, tiles - 1), tiles);                                                   \
        const float* bp_ = ep.bias + (tb_ % tiles_n) * kT2 + wn * 64 + li;                                             \
        bias_nx0 = bp_[0];                                                                                             \
        bias_nx1 = bp_[32];                                                                                            \
    }
#define PP_EPILOGUE()                                                                                                  \
    {                                                                                                                  \
        const int t_ = xcd_swizzle(ep_tile, tiles);                                                                    \
        const int64_t m0 = (int64_t)(t_ / tiles_n) * kT2;                                                              \
        const int n0 = (t_ % tiles_n) * kT2;                                                                           \
        const __amdgpu_buffer_rsrc_t rc =                                                                              \
            __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * ldc + n0), 0, 0x7FFFFFFF, 0x00020000);                  \
        const int voff_c = ((wm * 128 + 4 * kh) * ldci + wn * 64 + li) * 4;                                            \
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)(HAS_AUX ? xsrc + m0 * (int64_t)ldxi + n0 : C), 0, 0x7FFFFFFF, 0x00020000);                         \
        const int voff_x = ((wm * 128 + 4 * kh) * ldxi + wn * 64 + li) * 4;                                            \
        /* second residual (E_ADD2, two launches per step): fetched per 32x32 tile without a prefetch register set */  \
        const int ldx2i = (int)ep.ldadd2;                                                                              \
        const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(                                          \
            (void*)((EPI & E_ADD2) ? ep.add2 + m0 * (int64_t)ldx2i + n0 : C), 0, 0x7FFFFFFF, 0x00020000);              \
        const int voff_x2 = ((wm * 128 + 4 * kh) * ldx2i + wn * 64 + li) * 4;                                          \
        float aux[2][16];                                                                                              \
        if (HAS_AUX) {                                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) aux[0][r] = __builtin_bit_cast(                             \
                float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff_x, (((r & 3) + 8 * (r >> 2)) * ldxi) * 4, 0));    \
        }                                                                                                              \
        const int64_t row_base = m0 + wm * 128 + 4 * kh;                                                               \
        const int col_base = n0 + wn * 64 + li;                                                                        \
        /* this tile's two bias values were requested one output tile ago; request the next tile's now (a load per 32x32 \
           tile inside the store loop cost a full vmcnt(0) drain each: the stores alias C for the compiler) */         \
        const float bv_cur0 = bias_nx0, bv_cur1 = bias_nx1;                                                            \
        ep_tile += gridDim.x;                                                                                          \
        PP_BIAS_REQUEST()                                                                                              \
        /* bit mask of "output > 0" (gemm_common.h): written by the relu / dropout forward (E_MASKOUT), read instead of  \
           the M x N fp32 activation by the relu / dropout backward (E_GATEBITS): 1/32 of the bytes, 4 loads per tile */  \
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)((EPI & (E_MASKOUT | E_GATEBITS)) ? (void*)ep.mask : (void*)C), 0, 0x7FFFFFFF, 0x00020000);         \
        const int nw16 = (N >> 5) * 16;                                   /* bytes of one 4-row group of mask words */  \
        const int mrow4 = (int)((m0 + wm * 128) >> 2), mcb = (n0 >> 5) + wn * 2;                                       \
        u32x4 gb[2][4];                                                                                                \
        if (EPI & E_GATEBITS) {                                                                                        \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) gb[0][jj] = __builtin_bit_cast(u32x4,                     \
                __builtin_amdgcn_raw_buffer_load_b128(rm, kh * nw16, ((mrow4 + 2 * jj) * (N >> 5) + mcb) * 16, 0));    \
        }                                                                                                              \
        _Pragma("unroll") for (int tile = 0; tile < 8; ++tile) {                                                       \
            const int mt = tile >> 1, nt = tile & 1;                                                                   \
            /* dropout hash input of this lane's first row of the tile; the other 15 rows are multiples of nc1 away */ \
            const uint32_t x0t = (EPI & E_DROP) ? rng_x0(drop_se, (uint32_t)(row_base + mt * 32 + ep.row0) * (uint32_t)N + \
                                                                      (uint32_t)(col_base + nt * 32)) : 0u;            \
            if ((EPI & E_GATEBITS) && tile + 1 < 8) {                                                                  \
                const int mt2 = (tile + 1) >> 1, nt2 = (tile + 1) & 1;                                                 \
                _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) gb[(tile + 1) & 1][jj] = __builtin_bit_cast(u32x4,    \
                    __builtin_amdgcn_raw_buffer_load_b128(rm, kh * nw16,                                               \
                        ((mrow4 + mt2 * 8 + 2 * jj) * (N >> 5) + mcb + nt2) * 16, 0));                                 \
            }                                                                                                          \
            uint32_t mword = 0;                                                                                        \
            if (HAS_AUX && tile + 1 < 8) {                                                                             \
                const int mt2 = (tile + 1) >> 1, nt2 = (tile + 1) & 1;                                                 \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) aux[(tile + 1) & 1][r] = __builtin_bit_cast(            \
                    float, __builtin_amdgcn_raw_buffer_load_b32(                                                       \
                               rx, voff_x, ((mt2 * 32 + (r & 3) + 8 * (r >> 2)) * ldxi + nt2 * 32) * 4, 0));           \
            }                                                                                                          \
            float a2[16];                                                                                              \
            if (EPI & E_ADD2) {                                                                                        \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) a2[r] = __builtin_bit_cast(                             \
                    float, __builtin_amdgcn_raw_buffer_load_b32(                                                       \
                               rx2, voff_x2, ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldx2i + nt * 32) * 4, 0));          \
            }                                                                                                          \
            const int col = col_base + nt * 32;                                                                        \
            const float bv = nt ? bv_cur1 : bv_cur0;                                                                   \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                           \
                const int64_t row = row_base + mt * 32 + (r & 3) + 8 * (r >> 2);                                       \
                float v = acc[mt][nt][r] + bv;                                                                         \
                if (EPI & E_RELU) v = fmaxf(v, 0.0f);                                                                  \
                if (EPI & E_DROP)   /* == drop_scale(ep.seed, (row + ep.row0) * N + col, ..): thr > 0 on this path */   \
                    v *= rng_u24_from_x0(x0t + (uint32_t)((r & 3) + 8 * (r >> 2)) * nc1, drop_sh) >= ep.thr ? ep.inv_keep : 0.0f; \
                if (EPI & E_GATE) v *= (aux[tile & 1][r] > 0.0f ? ep.gate_scale : 0.0f);                               \
                if (EPI & E_ADD) v += aux[tile & 1][r];                                                                \
                if (EPI & E_ADD2) v += a2[r];                                                                          \
                if (EPI & E_GATEBITS) v = ((gb[tile & 1][r >> 2][r & 3] >> li) & 1u) ? v * ep.gate_scale : 0.0f;       \
                if (EPI & E_MASKOUT) {                                                                                 \
                    const uint64_t bal = __ballot(v > 0.0f);      /* low word: this row for kh = 0, high word: kh = 1 */ \
                    /* s_nop: the ballot is an SGPR pair written by a VALU compare; nothing pads hazards in front of asm */ \
                    asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"                 \
                                 : "+v"(mword) : "s"((uint32_t)bal), "n"(r), "s"((uint32_t)(bal >> 32)), "n"(16 + r));   \
                }                                                                                                      \
                if (ABL & 32) { asm volatile("" :: "v"(v)); } else                                                      \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff_c,                     \
                                                      ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldci + nt * 32) * 4,       \
                                                      (ABL & 64 ? 2 : 0) | (ABL & 128 ? 16 : 0) | (ABL & 256 ? 1 : 0)); \
                acc[mt][nt][r] = 0.0f;                                                                                 \
            }                                                                                                          \
            if ((EPI & E_MASKOUT) && lane < 32) {    /* lane = r + 16 kh holds the word of row (r & 3) + 8 (r >> 2) + 4 kh */ \
                const int jj = (lane >> 2) & 3, khh = lane >> 4;                                                       \
                __builtin_amdgcn_raw_buffer_store_b32(mword, rm, (2 * jj + khh) * nw16 + (lane & 3) * 4,               \
                                                      ((mrow4 + mt * 8) * (N >> 5) + mcb + nt) * 16, 0);               \
            }                                                                                                          \
        }                                                                                                              \
    }

    // (Round 4, measured and not kept: requesting K tile s+2 BEFORE the epilogue's stores, so that it does not queue behind
    // them -- as its own code path at tile boundaries it spilled, 2 x slower; as one path (store + request ahead of the
    // fragment reads in every phase) the memory phase loses the overlap of the fragment-read latency with the split
    // arithmetic: 5-7 % slower at every K.  profiles/r04_perf_log.md)
    // one phase pair for stream position s (RB_ = LDS buffer with K tile s, WB_ = buffer for tile s+1).  ONE raw-operand
    // register set: tile s+1 (requested in the previous memory phase) is split and stored, then the same registers are
    // reused for the request of tile s+2 -- a full MFMA phase + two barriers of latency cover.
#define PP_PHASES(RB_, WB_, PF0_, PF1_, S_)                                               \
    {                                                                             \
        if (kt == 0 && s > 0) PP_EPILOGUE()                                       \
        PP_READ_FRAGS(RB_)                                                        \
        PP_STORE(S_, WB_)                                                         \
        if (ABL & 8) asm volatile("" :: "v"(PF0_), "v"(PF1_));                    \
        PP_LOAD(S_, PF0_, PF1_)                                                   \
        PP_BARRIER()                                                              \
        if (!(ABL & 6144)) __builtin_amdgcn_s_setprio(1);                         \
        PP_MFMA(RB_)                                                              \
        if (!(ABL & 6144)) __builtin_amdgcn_s_setprio(0);                         \
        PP_BARRIER()                                                              \
        ++s;                                                                      \
        kt = (kt + 1 == T) ? 0 : kt + 1;                                          \
    }

    // V3 (ABL 1024): K tiles 2j and 2j+1 are the two 64-byte halves of the SAME 128-byte lines of both operands.  Requested
    // one K step apart (as above) the second half has left the L1 by the time it is asked for and every line crosses the
    // L2 -> L1 path twice; here both halves are requested back to back (sets x and y) in the even phases, none in the odd
    // ones: x (tile 2j) is split in the next phase, y (tile 2j+1) in the one after.
#define PP_PHASES3(RB_, WB_, S_, LOADPAIR_)                                       \
    {                                                                             \
        if (kt == 0 && s > 0) PP_EPILOGUE()                                       \
        PP_READ_FRAGS(RB_)                                                        \
        PP_STORE(S_, WB_)                                                         \
        if (LOADPAIR_) { PP_LOAD(x, pf0, pf1) PP_LOAD(y, pf2, pf3) }              \
        PP_BARRIER()                                                              \
        if (!(ABL & 6144)) __builtin_amdgcn_s_setprio(1);                         \
        PP_MFMA(RB_)                                                              \
        if (!(ABL & 6144)) __builtin_amdgcn_s_setprio(0);                         \
        PP_BARRIER()                                                              \
        ++s;                                                                      \
        kt = (kt + 1 == T) ? 0 : kt + 1;                                          \
    }

    // prologue: K tile 0 split into buffer 0, tile 1 requested
    int s = 0, kt = 0;
    float pf0 = 0.f, pf1 = 0.f, pf2 = 0.f, pf3 = 0.f;
    PP_BIAS_REQUEST()
    PP_LOAD(x, pf0, pf1)
    if (V3) { PP_LOAD(y, pf2, pf3) }                     // V3: tiles 0 (x) and 1 (y) requested together
    PP_STORE(x, buf0)
    if (!V3) { PP_LOAD(x, pf2, pf3) }
    if (V2 && !V3) { PP_LOAD(y, pf2, pf3) }              // V2: tiles 1 (x) and 2 (y) requested
    PP_BARRIER()
    if (wm == 1) { PP_BARRIER() }                        // group 1 falls one phase behind
    // ABL 2048: no priority flips at all; ABL 4096: static priority 1 for the younger wave group only (MI355X_MICROARCH.md,
    // "Two waves per SIMD", items 2 and 4)
    if ((ABL & 4096) && __builtin_amdgcn_readfirstlane(wm) == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    while (s < S) {
        if (V3) {
            PP_PHASES3(buf0, buf1, y, true)              // even phase: tile s+1 (y) stored, tiles s+2 (x), s+3 (y) requested
            PP_PHASES3(buf1, buf0, x, false)             // odd phase:  tile s+1 (x) stored
        } else if (V2) {
            PP_PHASES(buf0, buf1, pf0, pf1, x)
            PP_PHASES(buf1, buf0, pf2, pf3, y)
        } else {
            PP_PHASES(buf0, buf1, pf0, pf1, x)
            PP_PHASES(buf1, buf0, pf2, pf3, x)
        }
    }
    if (wm == 0) { PP_BARRIER() }                        // pairs with group 1's last barrier
    PP_EPILOGUE()
#undef PP_PHASES
#undef PP_PHASES3
#undef PP_EPILOGUE
#undef PP_BIAS_REQUEST
#undef PP_BARRIER
#undef PP_MFMA
#undef PP_MFMA_V1
#undef PP_MFMA_V2
#undef PP_READ_A
#undef PP_TERM
#undef PP_READ_FRAGS
#undef PP_STORE
#undef PP_ST1
#undef PP_LOAD
#undef PP_SET_SRC
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr int TM = 32;   // rows of the contraction (M) dimension per step

// FULL: N % 128 == 0, K % 128 == 0, M % 32 == 0 -> no bounds checks.
template <bool FULL>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_tn_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ B, int64_t ldb, int64_t M,
                                                                 int N, int K, int tiles_k, int64_t rows_per_split,
                                                                 float* __restrict__ ws, float* __restrict__ ws_bias) {
    __shared__ __attribute__((aligned(16))) float As[TM * BM];
    __shared__ __attribute__((aligned(16))) float Bs[TM * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * BM, k0 = tk * BN;
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);
    const bool want_bias = (ws_bias != nullptr) && tk == 0 && wn == 0;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[2] = {0.0f, 0.0f};

    const int ld_row = tid >> 5, ld_c4 = (tid & 31) * 4;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define TN_LOAD1(RA, RB, I, MM)                                                                              \
    {                                                                                                        \
        const int64_t r_ = (MM) + ld_row + 8 * (I);                                                          \
        if (FULL) {                                                                                          \
            RA = *reinterpret_cast<const float4*>(A + r_ * lda + n0 + ld_c4);                                \
            RB = *reinterpret_cast<const float4*>(B + r_ * ldb + k0 + ld_c4);                                \
        } else {                                                                                             \
            const bool rok_ = r_ < m_end;                                                                    \
            RA = (rok_ && n0 + ld_c4 < N) ? *reinterpret_cast<const float4*>(A + r_ * lda + n0 + ld_c4) : zero4; \
            RB = (rok_ && k0 + ld_c4 < K) ? *reinterpret_cast<const float4*>(B + r_ * ldb + k0 + ld_c4) : zero4; \
        }                                                                                                    \
    }
#define TN_LOAD(MM) TN_LOAD1(ra0, rb0, 0, MM) TN_LOAD1(ra1, rb1, 1, MM) TN_LOAD1(ra2, rb2, 2, MM) TN_LOAD1(ra3, rb3, 3, MM)
#define TN_STORE1(RA, RB, I)                                                   \
    *reinterpret_cast<float4*>(As + (ld_row + 8 * (I)) * BM + ld_c4) = RA;     \
    *reinterpret_cast<float4*>(Bs + (ld_row + 8 * (I)) * BN + ld_c4) = RB;
#define TN_STORE() TN_STORE1(ra0, rb0, 0) TN_STORE1(ra1, rb1, 1) TN_STORE1(ra2, rb2, 2) TN_STORE1(ra3, rb3, 3)

    if (m_begin < m_end) {
        TN_LOAD(m_begin)
        TN_STORE()
    }
    __syncthreads();
    const float* ap = As + kh * BM + wm * 64 + li;
    const float* bp = Bs + kh * BN + wn * 64 + li;
    for (int64_t mm = m_begin; mm < m_end; mm += TM) {
        const bool more = mm + TM < m_end;
        if (more) { TN_LOAD(mm + TM) }
        // operands of 4 contraction steps (8 rows of M) are fetched per batch: one LDS latency per 16 MFMAs
#pragma unroll
        for (int g = 0; g < TM / 8; ++g) {
            float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] = ap[(8 * g + 2 * u) * BM];
                a1[u] = ap[(8 * g + 2 * u) * BM + 32];
                b0[u] = bp[(8 * g + 2 * u) * BN];
                b1[u] = bp[(8 * g + 2 * u) * BN + 32];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the 8 reads ahead of the 16 MFMAs (hipcc sinks them otherwise)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc[1][1], 0, 0, 0);
                bsum[0] += a0[u];
                bsum[1] += a1[u];
            }
        }
        if (more) {
            __syncthreads();
            TN_STORE()
            __syncthreads();
        }
    }
#undef TN_LOAD
#undef TN_LOAD1
#undef TN_STORE
#undef TN_STORE1

    float* out = ws + (int64_t)blockIdx.y * N * K;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = k0 + wn * 64 + nt * 32 + li;
            if (!FULL && col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (FULL || row < N) out[(int64_t)row * K + col] = acc[mt][nt][r];
            }
        }
    }
    if (want_bias) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float tot = bsum[mt] + __shfl_xor(bsum[mt], 32, 64);
            const int row = n0 + wm * 64 + mt * 32 + li;
            if (kh == 0 && row < N) ws_bias[(int64_t)blockIdx.y * N + row] = tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x6 weight-gradient GEMM.  The contraction index (m) is the ROW index of both operands, while the bf16 MFMA wants
// 8 consecutive contraction elements per lane.  Each thread therefore stages a 2(m) x 4(n) block: the exact 3-way split
// is done in registers and the two rows are interleaved into dwords (row 2r in the low half, 2r+1 in the high half), so
// the LDS planes are [row pair][128 columns] dwords and a fragment is 4 conflict-free ds_read_b32 (no transposes).
constexpr int kTnRS = BM * 4 + 16;                    // bytes per row pair (128 dwords + 16 B pad)
constexpr int kTnPlane = (TM / 2) * kTnRS;            // bytes per plane

__device__ __forceinline__ void split3_pair4(const float4& r0, const float4& r1, uint4& h, uint4& m, uint4& l) {
    uint32_t h0, m0, l0, h1, m1, l1;
#define TN_SPLIT_E(E, X)                       \
    split3(r0.E, h0, m0, l0);                  \
    split3(r1.E, h1, m1, l1);                  \
    h.X = pack_hi(h0, h1);                     \
    m.X = pack_hi(m0, m1);                     \
    l.X = pack_hi(l0, l1);
    TN_SPLIT_E(x, x) TN_SPLIT_E(y, y) TN_SPLIT_E(z, z) TN_SPLIT_E(w, w)
#undef TN_SPLIT_E
}

// (bx, by) = (output tile, split of the contraction): blockIdx of the plain launch, looked up per problem in the grouped one
template <bool FULL, int MODE>
__device__ __forceinline__ void gemm_tn_x6_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                int64_t ldb, int64_t M, int N, int K, int tiles_k, int64_t rows_per_split,
                                                float* __restrict__ ws, float* __restrict__ ws_bias, int bx, int by) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[6 * kTnPlane];   // A_h A_m A_l B_h B_m B_l
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int tn = bx / tiles_k, tk = bx % tiles_k;
    const int n0 = tn * BM, k0 = tk * BN;
    const int64_t m_begin = (int64_t)by * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);
    const bool want_bias = (ws_bias != nullptr) && tk == 0;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);    // column sums of A for this thread's 4 columns

    // staging: thread = (column quad c4, row pair rp0 / rp0 + 8)
    const int c4 = (tid & 31) * 4, rp0 = tid >> 5;
    float4 a00, a01, a10, a11, b00, b01, b10, b11;    // [item][row of the pair]
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define X6_LD(DST, P, LD, COL0, LIM, ROW)                                                                   \
    DST = (FULL || ((ROW) < m_end && (COL0) + c4 < (LIM))) ? *reinterpret_cast<const float4*>((P) + (ROW) * (LD) + (COL0) + c4) : zero4;
#define X6_LOAD(MM)                                            \
    {                                                          \
        const int64_t r0_ = (MM) + 2 * rp0, r1_ = r0_ + 16;    \
        X6_LD(a00, A, lda, n0, N, r0_)                         \
        X6_LD(a01, A, lda, n0, N, r0_ + 1)                     \
        X6_LD(a10, A, lda, n0, N, r1_)                         \
        X6_LD(a11, A, lda, n0, N, r1_ + 1)                     \
        X6_LD(b00, B, ldb, k0, K, r0_)                         \
        X6_LD(b01, B, ldb, k0, K, r0_ + 1)                     \
        X6_LD(b10, B, ldb, k0, K, r1_)                         \
        X6_LD(b11, B, ldb, k0, K, r1_ + 1)                     \
    }
#define X6_ST(R0, R1, PLANE0, RP)                                                              \
    if (MODE == 2) {                                                                           \
        const uint4 h_ = make_uint4(cvt_pk_bf16(R0.x, R1.x), cvt_pk_bf16(R0.y, R1.y), cvt_pk_bf16(R0.z, R1.z), \
                                    cvt_pk_bf16(R0.w, R1.w));                                  \
        *reinterpret_cast<uint4*>(smem + (PLANE0) * kTnPlane + (RP) * kTnRS + c4 * 4) = h_;    \
    } else {                                                                                   \
        uint4 h_, m_, l_;                                                                      \
        split3_pair4(R0, R1, h_, m_, l_);                                                      \
        const int o_ = (RP) * kTnRS + c4 * 4;                                                  \
        *reinterpret_cast<uint4*>(smem + ((PLANE0) + 0) * kTnPlane + o_) = h_;                 \
        *reinterpret_cast<uint4*>(smem + ((PLANE0) + 1) * kTnPlane + o_) = m_;                 \
        *reinterpret_cast<uint4*>(smem + ((PLANE0) + 2) * kTnPlane + o_) = l_;                 \
    }
#define X6_STORE()                                                                             \
    X6_ST(a00, a01, 0, rp0) X6_ST(a10, a11, 0, rp0 + 8) X6_ST(b00, b01, 3, rp0) X6_ST(b10, b11, 3, rp0 + 8) \
    if (want_bias) {                                                                           \
        bsum.x += (a00.x + a01.x) + (a10.x + a11.x);                                           \
        bsum.y += (a00.y + a01.y) + (a10.y + a11.y);                                           \
        bsum.z += (a00.z + a01.z) + (a10.z + a11.z);                                           \
        bsum.w += (a00.w + a01.w) + (a10.w + a11.w);                                           \
    }

    if (m_begin < m_end) {
        X6_LOAD(m_begin)
        X6_STORE()
    }
    __syncthreads();
    // fragment of k16 step ks, lane (li, kh): row pairs ks*8 + kh*4 + 0..3, one dword each
    const unsigned char* abase = smem + (kh * 4) * kTnRS + (wm * 64 + li) * 4;
    const unsigned char* bbase = smem + 3 * kTnPlane + (kh * 4) * kTnRS + (wn * 64 + li) * 4;
    for (int64_t mm = m_begin; mm < m_end; mm += TM) {
        const bool more = mm + TM < m_end;
        if (more) { X6_LOAD(mm + TM) }
#pragma unroll
        for (int ks = 0; ks < TM / 16; ++ks) {
            constexpr int NP = MODE == 2 ? 1 : 3;              // planes in use
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    uint4 ua, ub;
                    const unsigned char* pa = abase + pc * kTnPlane + ks * 8 * kTnRS + tl * 128;
                    const unsigned char* pb = bbase + pc * kTnPlane + ks * 8 * kTnRS + tl * 128;
                    ua.x = *reinterpret_cast<const uint32_t*>(pa);
                    ua.y = *reinterpret_cast<const uint32_t*>(pa + kTnRS);
                    ua.z = *reinterpret_cast<const uint32_t*>(pa + 2 * kTnRS);
                    ua.w = *reinterpret_cast<const uint32_t*>(pa + 3 * kTnRS);
                    ub.x = *reinterpret_cast<const uint32_t*>(pb);
                    ub.y = *reinterpret_cast<const uint32_t*>(pb + kTnRS);
                    ub.z = *reinterpret_cast<const uint32_t*>(pb + 2 * kTnRS);
                    ub.w = *reinterpret_cast<const uint32_t*>(pb + 3 * kTnRS);
                    a[pc][tl] = __builtin_bit_cast(bf16x8, ua);
                    b[pc][tl] = __builtin_bit_cast(bf16x8, ub);
                }
#define X6_TERM(PA, PB)                                                                                        \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][0], acc[0][0], 0, 0, 0);                \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][1], acc[0][1], 0, 0, 0);                \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][0], acc[1][0], 0, 0, 0);                \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][1], acc[1][1], 0, 0, 0);
            if (MODE != 2) { X6_TERM(2, 0) X6_TERM(0, 2) X6_TERM(1, 1) X6_TERM(1, 0) X6_TERM(0, 1) }
            X6_TERM(0, 0)
#undef X6_TERM
        }
        if (more) {
            __syncthreads();
            X6_STORE()
            __syncthreads();
        }
    }
#undef X6_LD
#undef X6_LOAD
#undef X6_ST
#undef X6_STORE

    float* out = ws + (int64_t)by * N * K;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = k0 + wn * 64 + nt * 32 + li;
            if (!FULL && col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (FULL || row < N) out[(int64_t)row * K + col] = acc[mt][nt][r];
            }
        }
    }
    if (want_bias) {
        // the 8 threads that share a column quad (one per row-pair group) are combined through LDS in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [8][128]
        *reinterpret_cast<float4*>(red + rp0 * BM + c4) = bsum;
        __syncthreads();
        if (tid < BM) {
            float tot = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) tot += red[g * BM + tid];
            if (n0 + tid < N) ws_bias[(int64_t)by * N + n0 + tid] = tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <bool FULL, int MODE>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_tn_x6_kernel(const float* __restrict__ A, int64_t lda,
                                                                    const float* __restrict__ B, int64_t ldb, int64_t M,
                                                                    int N, int K, int tiles_k, int64_t rows_per_split,
                                                                    float* __restrict__ ws, float* __restrict__ ws_bias) {
    gemm_tn_x6_tile<FULL, MODE>(A, lda, B, ldb, M, N, K, tiles_k, rows_per_split, ws, ws_bias, (int)blockIdx.x, (int)blockIdx.y);
}

// GROUPED weight gradients (student / decoder steps: ~100 products of 3072 x 512 x 512 ... 2048 per step, each too small to
// fill the chip: 27 us + a 6 us partial-sum launch apiece).  The weight gradients of a backward pass are independent of each
// other and only the optimiser reads them, so the host defers them (ops.direct_weight_gradients) and issues up to kTnGroup
// products per launch: the problem table travels BY VALUE in the kernel arguments (graph-capturable, no device table to keep
// alive), a workgroup finds its problem by a scalar scan of the workgroup prefix.
constexpr int kTnGroup = 32;
struct TnGroupArgs {
    const float* A[kTnGroup];
    const float* B[kTnGroup];
    float* ws[kTnGroup];
    float* wsb[kTnGroup];
    int lda[kTnGroup], ldb[kTnGroup], M[kTnGroup], N[kTnGroup], K[kTnGroup], tiles_k[kTnGroup], tiles[kTnGroup], rows_per_split[kTnGroup];
    int wg_begin[kTnGroup + 1];
    int n;
};

template <bool FULL, int MODE>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_tn_x6_grouped_kernel(const TnGroupArgs g) {
    const int b = (int)blockIdx.x;
    int p = 0;
    for (int i = 1; i < g.n; ++i) p += (b >= g.wg_begin[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    const int local = b - g.wg_begin[p];
    gemm_tn_x6_tile<FULL, MODE>(g.A[p], g.lda[p], g.B[p], g.ldb[p], g.M[p], g.N[p], g.K[p], g.tiles_k[p], g.rows_per_split[p],
                                g.ws[p], g.wsb[p], local % g.tiles[p], local / g.tiles[p]);
}

// partial sums of a group -> the gradient buffers (fixed split order; accumulate = add to what the buffer holds)
struct RedGroupArgs {
    const float* ws[kTnGroup];
    float* out[kTnGroup];
    const float* ws2[kTnGroup];
    float* out2[kTnGroup];
    int count[kTnGroup], count2[kTnGroup], nsplit[kTnGroup];
    int blk_begin[kTnGroup + 1];
    int n, accumulate;
};

__global__ __launch_bounds__(256) void reduce_splits_grouped_kernel(const RedGroupArgs g) {
    const int b = (int)blockIdx.x;
    int p = 0;
    for (int i = 1; i < g.n; ++i) p += (b >= g.blk_begin[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    int64_t q = (int64_t)(b - g.blk_begin[p]) * 256 + threadIdx.x;                       // float4 index
    const float* ws = g.ws[p];
    float* out = g.out[p];
    int64_t count = g.count[p];
    const int64_t first = (count / 4 + 255) / 256 * 256;                                 // segment 2 (bias): workgroup boundary
    if (q >= first) {
        q -= first;
        ws = g.ws2[p];
        out = g.out2[p];
        count = g.count2[p];
    }
    if (q * 4 >= count) return;
    const int nsplit = g.nsplit[p];
    const float4* src = reinterpret_cast<const float4*>(ws) + q;
    const int64_t st4 = count / 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s_ = 0; s_ < nsplit; ++s_) {
        const float4 v = src[s_ * st4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(out) + q;
    if (g.accumulate) {
        const float4 prev = *o;
        acc.x += prev.x; acc.y += prev.y; acc.z += prev.z; acc.w += prev.w;
    }
    *o = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x6 weight-gradient GEMM, 256 x 256 output tile, 16 rows of M per step, 8 waves (2 x 4, wave tile 128 x 64), two LDS
// buffers, prefetch distance 2, branch-free steady-state loop (same structure as gemm_nt_x6_256_kernel; staging as
// gemm_tn_x6_kernel: 2(m) x 4(n) register blocks, rows interleaved into dwords -> fragments are 4 ds_read_b32).
constexpr int kW2TM = 16;                                  // contraction rows per step
constexpr int kW2RS = kT2 * 4 + 16;                        // bytes per row pair (256 dwords + pad)
constexpr int kW2Plane = (kW2TM / 2) * kW2RS;              // 8 320 B
constexpr int kW2Buf = 6 * kW2Plane;                       // 49 920 B; two buffers = 99 840 B
constexpr int kTPRS = (64 + 4) * 4;                        // ping-pong kernel: bytes per row pair of a 64-column block
constexpr int kTPBlk = (kW2TM / 2) * kTPRS;                // 2 176 B per column block
constexpr int kTPPlane = 4 * kTPBlk;                       // 8 704 B
constexpr int kTPBuf = 6 * kTPPlane;                       // 52 224 B; two buffers = 104 448 B

__global__ __launch_bounds__(kT2Threads, 2) void gemm_tn_x6_256_kernel(const float* __restrict__ A, int64_t lda,
                                                                      const float* __restrict__ B, int64_t ldb, int64_t M,
                                                                      int N, int K, int tiles_k, int64_t rows_per_split,
                                                                      float* __restrict__ ws, float* __restrict__ ws_bias) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * kT2, k0 = tk * kT2;
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);          // (m_end - m_begin) % 32 == 0 (host)
    const bool want_bias = (ws_bias != nullptr) && tk == 0;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging: thread = (column quad c4 of 64, row pair rp of 8): rows 2rp, 2rp+1 of the 16-row step
    const int c4 = (tid & 63) * 4, rp = tid >> 6;
    const float* a_src = A + (m_begin + 2 * rp) * lda + n0 + c4;
    const float* b_src = B + (m_begin + 2 * rp) * ldb + k0 + c4;
    float4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
#define W2_LOAD(S, MM)                                                                   \
    S##a0 = *reinterpret_cast<const float4*>(a_src + (int64_t)(MM) * lda);               \
    S##a1 = *reinterpret_cast<const float4*>(a_src + (int64_t)((MM) + 1) * lda);         \
    S##b0 = *reinterpret_cast<const float4*>(b_src + (int64_t)(MM) * ldb);               \
    S##b1 = *reinterpret_cast<const float4*>(b_src + (int64_t)((MM) + 1) * ldb);
#define W2_STORE(S, BUFP)                                                                \
    {                                                                                    \
        uint4 h_, m_, l_;                                                                \
        const int o_ = rp * kW2RS + c4 * 4;                                              \
        split3_pair4(S##a0, S##a1, h_, m_, l_);                                          \
        *reinterpret_cast<uint4*>((BUFP) + 0 * kW2Plane + o_) = h_;                      \
        *reinterpret_cast<uint4*>((BUFP) + 1 * kW2Plane + o_) = m_;                      \
        *reinterpret_cast<uint4*>((BUFP) + 2 * kW2Plane + o_) = l_;                      \
        split3_pair4(S##b0, S##b1, h_, m_, l_);                                          \
        *reinterpret_cast<uint4*>((BUFP) + 3 * kW2Plane + o_) = h_;                      \
        *reinterpret_cast<uint4*>((BUFP) + 4 * kW2Plane + o_) = m_;                      \
        *reinterpret_cast<uint4*>((BUFP) + 5 * kW2Plane + o_) = l_;                      \
        if (want_bias) {                                                                 \
            bsum.x += S##a0.x + S##a1.x;                                                 \
            bsum.y += S##a0.y + S##a1.y;                                                 \
            bsum.z += S##a0.z + S##a1.z;                                                 \
            bsum.w += S##a0.w + S##a1.w;                                                 \
        }                                                                                \
    }
#define W2_FRAG(DST, BASE)                                                               \
    {                                                                                    \
        uint4 u_;                                                                        \
        u_.x = *reinterpret_cast<const uint32_t*>(BASE);                                 \
        u_.y = *reinterpret_cast<const uint32_t*>((BASE) + kW2RS);                       \
        u_.z = *reinterpret_cast<const uint32_t*>((BASE) + 2 * kW2RS);                   \
        u_.w = *reinterpret_cast<const uint32_t*>((BASE) + 3 * kW2RS);                   \
        DST = __builtin_bit_cast(bf16x8, u_);                                            \
    }
#define W2_TERM(PA, PB)                                                                                                   \
    acc[hf * 2 + 0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][0], acc[hf * 2 + 0][0], 0, 0, 0);          \
    acc[hf * 2 + 0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][1], acc[hf * 2 + 0][1], 0, 0, 0);          \
    acc[hf * 2 + 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][0], acc[hf * 2 + 1][0], 0, 0, 0);          \
    acc[hf * 2 + 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][1], acc[hf * 2 + 1][1], 0, 0, 0);
#define W2_COMPUTE(BUFP)                                                                                                  \
    {                                                                                                                     \
        const unsigned char* ab_ = (BUFP) + (kh * 4) * kW2RS + (wm * 128 + li) * 4;                                       \
        const unsigned char* bb_ = (BUFP) + 3 * kW2Plane + (kh * 4) * kW2RS + (wn * 64 + li) * 4;                         \
        bf16x8 b[3][2];                                                                                                   \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                                  \
            _Pragma("unroll") for (int tl = 0; tl < 2; ++tl) W2_FRAG(b[pc][tl], bb_ + pc * kW2Plane + tl * 128)           \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                                \
            bf16x8 a[3][2];                                                                                               \
            _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                              \
                _Pragma("unroll") for (int tl = 0; tl < 2; ++tl) W2_FRAG(a[pc][tl], ab_ + pc * kW2Plane + (hf * 2 + tl) * 128) \
            W2_TERM(2, 0) W2_TERM(0, 2) W2_TERM(1, 1) W2_TERM(1, 0) W2_TERM(0, 1) W2_TERM(0, 0)                            \
        }                                                                                                                 \
    }

    unsigned char* const buf0 = smemw;
    unsigned char* const buf1 = smemw + kW2Buf;
    const int64_t rows = m_end - m_begin;                     // multiple of 32, >= 32 when non-empty
    if (rows > 0) {
        W2_LOAD(x, 0)
        W2_LOAD(y, kW2TM)
        W2_STORE(x, buf0)
        __syncthreads();
        for (int64_t mm = 0; mm < rows - 2 * kW2TM; mm += 2 * kW2TM) {
            W2_LOAD(x, mm + 2 * kW2TM)
            __builtin_amdgcn_sched_barrier(0);
            W2_COMPUTE(buf0)
            __builtin_amdgcn_sched_barrier(0);
            W2_STORE(y, buf1)
            __syncthreads();
            W2_LOAD(y, mm + 3 * kW2TM)
            __builtin_amdgcn_sched_barrier(0);
            W2_COMPUTE(buf1)
            __builtin_amdgcn_sched_barrier(0);
            W2_STORE(x, buf0)
            __syncthreads();
        }
        W2_COMPUTE(buf0)
        __builtin_amdgcn_sched_barrier(0);
        W2_STORE(y, buf1)
        __syncthreads();
        W2_COMPUTE(buf1)
    }
#undef W2_LOAD
#undef W2_STORE
#undef W2_FRAG
#undef W2_TERM
#undef W2_COMPUTE

    float* out = ws + (int64_t)blockIdx.y * N * K;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = k0 + wn * 64 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                out[(int64_t)row * K + col] = acc[mt][nt][r];
            }
        }
    }
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smemw);          // [8][256]
        *reinterpret_cast<float4*>(red + rp * kT2 + c4) = bsum;
        __syncthreads();
        if (tid < kT2) {
            float tot = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) tot += red[g * kT2 + tid];
            ws_bias[(int64_t)blockIdx.y * N + n0 + tid] = tot;
        }
    }
}

// Ping-pong variant of gemm_tn_x6_256_kernel (same tile, staging, LDS image and MFMA order => the same partial sums): the
// 16-row steps of the workgroup are cut into a memory phase (read the 18 fragments of step s, split + store step s+1,
// request step s+2) and an MFMA phase (48 MFMAs), each closed by a barrier; wave group 1 (the lower 128 output rows) runs one
// phase behind group 0, so that on every SIMD one wave issues MFMAs while the other one does its memory phase (the lockstep
// kernel kept the matrix pipes busy 64 % of the cycles: tools/pmc_gemm_tn.sh).
template <int NP>
__global__ __launch_bounds__(kT2Threads, 2) void gemm_tn_x6_pp_kernel(const float* __restrict__ A, int64_t lda,
                                                                      const float* __restrict__ B, int64_t ldb, int64_t M,
                                                                      int N, int K, int tiles_k, int64_t rows_per_split,
                                                                      float* __restrict__ ws, float* __restrict__ ws_bias) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    // (tile, split) of this workgroup.  The output tiles of ONE split read the same rows of A and B (they differ in the
    // column tile of one operand only); in dispatch order (x fastest) they would land on different XCDs, i.e. behind
    // different L2s, and every shared row would be fetched once per tile: 1.6 x the algorithmic HBM bytes.  Remapped so that
    // the tiles of a split are neighbours on one XCD (workgroup L of the linearised grid runs on XCD L % 8).
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.x > 1 && gridDim.y % 8 == 0) {
        const int L = by * (int)gridDim.x + bx;
        const int j = L >> 3;
        bx = j % (int)gridDim.x;
        by = (j / (int)gridDim.x) * 8 + (L & 7);
    }
    const int tn = bx / tiles_k, tk = bx % tiles_k;
    const int n0 = tn * kT2, k0 = tk * kT2;
    const int64_t m_begin = (int64_t)by * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);          // (m_end - m_begin) % 32 == 0 (host)
    const bool want_bias = (ws_bias != nullptr) && tk == 0;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging: thread = (column quad c4 of 64, row pair rp of 8): rows 2rp, 2rp+1 of the 16-row step
    const int c4 = (tid & 63) * 4, rp = tid >> 6;
    const float* a_src = A + (m_begin + 2 * rp) * lda + n0 + c4;
    const float* b_src = B + (m_begin + 2 * rp) * ldb + k0 + c4;
    float4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
#define W2_LOAD(S, MM)                                                                   \
    S##a0 = *reinterpret_cast<const float4*>(a_src + (int64_t)(MM) * lda);               \
    S##a1 = *reinterpret_cast<const float4*>(a_src + (int64_t)((MM) + 1) * lda);         \
    S##b0 = *reinterpret_cast<const float4*>(b_src + (int64_t)(MM) * ldb);               \
    S##b1 = *reinterpret_cast<const float4*>(b_src + (int64_t)((MM) + 1) * ldb);
#define W2_STORE(S, BUFP)                                                                \
    {                                                                                    \
        uint4 h_, m_, l_;                                                                \
        const int o_ = (c4 >> 6) * kTPBlk + rp * kTPRS + (c4 & 63) * 4;                  \
        if constexpr (NP == 2) { split2_pair4(S##a0, S##a1, h_, m_); } else { split3_pair4(S##a0, S##a1, h_, m_, l_); } \
        *reinterpret_cast<uint4*>((BUFP) + 0 * kTPPlane + o_) = h_;                      \
        *reinterpret_cast<uint4*>((BUFP) + 1 * kTPPlane + o_) = m_;                      \
        if constexpr (NP == 3) *reinterpret_cast<uint4*>((BUFP) + 2 * kTPPlane + o_) = l_; \
        if constexpr (NP == 2) { split2_pair4(S##b0, S##b1, h_, m_); } else { split3_pair4(S##b0, S##b1, h_, m_, l_); } \
        *reinterpret_cast<uint4*>((BUFP) + (NP + 0) * kTPPlane + o_) = h_;               \
        *reinterpret_cast<uint4*>((BUFP) + (NP + 1) * kTPPlane + o_) = m_;               \
        if constexpr (NP == 3) *reinterpret_cast<uint4*>((BUFP) + (NP + 2) * kTPPlane + o_) = l_; \
        if (want_bias && store_counts) {                                                 \
            bsum.x += S##a0.x + S##a1.x;                                                 \
            bsum.y += S##a0.y + S##a1.y;                                                 \
            bsum.z += S##a0.z + S##a1.z;                                                 \
            bsum.w += S##a0.w + S##a1.w;                                                 \
        }                                                                                \
    }
#define W2_FRAG(DST, BASE)                                                               \
    {                                                                                    \
        uint4 u_;                                                                        \
        u_.x = *reinterpret_cast<const uint32_t*>(BASE);                                 \
        u_.y = *reinterpret_cast<const uint32_t*>((BASE) + kW2RS);                       \
        u_.z = *reinterpret_cast<const uint32_t*>((BASE) + 2 * kW2RS);                   \
        u_.w = *reinterpret_cast<const uint32_t*>((BASE) + 3 * kW2RS);                   \
        DST = __builtin_bit_cast(bf16x8, u_);                                            \
    }
    // all 18 fragments of a 16-row step: 72 x ds_read_b32 in the memory phase (a[tile 0..3][plane], b[plane][tile 0..1])
    bf16x8 fa[4][NP], fb[NP][2];
    // LDS image of this kernel: plane = 4 column blocks of 64 columns, each [8 row pairs][64 + 4 dwords]: the four dwords of
    // a fragment (row pairs 4 kh .. 4 kh + 3 of one column) are 68 dwords apart, so TWO ds_read2_b32 fetch a fragment
    // straight into its four consecutive registers (36 reads per phase).  With the 256-column rows of the lockstep kernel
    // the dwords are 260 apart, out of ds_read2's reach: hipcc paired other dwords and needed 54 v_mov + 28 v_add per
    // phase to re-assemble the operands.  Inline asm (one base register per operand and plane, 8-bit dword offsets).
#define TP_FRAG_ASM(DST, ADDR, OFF)                                                                                       \
    {                                                                                                                     \
        unsigned long long p0_, p1_;                                                                                      \
        asm volatile("ds_read2_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read2_b32 %1, %2 offset0:%5 offset1:%6"            \
                     : "=&v"(p0_), "=&v"(p1_)                                                                             \
                     : "v"(ADDR), "n"((OFF) / 4), "n"((OFF) / 4 + kTPRS / 4), "n"((OFF) / 4 + 2 * (kTPRS / 4)),           \
                       "n"((OFF) / 4 + 3 * (kTPRS / 4)));                                                                 \
        const u32x4 u_ = {(unsigned)p0_, (unsigned)(p0_ >> 32), (unsigned)p1_, (unsigned)(p1_ >> 32)};                    \
        DST = __builtin_bit_cast(bf16x8, u_);                                                                             \
    }
#define TP_READ_FRAGS(BUFP)                                                                                               \
    {                                                                                                                     \
        const unsigned base_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(BUFP);             \
        _Pragma("unroll") for (int pc = 0; pc < NP; ++pc) {                                                               \
            const unsigned bb_ = base_ + (NP + pc) * kTPPlane + wn * kTPBlk + (kh * 4) * kTPRS + li * 4;                   \
            TP_FRAG_ASM(fb[pc][0], bb_, 0)                                                                                \
            TP_FRAG_ASM(fb[pc][1], bb_, 128)                                                                              \
            const unsigned ab0_ = base_ + pc * kTPPlane + (wm * 2) * kTPBlk + (kh * 4) * kTPRS + li * 4;                  \
            const unsigned ab1_ = ab0_ + kTPBlk;                                                                          \
            TP_FRAG_ASM(fa[0][pc], ab0_, 0)                                                                               \
            TP_FRAG_ASM(fa[1][pc], ab0_, 128)                                                                             \
            TP_FRAG_ASM(fa[2][pc], ab1_, 0)                                                                               \
            TP_FRAG_ASM(fa[3][pc], ab1_, 128)                                                                             \
        }                                                                                                                 \
    }
#define TP_TERM(PA, PB)                                                                                                   \
    _Pragma("unroll") for (int t4 = 0; t4 < 4; ++t4) {                                                                    \
        acc[t4][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t4][PA], fb[PB][0], acc[t4][0], 0, 0, 0);                 \
        acc[t4][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t4][PA], fb[PB][1], acc[t4][1], 0, 0, 0);                 \
    }
#define TP_MFMA() if constexpr (NP == 2) { TP_TERM(1, 0) TP_TERM(0, 1) TP_TERM(0, 0) } else { TP_TERM(NP - 1, 0) TP_TERM(0, NP - 1) TP_TERM(1, 1) TP_TERM(1, 0) TP_TERM(0, 1) TP_TERM(0, 0) }
#define TP_BARRIER()                          \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);
    // one phase pair for step s: RB_ holds step s, WB_ receives step s+1 (held by the raw set, requested one phase pair
    // ago), then the raw set is refilled with step s+2 (past the end: the last step again, never used)
#define TP_PHASES(RB_, WB_)                                                       \
    {                                                                             \
        TP_READ_FRAGS(RB_)                                                        \
        store_counts = st + 1 < nsteps;                                           \
        W2_STORE(x, WB_)                                                          \
        {                                                                         \
            const int64_t nx_ = min((int64_t)(st + 2), nsteps - 1) * kW2TM;       \
            W2_LOAD(x, nx_)                                                       \
        }                                                                         \
        TP_BARRIER()                                                              \
        __builtin_amdgcn_s_setprio(1);                                            \
        TP_MFMA()                                                                 \
        __builtin_amdgcn_s_setprio(0);                                            \
        TP_BARRIER()                                                              \
        ++st;                                                                     \
    }

    unsigned char* const buf0 = smemw;
    unsigned char* const buf1 = smemw + 2 * NP * kTPPlane;
    const int64_t rows = m_end - m_begin;                     // multiple of 32: an even number of 16-row steps
    const int64_t nsteps = rows / kW2TM;
    if (rows > 0) {
        // the bias sums must count every row exactly once: steps 0 .. nsteps-1 are stored once each (the store of the
        // clamped "step nsteps" in the last phase is skipped for the sums)
        int64_t st = 0;
        bool store_counts = true;
        W2_LOAD(x, 0)
        W2_STORE(x, buf0)
        W2_LOAD(x, min((int64_t)1, nsteps - 1) * kW2TM)
        TP_BARRIER()
        if (wm == 1) { TP_BARRIER() }                        // group 1 falls one phase behind
#pragma unroll 1
        while (st < nsteps) {
            TP_PHASES(buf0, buf1)
            TP_PHASES(buf1, buf0)
        }
        if (wm == 0) { TP_BARRIER() }                        // pairs with group 1's last barrier
    }
#undef TP_PHASES
#undef TP_BARRIER
#undef TP_MFMA
#undef TP_TERM
#undef TP_READ_FRAGS
#undef TP_FRAG_ASM
#undef W2_LOAD
#undef W2_STORE
#undef W2_FRAG

    float* out = ws + (int64_t)by * N * K;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = k0 + wn * 64 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                out[(int64_t)row * K + col] = acc[mt][nt][r];
            }
        }
    }
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smemw);          // [8][256]
        *reinterpret_cast<float4*>(red + rp * kT2 + c4) = bsum;
        __syncthreads();
        if (tid < kT2) {
            float tot = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) tot += red[g * kT2 + tid];
            ws_bias[(int64_t)by * N + n0 + tid] = tot;
        }
    }
}

// gemm_tn_x6_pp_kernel with a different LDS image (same products in the same order => bit-identical partial sums).
// Counters (tools/pmc_compare.py, profiles/r03_gemm_tn_lds.txt): per wave and 16-row step the kernel above spends 192 LDS-array
// cycles (36 ds_read2_b32 at 4 cycles + 6 ds_write_b128) against 113 in the NT kernel (18 ds_read_b128 + 12 ds_write_b64), and
// 400 instead of 250 cycles stalled on LDS issue -- the row-pair image makes every fragment four separate dwords.  Here a
// thread stages FOUR consecutive rows of four columns of ONE operand (waves 0-3: A, waves 4-7: B; 16-byte row loads as
// before), so the four m-values of a column are 8 contiguous bytes, and the image is [plane][col & 3][m >> 3][(m >> 2) & 1]
// [col >> 2] x 8 bytes (+64 B per col & 3 block): a fragment is two ds_read_b64 (2 LDS cycles each, conflict-free: the 32
// lanes of a group read 4 x 64 contiguous bytes in 4 distinct bank windows), the 12 ds_write_b64 of a thread are 512
// contiguous bytes per wave instruction.
template <int NP>
__global__ __launch_bounds__(kT2Threads, 2) void gemm_tn_x6_pq_kernel(const float* __restrict__ A, int64_t lda,
                                                                      const float* __restrict__ B, int64_t ldb, int64_t M,
                                                                      int N, int K, int tiles_k, int64_t rows_per_split,
                                                                      float* __restrict__ ws, float* __restrict__ ws_bias) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    // (tile, split) of this workgroup.  The output tiles of ONE split read the same rows of A and B (they differ in the
    // column tile of one operand only); in dispatch order (x fastest) they would land on different XCDs, i.e. behind
    // different L2s, and every shared row would be fetched once per tile: 1.6 x the algorithmic HBM bytes.  Remapped so that
    // the tiles of a split are neighbours on one XCD (workgroup L of the linearised grid runs on XCD L % 8).
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.x > 1 && gridDim.y % 8 == 0) {
        const int L = by * (int)gridDim.x + bx;
        const int j = L >> 3;
        bx = j % (int)gridDim.x;
        by = (j / (int)gridDim.x) * 8 + (L & 7);
    }
    const int tn = bx / tiles_k, tk = bx % tiles_k;
    const int n0 = tn * kT2, k0 = tk * kT2;
    const int64_t m_begin = (int64_t)by * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);          // (m_end - m_begin) % 32 == 0 (host)
    const bool want_bias = (ws_bias != nullptr) && tk == 0;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging: thread = (operand, row quad rq of 4, column quad c4 of 64): rows 4rq .. 4rq+3 of the 16-row step
    const int c4 = (tid & 63) * 4, rq = (tid >> 6) & 3, opnd = tid >> 8;          // opnd is uniform per wave group
    const int64_t ld_s = opnd ? ldb : lda;
    const float* s_src = (opnd ? B + k0 : A + n0) + (m_begin + 4 * rq) * ld_s + c4;
    float4 x0, x1, x2, x3;
    constexpr int kPQES = 2048 + 64;                  // bytes per (col & 3) block: [m >> 3][(m >> 2) & 1][64 slots] x 8 B, + 64 B
    constexpr int kPQPlane = 4 * kPQES;               // 8448 B
    const int st_off = opnd * NP * kPQPlane + (rq >> 1) * 1024 + (rq & 1) * 512 + (c4 >> 2) * 8;
#define W2_LOAD(S, MM)                                                                   \
    S##0 = *reinterpret_cast<const float4*>(s_src + (int64_t)(MM) * ld_s);               \
    S##1 = *reinterpret_cast<const float4*>(s_src + (int64_t)((MM) + 1) * ld_s);         \
    S##2 = *reinterpret_cast<const float4*>(s_src + (int64_t)((MM) + 2) * ld_s);         \
    S##3 = *reinterpret_cast<const float4*>(s_src + (int64_t)((MM) + 3) * ld_s);
#define PQ_COL(S, E, EI, BUFP)                                                           \
    {                                                                                    \
        uint2 h_, m_, l_;                                                                \
        if constexpr (NP == 2) {                                                         \
            split2_pair((S##0).E, (S##1).E, h_.x, m_.x);                                     \
            split2_pair((S##2).E, (S##3).E, h_.y, m_.y);                                     \
        } else {                                                                         \
            uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                       \
            split3((S##0).E, h0_, m0_, l0_);                                               \
            split3((S##1).E, h1_, m1_, l1_);                                               \
            h_.x = pack_hi(h0_, h1_); m_.x = pack_hi(m0_, m1_); l_.x = pack_hi(l0_, l1_); \
            split3((S##2).E, h0_, m0_, l0_);                                               \
            split3((S##3).E, h1_, m1_, l1_);                                               \
            h_.y = pack_hi(h0_, h1_); m_.y = pack_hi(m0_, m1_); l_.y = pack_hi(l0_, l1_); \
        }                                                                                \
        unsigned char* d_ = (BUFP) + st_off + (EI) * kPQES;                              \
        *reinterpret_cast<uint2*>(d_ + 0 * kPQPlane) = h_;                               \
        *reinterpret_cast<uint2*>(d_ + 1 * kPQPlane) = m_;                               \
        if constexpr (NP == 3) *reinterpret_cast<uint2*>(d_ + 2 * kPQPlane) = l_;        \
    }
#define W2_STORE(S, BUFP)                                                                \
    {                                                                                    \
        PQ_COL(S, x, 0, BUFP) PQ_COL(S, y, 1, BUFP) PQ_COL(S, z, 2, BUFP) PQ_COL(S, w, 3, BUFP) \
        if (want_bias && store_counts && opnd == 0) {                                    \
            bsum.x += ((S##0).x + (S##1).x) + ((S##2).x + (S##3).x);                             \
            bsum.y += ((S##0).y + (S##1).y) + ((S##2).y + (S##3).y);                             \
            bsum.z += ((S##0).z + (S##1).z) + ((S##2).z + (S##3).z);                             \
            bsum.w += ((S##0).w + (S##1).w) + ((S##2).w + (S##3).w);                             \
        }                                                                                \
    }
#define W2_FRAG(DST, BASE)                                                               \
    {                                                                                    \
        uint4 u_;                                                                        \
        u_.x = *reinterpret_cast<const uint32_t*>(BASE);                                 \
        u_.y = *reinterpret_cast<const uint32_t*>((BASE) + kW2RS);                       \
        u_.z = *reinterpret_cast<const uint32_t*>((BASE) + 2 * kW2RS);                   \
        u_.w = *reinterpret_cast<const uint32_t*>((BASE) + 3 * kW2RS);                   \
        DST = __builtin_bit_cast(bf16x8, u_);                                            \
    }
    // all 18 fragments of a 16-row step: 72 x ds_read_b32 in the memory phase (a[tile 0..3][plane], b[plane][tile 0..1])
    bf16x8 fa[4][NP], fb[NP][2];
    // LDS image of this kernel: plane = 4 column blocks of 64 columns, each [8 row pairs][64 + 4 dwords]: the four dwords of
    // a fragment (row pairs 4 kh .. 4 kh + 3 of one column) are 68 dwords apart, so TWO ds_read2_b32 fetch a fragment
    // straight into its four consecutive registers (36 reads per phase).  With the 256-column rows of the lockstep kernel
    // the dwords are 260 apart, out of ds_read2's reach: hipcc paired other dwords and needed 54 v_mov + 28 v_add per
    // phase to re-assemble the operands.  Inline asm (one base register per operand and plane, 8-bit dword offsets).
#define TP_FRAG_ASM(DST, ADDR, OFF)                                                                                       \
    {                                                                                                                     \
        unsigned long long p0_, p1_;                                                                                      \
        asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"                                      \
                     : "=&v"(p0_), "=&v"(p1_) : "v"(ADDR), "n"(OFF), "n"((OFF) + 512));                                   \
        const u32x4 u_ = {(unsigned)p0_, (unsigned)(p0_ >> 32), (unsigned)p1_, (unsigned)(p1_ >> 32)};                    \
        DST = __builtin_bit_cast(bf16x8, u_);                                                                             \
    }
    // fragment of output row / column n (lane li within a 32-tile): block (n & 3) = li & 3, slot n >> 2, half by the offset
#define TP_READ_FRAGS(BUFP)                                                                                               \
    {                                                                                                                     \
        const unsigned base_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(BUFP) +            \
                               (li & 3) * kPQES + kh * 1024 + (li >> 2) * 8;                                              \
        _Pragma("unroll") for (int pc = 0; pc < NP; ++pc) {                                                               \
            const unsigned bb_ = base_ + (NP + pc) * kPQPlane + wn * 128;                                                 \
            TP_FRAG_ASM(fb[pc][0], bb_, 0)                                                                                \
            TP_FRAG_ASM(fb[pc][1], bb_, 64)                                                                               \
            const unsigned ab_ = base_ + pc * kPQPlane + wm * 256;                                                        \
            TP_FRAG_ASM(fa[0][pc], ab_, 0)                                                                                \
            TP_FRAG_ASM(fa[1][pc], ab_, 64)                                                                               \
            TP_FRAG_ASM(fa[2][pc], ab_, 128)                                                                              \
            TP_FRAG_ASM(fa[3][pc], ab_, 192)                                                                              \
        }                                                                                                                 \
    }
#define TP_TERM(PA, PB)                                                                                                   \
    _Pragma("unroll") for (int t4 = 0; t4 < 4; ++t4) {                                                                    \
        acc[t4][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t4][PA], fb[PB][0], acc[t4][0], 0, 0, 0);                 \
        acc[t4][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t4][PA], fb[PB][1], acc[t4][1], 0, 0, 0);                 \
    }
#define TP_MFMA() if constexpr (NP == 2) { TP_TERM(1, 0) TP_TERM(0, 1) TP_TERM(0, 0) } else { TP_TERM(NP - 1, 0) TP_TERM(0, NP - 1) TP_TERM(1, 1) TP_TERM(1, 0) TP_TERM(0, 1) TP_TERM(0, 0) }
#define TP_BARRIER()                          \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);
    // one phase pair for step s: RB_ holds step s, WB_ receives step s+1 (held by the raw set, requested one phase pair
    // ago), then the raw set is refilled with step s+2 (past the end: the last step again, never used)
#define TP_PHASES(RB_, WB_)                                                       \
    {                                                                             \
        TP_READ_FRAGS(RB_)                                                        \
        store_counts = st + 1 < nsteps;                                           \
        W2_STORE(x, WB_)                                                          \
        {                                                                         \
            const int64_t nx_ = min((int64_t)(st + 2), nsteps - 1) * kW2TM;       \
            W2_LOAD(x, nx_)                                                       \
        }                                                                         \
        TP_BARRIER()                                                              \
        __builtin_amdgcn_s_setprio(1);                                            \
        TP_MFMA()                                                                 \
        __builtin_amdgcn_s_setprio(0);                                            \
        TP_BARRIER()                                                              \
        ++st;                                                                     \
    }

    unsigned char* const buf0 = smemw;
    unsigned char* const buf1 = smemw + 2 * NP * kPQPlane;
    const int64_t rows = m_end - m_begin;                     // multiple of 32: an even number of 16-row steps
    const int64_t nsteps = rows / kW2TM;
    if (rows > 0) {
        // the bias sums must count every row exactly once: steps 0 .. nsteps-1 are stored once each (the store of the
        // clamped "step nsteps" in the last phase is skipped for the sums)
        int64_t st = 0;
        bool store_counts = true;
        W2_LOAD(x, 0)
        W2_STORE(x, buf0)
        W2_LOAD(x, min((int64_t)1, nsteps - 1) * kW2TM)
        TP_BARRIER()
        if (wm == 1) { TP_BARRIER() }                        // group 1 falls one phase behind
#pragma unroll 1
        while (st < nsteps) {
            TP_PHASES(buf0, buf1)
            TP_PHASES(buf1, buf0)
        }
        if (wm == 0) { TP_BARRIER() }                        // pairs with group 1's last barrier
    }
#undef TP_PHASES
#undef TP_BARRIER
#undef TP_MFMA
#undef TP_TERM
#undef TP_READ_FRAGS
#undef TP_FRAG_ASM
#undef W2_LOAD
#undef W2_STORE
#undef PQ_COL
#undef W2_FRAG

    float* out = ws + (int64_t)by * N * K;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = k0 + wn * 64 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                out[(int64_t)row * K + col] = acc[mt][nt][r];
            }
        }
    }
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smemw);          // [4 row quads][256]
        if (opnd == 0) *reinterpret_cast<float4*>(red + rq * kT2 + c4) = bsum;
        __syncthreads();
        if (tid < kT2) {
            float tot = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) tot += red[g * kT2 + tid];
            ws_bias[(int64_t)by * N + n0 + tid] = tot;
        }
    }
}

// GEMM arithmetic mode: 0 = fp32 MFMA (exact fp32), 1 = bf16x6 split on the bf16 MFMA (fp32-class accuracy, 2.67x rate),
// 2 = plain bf16 operands (one MFMA per product, fp32 accumulate; BASELINE configs[4] names bf16): 128-tile kernels only.
// Initialised from VQCPC_GEMM_MODE, changeable through vqcpc_gemm_set_mode().
static std::atomic<int> g_gemm_mode{-1};
// Gradient arithmetic (opt-in): products per fp32 product of the 256-tile bf16x6 kernels while a gradient scope is open
// (vqcpc_gemm_gradient_scope: the trainers open it around loss.backward()).  6 = the forward's exact split (default);
// 3 = two rounded planes per operand, hh + hm + mh (gemm_nt_x6_pp_kernel<.., NP = 2>): ~2^-17 per product.
static std::atomic<int> g_grad_products{6};
static std::atomic<int> g_grad_scope{0};
static int gradient_products_now() {
    return g_grad_scope.load(std::memory_order_relaxed) > 0 ? g_grad_products.load(std::memory_order_relaxed) : 6;
}
// A/B switches of the kernel selection exist in LAB builds only (round 5: the product library keeps no such process-wide value;
// what is left there is the arithmetic mode and the bf16-pair gradient scope, include/vqcpc.h "PROCESS-WIDE STATE")
#if VQCPC_LAB
static std::atomic<int> g_use_pp{1};   // bf16x6 NT 256-tile: ping-pong wave groups
static std::atomic<int> g_use_sw{0};   // bf16x6 NT 256-tile: software-pipelined one-wave-per-SIMD kernel (gemm_sw.hip)
static std::atomic<int> g_use_dma{0};  // bf16x6 NT 256-tile: LDS-DMA operand delivery (gemm_dma.hip) instead of register staging
static std::atomic<int> g_use_t2{1};   // bf16x6 NT: use the 256x256 tile kernel where shapes allow
static inline bool pp_enabled() { return g_use_pp.load(std::memory_order_relaxed) != 0; }
static inline bool t2_enabled() { return g_use_t2.load(std::memory_order_relaxed) != 0; }
#else
static constexpr bool pp_enabled() { return true; }
static constexpr bool t2_enabled() { return true; }
#endif
static int gemm_mode() {
    int m = g_gemm_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("VQCPC_GEMM_MODE");
        // "0" / "f32" -> 0;  "1" / "x6" / "bf16x6" -> 1;  "8" / "2" / "bf16" -> 2
        m = 0;
        if (e) {
            if (!strcmp(e, "1") || !strcmp(e, "x6") || !strcmp(e, "bf16x6")) m = 1;
            else if (!strcmp(e, "8") || !strcmp(e, "2") || !strcmp(e, "bf16")) m = 2;
        }
        g_gemm_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

static int tn_splits(int64_t M, int N, int K) {
    const int64_t tiles = ceil_div(N, BM) * ceil_div(K, BN);
    // short contractions (student step: 3072 rows) are prologue / epilogue-bound: fewer, longer splits (72 vs 76-81 us
    // for the 2048 x 512 weights, tools/bench_tn_small.py)
    int64_t s = ceil_div(M < 8192 ? 512 : 1024, tiles);
    s = std::min<int64_t>(s, ceil_div(M, 8 * TM));   // at least 256 rows per split
    return (int)std::max<int64_t>(s, 1);
}
// 256-tile kernel: one workgroup per CU, so tiles * splits must not exceed the CU count (a 257th workgroup would wait
// for a whole round)
static int tn_splits_256(int64_t M, int N, int K) {
    const int64_t tiles = (int64_t)(N / kT2) * (K / kT2);
    int64_t s = std::max<int64_t>(1, kNumCU / tiles);
    s = std::min<int64_t>(s, std::max<int64_t>(1, M / 256));
    return (int)s;
}
// ... and when M is too short for the split count to make up for few output tiles (student / decoder steps: 3072 /
// 12 288 rows, 512 x 512 or 256 x 512 weights: 24-96 workgroups), the 128-tile kernel's 4x more tiles win:
// 50 -> 30 us, 45 -> 27 us at M = 3072 (tools/bench_tn_small.py)
static bool tn_can_use_256(int64_t M, int N, int K) {
    if ((N % kT2) || (K % kT2) || (M % 32) || M < 8192) return false;
    return (int64_t)(N / kT2) * (K / kT2) * tn_splits_256(M, N, K) >= kNumCU / 2;
}

// =====================================================================================================================
// Skinny NT GEMM for latency-bound small-M products (GRU recurrence: M = batch = 256, the per-event output heads of the
// student step: M = 8).  The 128 / 256 tiles give such shapes a handful of workgroups with long K loops (80-160 us);
// here a workgroup owns ONE 32 x 32 output tile and its NW waves split K (wave w: [w K/NW, (w+1) K/NW)), the partial
// tiles meet in LDS.  v_mfma_f32_32x32x2_f32 on fp32 operands in both GEMM modes (exact products).  The MFMA k index is
// a summation index, so lane half g takes a contiguous K range of its operand row: A / B fragments are float4 loads
// straight from global memory (both operands are K-contiguous in the NT layout), no LDS staging.
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_nt_skinny_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ add, int64_t ldadd) {
    __shared__ float red[NW - 1][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, l31 = lane & 31;
    const int i = blockIdx.y * 32 + l31, j = blockIdx.x * 32 + l31;
    const int len = K / (2 * NW);                               // floats per lane half (multiple of 4)
    const int k0 = (wave * 2 + g) * len;
    const float* ap = A + (int64_t)min(i, M - 1) * lda + k0;
    const float* bp = B + (int64_t)min(j, N - 1) * ldb + k0;
    floatx16 acc = {0};
    int kk = 0;
    for (; kk + 16 <= len; kk += 16) {
        float4 a[4], b[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[v] = *reinterpret_cast<const float4*>(ap + kk + 4 * v);
            b[v] = *reinterpret_cast<const float4*>(bp + kk + 4 * v);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].x, b[v].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].y, b[v].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].z, b[v].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[v].w, b[v].w, acc, 0, 0, 0);
        }
    }
    for (; kk < len; kk += 4) {
        const float4 a = *reinterpret_cast<const float4*>(ap + kk);
        const float4 b = *reinterpret_cast<const float4*>(bp + kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int col = blockIdx.x * 32 + l31;
        const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
#pragma unroll
            for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];
            const int row = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (row < M && col < N) {
                v += bv;
                if (add) v += add[(int64_t)row * ldadd + col];
                C[(int64_t)row * ldc + col] = v;
            }
        }
    }
}

static bool skinny_ok(int64_t M, int N, int K, int flags, int* nw) {
    if (flags & ~(E_BIAS | E_ADD)) return false;
    // narrow outputs (N <= 64: output_linear to the codebook dimension, the upscaler's second layer, d x of the first GRU layer):
    // a 128-wide tile computes 2-4x the columns that exist and leaves one workgroup per 128 rows to walk K alone
    // (34 816 x 32 x 512: 69 us, 2048 x 32 x 1536: 115 us); 32 x 32 tiles with the K range over the waves: 3-6x faster
    const bool narrow = N <= 64 && K >= 128 && M <= (1 << 21);          // gridDim.y = M / 32
    if (!narrow && (M > 1024 || ceil_div(M, BM) * ceil_div(N, BN) >= 128)) return false;     // enough big tiles: use them
    // bf16x6 mode, >= 96 tiles of 64 x 64: the LDS-staged 64-tile kernel reads full lines and beats the fragment-shaped global
    // loads of this kernel from there on (768 x 1536 x 512: 37.7 -> 18.5 us, 768 x 512 x 512: 16.1 -> 14.7; tools/bench_s64.py)
    if (!narrow && gemm_mode() == 1 && M % 64 == 0 && N % 64 == 0 && K % BK == 0 && (M / 64) * (N / 64) >= 96) return false;
    *nw = K >= 1024 ? 8 : 4;
    return K % (8 * *nw) == 0;
}

}  // namespace vq

using namespace vq;

static int gemm_nt_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                          int K, const EpiParams& ep_in, hipStream_t st, bool may_split) {
    EpiParams ep = ep_in;
    const int tiles_n = (int)ceil_div(N, BN);
    const int64_t tiles = ceil_div(M, BM) * tiles_n;
    VQ_REQUIRE(tiles < (1ll << 31), "gemm_nt: too many tiles");
    const float* bias = ep.bias;
    const float* gate = ep.gate;
    const float* add = ep.add;
    const float* add2 = ep.add2;
    const int64_t ldadd = ep.ldadd;
    const bool full = (M % BM == 0) && (N % BN == 0) && (K % BK == 0);
    const int flags = (bias ? E_BIAS : 0) | (ep.act == 1 ? E_RELU : 0) | (ep.thr ? E_DROP : 0) | (gate ? E_GATE : 0) |
                      (add ? E_ADD : 0) | (add2 ? E_ADD2 : 0);
    const dim3 grid((unsigned)tiles), block(kGemmThreads);
    const int mode = gemm_mode();
    int nw = 0;
    if (skinny_ok(M, N, K, flags, &nw)) {
        const dim3 sgrid((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, 32));
        if (nw == 8)
            hipLaunchKernelGGL(gemm_nt_skinny_kernel<8>, sgrid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, (int)M, N, K, bias, add,
                               ldadd);
        else
            hipLaunchKernelGGL(gemm_nt_skinny_kernel<4>, sgrid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, (int)M, N, K, bias, add,
                               ldadd);
        VQ_CHECK_LAUNCH("gemm_nt_skinny");
        return VQCPC_OK;
    }
    // under-filled bf16x6 launches (at most one 128-tile per CU): 64 x 64 tiles, four times the workgroups
    static const int s64_max_tiles = lab_env_int("VQCPC_S64_MAX_TILES", 256);
    if (mode == 1 && tiles <= s64_max_tiles && (M % kS64 == 0) && (N % kS64 == 0) && (K % BK == 0) && M >= kS64) {
        const int tn64 = N / kS64;
        const dim3 g64((unsigned)((M / kS64) * tn64));
#define S64_CASE(EPIV)                                                                                                     \
    case EPIV:                                                                                                             \
        hipLaunchKernelGGL((gemm_nt_x6_s64_kernel<EPIV>), g64, block, 0, st, A, lda, B, ldb, C, ldc, M, N, K, tn64, ep);   \
        VQ_CHECK_LAUNCH("gemm_nt_x6_s64");                                                                                 \
        return VQCPC_OK;
        switch (flags) {
            S64_CASE(0)
            S64_CASE(E_BIAS)
            S64_CASE(E_BIAS | E_RELU)
            S64_CASE(E_BIAS | E_RELU | E_DROP)
            S64_CASE(E_GATE)
            S64_CASE(E_ADD)
            S64_CASE(E_BIAS | E_ADD)
            S64_CASE(E_BIAS | E_DROP | E_ADD)
            default: break;
        }
#undef S64_CASE
    }
    // bf16x6, full 256 x 256 tiles: the high-arithmetic-intensity kernel (one workgroup of 8 waves per CU)
    // the 256-tile kernel runs ONE workgroup per CU: pick it only when its last (partial) round of tiles does not waste
    // more than the ~8 % it gains per tile over the 128-tile kernel (2 workgroups per CU, 4x more tiles)
    bool t2_ok = mode == 1 && t2_enabled() && (M % kT2 == 0) && (N % kT2 == 0) &&
                 (K % (2 * kT2BK) == 0) && (!add2 || (flags == (E_ADD | E_ADD2) && pp_enabled()));
    if (t2_ok && may_split) {
        // cost model in units of one round of 256-tiles (256 workgroups): a round of the 128-tile kernel (512 workgroups)
        // does half the work ~8 % less efficiently.  A partial last round wastes whole CUs, so a GEMM of 2.1 rounds is cut by
        // rows into the full rounds (256-tile kernel) + the remaining rows (128-tile kernel).
        const int64_t mt = M / kT2, tn = N / kT2, t256 = mt * tn;
        const double c256 = ceil((double)t256 / kNumCU);
        const double c128 = ceil((double)tiles / (2 * kNumCU)) * 0.54;
        const int64_t main_mt = (t256 / kNumCU) * kNumCU / tn;          // row tiles that fill whole rounds
        double csplit = 1e30;
        if (main_mt > 0 && main_mt < mt) {
            const int64_t rem_tiles128 = (mt - main_mt) * 2 * ceil_div(N, BN);
            csplit = ceil((double)(main_mt * tn) / kNumCU) + ceil((double)rem_tiles128 / (2 * kNumCU)) * 0.54 + 0.03;
        }
        if (csplit < c256 && csplit < c128) {
            const int64_t m_main = main_mt * kT2;
            int rc = gemm_nt_launch(A, lda, B, ldb, C, ldc, m_main, N, K, ep, st, false);
            if (rc) return rc;
            EpiParams e2 = ep;
            if (e2.gate) e2.gate += m_main * e2.ldgate;
            if (e2.add) e2.add += m_main * e2.ldadd;
            if (e2.add2) e2.add2 += m_main * e2.ldadd2;
            e2.row0 = ep.row0 + m_main;
            return gemm_nt_launch(A + m_main * lda, lda, B, ldb, C + m_main * ldc, ldc, M - m_main, N, K, e2, st, false);
        }
        t2_ok = c256 <= c128;
    } else if (t2_ok) {
        const double r256 = (double)((M / kT2) * (N / kT2)) / kNumCU, r128 = (double)tiles / (2 * kNumCU);
        const double eff256 = r256 / ceil(r256), eff128 = r128 / ceil(r128);
        t2_ok = eff256 * 1.08 >= eff128;
    }
#if VQCPC_LAB      // rejected designs kept for A/B measurements (gemm_sw.hip, gemm_dma.hip): lab builds only
    if (t2_ok && g_use_sw.load(std::memory_order_relaxed) && gemm_nt_sw_ok(M, N, K, flags))
        return gemm_nt_sw_launch(A, lda, B, ldb, C, ldc, M, N, K, flags, ep, st);
    if (t2_ok && g_use_dma.load(std::memory_order_relaxed) && gemm_nt_dma_ok(M, N, K, flags))
        return gemm_nt_dma_launch(A, lda, B, ldb, C, ldc, M, N, K, flags, ep, st);
#endif
    if (t2_ok) {
        const int tn2 = N / kT2;
        const int tiles2 = (int)((M / kT2) * tn2);
        // VQCPC_PP_GRID (measurement only): fewer persistent workgroups than CUs, to tell a per-CU store limit from a chip-wide burst
        static const int grid_cap = lab_env_int("VQCPC_PP_GRID", kNumCU);
        const dim3 grid2((unsigned)std::min(tiles2, grid_cap)), block2(kT2Threads);
        // the gate epilogue (relu / dropout backward: reads an M x N operand, N = 4 K) is HBM-bound; all eight waves storing
        // together (lockstep kernel) keep more bytes in flight than one wave group at a time: 0.92 vs 1.21 ms at C1
        const bool use_pp = pp_enabled() && flags != E_GATE;
        const size_t lds2 = 2 * kT2Buf;
        const size_t lds_pp = 2 * 6 * kT2 * 32;          // ping-pong kernel: unpadded swizzled planes (96 KB)
        const size_t lds_pp2 = 2 * 4 * kT2 * 32;         // its two-plane gradient variant (64 KB)
#if VQCPC_LAB
// ablation variants of the ping-pong kernel (tools/ablate_pp_gemm.py, VQCPC_PP_ABL; bias epilogue only): lab builds only
#define T2_ABL(EPIV)                                                                                                       \
    {                                                                                                                      \
        static const int abl = lab_env_int("VQCPC_PP_ABL", 0);                                                             \
        if (use_pp && abl && (EPIV) == E_BIAS) {                                                                           \
            if (abl == 1) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 1>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 2) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 2>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 3) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 3>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 8) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 8>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 32) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 32>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 16) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 16>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 64) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 64>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 128) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 128>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 192) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 192>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 256) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 256>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 1024) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 1024>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 2048) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 2048>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 2048>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 4096) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 4096>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 4096>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 512) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 512>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            if (abl == 4) { (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_BIAS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp); hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_BIAS, 4>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); } \
            VQ_CHECK_LAUNCH("gemm_nt_x6_pp (ablation)");                                                                   \
            return VQCPC_OK;                                                                                               \
        }                                                                                                                  \
    }
#else
#define T2_ABL(EPIV)
#endif
#define T2_LAUNCH(EPIV)                                                                                                    \
    {                                                                                                                      \
        static bool attr_done = false;                                                                                     \
        if (!attr_done) {                                                                                                  \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_256_kernel<EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds2);                                                                          \
            attr_done = true;                                                                                              \
        }                                                                                                                  \
        static bool attr_pp = false;                                                                                       \
        if (!attr_pp) {                                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                      (int)lds_pp);                                                                        \
            attr_pp = true;                                                                                                \
        }                                                                                                                  \
        if (use_pp && ((EPIV) == 0 || (EPIV) == E_ADD || (EPIV) == (E_ADD | E_ADD2)) && gradient_products_now() == 3) {    \
            /* opt-in gradient arithmetic (input-gradient GEMMs inside a gradient scope): two planes, three products */    \
            constexpr int EG = ((EPIV) == 0 || (EPIV) == E_ADD || (EPIV) == (E_ADD | E_ADD2)) ? (EPIV) : 0;                \
            static bool attr_g = false;                                                                                    \
            if (!attr_g) {                                                                                                 \
                (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<EG, 0, 2>,                                     \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp2);                       \
                attr_g = true;                                                                                             \
            }                                                                                                              \
            hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<EG, 0, 2>), grid2, block2, lds_pp2, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); \
            VQ_CHECK_LAUNCH("gemm_nt_x6_pp (gradient arithmetic)");                                                        \
            return VQCPC_OK;                                                                                               \
        }                                                                                                                  \
        T2_ABL(EPIV)                                                                                                       \
        if (use_pp)                                                                                                        \
            hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<EPIV>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); \
        else                                                                                                               \
            hipLaunchKernelGGL((gemm_nt_x6_256_kernel<EPIV>), grid2, block2, lds2, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); \
        VQ_CHECK_LAUNCH("gemm_nt_x6_256");                                                                                 \
        return VQCPC_OK;                                                                                                   \
    }
        switch (flags) {
            case 0: T2_LAUNCH(0)
            case E_BIAS: T2_LAUNCH(E_BIAS)
            case E_BIAS | E_RELU: T2_LAUNCH(E_BIAS | E_RELU)
            case E_BIAS | E_RELU | E_DROP: T2_LAUNCH(E_BIAS | E_RELU | E_DROP)
            case E_GATE: T2_LAUNCH(E_GATE)
            case E_ADD: T2_LAUNCH(E_ADD)
            case E_ADD | E_ADD2: T2_LAUNCH(E_ADD | E_ADD2)
            // s = x + dropout(a W^T + b): the residual sum that LayerNorm normalises (transformer_custom.py:282-283,288-289),
            // produced by the out-proj / FFN2 epilogue so that the LayerNorm kernels read ONE input stream
            case E_BIAS | E_ADD: T2_LAUNCH(E_BIAS | E_ADD)
            case E_BIAS | E_DROP | E_ADD: T2_LAUNCH(E_BIAS | E_DROP | E_ADD)
            default: break;
        }
#undef T2_LAUNCH
#undef T2_ABL
    }
#define NT_LAUNCH(FULLV, EPIV)                                                                                            \
    if (mode == 1)                                                                                                        \
        hipLaunchKernelGGL((gemm_nt_kernel<FULLV, EPIV, 1>), grid, block, 0, st, A, lda, B, ldb, C, ldc, M, N, K, tiles_n, ep); \
    else if (mode == 2)                                                                                                   \
        hipLaunchKernelGGL((gemm_nt_kernel<FULLV, EPIV, 2>), grid, block, 0, st, A, lda, B, ldb, C, ldc, M, N, K, tiles_n, ep); \
    else                                                                                                                  \
        hipLaunchKernelGGL((gemm_nt_kernel<FULLV, EPIV, 0>), grid, block, 0, st, A, lda, B, ldb, C, ldc, M, N, K, tiles_n, ep)
#define NT_CASE(EPIV)                       \
    case EPIV:                              \
        if (full) { NT_LAUNCH(true, EPIV); }  \
        else { NT_LAUNCH(false, EPIV); }      \
        break;
    switch (flags) {
        NT_CASE(0)
        NT_CASE(E_BIAS)
        NT_CASE(E_BIAS | E_RELU)
        NT_CASE(E_BIAS | E_RELU | E_DROP)
        NT_CASE(E_GATE)
        NT_CASE(E_ADD)
        NT_CASE(E_ADD | E_ADD2)
        NT_CASE(E_BIAS | E_ADD)
        NT_CASE(E_BIAS | E_DROP | E_ADD)
        default:
            if (full) { NT_LAUNCH(true, E_RUNTIME); }
            else { NT_LAUNCH(false, E_RUNTIME); }
            break;
    }
#undef NT_CASE
#undef NT_LAUNCH
    VQ_CHECK_LAUNCH("gemm_nt");
    return VQCPC_OK;
}

extern "C" {

int vqcpc_gemm_set_gradient_products(int products) {
    VQ_REQUIRE(products == 3 || products == 6, "gemm_set_gradient_products: 3 or 6, got %d", products);
    g_grad_products.store(products, std::memory_order_relaxed);
    return VQCPC_OK;
}

int vqcpc_gemm_get_gradient_products(void) { return g_grad_products.load(std::memory_order_relaxed); }

int vqcpc_gemm_gradient_scope(int open) {
    if (open) g_grad_scope.fetch_add(1, std::memory_order_relaxed);
    else if (g_grad_scope.load(std::memory_order_relaxed) > 0) g_grad_scope.fetch_sub(1, std::memory_order_relaxed);
    return VQCPC_OK;
}

int vqcpc_gemm_set_mode(int mode) {
    // bit 0: arithmetic (0 fp32 MFMA, 1 bf16x6); bit 1 set: bf16x6 WITHOUT the 256x256-tile kernels (A/B testing);
    // 8 = plain bf16 operands (one bf16 MFMA per product, fp32 accumulation)
    // +16: 256-tile NT kernel with LDS-DMA operand delivery (gemm_dma.hip) instead of register staging (A/B switch; the
    // register-staged ping-pong kernel is 3-5 % faster: an LDS-DMA instruction costs ~90 issue cycles on its SIMD)
    // +32: software-pipelined one-wave-per-SIMD 256-tile NT kernel (gemm_sw.hip; A/B switch)
#if VQCPC_LAB
    const int use_sw = (mode >= 32 && mode < 48) ? 1 : 0;
    if (use_sw) mode -= 32;
    g_use_sw.store(use_sw, std::memory_order_relaxed);
    const int use_dma = (mode >= 16 && mode < 32) ? 1 : 0;
    if (use_dma) mode -= 16;
    g_use_dma.store(use_dma, std::memory_order_relaxed);
#endif
    VQ_REQUIRE((mode >= 0 && mode <= 7) || mode == 8,
               "gemm_set_mode: mode must be 0 (fp32 MFMA), 1 (bf16x6) [+2: 128-tile only, +4: no ping-pong; lab builds: +16 LDS-DMA "
               "kernel, +32 one-wave-per-SIMD kernel] or 8 (bf16)");
    if (mode == 8) {
        g_gemm_mode.store(2, std::memory_order_relaxed);
        return VQCPC_OK;
    }
#if VQCPC_LAB
    g_use_t2.store((mode & 2) ? 0 : 1, std::memory_order_relaxed);
    g_use_pp.store((mode & 4) ? 0 : 1, std::memory_order_relaxed);
#else
    VQ_REQUIRE((mode & 6) == 0, "gemm_set_mode: the kernel-selection switches (+2, +4, +16, +32) exist in lab builds only");
#endif
    mode &= 1;
    g_gemm_mode.store(mode, std::memory_order_relaxed);
    return VQCPC_OK;
}

int vqcpc_gemm_get_mode(void) { return gemm_mode(); }

int vqcpc_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                  const float* bias, int act, float drop_p, uint64_t seed, const float* gate, int64_t ldgate,
                  float gate_scale, const float* add, int64_t ldadd, const float* add2, int64_t ldadd2, void* stream) {
    if (M == 0) return VQCPC_OK;
    VQ_REQUIRE(A && B && C, "gemm_nt: null pointer");
    VQ_REQUIRE(M >= 0 && N >= 1 && K >= 4 && K % 4 == 0, "gemm_nt: bad shape M=%lld N=%d K=%d (K %% 4 == 0 required)",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N, "gemm_nt: bad leading dimensions");
    VQ_REQUIRE(aligned16(A) && aligned16(B), "gemm_nt: A and B must be 16-byte aligned");
    VQ_REQUIRE(act == 0 || act == 1, "gemm_nt: act must be 0 (none) or 1 (relu)");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm_nt: bad dropout probability");
    VQ_REQUIRE(ldc < (1 << 22) && ldgate < (1 << 22) && ldadd < (1 << 22), "gemm_nt: leading dimension too large");
    VQ_REQUIRE((!gate || ldgate >= N) && (!add || ldadd >= N) && (!add2 || (add && ldadd2 >= N)),
               "gemm_nt: bad gate/add strides (add2 needs add)");
    EpiParams ep{bias, act, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, gate, ldgate, gate_scale, add, ldadd, add2, ldadd2, 0};
    return gemm_nt_launch(A, lda, B, ldb, C, ldc, M, N, K, ep, (hipStream_t)stream, true);
}

// ---- split-K NT GEMM for launches that cannot fill the chip ----------------------------------------------------------
// The student step (configs[3]: 8 sequences x 384 tokens = 3072 rows, 768 at the event level) projects back to d_model =
// 512 columns with K = 1536 / 2048: 96 (24) tiles of 128 x 128 for 256 CUs, each walking 48-64 K tiles alone on its CU.
// Here the K range is cut over blockIdx.y into `s` partial planes (tiles x s ~ 400-500 workgroups, two per CU) and one
// float4-per-lane pass sums the planes in a fixed order and applies bias / residual: deterministic, fp32-class (the
// partial sums are rounded once more than a single accumulation chain would be).
namespace vq {

static int splitk_choose(int64_t M, int N, int K) {
    static const int min_k = lab_env_int("VQCPC_SPLITK_MIN_K", 768);
    if (gemm_mode() != 1 || M % BM || N % BN || K % BK || K < min_k) return 0;
    const int64_t tiles = (M / BM) * (N / BN);
    if (tiles > 160) return 0;
    const int kt = K / BK;
    int best = 0;
    // at most 16 tiles (the GRU recurrence: 256 x 512 x 1536, one launch per time step on the critical path): planes of
    // 64 k keep the lone workgroup of a CU to a handful of K tiles (21 -> 15.5 us)
    static const int min_ks = lab_env_int("VQCPC_SPLITK_MIN_KS", 64);
    for (int s = 2; s <= 16; ++s)
        if (kt % s == 0 && K / s >= (tiles <= 16 ? min_ks : 256) && tiles * s <= 512) best = s;
    return best;
}

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ ws, int64_t plane, int nsplit,
                                                              float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                              const float* __restrict__ bias, const float* __restrict__ add,
                                                              int64_t ldadd) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N / 4;
    if (q >= M * n4) return;
    const int64_t row = q / n4;
    const int c = (int)(q % n4) * 4;
    const float4* p = reinterpret_cast<const float4*>(ws + row * N + c);
    const int64_t st4 = plane / 4;
#define SK_ADD(X, Y) make_float4(X.x + Y.x, X.y + Y.y, X.z + Y.z, X.w + Y.w)
    float4 acc = p[0];
    int s = 1;
    for (; s + 2 < nsplit; s += 3) {
        const float4 v0 = p[(s + 0) * st4], v1 = p[(s + 1) * st4], v2 = p[(s + 2) * st4];
        const float4 a = SK_ADD(v0, v1);
        const float4 b = SK_ADD(a, v2);
        acc = SK_ADD(acc, b);
    }
    for (; s < nsplit; ++s) {
        const float4 v = p[s * st4];
        acc = SK_ADD(acc, v);
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + c);
        acc = SK_ADD(acc, b);
    }
    if (add) {
        const float4 r = *reinterpret_cast<const float4*>(add + row * ldadd + c);
        acc = SK_ADD(acc, r);
    }
#undef SK_ADD
    *reinterpret_cast<float4*>(C + row * ldc + c) = acc;
}

}  // namespace vq

// Rows of an (M, N, K) product that vqcpc_gemm_nt gives to whole rounds of the 256-tile kernel when it cuts the launch by
// rows (M when it does not cut): a caller with a workspace can run the remaining rows through vqcpc_gemm_nt_splitk instead
// of the single under-filled 128-tile launch (bias / residual epilogues, K >= 1024).
int64_t vqcpc_gemm_nt_main_rows(int64_t M, int N, int K) {
    if (gemm_mode() != 1 || !t2_enabled() || M % kT2 || N % kT2 || K % (2 * kT2BK)) return M;
    const int64_t mt = M / kT2, tn = N / kT2, t256 = mt * tn;
    const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
    const double c256 = ceil((double)t256 / kNumCU);
    const double c128 = ceil((double)tiles / (2 * kNumCU)) * 0.54;
    const int64_t main_mt = (t256 / kNumCU) * kNumCU / tn;
    if (main_mt <= 0 || main_mt >= mt) return M;
    const int64_t rem_tiles128 = (mt - main_mt) * 2 * ceil_div(N, BN);
    const double csplit = ceil((double)(main_mt * tn) / kNumCU) + ceil((double)rem_tiles128 / (2 * kNumCU)) * 0.54 + 0.03;
    return (csplit < c256 && csplit < c128) ? main_mt * kT2 : M;
}

int64_t vqcpc_gemm_nt_splitk_workspace(int64_t M, int N, int K) {
    const int s = splitk_choose(M, N, K);
    return s ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

int vqcpc_gemm_nt_splitk(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                         int K, const float* bias, const float* add, int64_t ldadd, void* workspace, int64_t workspace_bytes,
                         void* stream) {
    VQ_REQUIRE(A && B && C && workspace, "gemm_nt_splitk: null pointer");
    const int s = splitk_choose(M, N, K);
    VQ_REQUIRE(s > 0, "gemm_nt_splitk: shape M=%lld N=%d K=%d is not a split-K shape in this GEMM mode (query "
               "vqcpc_gemm_nt_splitk_workspace first)", (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && ldc % 4 == 0 && (!add || (ldadd >= N && ldadd % 4 == 0)),
               "gemm_nt_splitk: bad leading dimensions");
    VQ_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && aligned16(workspace) && (!add || aligned16(add)) &&
               (!bias || aligned16(bias)), "gemm_nt_splitk: operands must be 16-byte aligned");
    if (workspace_bytes < (int64_t)s * M * N * (int64_t)sizeof(float)) {
        set_error("gemm_nt_splitk: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    EpiParams ep{};
    ep.inv_keep = 1.0f;
    ep.split_plane = M * N;
    const int tiles_n = N / BN;
    hipLaunchKernelGGL((gemm_nt_kernel<true, 0, 1>), dim3((unsigned)((M / BM) * tiles_n), (unsigned)s), dim3(kGemmThreads), 0,
                       st, A, lda, B, ldb, (float*)workspace, (int64_t)N, M, N, K / s, tiles_n, ep);
    VQ_CHECK_LAUNCH("gemm_nt_splitk");
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)ceil_div(M * (N / 4), 256)), dim3(256), 0, st,
                       (const float*)workspace, M * N, s, C, ldc, M, N, bias, add, ldadd);
    VQ_CHECK_LAUNCH("splitk_epilogue");
    return VQCPC_OK;
}

int64_t vqcpc_gemm_tn_workspace(int64_t M, int N, int K) {
    int s = tn_splits(std::max<int64_t>(M, 1), N, K);
    if (tn_can_use_256(M, N, K)) s = std::max(s, tn_splits_256(M, N, K));
    return (int64_t)s * ((int64_t)N * K + N) * (int64_t)sizeof(float);
}

int vqcpc_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* dW, float* db, int64_t M, int N, int K,
                  int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(A && B && dW && workspace, "gemm_tn: null pointer");
    VQ_REQUIRE(M >= 1 && N >= 4 && K >= 4 && N % 4 == 0 && K % 4 == 0, "gemm_tn: bad shape M=%lld N=%d K=%d", (long long)M,
               N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= N && ldb >= K, "gemm_tn: bad leading dimensions");
    VQ_REQUIRE(aligned16(A) && aligned16(B), "gemm_tn: A and B must be 16-byte aligned");
    if (workspace_bytes < vqcpc_gemm_tn_workspace(M, N, K)) {
        set_error("gemm_tn: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const bool use256 = gemm_mode() == 1 && t2_enabled() && tn_can_use_256(M, N, K);
    const int splits = use256 ? tn_splits_256(M, N, K) : tn_splits(M, N, K);
    const int64_t rows_per_split = round_up(ceil_div(M, splits), TM);
    const int tiles_k = (int)ceil_div(K, BN);
    const int tiles = (int)ceil_div(N, BM) * tiles_k;
    float* ws = (float*)workspace;
    float* ws_bias = db ? ws + (int64_t)splits * N * K : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const bool full_tn = (N % BM == 0) && (K % BN == 0) && (M % TM == 0);
    if (use256) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_x6_256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      2 * kW2Buf);
            attr_done = true;
        }
        const int tk2 = K / kT2;
        if (pp_enabled()) {
            static bool attr_pp = false;
            constexpr int kPQBuf3 = 2 * 6 * 4 * (2048 + 64), kPQBuf2 = 2 * 4 * 4 * (2048 + 64);       // gemm_tn_x6_pq_kernel images
            if (!attr_pp) {
                (void)hipFuncSetAttribute((const void*)gemm_tn_x6_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          2 * kTPBuf);
                (void)hipFuncSetAttribute((const void*)gemm_tn_x6_pp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          2 * 4 * kTPPlane);
                (void)hipFuncSetAttribute((const void*)gemm_tn_x6_pq_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kPQBuf3);
                (void)hipFuncSetAttribute((const void*)gemm_tn_x6_pq_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kPQBuf2);
                attr_pp = true;
            }
#if VQCPC_LAB
            // A/B switch of lab builds: VQCPC_TN_PQ=0 keeps the row-pair LDS image of gemm_tn_x6_pp_kernel (read per launch:
            // the lab test flips it inside one process)
            const char* pq_env = getenv("VQCPC_TN_PQ");
            const bool use_pq = !(pq_env && pq_env[0] == '0');
#else
            constexpr bool use_pq = true;                       // the quad-row LDS image (gemm_tn_x6_pq_kernel)
#endif
            const dim3 g_((N / kT2) * tk2, splits), b_(kT2Threads);
            if (gradient_products_now() == 3) {        // opt-in gradient arithmetic, inside a gradient scope only
                if (use_pq)
                    hipLaunchKernelGGL(gemm_tn_x6_pq_kernel<2>, g_, b_, kPQBuf2, s, A, lda, B, ldb, M, N, K, tk2, rows_per_split,
                                       ws, ws_bias);
                else
                    hipLaunchKernelGGL(gemm_tn_x6_pp_kernel<2>, g_, b_, 2 * 4 * kTPPlane, s, A, lda, B, ldb, M, N, K, tk2,
                                       rows_per_split, ws, ws_bias);
            } else if (use_pq)
                hipLaunchKernelGGL(gemm_tn_x6_pq_kernel<3>, g_, b_, kPQBuf3, s, A, lda, B, ldb, M, N, K, tk2, rows_per_split, ws,
                                   ws_bias);
            else
                hipLaunchKernelGGL(gemm_tn_x6_pp_kernel<3>, g_, b_, 2 * kTPBuf, s, A, lda, B, ldb, M, N, K, tk2, rows_per_split,
                                   ws, ws_bias);
        } else
        hipLaunchKernelGGL(gemm_tn_x6_256_kernel, dim3((N / kT2) * tk2, splits), dim3(kT2Threads), 2 * kW2Buf, s, A, lda, B,
                           ldb, M, N, K, tk2, rows_per_split, ws, ws_bias);
    } else if (gemm_mode() == 1) {
        if (full_tn)
            hipLaunchKernelGGL((gemm_tn_x6_kernel<true, 1>), dim3(tiles, splits), dim3(kGemmThreads), 0, s, A, lda, B, ldb, M, N,
                               K, tiles_k, rows_per_split, ws, ws_bias);
        else
            hipLaunchKernelGGL((gemm_tn_x6_kernel<false, 1>), dim3(tiles, splits), dim3(kGemmThreads), 0, s, A, lda, B, ldb, M,
                               N, K, tiles_k, rows_per_split, ws, ws_bias);
    } else if (gemm_mode() == 2) {
        if (full_tn)
            hipLaunchKernelGGL((gemm_tn_x6_kernel<true, 2>), dim3(tiles, splits), dim3(kGemmThreads), 0, s, A, lda, B, ldb, M, N,
                               K, tiles_k, rows_per_split, ws, ws_bias);
        else
            hipLaunchKernelGGL((gemm_tn_x6_kernel<false, 2>), dim3(tiles, splits), dim3(kGemmThreads), 0, s, A, lda, B, ldb, M,
                               N, K, tiles_k, rows_per_split, ws, ws_bias);
    } else if (full_tn)
        hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(tiles, splits), dim3(kGemmThreads), 0, s, A, lda, B, ldb, M, N, K,
                           tiles_k, rows_per_split, ws, ws_bias);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(tiles, splits), dim3(kGemmThreads), 0, s, A, lda, B, ldb, M, N, K,
                           tiles_k, rows_per_split, ws, ws_bias);
    VQ_CHECK_LAUNCH("gemm_tn");
    if (accumulate == 2) return VQCPC_OK;       // partial sums stay in `workspace` (vqcpc_gemm_tn_deferred_splits): the caller reduces
    return launch_reduce_splits2(ws, (int64_t)N * K, splits, dW, (int64_t)N * K, ws_bias, N, db, db ? N : 0, accumulate, s);
}

// Deferred reduction (accumulate == 2): vqcpc_gemm_tn leaves its `splits` partial sums in the workspace -- dW partials at
// float offset s * N * K, db partials at splits * N * K + s * N -- for ONE vqcpc_reduce_grouped_vec launch over many products
// at the end of a backward pass.  Returns that split count, or 0 when the immediate reduction would not take the float4 kernel
// (small products): only then is the deferred result bit-identical to the immediate one.
int vqcpc_gemm_tn_deferred_splits(int64_t M, int N, int K) {
    if (M < 1 || N < 4 || K < 4 || (N % 4) || (K % 4)) return 0;
    const bool use256 = gemm_mode() == 1 && t2_enabled() && tn_can_use_256(M, N, K);
    const int splits = use256 ? tn_splits_256(M, N, K) : tn_splits(M, N, K);
    return ((int64_t)N * K >= (1 << 16) && splits <= 64) ? splits : 0;
}


// ---- grouped weight gradients (see gemm_tn_x6_grouped_kernel) ---------------------------------------------------------
// A product is "groupable" when the single launch would take the 128-tile bf16x6 kernel (few rows: student / decoder steps);
// the large products of the CPC step fill the chip on their own and keep their launches.
int vqcpc_gemm_tn_groupable(int64_t M, int N, int K) {
    const int mode = gemm_mode();
    if (mode != 1 && mode != 2) return 0;
    if (M < 1 || N < 4 || K < 4 || (N % 4) || (K % 4)) return 0;
    if (mode == 1 && t2_enabled() && tn_can_use_256(M, N, K)) return 0;
    // deferred problems keep both operands alive until the gradient scope closes and run on ~64 workgroups each: only the
    // genuinely small ones (student / decoder steps, narrow projections) are worth grouping -- at most 64 MB of operands
    return (M <= (1 << 20) && M * ((int64_t)N + K) * 4 <= (64ll << 20)) ? 1 : 0;
}

// split count of a problem inside a group: by its own shape only (results do not depend on what else is in the group);
// ~64 workgroups per problem, at least 256 rows per split
static int tn_group_splits(int64_t M, int N, int K) {
    const int64_t tiles = ceil_div(N, BM) * ceil_div(K, BN);
    int64_t s = ceil_div(64, tiles);
    s = std::min<int64_t>(s, std::max<int64_t>(1, M / 256));
    return (int)std::max<int64_t>(s, 1);
}

int64_t vqcpc_gemm_tn_grouped_workspace(int n, const int64_t* M, const int* N, const int* K) {
    int64_t floats = 0;
    for (int i = 0; i < n; ++i)
        floats += (int64_t)tn_group_splits(std::max<int64_t>(M[i], 1), N[i], K[i]) * ((int64_t)N[i] * K[i] + N[i]);
    return floats * (int64_t)sizeof(float);
}

int vqcpc_gemm_tn_grouped(int n, const void* const* A, const int64_t* lda, const void* const* B, const int64_t* ldb,
                          void* const* dW, void* const* db, const int64_t* M, const int* N, const int* K, int accumulate,
                          void* workspace, int64_t workspace_bytes, void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(n > 0 && A && lda && B && ldb && dW && db && M && N && K && workspace, "gemm_tn_grouped: null pointer");
    if (workspace_bytes < vqcpc_gemm_tn_grouped_workspace(n, M, N, K)) {
        set_error("gemm_tn_grouped: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const int mode = gemm_mode();
    for (int i = 0; i < n; ++i) {
        VQ_REQUIRE(A[i] && B[i] && dW[i], "gemm_tn_grouped: null operand in problem %d", i);
        VQ_REQUIRE(vqcpc_gemm_tn_groupable(M[i], N[i], K[i]), "gemm_tn_grouped: problem %d (M=%lld N=%d K=%d) is not groupable in "
                   "this GEMM mode (query vqcpc_gemm_tn_groupable first)", i, (long long)M[i], N[i], K[i]);
        VQ_REQUIRE(lda[i] % 4 == 0 && ldb[i] % 4 == 0 && lda[i] >= N[i] && ldb[i] >= K[i] && lda[i] < (1ll << 31) &&
                       ldb[i] < (1ll << 31),
                   "gemm_tn_grouped: bad leading dimensions in problem %d", i);
        VQ_REQUIRE(aligned16(A[i]) && aligned16(B[i]) && aligned16(dW[i]) && (!db[i] || aligned16(db[i])),
                   "gemm_tn_grouped: operands of problem %d must be 16-byte aligned", i);
    }
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    // full-tile problems first (they share the kernel without bounds checks), the ragged ones after them; the order inside
    // each class is the caller's
    std::vector<int> order;
    order.reserve(n);
    auto is_full = [&](int q) { return (N[q] % BM == 0) && (K[q] % BN == 0) && (M[q] % TM == 0); };
    for (int q = 0; q < n; ++q) if (is_full(q)) order.push_back(q);
    const int n_full = (int)order.size();
    for (int q = 0; q < n; ++q) if (!is_full(q)) order.push_back(q);
    int pos = 0;
    while (pos < n) {
        // a chunk: up to kTnGroup problems of one class with pairwise distinct gradient buffers (a repeated buffer waits for the
        // next launch pair, which is stream-ordered behind this one: accumulation order = issue order)
        TnGroupArgs g;
        RedGroupArgs r;
        int c = 0, wg = 0, rb = 0;
        bool full = true;
        const int class_end = pos < n_full ? n_full : n;
        while (pos < class_end && c < kTnGroup) {
            const int i = order[pos];
            bool dup = false;
            for (int j = 0; j < c; ++j) dup = dup || r.out[j] == (float*)dW[i] || (db[i] && r.out2[j] == (float*)db[i]);
            if (dup) break;
            const int splits = tn_group_splits(M[i], N[i], K[i]);
            const int tiles_k = (int)ceil_div(K[i], BN);
            const int tiles = (int)ceil_div(N[i], BM) * tiles_k;
            g.A[c] = (const float*)A[i];
            g.B[c] = (const float*)B[i];
            g.ws[c] = ws;
            g.wsb[c] = db[i] ? ws + (int64_t)splits * N[i] * K[i] : nullptr;
            g.lda[c] = (int)lda[i];
            g.ldb[c] = (int)ldb[i];
            g.M[c] = (int)M[i];
            g.N[c] = N[i];
            g.K[c] = K[i];
            g.tiles_k[c] = tiles_k;
            g.tiles[c] = tiles;
            g.rows_per_split[c] = (int)round_up(ceil_div(M[i], splits), TM);
            g.wg_begin[c] = wg;
            wg += tiles * splits;
            full = full && (N[i] % BM == 0) && (K[i] % BN == 0) && (M[i] % TM == 0);
            r.ws[c] = g.ws[c];
            r.out[c] = (float*)dW[i];
            r.ws2[c] = g.wsb[c];
            r.out2[c] = (float*)db[i];
            r.count[c] = N[i] * K[i];
            r.count2[c] = db[i] ? N[i] : 0;
            r.nsplit[c] = splits;
            r.blk_begin[c] = rb;
            rb += (int)(ceil_div((int64_t)N[i] * K[i] / 4, 256) + (db[i] ? ceil_div(N[i] / 4, 256) : 0));
            ws += (int64_t)splits * ((int64_t)N[i] * K[i] + N[i]);
            ++c;
            ++pos;
        }
        for (int j = c; j <= kTnGroup; ++j) g.wg_begin[j] = wg, r.blk_begin[j] = rb;
        for (int j = c; j < kTnGroup; ++j) {          // unused slots: never selected (their prefix equals the total)
            g.A[j] = g.B[j] = nullptr; g.ws[j] = g.wsb[j] = nullptr;
            g.lda[j] = g.ldb[j] = g.M[j] = g.N[j] = g.K[j] = g.tiles_k[j] = g.tiles[j] = g.rows_per_split[j] = 1;
            r.ws[j] = r.ws2[j] = nullptr; r.out[j] = r.out2[j] = nullptr;
            r.count[j] = r.count2[j] = r.nsplit[j] = 0;
        }
        g.n = c;
        r.n = c;
        r.accumulate = accumulate;
        const dim3 grid((unsigned)wg), block(kGemmThreads);
        if (mode == 1) {
            if (full) hipLaunchKernelGGL((gemm_tn_x6_grouped_kernel<true, 1>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_tn_x6_grouped_kernel<false, 1>), grid, block, 0, s, g);
        } else {
            if (full) hipLaunchKernelGGL((gemm_tn_x6_grouped_kernel<true, 2>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_tn_x6_grouped_kernel<false, 2>), grid, block, 0, s, g);
        }
        VQ_CHECK_LAUNCH("gemm_tn_grouped");
        hipLaunchKernelGGL(reduce_splits_grouped_kernel, dim3((unsigned)rb), dim3(256), 0, s, r);
        VQ_CHECK_LAUNCH("reduce_splits_grouped");
    }
    return VQCPC_OK;
}


// ---- relu / dropout gate as a bit mask (gemm_common.h: EpiParams::mask) ---------------------------------------------
// The forward of F.relu + dropout after linear1 (transformer_custom.py:285) writes, next to its fp32 output, one bit per
// element "output > 0"; the backward's epilogue d_h = (d_y . W2) * [h > 0] / (1 - p) reads the bits instead of the
// M x 4d fp32 activation (2.3 GB per layer at C1).  256-tile ping-pong kernel only: M, N multiples of 256, K of 32, bf16x6.
int vqcpc_gemm_gatebits_supported(int64_t M, int N, int K) {
    return (gemm_mode() == 1 && t2_enabled() && pp_enabled() &&
            M >= kT2 && M % kT2 == 0 && N % kT2 == 0 && K % 32 == 0 && K >= 32)
               ? 1 : 0;
}

int64_t vqcpc_gemm_gatebits_bytes(int64_t M, int N) { return M * (int64_t)(N / 32) * 4; }

static int gatebits_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                           int K, const EpiParams& ep, int flags, hipStream_t st) {
    const int tn2 = N / kT2;
    const int tiles2 = (int)((M / kT2) * tn2);
    const dim3 grid2((unsigned)std::min(tiles2, kNumCU)), block2(kT2Threads);
    const size_t lds_pp = 2 * 6 * kT2 * 32;
    if (flags == E_GATEBITS && gradient_products_now() == 3) {      // opt-in gradient arithmetic (see gemm_nt_launch)
        static bool attr_g = false;
        if (!attr_g) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<E_GATEBITS, 0, 2>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * kT2 * 32);
            attr_g = true;
        }
        hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<E_GATEBITS, 0, 2>), grid2, block2, 2 * 4 * kT2 * 32, st, A, lda, B, ldb, C, ldc, M,
                           N, K, tn2, tiles2, ep);
        VQ_CHECK_LAUNCH("gemm_nt_x6_pp (mask, gradient arithmetic)");
        return VQCPC_OK;
    }
#define GB_LAUNCH(EPIV)                                                                                                   \
    {                                                                                                                      \
        static bool attr_pp = false;                                                                                       \
        if (!attr_pp) {                                                                                                    \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_pp_kernel<EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                      (int)lds_pp);                                                                        \
            attr_pp = true;                                                                                                \
        }                                                                                                                  \
        hipLaunchKernelGGL((gemm_nt_x6_pp_kernel<EPIV>), grid2, block2, lds_pp, st, A, lda, B, ldb, C, ldc, M, N, K, tn2, tiles2, ep); \
        VQ_CHECK_LAUNCH("gemm_nt_x6_pp (mask)");                                                                           \
        return VQCPC_OK;                                                                                                   \
    }
    switch (flags) {
        case E_BIAS | E_RELU | E_MASKOUT: GB_LAUNCH(E_BIAS | E_RELU | E_MASKOUT)
        case E_BIAS | E_RELU | E_DROP | E_MASKOUT: GB_LAUNCH(E_BIAS | E_RELU | E_DROP | E_MASKOUT)
        case E_GATEBITS: GB_LAUNCH(E_GATEBITS)
        default: break;
    }
#undef GB_LAUNCH
    set_error("gemm gate-bits: epilogue combination %d not instantiated", flags);
    return VQCPC_EINVAL;
}

int vqcpc_gemm_nt_relu_mask(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                            int K, const float* bias, float drop_p, uint64_t seed, void* mask, void* stream) {
    VQ_REQUIRE(A && B && C && bias && mask, "gemm_nt_relu_mask: null pointer");
    VQ_REQUIRE(vqcpc_gemm_gatebits_supported(M, N, K), "gemm_nt_relu_mask: needs the bf16x6 mode and M, N multiples of 256, K of 32");
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && aligned16(A) && aligned16(B) && aligned16(mask),
               "gemm_nt_relu_mask: bad leading dimensions / alignment");
    VQ_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "gemm_nt_relu_mask: drop_p out of range");
    EpiParams ep{};
    ep.bias = bias;
    ep.act = 1;
    ep.thr = drop_threshold(drop_p);
    ep.inv_keep = 1.0f / (1.0f - drop_p);
    ep.seed = seed;
    ep.mask = (uint32_t*)mask;
    return gatebits_launch(A, lda, B, ldb, C, ldc, M, N, K, ep, E_BIAS | E_RELU | (ep.thr ? E_DROP : 0) | E_MASKOUT,
                           (hipStream_t)stream);
}

int vqcpc_gemm_nt_gatebits(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                           int K, const void* mask, float gate_scale, void* stream) {
    VQ_REQUIRE(A && B && C && mask, "gemm_nt_gatebits: null pointer");
    VQ_REQUIRE(vqcpc_gemm_gatebits_supported(M, N, K), "gemm_nt_gatebits: needs the bf16x6 mode and M, N multiples of 256, K of 32");
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && aligned16(A) && aligned16(B) && aligned16(mask),
               "gemm_nt_gatebits: bad leading dimensions / alignment");
    EpiParams ep{};
    ep.gate_scale = gate_scale;
    ep.mask = (uint32_t*)const_cast<void*>(mask);
    return gatebits_launch(A, lda, B, ldb, C, ldc, M, N, K, ep, E_GATEBITS, (hipStream_t)stream);
}

}  // extern "C"
