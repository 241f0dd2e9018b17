// bf16x6 NT GEMM, software-pipelined, ONE wavefront per SIMD (512-register kernel).
//
// C[M,N] = epi(A[M,K] . B[N,K]^T), fp32 in / fp32 out, every product = six bf16 MFMAs on the exact 3-way split of its
// operands (gemm_common.h).  256 x 256 x 16 tile like gemm_nt_x6_pp_kernel, but 4 waves (2 x 2, wave tile 128 x 128: the 16
// accumulator tiles are the 256 AGPRs) instead of 8, so that a wave owns its SIMD: there is no second wave to overlap
// with, and no memory phase -- every load, split, LDS write and fragment read of the pipeline is placed in the shadow of
// the wave's own MFMAs (a 32x32x16 bf16 MFMA occupies the pipe for 32 cycles = 8 issue slots, of which <= 5 can be used
// by other instructions without slowing it: MI355X_MICROARCH.md).  Per K tile a wave issues 96 MFMAs and ~250 other
// instructions (8 global loads, 8 x 22 split VALU, 24 ds_write_b64, 24 ds_read_b128, one barrier): 2.6 per MFMA.
//
// Stream position s (K tile s of the persistent tile stream), LDS buffers cur = s % 2 (planes of tile s, complete) and
// nxt; the six MFMA groups of a K tile (16 MFMAs each, smallest terms first as in the other kernels) carry:
//   G1 (A.l x B.h)  read A.h(s);                     split + store rows 0,1 of A(s+1)
//   G2 (A.h x B.l)  request rows 0,1 of A(s+2);      split + store rows 2,3 of A(s+1)
//   G3 (A.m x B.m)  request rows 2,3 of A(s+2);      split + store rows 0,1 of B(s+1)
//   G4 (A.m x B.h)  request rows 0,1 of B(s+2);      split + store rows 2,3 of B(s+1);  then the ONE barrier of the K tile
//   G5 (A.h x B.m)  request rows 2,3 of B(s+2);      read A.l, B.l, B.h' (s+1)
//   G6 (A.h x B.h)                                   read A.m, B.m (s+1)
// ("rows i" = the thread's i-th staged row, 64 rows apart).  Fragment registers are re-used as their plane's last term
// retires: A.l after G1, B.l after G2, A.m after G4, B.m after G5; B.h is live in G1, G4 and G6, so the next tile's B.h
// goes to a second register set (the two alternate); A.h is read at the start of its own K tile (G1) for G2.
//   LDS hazards: tile s+1 is written (G1-G4 of iteration s) into the buffer that held tile s-1, whose last reads (A.h in
//   G1 of iteration s-1, everything else earlier) precede the barrier of iteration s-1; tile s+1 is read from G5 of
//   iteration s on, behind the barrier that closes its writes.
#include "gemm_common.h"

namespace vq {

constexpr int kS = 256;
constexpr int kSBK = 16;
constexpr int kSThreads = 256;
constexpr int kSPlane = kS * 32;       // 8 KB: 256 rows x 16 bf16
constexpr int kSBuf = 6 * kSPlane;     // 48 KB

#define SG_MFMA 0x8
#define SG_VALU 0x2
#define SG_VMEM_RD 0x20
#define SG_DS_RD 0x100
#define SG_DS_WR 0x200

template <int EPI>
__global__ __launch_bounds__(kSThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_x6_sw_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    int64_t M, int N, int K, int tiles_n, int tiles, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int T = K / kSBK;
    const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    floatx16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- load cursor: thread stages rows ld_row + 64 i (i = 0..3), 4 k's at ld_c4, of both operands ----
    const int ld_row = tid >> 2, ld_c4 = (tid & 3) * 4;
    int ld_tile = blockIdx.x, ld_k = 0;
    const float* a_src;
    const float* b_src;
#define SW_SET_SRC()                                                        \
    {                                                                       \
        const int t_ = xcd_swizzle(min(ld_tile, tiles - 1), tiles);         \
        a_src = A + ((int64_t)(t_ / tiles_n) * kS + ld_row) * lda + ld_c4;  \
        b_src = B + ((int64_t)(t_ % tiles_n) * kS + ld_row) * ldb + ld_c4;  \
    }
    SW_SET_SRC()
    float4 ra[4], rb[4];
#define SW_LOAD_A(I) ra[I] = *reinterpret_cast<const float4*>(a_src + (int64_t)(64 * (I)) * lda + ld_k);
#define SW_LOAD_B(I) rb[I] = *reinterpret_cast<const float4*>(b_src + (int64_t)(64 * (I)) * ldb + ld_k);
#define SW_ADVANCE()                                                                           \
    ld_k += kSBK;                                                                              \
    if (ld_k == K) {                                                                           \
        ld_k = 0;                                                                              \
        ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */ \
        SW_SET_SRC()                                                                           \
    }
    // LDS image of gemm_nt_x6_pp_kernel: unpadded 32-byte rows, the two 16-byte chunks XOR-swizzled by bit 3 of the row
#define SW_ST1(R, PLANE0, ROW, BUFP)                                                 \
    {                                                                                \
        uint2 h_, m_, l_;                                                            \
        split3x4(R, h_, m_, l_);                                                     \
        const int o_ = (ROW) * 32 + ((((ld_c4 >> 3) ^ ((ROW) >> 3)) & 1) << 4) + (ld_c4 & 7) * 2; \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 0) * kSPlane + o_) = h_;      \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 1) * kSPlane + o_) = m_;      \
        *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 2) * kSPlane + o_) = l_;      \
    }
#define SW_STORE_A(I, BUFP) SW_ST1(ra[I], 0, ld_row + 64 * (I), BUFP)
#define SW_STORE_B(I, BUFP) SW_ST1(rb[I], 3, ld_row + 64 * (I), BUFP)

    const int swz = ((kh ^ (li >> 3)) & 1) << 4;
    const int a_off = (wm * 128 + li) * 32 + swz;
    const int b_off = 3 * kSPlane + (wn * 128 + li) * 32 + swz;
    unsigned char* const buf0 = smem_s;
    unsigned char* const buf1 = smem_s + kSBuf;
    // fragments: plane 0 = high, 1 = mid, 2 = low; fbh2 = the alternate high-plane set of B
    bf16x8 fa[3][4], fb[3][4], fbh2[4];
#define SW_READ_A(P, BUFP)                                                                                           \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                    \
        fa[P][t] = *reinterpret_cast<const bf16x8*>((BUFP) + a_off + (P) * kSPlane + t * 32 * 32);
#define SW_READ_B(DST, P, BUFP)                                                                                      \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                    \
        DST[t] = *reinterpret_cast<const bf16x8*>((BUFP) + b_off + (P) * kSPlane + t * 32 * 32);
#define SW_TERM(FA, FB)                                                                                              \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                                 \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                             \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[mt], FB[nt], acc[mt][nt], 0, 0, 0);
    // ---- hand-placed interleave: every MFMA is followed by one "filler" chunk and a scheduling fence, so the emitted
    // stream is exactly this order (the sched_group_barrier solver left 40-60 instruction clumps between MFMA runs).
    // Chunks of one staged float4: two elements' split (8 VALU) | the other two (8 VALU) | 6 packs + 3 ds_write_b64.
#define PIN __builtin_amdgcn_sched_barrier(0);
    uint32_t sh0, sm0, sl0, sh1, sm1, sl1, sh2, sm2, sl2, sh3, sm3, sl3;
#define SP0(R) split3(R.x, sh0, sm0, sl0);
#define SP1(R) split3(R.y, sh1, sm1, sl1);
#define SP2(R) split3(R.z, sh2, sm2, sl2);
#define SP3(R) split3(R.w, sh3, sm3, sl3);
#define ST_OFF(ROW) ((ROW) * 32 + ((((ld_c4 >> 3) ^ ((ROW) >> 3)) & 1) << 4) + (ld_c4 & 7) * 2)
#define ST_H(PLANE0, ROW, BUFP) *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 0) * kSPlane + ST_OFF(ROW)) = make_uint2(pack_hi(sh0, sh1), pack_hi(sh2, sh3));
#define ST_M(PLANE0, ROW, BUFP) *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 1) * kSPlane + ST_OFF(ROW)) = make_uint2(pack_hi(sm0, sm1), pack_hi(sm2, sm3));
#define ST_L(PLANE0, ROW, BUFP) *reinterpret_cast<uint2*>((BUFP) + ((PLANE0) + 2) * kSPlane + ST_OFF(ROW)) = make_uint2(pack_hi(sl0, sl1), pack_hi(sl2, sl3));
#define RD_A(P, T_, BUFP) fa[P][T_] = *reinterpret_cast<const bf16x8*>((BUFP) + a_off + (P) * kSPlane + (T_) * 32 * 32);
#define RD_B(DST, P, T_, BUFP) DST[T_] = *reinterpret_cast<const bf16x8*>((BUFP) + b_off + (P) * kSPlane + (T_) * 32 * 32);
#define MF(I, FA, FB) acc[(I) >> 2][(I) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[(I) >> 2], FB[(I) & 3], acc[(I) >> 2][(I) & 3], 0, 0, 0);
    // a group of 16 MFMAs FA x FB with the fillers F0 .. F15 behind them
#define GROUP(FA, FB, F0, F1, F2, F3, F4, F5, F6, F7, F8, F9, F10, F11, F12, F13, F14, F15)                            \
    MF(0, FA, FB) F0 PIN MF(1, FA, FB) F1 PIN MF(2, FA, FB) F2 PIN MF(3, FA, FB) F3 PIN                                \
    MF(4, FA, FB) F4 PIN MF(5, FA, FB) F5 PIN MF(6, FA, FB) F6 PIN MF(7, FA, FB) F7 PIN                                \
    MF(8, FA, FB) F8 PIN MF(9, FA, FB) F9 PIN MF(10, FA, FB) F10 PIN MF(11, FA, FB) F11 PIN                            \
    MF(12, FA, FB) F12 PIN MF(13, FA, FB) F13 PIN MF(14, FA, FB) F14 PIN MF(15, FA, FB) F15 PIN
#define NOP_

    // ---- epilogue of the output tile `ep_tile` (direct stores in the accumulator layout, as gemm_nt_x6_pp_kernel) ----
    int ep_tile = blockIdx.x;
    const int ldci = (int)ldc;
#define SW_EPILOGUE()                                                                                                  \
    {                                                                                                                  \
        const int t_ = xcd_swizzle(ep_tile, tiles);                                                                    \
        const int64_t m0 = (int64_t)(t_ / tiles_n) * kS;                                                               \
        const int n0 = (t_ % tiles_n) * kS;                                                                            \
        const __amdgpu_buffer_rsrc_t rc =                                                                              \
            __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * ldc + n0), 0, 0x7FFFFFFF, 0x00020000);                  \
        const int voff_c = ((wm * 128 + 4 * kh) * ldci + wn * 128 + li) * 4;                                           \
        float bv[4];                                                                                                   \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                               \
            bv[nt] = (EPI & E_BIAS) ? ep.bias[n0 + wn * 128 + nt * 32 + li] : 0.0f;                                    \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                               \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                         \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                       \
                    float v = acc[mt][nt][r] + bv[nt];                                                                 \
                    if (EPI & E_RELU) v = fmaxf(v, 0.0f);                                                              \
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff_c,                 \
                                                          ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldci + nt * 32) * 4, 0); \
                    acc[mt][nt][r] = 0.0f;                                                                             \
                }                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);      /* one accumulator tile's 16 temporaries at a time */          \
            }                                                                                                          \
        ep_tile += gridDim.x;                                                                                          \
    }

    // one K tile: CUR_ holds tile s, NXT_ receives tile s+1; BH_ = high-plane B fragments of tile s, BHN_ = those of s+1
#define SW_ITER(CUR_, NXT_, BH_, BHN_)                                                                                 \
    {                                                                                                                  \
        /* G1: A.l x B.h | read A.h(s); rows 0, 1 of A(s+1) */                                                         \
        GROUP(fa[2], BH_, RD_A(0, 0, CUR_) RD_A(0, 1, CUR_), RD_A(0, 2, CUR_) RD_A(0, 3, CUR_),                        \
              SP0(ra[0]), SP1(ra[0]), SP2(ra[0]), SP3(ra[0]), ST_H(0, ld_row, NXT_), ST_M(0, ld_row, NXT_), ST_L(0, ld_row, NXT_),                                                                                                      \
              SP0(ra[1]), SP1(ra[1]), SP2(ra[1]), SP3(ra[1]), ST_H(0, ld_row + 64, NXT_), ST_M(0, ld_row + 64, NXT_), ST_L(0, ld_row + 64, NXT_))                                                                                                      \
        /* G2: A.h x B.l | request rows 0, 1 of A(s+2); rows 2, 3 of A(s+1) */                                         \
        GROUP(fa[0], fb[2], SW_ADVANCE() SW_LOAD_A(0), SW_LOAD_A(1),                                                   \
              SP0(ra[2]), SP1(ra[2]), SP2(ra[2]), SP3(ra[2]), ST_H(0, ld_row + 128, NXT_), ST_M(0, ld_row + 128, NXT_), ST_L(0, ld_row + 128, NXT_),                                                                                                      \
              SP0(ra[3]), SP1(ra[3]), SP2(ra[3]), SP3(ra[3]), ST_H(0, ld_row + 192, NXT_), ST_M(0, ld_row + 192, NXT_), ST_L(0, ld_row + 192, NXT_))                                                                                                      \
        /* G3: A.m x B.m | request rows 2, 3 of A(s+2); rows 0, 1 of B(s+1) */                                         \
        GROUP(fa[1], fb[1], SW_LOAD_A(2), SW_LOAD_A(3),                                                                \
              SP0(rb[0]), SP1(rb[0]), SP2(rb[0]), SP3(rb[0]), ST_H(3, ld_row, NXT_), ST_M(3, ld_row, NXT_), ST_L(3, ld_row, NXT_),                                                                                                      \
              SP0(rb[1]), SP1(rb[1]), SP2(rb[1]), SP3(rb[1]), ST_H(3, ld_row + 64, NXT_), ST_M(3, ld_row + 64, NXT_), ST_L(3, ld_row + 64, NXT_))                                                                                                      \
        /* G4: A.m x B.h | request rows 0, 1 of B(s+2); rows 2, 3 of B(s+1); the barrier */                            \
        GROUP(fa[1], BH_, SW_LOAD_B(0), SW_LOAD_B(1),                                                                  \
              SP0(rb[2]), SP1(rb[2]), SP2(rb[2]), SP3(rb[2]), ST_H(3, ld_row + 128, NXT_), ST_M(3, ld_row + 128, NXT_), ST_L(3, ld_row + 128, NXT_),                                                                                                      \
              SP0(rb[3]), SP1(rb[3]), SP2(rb[3]), SP3(rb[3]), ST_H(3, ld_row + 192, NXT_), ST_M(3, ld_row + 192, NXT_), ST_L(3, ld_row + 192, NXT_))                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                \
        PIN                                                                                                            \
        /* G5: A.h x B.m | request rows 2, 3 of B(s+2); read A.l, B.l, B.h' (s+1) */                                   \
        GROUP(fa[0], fb[1], SW_LOAD_B(2), SW_LOAD_B(3), RD_A(2, 0, NXT_), RD_A(2, 1, NXT_), RD_A(2, 2, NXT_),          \
              RD_A(2, 3, NXT_), RD_B(fb[2], 2, 0, NXT_), RD_B(fb[2], 2, 1, NXT_), RD_B(fb[2], 2, 2, NXT_),             \
              RD_B(fb[2], 2, 3, NXT_), RD_B(BHN_, 0, 0, NXT_), RD_B(BHN_, 0, 1, NXT_), RD_B(BHN_, 0, 2, NXT_),         \
              RD_B(BHN_, 0, 3, NXT_), NOP_, NOP_)                                                                      \
        /* G6: A.h x B.h | read A.m, B.m (s+1): their registers retired with G4 / G5 */                                \
        GROUP(fa[0], BH_, RD_A(1, 0, NXT_), NOP_, RD_A(1, 1, NXT_), NOP_, RD_A(1, 2, NXT_), NOP_, RD_A(1, 3, NXT_),    \
              NOP_, RD_B(fb[1], 1, 0, NXT_), NOP_, RD_B(fb[1], 1, 1, NXT_), NOP_, RD_B(fb[1], 1, 2, NXT_), NOP_,       \
              RD_B(fb[1], 1, 3, NXT_), NOP_)                                                                           \
    }

    // prologue: K tile 0 -> buffer 0, all of its fragments except A.h read; K tile 1 requested
    SW_LOAD_A(0) SW_LOAD_A(1) SW_LOAD_A(2) SW_LOAD_A(3) SW_LOAD_B(0) SW_LOAD_B(1) SW_LOAD_B(2) SW_LOAD_B(3)
    SW_STORE_A(0, buf0) SW_STORE_A(1, buf0) SW_STORE_A(2, buf0) SW_STORE_A(3, buf0)
    SW_STORE_B(0, buf0) SW_STORE_B(1, buf0) SW_STORE_B(2, buf0) SW_STORE_B(3, buf0)
    SW_ADVANCE()
    SW_LOAD_A(0) SW_LOAD_A(1) SW_LOAD_A(2) SW_LOAD_A(3) SW_LOAD_B(0) SW_LOAD_B(1) SW_LOAD_B(2) SW_LOAD_B(3)
    __syncthreads();
    SW_READ_A(2, buf0) SW_READ_A(1, buf0) SW_READ_B(fb[2], 2, buf0) SW_READ_B(fb[1], 1, buf0) SW_READ_B(fb[0], 0, buf0)
    // nested loops (an epilogue INSIDE the K-tile loop makes hipcc shuttle accumulator tiles between AGPRs and VGPRs at the
    // loop header); the load / fragment pipeline simply runs on across the output tiles.  T is even (K % 32 == 0).
#pragma unroll 1
    for (int it = 0; it < my_tiles; ++it) {
#pragma unroll 1
        for (int kp = 0; kp < T; kp += 2) {
            SW_ITER(buf0, buf1, fb[0], fbh2)
            SW_ITER(buf1, buf0, fbh2, fb[0])
        }
        SW_EPILOGUE()
    }
#undef SW_ITER
#undef SW_EPILOGUE
#undef GROUP
#undef MF
#undef RD_B
#undef RD_A
#undef ST_L
#undef ST_M
#undef ST_H
#undef ST_OFF
#undef SP3
#undef SP2
#undef SP1
#undef SP0
#undef PIN
#undef NOP_
#undef SW_TERM
#undef SW_READ_B
#undef SW_READ_A
#undef SW_STORE_B
#undef SW_STORE_A
#undef SW_ST1
#undef SW_ADVANCE
#undef SW_LOAD_B
#undef SW_LOAD_A
#undef SW_SET_SRC
}

bool gemm_nt_sw_ok(int64_t M, int N, int K, int flags) {
    if (M % kS || N % kS || K % 32 || K < 32) return false;
    return flags == 0 || flags == E_BIAS;
}

int gemm_nt_sw_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                      int K, int flags, const EpiParams& ep, hipStream_t st) {
    const int tn = N / kS;
    const int tiles = (int)((M / kS) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kSThreads);
    const size_t lds = 2 * kSBuf;
#define SW_LAUNCH(EPIV)                                                                                               \
    {                                                                                                                 \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_sw_kernel<EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                      \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((gemm_nt_x6_sw_kernel<EPIV>), grid, block, lds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        VQ_CHECK_LAUNCH("gemm_nt_x6_sw");                                                                             \
        return VQCPC_OK;                                                                                              \
    }
    switch (flags) {
        case 0: SW_LAUNCH(0)
        case E_BIAS: SW_LAUNCH(E_BIAS)
        default: break;
    }
#undef SW_LAUNCH
    set_error("gemm_nt_sw: epilogue combination %d not instantiated", flags);
    return VQCPC_EINVAL;
}

}  // namespace vq
