// bf16 GEMMs for BASELINE configs[4] (reduced precision: bf16 operands, fp32 accumulation, one v_mfma_f32_32x32x16_bf16 per
// product).  Unlike MODE 2 of the 128-tile kernels in gemm.hip (fp32 operands loaded from HBM and rounded while they are
// staged), the operands here ARE bf16 in HBM: they go global -> VGPR -> LDS without conversion, and an output that only
// feeds another GEMM (the FFN hidden activation, its gradient) is written as bf16 and never exists in fp32.
//
//   vqcpc_cast_bf16      fp32 (row stride) -> dense bf16, round-to-nearest-even like torch's .bfloat16()
//   gemm_nt_bf16_kernel  C[M,N] = epi(A[M,K] . B[N,K]^T): 256 x 256 tile, 8 waves (2 x 4, wave tile 128 x 64), persistent
//                        over tiles, K tiles of 32 (64-byte rows).  Ping-pong wave groups as gemm_nt_x6_pp_kernel: group 1
//                        (rows 128..255) runs one phase behind group 0, so one wave of a SIMD issues its 16 MFMAs while the
//                        other one reads fragments / stores the next K tile.  Without the three bf16 planes of the split
//                        kernels there is room for TWO raw-operand register sets: a K tile is requested two phase pairs
//                        before it is written to LDS.
//                        LDS image of a K tile: A rows 0..255 then B rows 0..255, 64 bytes each, 16-byte chunk c of row r
//                        at chunk position c ^ ((r >> 2) & 3): fragment reads (ds_read_b128, 16 rows per lane group) and
//                        staging writes (ds_write_b128, 2 rows per 8-lane group) are conflict free.
//                        Epilogue through a 4 KB LDS scratch per wave (see gemm_dma.hip): dwordx4 stores of fp32 and / or
//                        dwordx2 stores of bf16.
#include <stdlib.h>

#include <atomic>

#include "gemm_common.h"

namespace vq {

typedef unsigned short bf16_t;
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kB = 256;                      // tile edge
constexpr int kBBK = 32;                     // k (bf16 elements) per stage
constexpr int kBRowB = kBBK * 2;             // 64 bytes per operand row per stage
constexpr int kBOperand = kB * kBRowB;       // 16 KB
constexpr int kBStage = 2 * kBOperand;       // 32 KB
constexpr int kBThreads = 512;
constexpr int kBLds = 2 * kBStage + 8 * 4096;   // two stages + a 4 KB epilogue scratch per wave = 96 KB

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, int64_t ld, bf16_t* __restrict__ out,
                                                        int64_t rows, int cols) {
    const int64_t n4 = rows * (cols / 4);
    const int c4n = cols / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        const float4 v = *reinterpret_cast<const float4*>(in + r * ld + c);
        *reinterpret_cast<uint2*>(out + r * cols + c) = round4_bf16(v);
    }
}

// extra epilogue features of the bf16 kernel (on top of gemm_common.h's E_*)
enum { B_OUT_F32 = 1, B_OUT_BF16 = 2, B_GATE_BF16 = 4, B_ADD_BF16 = 8 };   // B_ADD_BF16: the residual operand (E_ADD) is bf16, in `gate_b`
// cache policy of the output stores (" nt", " sc1", ..: measurement builds, tools/build_lab_variants.sh)
#ifndef VQ_BF16_STORE_POL
#define VQ_BF16_STORE_POL ""
#endif

struct Bf16Out {
    float* c;            // fp32 output (B_OUT_F32)
    int64_t ldc;
    bf16_t* cb;          // bf16 output (B_OUT_BF16), dense or strided
    int64_t ldcb;
    const bf16_t* gate_b;   // bf16 gate operand (B_GATE_BF16): only its sign is used; or the bf16 residual operand (B_ADD_BF16)
    int64_t ldgate_b;
    int stagger;            // start delay of workgroup b: ((b >> 3) & 3) * stagger * 64 clocks (see the kernel); -1 = measurement
                            // variant without the epilogue's global stores (VQCPC_BF16_STAGGER=-1, tools/bench_gemm_bf16.py)
};

// =====================================================================================================================
// Epilogue of the 256 x 256 bf16 NT kernels (shared by gemm_nt_bf16_kernel and gemm_nt_bf16_k64_kernel; expands inside the
// kernel body and uses its locals: smem, wave, lane, li, kh, wm, wn, o, ep, N, tiles, tiles_n, acc, EPI, OUT).
// One 32 x 32 accumulator tile at a time goes through a 4 KB LDS scratch of its wave (16 ds_write_b32, 4 ds_read_b128:
// lane -> row lane >> 3 (+ 8 j), columns 4 (lane & 7) ..) and leaves as dwordx4 stores of fp32 and / or dwordx2 stores of
// bf16 (8 row segments of 128 / 64 bytes per instruction); bias / gate / residual operands are fetched in the same shape
// (fp32: 4 x dwordx4; bf16 gate: 4 x dwordx2 kept RAW, only the sign is looked at).  Software-pipelined: while tile t is
// finished (bias / activation / dropout / gate / residual, stores) the scratch round trip of tile t + 1 and the operand
// loads of tile t + 3 are in flight.  A bf16 gate operand (24 VGPRs for three tiles) is requested one whole phase pair
// earlier (B_AUX_PREFETCH in the last memory phase of the tile); fp32 operands (48 VGPRs) only in the epilogue itself.
// LDS accesses and stores are inline asm (see gemm_dma.hip: no compiler-inserted vmcnt(0), explicit wait states).
// SCR_OFF = byte offset of the 8 x 4 KB scratch area inside the kernel's dynamic LDS.
#define B_EPI_DECLS(SCR_OFF)                                                                                           \
    int ep_tile = blockIdx.x;                                                                                          \
    constexpr bool HAS_AUX = (EPI & (E_GATE | E_ADD)) != 0;                                                            \
    constexpr bool AUX_B16 = (OUT & (B_GATE_BF16 | B_ADD_BF16)) != 0;  /* gate (sign only) / residual operand is bf16 */     \
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;                 \
    const unsigned scr = lds0 + (SCR_OFF) + wave * 4096;                                                               \
    const unsigned scr_w = scr + ((4 * kh) * 32 + li) * 4;                                                             \
    const unsigned scr_r = scr + ((lane >> 3) * 32 + (lane & 7) * 4) * 4;                                              \
    const int e_row = wm * 128 + (lane >> 3), e_col = wn * 64 + 4 * (lane & 7);                                        \
    const int ldci = (int)o.ldc, ldcbi = (int)o.ldcb;                                                                  \
    const float* xsrc = (EPI & E_GATE) ? ep.gate : ep.add;                                                             \
    const int ldxi = (int)(AUX_B16 ? o.ldgate_b : ((EPI & E_GATE) ? ep.ldgate : ep.ldadd));                            \
    union AuxT {                                                                                                       \
        fx4 f[4];                                                                                                      \
        u32x2 h[4];                                                                                                    \
    };                                                                                                                 \
    AuxT aux0, aux1, aux2;
#define B_SCR_WRITE(MT, NT)                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                     \
        asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(scr_w), "v"(acc[MT][NT][r]), "i"(((r & 3) + 8 * (r >> 2)) * 128));
#define B_SCR_READ(V)                                                                                                  \
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(V[0]) : "v"(scr_r));                                            \
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(V[1]) : "v"(scr_r));                                         \
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(V[2]) : "v"(scr_r));                                         \
    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(V[3]) : "v"(scr_r));
#define B_SCR_WAIT(V) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]));
#define B_AUX_LOAD(DST, TILE)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
        if (AUX_B16)                                                                                                   \
            DST.h[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(                                 \
                rx, voff_x, ((((TILE) >> 1) * 32 + 8 * j) * ldxi + ((TILE) & 1) * 32) * 2, 0));                        \
        else                                                                                                           \
            DST.f[j] = __builtin_bit_cast(fx4, __builtin_amdgcn_raw_buffer_load_b128(                                  \
                rx, voff_x, ((((TILE) >> 1) * 32 + 8 * j) * ldxi + ((TILE) & 1) * 32) * 4, 0));                        \
    }
#define B_EPI_TILE(V, AUX, TILE)                                                                                       \
    {                                                                                                                  \
        constexpr int MT = (TILE) >> 1, NT = (TILE) & 1;                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
            const int64_t row = m0 + e_row + MT * 32 + 8 * j;                                                          \
            const int col = n0 + e_col + NT * 32;                                                                      \
            fx4 ov;                                                                                                    \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                            \
                float v = V[j][c];                                                                                     \
                if (EPI & E_BIAS) v += bias4[NT][c];                                                                   \
                if (EPI & E_RELU) v = fmaxf(v, 0.0f);                                                                  \
                if (EPI & E_DROP)   /* == drop_scale(ep.seed, (row + ep.row0) * N + col + c, ..); thr > 0 on this path */  \
                    v *= rng_u24_from_x0(x0_lane + (uint32_t)((MT * 32 + 8 * j) * N + NT * 32 + c) * kRngMul, drop_sh) >= ep.thr \
                             ? ep.inv_keep : 0.0f;                                                                     \
                if (EPI & E_GATE) {                                                                                    \
                    bool pos;                                                                                          \
                    if (AUX_B16) {                                                                                     \
                        const unsigned hw = (AUX.h[j][c >> 1] >> (16 * (c & 1))) & 0xFFFFu;       /* bf16 > 0 */       \
                        pos = hw != 0 && hw < 0x8000u;                                                                 \
                    } else {                                                                                           \
                        pos = AUX.f[j][c] > 0.0f;                                                                      \
                    }                                                                                                  \
                    v *= pos ? ep.gate_scale : 0.0f;                                                                   \
                }                                                                                                      \
                if (EPI & E_ADD) v += AUX_B16 ? __uint_as_float(((AUX.h[j][c >> 1] >> (16 * (c & 1))) & 0xFFFFu) << 16) : AUX.f[j][c]; \
                ov[c] = v;                                                                                             \
            }                                                                                                          \
            if ((OUT & B_OUT_F32) && o.stagger != -1)                                                                  \
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen" VQ_BF16_STORE_POL "\n\ts_nop 1" ::"v"(ov), "v"(voff_c), \
                             "s"(rc), "s"(((MT * 32 + 8 * j) * ldci + NT * 32) * 4) : "memory");                       \
            if ((OUT & B_OUT_BF16) && o.stagger != -1) {                                                               \
                u32x2 pk;                                                                                              \
                pk[0] = cvt_pk_bf16(ov[0], ov[1]);                                                                     \
                pk[1] = cvt_pk_bf16(ov[2], ov[3]);                                                                     \
                asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, %3 offen" VQ_BF16_STORE_POL "\n\ts_nop 1" ::"v"(pk), "v"(voff_cb), \
                             "s"(rcb), "s"(((MT * 32 + 8 * j) * ldcbi + NT * 32) * 2) : "memory");                     \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[MT][NT][r] = 0.0f;                                          \
    }
#define B_EPI_DESC()                                                                                                   \
    const int t_ = xcd_swizzle(ep_tile, tiles);                                                                        \
    const int64_t m0 = (int64_t)(t_ / tiles_n) * kTileRows;                                                            \
    const int n0 = (t_ % tiles_n) * kB;                                                                                \
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(                                               \
        (void*)(!HAS_AUX ? (const void*)smem                                                                           \
                         : AUX_B16 ? (const void*)(o.gate_b + m0 * (int64_t)ldxi + n0)                                 \
                                   : (const void*)(xsrc + m0 * (int64_t)ldxi + n0)),                                   \
        0, 0x7FFFFFFF, 0x00020000);                                                                                    \
    const int voff_x = (e_row * ldxi + e_col) * (AUX_B16 ? 2 : 4);
#define B_AUX_PREFETCH()                                                                                               \
    {                                                                                                                  \
        B_EPI_DESC()                                                                                                   \
        B_AUX_LOAD(aux0, 0) B_AUX_LOAD(aux1, 1) B_AUX_LOAD(aux2, 2)                                                    \
    }
#define B_EPI_STEP(TILE, VCUR, VNXT, AUXC)                                                                             \
    B_SCR_WAIT(VCUR)                                                                                                   \
    if ((TILE) + 1 < 8) {                                                                                              \
        B_SCR_WRITE(((TILE) + 1) >> 1, ((TILE) + 1) & 1)                                                               \
        B_SCR_READ(VNXT)                                                                                               \
    }                                                                                                                  \
    B_EPI_TILE(VCUR, AUXC, TILE)                                                                                       \
    if (HAS_AUX && (TILE) + 3 < 8) { B_AUX_LOAD(AUXC, (TILE) + 3) }
#define B_EPILOGUE_P(PREFETCHED)                                                                                       \
    {                                                                                                                  \
        B_EPI_DESC()                                                                                                   \
        /* dropout hash input of this lane's first element of the output tile; the others are constant offsets away */ \
        const uint64_t drop_se = rng_seed_eff(ep.seed);                                                                \
        const uint32_t drop_sh = (uint32_t)(drop_se >> 32);                                                            \
        const uint32_t x0_lane = rng_x0(drop_se, (uint32_t)(m0 + e_row + ep.row0) * (uint32_t)N + (uint32_t)(n0 + e_col)); \
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)((OUT & B_OUT_F32) ? o.c + m0 * o.ldc + n0 : (float*)smem), 0, 0x7FFFFFFF, 0x00020000);             \
        const __amdgpu_buffer_rsrc_t rcb = __builtin_amdgcn_make_buffer_rsrc(                                          \
            (void*)((OUT & B_OUT_BF16) ? o.cb + m0 * o.ldcb + n0 : (bf16_t*)smem), 0, 0x7FFFFFFF, 0x00020000);         \
        const int voff_c = (e_row * ldci + e_col) * 4;                                                                 \
        const int voff_cb = (e_row * ldcbi + e_col) * 2;                                                               \
        fx4 bias4[2];                                                                                                  \
        if (EPI & E_BIAS) {                                                                                            \
            bias4[0] = *reinterpret_cast<const fx4*>(ep.bias + n0 + e_col);                                            \
            bias4[1] = *reinterpret_cast<const fx4*>(ep.bias + n0 + e_col + 32);                                       \
        }                                                                                                              \
        if (HAS_AUX && !(PREFETCHED)) { B_AUX_LOAD(aux0, 0) B_AUX_LOAD(aux1, 1) B_AUX_LOAD(aux2, 2) }                   \
        fx4 va[4], vb[4];                                                                                              \
        B_SCR_WRITE(0, 0)                                                                                              \
        B_SCR_READ(va)                                                                                                 \
        B_EPI_STEP(0, va, vb, aux0) B_EPI_STEP(1, vb, va, aux1) B_EPI_STEP(2, va, vb, aux2) B_EPI_STEP(3, vb, va, aux0) \
        B_EPI_STEP(4, va, vb, aux1) B_EPI_STEP(5, vb, va, aux2) B_EPI_STEP(6, va, vb, aux0) B_EPI_STEP(7, vb, va, aux1) \
        ep_tile += gridDim.x;                                                                                          \
    }
#define B_EPILOGUE() B_EPILOGUE_P(AUX_B16)   /* callers that request a bf16 gate operand with B_AUX_PREFETCH */

template <int EPI, int OUT>
__global__ __launch_bounds__(kBThreads, 2) void gemm_nt_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const bf16_t* __restrict__ B, int64_t ldb, Bf16Out o,
                                                                   int64_t M, int N, int K, int tiles_n, int tiles,
                                                                   EpiParams ep) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                       // wm = wave group: 0 leads, 1 runs one phase behind
    const int li = lane & 31, kh = lane >> 5;
    const int T = K / kBBK;                                        // K tiles per output tile (K % 64 == 0: T even)
    const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * T;

    // Persistent workgroups with identical tiles run in lockstep: all 256 CUs store their output tiles (64 MB of fp32 per
    // round) at the same time, and at configs[4]'s row count those bytes go to HBM, not to the 256 MB MALL -- the chip
    // alternates between a phase that only computes and a phase that only writes.  Starting the four workgroups that share
    // an XCD slot group a quarter of a tile apart keeps the write stream continuous under the other workgroups' K loops.
    if (o.stagger > 0) {
        const int n = (((int)blockIdx.x >> 3) & 3) * o.stagger;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
    }

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- staging (group-local): group g moves rows [128 g, 128 g + 128) of both operands; thread -> 16-byte chunks q, q + 256
    const int tg = tid & 255;
    const int st_row0 = tg >> 2, st_c = tg & 3;                    // chunk q = tg: row tg >> 2, chunk tg & 3; q + 256: row + 64
    int ld_tile = blockIdx.x, ld_k = 0;
    const bf16_t* a_src;
    const bf16_t* b_src;
#define B_SET_SRC()                                                                              \
    {                                                                                            \
        const int t_ = xcd_swizzle(min(ld_tile, tiles - 1), tiles);                              \
        a_src = A + ((int64_t)(t_ / tiles_n) * kB + wm * 128 + st_row0) * lda + st_c * 8;        \
        b_src = B + ((int64_t)(t_ % tiles_n) * kB + wm * 128 + st_row0) * ldb + st_c * 8;        \
    }
    B_SET_SRC()
    uint4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
#define B_LOAD(S_)                                                                        \
    S_##a0 = *reinterpret_cast<const uint4*>(a_src + ld_k);                               \
    S_##a1 = *reinterpret_cast<const uint4*>(a_src + (int64_t)64 * lda + ld_k);           \
    S_##b0 = *reinterpret_cast<const uint4*>(b_src + ld_k);                               \
    S_##b1 = *reinterpret_cast<const uint4*>(b_src + (int64_t)64 * ldb + ld_k);           \
    ld_k += kBBK;                                                                         \
    if (ld_k == K) {                                                                      \
        ld_k = 0;                                                                         \
        ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */ \
        B_SET_SRC()                                                                       \
    }
    // LDS position of chunk (row, c): row * 64 + ((c ^ ((row >> 2) & 3)) << 4); rows st_row0 and st_row0 + 64 share the swizzle
    const int st_off = (wm * 128 + st_row0) * kBRowB + ((st_c ^ ((st_row0 >> 2) & 3)) << 4);
#define B_STORE(S_, BUFP)                                                                 \
    *reinterpret_cast<uint4*>((BUFP) + st_off) = S_##a0;                                  \
    *reinterpret_cast<uint4*>((BUFP) + st_off + 64 * kBRowB) = S_##a1;                    \
    *reinterpret_cast<uint4*>((BUFP) + kBOperand + st_off) = S_##b0;                      \
    *reinterpret_cast<uint4*>((BUFP) + kBOperand + st_off + 64 * kBRowB) = S_##b1;

    // ---- fragments: lane (row li, k group kh) of k16 step ks reads chunk 2 ks + kh of its row ----
    const int fsw = (li >> 2) & 3;
    const int f_off0 = li * kBRowB + (((0 + kh) ^ fsw) << 4);      // ks = 0
    const int f_off1 = li * kBRowB + (((2 + kh) ^ fsw) << 4);      // ks = 1
    const int a_base = (wm * 128) * kBRowB;
    const int b_base = kBOperand + (wn * 64) * kBRowB;
    bf16x8 fb[2][2], fa[2][4];                                     // [ks][tile]
#define B_READ_FRAGS(BUFP)                                                                                   \
    _Pragma("unroll") for (int tl = 0; tl < 2; ++tl) {                                                       \
        fb[0][tl] = *reinterpret_cast<const bf16x8*>((BUFP) + b_base + tl * 32 * kBRowB + f_off0);           \
        fb[1][tl] = *reinterpret_cast<const bf16x8*>((BUFP) + b_base + tl * 32 * kBRowB + f_off1);           \
    }                                                                                                        \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                       \
        fa[0][mt] = *reinterpret_cast<const bf16x8*>((BUFP) + a_base + mt * 32 * kBRowB + f_off0);           \
        fa[1][mt] = *reinterpret_cast<const bf16x8*>((BUFP) + a_base + mt * 32 * kBRowB + f_off1);           \
    }
#define B_MFMA()                                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                   \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt], fb[ks][0], acc[mt][0], 0, 0, 0); \
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt], fb[ks][1], acc[mt][1], 0, 0, 0); \
        }
#define B_BARRIER()                                                  \
    __builtin_amdgcn_sched_barrier(0);                               \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue through the wave's LDS scratch: macros B_EPILOGUE / B_AUX_PREFETCH at file scope above ----
    constexpr int kTileRows = kB;
    B_EPI_DECLS(2 * kBStage)
    // one phase pair for stream position s: RB_ = LDS buffer with K tile s, WB_ = buffer for tile s+1; register set SET_
    // holds tile s+1 (requested two phase pairs ago), is written to LDS and re-used for the request of tile s+3
#define B_PHASES(RB_, WB_, SET_)                                                  \
    {                                                                             \
        if (kt == 0 && s > 0) B_EPILOGUE()                                        \
        if (HAS_AUX && AUX_B16 && kt == T - 1) B_AUX_PREFETCH()                   \
        B_READ_FRAGS(RB_)                                                         \
        B_STORE(SET_, WB_)                                                        \
        B_LOAD(SET_)                                                              \
        B_BARRIER()                                                               \
        __builtin_amdgcn_s_setprio(1);                                            \
        B_MFMA()                                                                  \
        __builtin_amdgcn_s_setprio(0);                                            \
        B_BARRIER()                                                               \
        ++s;                                                                      \
        kt = (kt + 1 == T) ? 0 : kt + 1;                                          \
    }

    unsigned char* const buf0 = smem;
    unsigned char* const buf1 = smem + kBStage;
    // prologue: K tile 0 in buffer 0, tiles 1 (set y) and 2 (set x) requested
    B_LOAD(x)
    B_LOAD(y)
    B_STORE(x, buf0)
    B_LOAD(x)
    B_BARRIER()
    if (wm == 1) { B_BARRIER() }                         // group 1 falls one phase behind
    int s = 0, kt = 0;
#pragma unroll 1
    while (s < S) {
        B_PHASES(buf0, buf1, y)                          // tile s+1 is in set y (s even), in set x (s odd)
        B_PHASES(buf1, buf0, x)
    }
    if (wm == 0) { B_BARRIER() }                         // pairs with group 1's last barrier
    B_EPILOGUE()
#undef B_PHASES
#undef B_BARRIER
#undef B_MFMA
#undef B_READ_FRAGS
#undef B_STORE
#undef B_LOAD
#undef B_SET_SRC
}


// ---------------------------------------------------------------------------------------------------------------------
// gemm_nt_bf16_k64_kernel: 256 x 256 tile, 8 waves (2 x 4, wave tile 128 x 64), persistent, K tiles of 64 (128-byte operand
// rows = whole cache lines per request) delivered global -> LDS by DMA into a ring of TWO 64 KB slots.
//
// Counters and ablations of the K-tile-32 kernels (profiles/r04_gemm_bf16.md): every L1 -> L2 read request is a 64-byte
// half line (2.9e8 requests for 17 GB), waves sit in issue stalls for half of their cycles while the matrix pipes are busy
// 50-59 %, and a DMA variant of the same loop WITHOUT its operand delivery ran at 1 450 TFLOP/s (with it: 930): the request
// stream, not the MFMAs, paces them.  128-byte rows halve the requests per byte.  (The same round measured a 128 x 256 tile on
// two independent four-wave workgroups per CU, meant to overlap one workgroup's epilogue with the other's K loop: it moves a
// third more operand bytes per MFMA and was slower for that reason; and an LDS-free epilogue on transposed accumulators,
// whose 32-byte row pieces per store were slower than the LDS-transposed full-line stores.)
//
// LDS: A slot 0 | A slot 1 | B slot 0 | B slot 1 (32 KB each) | 8 x 4 KB epilogue scratch.  16-byte chunk c of row r at chunk
// position c ^ ((r >> 1) & 7): a fragment read (ds_read_b128, lane = row li, chunk 2 ks + kh) touches 16 distinct 16-byte
// slots of the 256-byte bank row in every lane group.  A DMA instruction writes 1 KB = 8 rows lane-linearly, so the swizzle is
// applied to the per-lane SOURCE address.
// No ping-pong: all eight waves run the same stream, ONE workgroup barrier per K tile.  A K tile is consumed in two halves
// (k16 steps 0-1 from register set X, 2-3 from set Y: 48 registers each, the 24 fragments of a whole tile would not fit):
//   half 0 of tile s:  read Y <- (s, steps 2-3);  16 MFMAs on X;  lgkmcnt(0)
//   half 1 of tile s:  vmcnt: own DMA pieces of tile s+1 landed;  s_barrier  (now everybody's have, and everybody has
//                      finished reading tile s);  DMA tile s+2 -> the slot of tile s;  read X <- (s+1, steps 0-1);
//                      16 MFMAs on Y;  lgkmcnt(0)
// so the LDS reads of the next half and the DMA issue always run under 16 MFMAs of the same wave, and a requested tile has
// 32 MFMAs (~2 000 cycles at two waves per SIMD) to land.  At an output-tile boundary the epilogue (B_EPILOGUE above) runs
// first and X is read after it (its registers are the epilogue's); the DMA of the next tile's first two K tiles is in flight
// meanwhile, and the wait that follows allows for the epilogue's own stores (vmcnt counts in order).
constexpr int kKRowB = 128;                    // bytes per operand row per K tile
constexpr int kKBK = 64;
constexpr int kKOperand = kB * kKRowB;         // 32 KB
constexpr int kKLds = 4 * kKOperand + 8 * 4096;   // 160 KB

template <int EPI, int OUT>
__global__ __launch_bounds__(kBThreads, 2) void gemm_nt_bf16_k64_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                       const bf16_t* __restrict__ B, int64_t ldb, Bf16Out o,
                                                                       int64_t M, int N, int K, int tiles_n, int tiles,
                                                                       EpiParams ep) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    const int T = K / kKBK;                                        // K tiles per output tile
    const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * T;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- DMA: wave w delivers rows [32 w, 32 w + 32) of A and of B: 4 + 4 instructions of 8 rows x 128 bytes ----
    // lane l -> row l >> 3 of the 8-row block j, chunk position l & 7 <- logical chunk (l & 7) ^ (((8 j + (l >> 3)) >> 1) & 7)
    //        = (l & 7) ^ ((4 (j & 1) + (l >> 4)) & 7): one per-lane offset for even j, one for odd j
    const int dl_row = lane >> 3;
    const int dl_ce = (lane & 7) ^ (lane >> 4), dl_co = (lane & 7) ^ (4 + (lane >> 4));
    const int ldai = (int)lda, ldbi = (int)ldb;
    const int voff_ae = (dl_row * ldai + dl_ce * 8) * 2, voff_ao = (dl_row * ldai + dl_co * 8) * 2;
    const int voff_be = (dl_row * ldbi + dl_ce * 8) * 2, voff_bo = (dl_row * ldbi + dl_co * 8) * 2;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    int ld_tile = blockIdx.x, ld_k = 0;
    __amdgpu_buffer_rsrc_t rs_a, rs_b;
#define K_SET_SRC()                                                                                                     \
    {                                                                                                                   \
        const int t_ = xcd_swizzle(min(ld_tile, tiles - 1), tiles);                                                     \
        rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + ((int64_t)(t_ / tiles_n) * kB + 32 * wave) * lda), 0,       \
                                                 0x7FFFFFFF, 0x00020000);                                               \
        rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(B + ((int64_t)(t_ % tiles_n) * kB + 32 * wave) * ldb), 0,       \
                                                 0x7FFFFFFF, 0x00020000);                                               \
    }
    K_SET_SRC()
#define K_ISSUE(SLOT)                                                                                                   \
    {                                                                                                                   \
        unsigned char* da_ = smem + (SLOT) * kKOperand + wave * 4096;                                                   \
        unsigned char* db_ = smem + (2 + (SLOT)) * kKOperand + wave * 4096;                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(da_ + j * 1024), 16, (j & 1) ? voff_ao : voff_ae, \
                                                     (j * 8 * ldai + ld_k) * 2, 0, 0);                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(db_ + j * 1024), 16, (j & 1) ? voff_bo : voff_be, \
                                                     (j * 8 * ldbi + ld_k) * 2, 0, 0);                                  \
        ld_k += kKBK;                                                                                                   \
        if (ld_k == K) {                                                                                                \
            ld_k = 0;                                                                                                   \
            ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */                     \
            K_SET_SRC()                                                                                                 \
        }                                                                                                               \
    }

    // ---- fragments (inline asm: no compiler-inserted vmcnt(0) while DMA is in flight) ----
    const unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int fsw = (li >> 1) & 7;
    const unsigned a_row = ldsb + (wm * 128 + li) * kKRowB;
    const unsigned b_row = ldsb + 2 * kKOperand + (wn * 64 + li) * kKRowB;
    const unsigned a_ad0 = a_row + (((0 + kh) ^ fsw) << 4), a_ad1 = a_row + (((2 + kh) ^ fsw) << 4);
    const unsigned a_ad2 = a_row + (((4 + kh) ^ fsw) << 4), a_ad3 = a_row + (((6 + kh) ^ fsw) << 4);
    const unsigned b_ad0 = b_row + (((0 + kh) ^ fsw) << 4), b_ad1 = b_row + (((2 + kh) ^ fsw) << 4);
    const unsigned b_ad2 = b_row + (((4 + kh) ^ fsw) << 4), b_ad3 = b_row + (((6 + kh) ^ fsw) << 4);
    fx4 xb[2][2], xa[2][4], yb[2][2], ya[2][4];                    // X: k16 steps 0, 1; Y: steps 2, 3; [step][tile]
#define K_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
#define K_READ(S_, SLOT, AD0A, AD1A, AD0B, AD1B)                                                                        \
    K_RD(S_##b[0][0], AD0B, (SLOT) * kKOperand);                                                                        \
    K_RD(S_##b[0][1], AD0B, (SLOT) * kKOperand + 32 * kKRowB);                                                          \
    K_RD(S_##a[0][0], AD0A, (SLOT) * kKOperand);                                                                        \
    K_RD(S_##a[0][1], AD0A, (SLOT) * kKOperand + 32 * kKRowB);                                                          \
    K_RD(S_##a[0][2], AD0A, (SLOT) * kKOperand + 64 * kKRowB);                                                          \
    K_RD(S_##a[0][3], AD0A, (SLOT) * kKOperand + 96 * kKRowB);                                                          \
    K_RD(S_##b[1][0], AD1B, (SLOT) * kKOperand);                                                                        \
    K_RD(S_##b[1][1], AD1B, (SLOT) * kKOperand + 32 * kKRowB);                                                          \
    K_RD(S_##a[1][0], AD1A, (SLOT) * kKOperand);                                                                        \
    K_RD(S_##a[1][1], AD1A, (SLOT) * kKOperand + 32 * kKRowB);                                                          \
    K_RD(S_##a[1][2], AD1A, (SLOT) * kKOperand + 64 * kKRowB);                                                          \
    K_RD(S_##a[1][3], AD1A, (SLOT) * kKOperand + 96 * kKRowB);
#define K_READ_X(SLOT) K_READ(x, SLOT, a_ad0, a_ad1, b_ad0, b_ad1)
#define K_READ_Y(SLOT) K_READ(y, SLOT, a_ad2, a_ad3, b_ad2, b_ad3)
#define K_LGKM_WAIT(S_)                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                 : "+v"(S_##b[0][0]), "+v"(S_##b[0][1]), "+v"(S_##b[1][0]), "+v"(S_##b[1][1]), "+v"(S_##a[0][0]),       \
                   "+v"(S_##a[0][1]), "+v"(S_##a[0][2]), "+v"(S_##a[0][3]), "+v"(S_##a[1][0]), "+v"(S_##a[1][1]),       \
                   "+v"(S_##a[1][2]), "+v"(S_##a[1][3]));
#define K_MFMA(S_)                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                    \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                              \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, S_##a[ks][mt]),             \
                                                                 __builtin_bit_cast(bf16x8, S_##b[ks][0]), acc[mt][0], 0, 0, 0); \
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, S_##a[ks][mt]),             \
                                                                 __builtin_bit_cast(bf16x8, S_##b[ks][1]), acc[mt][1], 0, 0, 0); \
        }                                                                                                               \
    __builtin_amdgcn_s_setprio(0);
    // stores of the epilogue issued by this wave since its last DMA: the wait for that DMA may leave them outstanding
    constexpr int kEpiStores = ((OUT & B_OUT_F32) ? 32 : 0) + ((OUT & B_OUT_BF16) ? 32 : 0);
    constexpr int kBoundaryVm = kEpiStores > 63 ? 63 : kEpiStores;
#define K_SYNC(AFTER_EPILOGUE)                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (AFTER_EPILOGUE) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kBoundaryVm) : "memory");                 \
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                                  \
    __builtin_amdgcn_sched_barrier(0);

    constexpr int kTileRows = kB;
    B_EPI_DECLS(4 * kKOperand)

    // stream position s (K tile kt of the current output tile): RB_ = its slot, WB_ = the other one.
    // Round 5: the 12 fragment reads of a half and the 8 DMA requests of a K tile are issued BETWEEN the MFMAs (one or two per
    // MFMA pair, fenced), not in a block in front of them.  The two waves of a SIMD leave every barrier together and run the same
    // stream: a block of 8 LDS-DMA requests + 12 reads in front of 16 MFMAs is a block in front of BOTH waves' MFMAs -- the matrix
    // pipe idled for its issue time (60-180 cycles per DMA piece) once per K tile (the lesson of csrc/gemm_grad.hip, where the
    // grouped schedule took the sum of matrix and memory-side time).  Same products in the same order: bit-identical results.
#define K_PAIR(S_, KS, MT)                                                                                              \
    acc[MT][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, S_##a[KS][MT]),                     \
                                                         __builtin_bit_cast(bf16x8, S_##b[KS][0]), acc[MT][0], 0, 0, 0); \
    acc[MT][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, S_##a[KS][MT]),                     \
                                                         __builtin_bit_cast(bf16x8, S_##b[KS][1]), acc[MT][1], 0, 0, 0);
#define K_F() __builtin_amdgcn_sched_barrier(0);
    // MFMAs of the X half with the reads of the Y half (k16 steps 2, 3 of slot SLOT) between them
#define K_MFMA_X_READ_Y(SLOT)                                                                                           \
    K_PAIR(x, 0, 0) K_F() K_RD(yb[0][0], b_ad2, (SLOT) * kKOperand); K_RD(yb[0][1], b_ad2, (SLOT) * kKOperand + 32 * kKRowB); K_F() \
    K_PAIR(x, 0, 1) K_F() K_RD(ya[0][0], a_ad2, (SLOT) * kKOperand); K_RD(ya[0][1], a_ad2, (SLOT) * kKOperand + 32 * kKRowB); K_F() \
    K_PAIR(x, 0, 2) K_F() K_RD(ya[0][2], a_ad2, (SLOT) * kKOperand + 64 * kKRowB); K_RD(ya[0][3], a_ad2, (SLOT) * kKOperand + 96 * kKRowB); K_F() \
    K_PAIR(x, 0, 3) K_F() K_RD(yb[1][0], b_ad3, (SLOT) * kKOperand); K_RD(yb[1][1], b_ad3, (SLOT) * kKOperand + 32 * kKRowB); K_F() \
    K_PAIR(x, 1, 0) K_F() K_RD(ya[1][0], a_ad3, (SLOT) * kKOperand); K_F()                                              \
    K_PAIR(x, 1, 1) K_F() K_RD(ya[1][1], a_ad3, (SLOT) * kKOperand + 32 * kKRowB); K_F()                                \
    K_PAIR(x, 1, 2) K_F() K_RD(ya[1][2], a_ad3, (SLOT) * kKOperand + 64 * kKRowB); K_F()                                \
    K_PAIR(x, 1, 3) K_F() K_RD(ya[1][3], a_ad3, (SLOT) * kKOperand + 96 * kKRowB); K_F()
    // one DMA piece (8 rows x 128 bytes) of operand OP (a | b), piece J of the K tile at the load cursor, into slot DSLOT
#define K_DMA(OP, J, DSLOT, OPBASE)                                                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_##OP, (lds_ptr_t)(smem + ((OPBASE) + (DSLOT)) * kKOperand + wave * 4096 + (J) * 1024), \
                                             16, ((J) & 1) ? voff_##OP##o : voff_##OP##e, ((J) * 8 * ld##OP##i + ld_k) * 2, 0, 0);
#define K_ADVANCE()                                                                                                     \
    ld_k += kKBK;                                                                                                       \
    if (ld_k == K) {                                                                                                    \
        ld_k = 0;                                                                                                       \
        ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */                         \
        K_SET_SRC()                                                                                                     \
    }
    // MFMAs of the Y half with the DMA requests of the next K tile for slot DSLOT and (RD) the reads of the X half (k16 steps 0, 1)
    // of slot RSLOT between them
#define K_MFMA_Y_ISSUE_READ_X(DSLOT, RSLOT, RD)                                                                         \
    K_PAIR(y, 0, 0) K_F() K_DMA(a, 0, DSLOT, 0) if (RD) { K_RD(xb[0][0], b_ad0, (RSLOT) * kKOperand); K_RD(xb[0][1], b_ad0, (RSLOT) * kKOperand + 32 * kKRowB); } K_F() \
    K_PAIR(y, 0, 1) K_F() K_DMA(a, 1, DSLOT, 0) if (RD) { K_RD(xa[0][0], a_ad0, (RSLOT) * kKOperand); K_RD(xa[0][1], a_ad0, (RSLOT) * kKOperand + 32 * kKRowB); } K_F() \
    K_PAIR(y, 0, 2) K_F() K_DMA(a, 2, DSLOT, 0) if (RD) { K_RD(xa[0][2], a_ad0, (RSLOT) * kKOperand + 64 * kKRowB); K_RD(xa[0][3], a_ad0, (RSLOT) * kKOperand + 96 * kKRowB); } K_F() \
    K_PAIR(y, 0, 3) K_F() K_DMA(a, 3, DSLOT, 0) if (RD) { K_RD(xb[1][0], b_ad1, (RSLOT) * kKOperand); K_RD(xb[1][1], b_ad1, (RSLOT) * kKOperand + 32 * kKRowB); } K_F() \
    K_PAIR(y, 1, 0) K_F() K_DMA(b, 0, DSLOT, 2) if (RD) { K_RD(xa[1][0], a_ad1, (RSLOT) * kKOperand); } K_F()           \
    K_PAIR(y, 1, 1) K_F() K_DMA(b, 1, DSLOT, 2) if (RD) { K_RD(xa[1][1], a_ad1, (RSLOT) * kKOperand + 32 * kKRowB); } K_F() \
    K_PAIR(y, 1, 2) K_F() K_DMA(b, 2, DSLOT, 2) if (RD) { K_RD(xa[1][2], a_ad1, (RSLOT) * kKOperand + 64 * kKRowB); } K_F() \
    K_PAIR(y, 1, 3) K_F() K_DMA(b, 3, DSLOT, 2) if (RD) { K_RD(xa[1][3], a_ad1, (RSLOT) * kKOperand + 96 * kKRowB); } K_ADVANCE() K_F()
#define K_TILE(RB_, WB_)                                                                                                \
    {                                                                                                                   \
        const bool boundary = (kt == 0 && s > 0);                                                                       \
        if (boundary) {                                                                                                 \
            B_EPILOGUE_P(false)                                                                                         \
            K_READ_X(RB_)                                                                                               \
            K_LGKM_WAIT(x)                                                                                              \
        }                                                                                                               \
        K_MFMA_X_READ_Y(RB_)                                                                                            \
        K_LGKM_WAIT(y)                                                                                                  \
        K_SYNC(boundary)                                                                                                \
        ++s;                                                                                                            \
        kt = (kt + 1 == T) ? 0 : kt + 1;                                                                                \
        K_MFMA_Y_ISSUE_READ_X(RB_, WB_, kt != 0)                                                                        \
        if (kt != 0) { K_LGKM_WAIT(x) }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }

    // prologue: K tiles 0 and 1 requested, tile 0 landed for everybody, X <- (0, steps 0-1)
    K_ISSUE(0)
    K_ISSUE(1)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    K_READ_X(0)
    K_LGKM_WAIT(x)
    int s = 0, kt = 0;
#pragma unroll 1
    while (s < S) {                                      // S is even (T is: K % 128 == 0, host)
        K_TILE(0, 1)
        K_TILE(1, 0)
    }
    B_EPILOGUE_P(false)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead DMAs must not outlive the workgroup's LDS
#undef K_TILE
#undef K_MFMA_Y_ISSUE_READ_X
#undef K_ADVANCE
#undef K_DMA
#undef K_MFMA_X_READ_Y
#undef K_F
#undef K_PAIR
#undef K_SYNC
#undef K_MFMA
#undef K_LGKM_WAIT
#undef K_READ_Y
#undef K_READ_X
#undef K_READ
#undef K_RD
#undef K_ISSUE
#undef K_SET_SRC
}


// ---------------------------------------------------------------------------------------------------------------------
// bf16 weight-gradient GEMM: dW[N,K] = A[M,N]^T . B[M,K], both operands bf16 in HBM, contraction over the rows M.
// 256 x 256 output tile, 8 waves (2 x 4, wave tile 128 x 64), 32 rows of M per step (two register sets = two steps ahead), M split over blockIdx.y (one
// workgroup per CU), deterministic reduction of the fp32 partials by the caller.  The contraction index is the ROW
// index of both operands while the MFMA wants 8 consecutive contraction elements per lane, so -- as in gemm_tn_x6_256 --
// staging interleaves row pairs into dwords (row 2r in the low half, 2r + 1 in the high half): LDS planes are
// [row pair][256 columns] dwords and a fragment is 4 conflict-free ds_read_b32.  With bf16 sources the interleave is
// 8 v_perm per 2 x 8 block and there is nothing else to compute.
constexpr int kTBM = 32;                                   // contraction rows per step (64: two register sets spill)
constexpr int kTBJ = kTBM / 32;                            // row pairs per thread, operand and step
constexpr int kTBRS = kB * 4 + 16;                         // bytes per row pair (256 dwords + pad)
constexpr int kTBPlane = (kTBM / 2) * kTBRS;               // 16 640 B per operand
constexpr int kTBBuf = 2 * kTBPlane;                       // 33 280 B; two buffers = 66 560 B

__global__ __launch_bounds__(kBThreads, 2) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const bf16_t* __restrict__ B, int64_t ldb, int64_t M,
                                                                   int N, int K, int tiles_k, int64_t rows_per_split,
                                                                   float* __restrict__ ws, float* __restrict__ ws_bias) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemt[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    // (the XCD-aware (tile, split) mapping of gemm_tn_x6_pp_kernel -- tiles of one split on one XCD, HBM fetch 1.6 -> 1.0 x
    // algorithmic there -- makes THIS kernel slower, 834-879 -> 802-821 TFLOP/s at configs[4]: with 6 x less MFMA time
    // per operand byte it lives on L2 bandwidth, and sixteen tiles pulling the same rows through one XCD's L2 is worse
    // than eight L2s each serving two of them out of the MALL; dispatch order kept)
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * kB, k0 = tk * kB;
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);          // (m_end - m_begin) % (2 kTBM) == 0 (host)
    const bool want_bias = (ws_bias != nullptr) && tk == 0;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};         // column sums of A for this thread's 8 columns

    // staging: thread = (column oct c8 of 32, row pair rp of 16); two row pairs (rp, rp + 16) per operand and step
    const int c8 = (tid & 31) * 8, rp = tid >> 5;
    const bf16_t* a_src = A + (m_begin + 2 * rp) * lda + n0 + c8;
    const bf16_t* b_src = B + (m_begin + 2 * rp) * ldb + k0 + c8;
    uint4 xa[2 * kTBJ], xb[2 * kTBJ], ya[2 * kTBJ], yb[2 * kTBJ];     // [pair j][row]: rows 2 rp (+ 32 j), + 1
#define TB_LOAD(S, MM)                                                                          \
    _Pragma("unroll") for (int j = 0; j < kTBJ; ++j) {                                             \
        S##a[2 * j] = *reinterpret_cast<const uint4*>(a_src + (int64_t)((MM) + 32 * j) * lda);       \
        S##a[2 * j + 1] = *reinterpret_cast<const uint4*>(a_src + (int64_t)((MM) + 32 * j + 1) * lda); \
        S##b[2 * j] = *reinterpret_cast<const uint4*>(b_src + (int64_t)((MM) + 32 * j) * ldb);       \
        S##b[2 * j + 1] = *reinterpret_cast<const uint4*>(b_src + (int64_t)((MM) + 32 * j + 1) * ldb); \
    }
    // dword c of the interleaved block = (row0[c] low half, row1[c] high half)
#define TB_ILV(R0, R1, LO, HI)                                                                  \
    LO = make_uint4(__builtin_amdgcn_perm(R1.x, R0.x, 0x05040100u), __builtin_amdgcn_perm(R1.x, R0.x, 0x07060302u), \
                    __builtin_amdgcn_perm(R1.y, R0.y, 0x05040100u), __builtin_amdgcn_perm(R1.y, R0.y, 0x07060302u)); \
    HI = make_uint4(__builtin_amdgcn_perm(R1.z, R0.z, 0x05040100u), __builtin_amdgcn_perm(R1.z, R0.z, 0x07060302u), \
                    __builtin_amdgcn_perm(R1.w, R0.w, 0x05040100u), __builtin_amdgcn_perm(R1.w, R0.w, 0x07060302u));
#define TB_BSUM(Q, U0, U1)                                                                      \
    bsum[2 * (Q)] += __uint_as_float((U0) << 16) + __uint_as_float((U1) << 16);                 \
    bsum[2 * (Q) + 1] += __uint_as_float((U0) & 0xFFFF0000u) + __uint_as_float((U1) & 0xFFFF0000u);
#define TB_STORE(S, BUFP)                                                                       \
    _Pragma("unroll") for (int j = 0; j < kTBJ; ++j) {                                             \
        uint4 lo_, hi_;                                                                         \
        const int o_ = (rp + 16 * j) * kTBRS + c8 * 4;                                          \
        TB_ILV(S##a[2 * j], S##a[2 * j + 1], lo_, hi_)                                          \
        *reinterpret_cast<uint4*>((BUFP) + o_) = lo_;                                           \
        *reinterpret_cast<uint4*>((BUFP) + o_ + 16) = hi_;                                      \
        if (want_bias) {                                                                        \
            TB_BSUM(0, S##a[2 * j].x, S##a[2 * j + 1].x) TB_BSUM(1, S##a[2 * j].y, S##a[2 * j + 1].y)    \
            TB_BSUM(2, S##a[2 * j].z, S##a[2 * j + 1].z) TB_BSUM(3, S##a[2 * j].w, S##a[2 * j + 1].w)    \
        }                                                                                       \
        TB_ILV(S##b[2 * j], S##b[2 * j + 1], lo_, hi_)                                          \
        *reinterpret_cast<uint4*>((BUFP) + kTBPlane + o_) = lo_;                                \
        *reinterpret_cast<uint4*>((BUFP) + kTBPlane + o_ + 16) = hi_;                           \
    }
#define TB_FRAG(DST, BASE)                                                               \
    {                                                                                    \
        uint4 u_;                                                                        \
        u_.x = *reinterpret_cast<const uint32_t*>(BASE);                                 \
        u_.y = *reinterpret_cast<const uint32_t*>((BASE) + kTBRS);                       \
        u_.z = *reinterpret_cast<const uint32_t*>((BASE) + 2 * kTBRS);                   \
        u_.w = *reinterpret_cast<const uint32_t*>((BASE) + 3 * kTBRS);                   \
        DST = __builtin_bit_cast(bf16x8, u_);                                            \
    }
    // k16 step ks of the 64-row stage: row pairs 8 ks + 4 kh + 0..3
#define TB_COMPUTE(BUFP)                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < kTBM / 16; ++ks) {                                                            \
        const unsigned char* ab_ = (BUFP) + (8 * ks + 4 * kh) * kTBRS + (wm * 128 + li) * 4;                              \
        const unsigned char* bb_ = (BUFP) + kTBPlane + (8 * ks + 4 * kh) * kTBRS + (wn * 64 + li) * 4;                    \
        bf16x8 a_[4], b_[2];                                                                                              \
        _Pragma("unroll") for (int tl = 0; tl < 2; ++tl) TB_FRAG(b_[tl], bb_ + tl * 128)                                  \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) TB_FRAG(a_[mt], ab_ + mt * 128)                                  \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                                \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[mt], b_[0], acc[mt][0], 0, 0, 0);                     \
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[mt], b_[1], acc[mt][1], 0, 0, 0);                     \
        }                                                                                                                 \
        if (ks & 1) __builtin_amdgcn_sched_barrier(0);     /* at most two k16 steps of fragments (48 VGPRs) in flight */  \
    }

    unsigned char* const buf0 = smemt;
    unsigned char* const buf1 = smemt + kTBBuf;
    const int64_t rows = m_end - m_begin;                     // multiple of 2 kTBM, >= 2 kTBM when non-empty
    if (rows > 0) {
        TB_LOAD(x, 0)
        TB_LOAD(y, kTBM)
        TB_STORE(x, buf0)
        __syncthreads();
        for (int64_t mm = 0; mm < rows - 2 * kTBM; mm += 2 * kTBM) {
            TB_LOAD(x, mm + 2 * kTBM)
            __builtin_amdgcn_sched_barrier(0);
            TB_COMPUTE(buf0)
            __builtin_amdgcn_sched_barrier(0);
            TB_STORE(y, buf1)
            __syncthreads();
            TB_LOAD(y, mm + 3 * kTBM)
            __builtin_amdgcn_sched_barrier(0);
            TB_COMPUTE(buf1)
            __builtin_amdgcn_sched_barrier(0);
            TB_STORE(x, buf0)
            __syncthreads();
        }
        TB_COMPUTE(buf0)
        __builtin_amdgcn_sched_barrier(0);
        TB_STORE(y, buf1)
        __syncthreads();
        TB_COMPUTE(buf1)
    }
#undef TB_LOAD
#undef TB_ILV
#undef TB_STORE
#undef TB_BSUM
#undef TB_FRAG
#undef TB_COMPUTE

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
        float* red = reinterpret_cast<float*>(smemt);          // [16 row-pair groups][256 columns]
#pragma unroll
        for (int q = 0; q < 8; ++q) red[rp * kB + c8 + q] = bsum[q];
        __syncthreads();
        if (tid < kB) {
            float tot = 0.0f;
#pragma unroll
            for (int g = 0; g < 16; ++g) tot += red[g * kB + tid];
            ws_bias[(int64_t)blockIdx.y * N + n0 + tid] = tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_tn_bf16_tr_kernel: the same weight gradient with operands delivered global -> LDS by DMA in their natural row-major
// form and the fragments read with ds_read_b64_tr_b16 (gfx950's transposing LDS read).
//
// The kernel above needs 8 consecutive CONTRACTION elements (rows m) per lane while memory holds rows: it interleaves row
// pairs with v_perm in registers, writes them to LDS and reads a fragment as four ds_read_b32 -- 48 LDS reads per wave and
// 16 MFMAs at 128 B/clk plus the staging writes: the LDS, not the matrix pipe, paces it (0.33 of the bf16 peak).  Here:
//   * a slot = 64 rows x 256 columns (512 bytes per row) of A and of B, two slots per operand (128 KB), filled by
//     buffer_load_dwordx4 ... lds: whole 128-byte lines per request, no staging registers, no v_perm, no LDS writes;
//   * ds_read_b64_tr_b16 (measured semantics, tools/micro/tr_probe.hip): within a 16-lane group, source lane j = 4 e + c
//     supplies the address of 4 consecutive bf16, and lane i = 4 c + k receives element k of source lanes c, 4 + c, 8 + c,
//     12 + c (e = 0..3).  With source lane (e, c) pointing at X[m0 + e][n0 + 4 c ..] the group's lane i holds column n0 + i
//     for rows m0 .. m0 + 3: two such reads are the 8 contraction elements of a 32 x 32 x 16 MFMA fragment (lane groups
//     0 / 1 = columns 0-15 / 16-31 at rows m0 .. m0 + 7, groups 2 / 3 the same columns at rows m0 + 8 .. m0 + 15), at
//     256 B/clk and half the instruction count of the ds_read_b32 form;
//   * 16-byte chunk c of row r sits at chunk position c ^ (4 (r & 3)) (applied to the DMA's per-lane SOURCE address): the four
//     rows of a read's 32-lane half then touch four different 64-byte segments of the 256-byte bank row;
//   * the schedule of gemm_nt_bf16_k64_kernel: fragments of a slot in two 48-register halves (k16 steps 0-1 / 2-3), the next
//     half's reads and the next slot's DMA under 16 MFMAs of the same wave, ONE barrier per 64 rows.
// The bias gradient (column sums of A) comes from the A fragments of the waves with wn == 0.
constexpr int kTRRows = 64;                         // contraction rows per slot
constexpr int kTRRowB = kB * 2;                     // 512 bytes per operand row
constexpr int kTROperand = kTRRows * kTRRowB;       // 32 KB
constexpr int kTRLds = 4 * kTROperand;              // A slot 0 | A slot 1 | B slot 0 | B slot 1

__global__ __launch_bounds__(kBThreads, 2) void gemm_tn_bf16_tr_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                      const bf16_t* __restrict__ B, int64_t ldb, int64_t M,
                                                                      int N, int K, int tiles_k, int64_t rows_per_split,
                                                                      float* __restrict__ ws, float* __restrict__ ws_bias) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * kB, k0 = tk * kB;
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);           // (m_end - m_begin) % 128 == 0 (host)
    const int S = m_end > m_begin ? (int)((m_end - m_begin) / kTRRows) : 0;      // slots of this split: even
    const bool want_bias = (ws_bias != nullptr) && tk == 0 && wn == 0;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};                             // column sums of A: column 128 wm + 32 mt + li, rows of this lane's kh

    // ---- DMA: wave w delivers rows [8 w, 8 w + 8) of a slot of A and of B: 4 + 4 instructions of 2 rows x 512 bytes ----
    // lane l -> row l >> 5 of the 2-row block j, chunk position l & 31 <- logical chunk (l & 31) ^ (4 ((2 j + (l >> 5)) & 3))
    const int dl_row = lane >> 5;
    const int dl_ce = (lane & 31) ^ (4 * dl_row), dl_co = (lane & 31) ^ (4 * (2 + dl_row));
    const int ldai = (int)lda, ldbi = (int)ldb;
    const int voff_ae = (dl_row * ldai + dl_ce * 8) * 2, voff_ao = (dl_row * ldai + dl_co * 8) * 2;
    const int voff_be = (dl_row * ldbi + dl_ce * 8) * 2, voff_bo = (dl_row * ldbi + dl_co * 8) * 2;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(A + (m_begin + 8 * wave) * lda + n0), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(B + (m_begin + 8 * wave) * ldb + k0), 0, 0x7FFFFFFF, 0x00020000);
    int ld_slot = 0;                                                  // next slot to request (clamped to the last one)
#define R_ISSUE(SLOT)                                                                                                   \
    {                                                                                                                   \
        unsigned char* da_ = smem + (SLOT) * kTROperand + wave * 4096;                                                  \
        unsigned char* db_ = smem + (2 + (SLOT)) * kTROperand + wave * 4096;                                            \
        const int row0_ = min(ld_slot, S - 1) * kTRRows;                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(da_ + j * 1024), 16, (j & 1) ? voff_ao : voff_ae, \
                                                     (row0_ + 2 * j) * ldai * 2, 0, 0);                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(db_ + j * 1024), 16, (j & 1) ? voff_bo : voff_be, \
                                                     (row0_ + 2 * j) * ldbi * 2, 0, 0);                                 \
        ++ld_slot;                                                                                                      \
    }

    // ---- fragment addressing: lane = (group g = lane >> 4: column half g & 1, row half kh = g >> 1; source role e = (lane >> 2) & 3
    // (row m0 + e), c = lane & 3 (columns 4 c ..)) ----
    const unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int te = (lane >> 2) & 3, tc = lane & 3, tg = (lane >> 4) & 1;
    const unsigned row_off = (unsigned)((8 * kh + te) * kTRRowB + (tc & 1) * 8);
    unsigned a_ad[4], b_ad[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        a_ad[mt] = ldsb + row_off + ((unsigned)((16 * wm + 4 * mt + 2 * tg + (tc >> 1)) ^ (4 * te)) << 4);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        b_ad[nt] = ldsb + 2 * kTROperand + row_off + ((unsigned)((8 * wn + 4 * nt + 2 * tg + (tc >> 1)) ^ (4 * te)) << 4);
    uint2 xa[2][4][2], xb[2][2][2], ya[2][4][2], yb[2][2][2];         // [k16 step of the half][tile][rows 0-3 | 4-7]
#define R_RD(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
    // k16 step KS (0..3) of slot SLOT: rows 16 KS + 8 kh + 4 r + e
#define R_READ_STEP(S_, H, SLOT, KS)                                                                                    \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                                  \
        R_RD(S_##b[H][nt][0], b_ad[nt], (SLOT) * kTROperand + (16 * (KS)) * kTRRowB);                                   \
        R_RD(S_##b[H][nt][1], b_ad[nt], (SLOT) * kTROperand + (16 * (KS) + 4) * kTRRowB);                               \
    }                                                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                                  \
        R_RD(S_##a[H][mt][0], a_ad[mt], (SLOT) * kTROperand + (16 * (KS)) * kTRRowB);                                   \
        R_RD(S_##a[H][mt][1], a_ad[mt], (SLOT) * kTROperand + (16 * (KS) + 4) * kTRRowB);                               \
    }
#define R_READ_X(SLOT) R_READ_STEP(x, 0, SLOT, 0) R_READ_STEP(x, 1, SLOT, 1)
#define R_READ_Y(SLOT) R_READ_STEP(y, 0, SLOT, 2) R_READ_STEP(y, 1, SLOT, 3)
#define R_W2(V) "+v"(V[0]), "+v"(V[1])
#define R_LGKM_WAIT(S_)                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                 : R_W2(S_##b[0][0]), R_W2(S_##b[0][1]), R_W2(S_##b[1][0]), R_W2(S_##b[1][1]), R_W2(S_##a[0][0]),       \
                   R_W2(S_##a[0][1]), R_W2(S_##a[0][2]), R_W2(S_##a[0][3]), R_W2(S_##a[1][0]), R_W2(S_##a[1][1]),       \
                   R_W2(S_##a[1][2]), R_W2(S_##a[1][3]));
#define R_FRAG(V) __builtin_bit_cast(bf16x8, make_uint4(V[0].x, V[0].y, V[1].x, V[1].y))
#define R_BSUM1(U) (__uint_as_float((U) << 16) + __uint_as_float((U) & 0xFFFF0000u))
#define R_MFMA(S_)                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                                      \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                       \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                              \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R_FRAG(S_##a[h][mt]), R_FRAG(S_##b[h][0]), acc[mt][0], 0, 0, 0); \
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R_FRAG(S_##a[h][mt]), R_FRAG(S_##b[h][1]), acc[mt][1], 0, 0, 0); \
        }                                                                                                               \
    __builtin_amdgcn_s_setprio(0);                                                                                      \
    if (want_bias) {                                                                                                    \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                            \
                bsum[mt] += (R_BSUM1(S_##a[h][mt][0].x) + R_BSUM1(S_##a[h][mt][0].y)) +                                 \
                            (R_BSUM1(S_##a[h][mt][1].x) + R_BSUM1(S_##a[h][mt][1].y));                                  \
    }
#define R_SYNC()                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                                       \
    __builtin_amdgcn_sched_barrier(0);
    // Round 5 (as gemm_nt_bf16_k64_kernel above): the 24 transposing reads of a half and the 8 DMA requests of a slot are issued
    // BETWEEN the MFMAs (three reads, one request per MFMA pair, fenced) instead of in a block in front of both waves' MFMAs.
    // Same products in the same order: bit-identical partial sums.
#define R_PAIR(S_, H, MT)                                                                                               \
    acc[MT][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R_FRAG(S_##a[H][MT]), R_FRAG(S_##b[H][0]), acc[MT][0], 0, 0, 0); \
    acc[MT][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R_FRAG(S_##a[H][MT]), R_FRAG(S_##b[H][1]), acc[MT][1], 0, 0, 0);
#define R_F() __builtin_amdgcn_sched_barrier(0);
#define R_DMA(OP, J, DSLOT, OPBASE)                                                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_##OP, (lds_ptr_t)(smem + ((OPBASE) + (DSLOT)) * kTROperand + wave * 4096 + (J) * 1024), \
                                             16, ((J) & 1) ? voff_##OP##o : voff_##OP##e,                              \
                                             (min(ld_slot, S - 1) * kTRRows + 2 * (J)) * ld##OP##i * 2, 0, 0);
#define R_BIAS(S_)                                                                                                      \
    if (want_bias) {                                                                                                    \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                            \
                bsum[mt] += (R_BSUM1(S_##a[h][mt][0].x) + R_BSUM1(S_##a[h][mt][0].y)) +                                 \
                            (R_BSUM1(S_##a[h][mt][1].x) + R_BSUM1(S_##a[h][mt][1].y));                                  \
    }
#define R_MFMA_X_READ_Y(SLOT)                                                                                           \
    R_PAIR(x, 0, 0) R_F() R_RD(yb[0][0][0], b_ad[0], (SLOT) * kTROperand + (16 * (2)) * kTRRowB); R_RD(yb[0][0][1], b_ad[0], (SLOT) * kTROperand + (16 * (2) + 4) * kTRRowB); R_RD(yb[0][1][0], b_ad[1], (SLOT) * kTROperand + (16 * (2)) * kTRRowB); R_F() \
    R_PAIR(x, 0, 1) R_F() R_RD(yb[0][1][1], b_ad[1], (SLOT) * kTROperand + (16 * (2) + 4) * kTRRowB); R_RD(ya[0][0][0], a_ad[0], (SLOT) * kTROperand + (16 * (2)) * kTRRowB); R_RD(ya[0][0][1], a_ad[0], (SLOT) * kTROperand + (16 * (2) + 4) * kTRRowB); R_F() \
    R_PAIR(x, 0, 2) R_F() R_RD(ya[0][1][0], a_ad[1], (SLOT) * kTROperand + (16 * (2)) * kTRRowB); R_RD(ya[0][1][1], a_ad[1], (SLOT) * kTROperand + (16 * (2) + 4) * kTRRowB); R_RD(ya[0][2][0], a_ad[2], (SLOT) * kTROperand + (16 * (2)) * kTRRowB); R_F() \
    R_PAIR(x, 0, 3) R_F() R_RD(ya[0][2][1], a_ad[2], (SLOT) * kTROperand + (16 * (2) + 4) * kTRRowB); R_RD(ya[0][3][0], a_ad[3], (SLOT) * kTROperand + (16 * (2)) * kTRRowB); R_RD(ya[0][3][1], a_ad[3], (SLOT) * kTROperand + (16 * (2) + 4) * kTRRowB); R_F() \
    R_PAIR(x, 1, 0) R_F() R_RD(yb[1][0][0], b_ad[0], (SLOT) * kTROperand + (16 * (3)) * kTRRowB); R_RD(yb[1][0][1], b_ad[0], (SLOT) * kTROperand + (16 * (3) + 4) * kTRRowB); R_RD(yb[1][1][0], b_ad[1], (SLOT) * kTROperand + (16 * (3)) * kTRRowB); R_F() \
    R_PAIR(x, 1, 1) R_F() R_RD(yb[1][1][1], b_ad[1], (SLOT) * kTROperand + (16 * (3) + 4) * kTRRowB); R_RD(ya[1][0][0], a_ad[0], (SLOT) * kTROperand + (16 * (3)) * kTRRowB); R_RD(ya[1][0][1], a_ad[0], (SLOT) * kTROperand + (16 * (3) + 4) * kTRRowB); R_F() \
    R_PAIR(x, 1, 2) R_F() R_RD(ya[1][1][0], a_ad[1], (SLOT) * kTROperand + (16 * (3)) * kTRRowB); R_RD(ya[1][1][1], a_ad[1], (SLOT) * kTROperand + (16 * (3) + 4) * kTRRowB); R_RD(ya[1][2][0], a_ad[2], (SLOT) * kTROperand + (16 * (3)) * kTRRowB); R_F() \
    R_PAIR(x, 1, 3) R_F() R_RD(ya[1][2][1], a_ad[2], (SLOT) * kTROperand + (16 * (3) + 4) * kTRRowB); R_RD(ya[1][3][0], a_ad[3], (SLOT) * kTROperand + (16 * (3)) * kTRRowB); R_RD(ya[1][3][1], a_ad[3], (SLOT) * kTROperand + (16 * (3) + 4) * kTRRowB); R_F() \
    R_BIAS(x)
#define R_MFMA_Y_ISSUE_READ_X(DSLOT, RSLOT)                                                                             \
    R_PAIR(y, 0, 0) R_F() R_DMA(a, 0, DSLOT, 0) R_RD(xb[0][0][0], b_ad[0], (RSLOT) * kTROperand + (16 * (0)) * kTRRowB); R_RD(xb[0][0][1], b_ad[0], (RSLOT) * kTROperand + (16 * (0) + 4) * kTRRowB); R_RD(xb[0][1][0], b_ad[1], (RSLOT) * kTROperand + (16 * (0)) * kTRRowB); R_F() \
    R_PAIR(y, 0, 1) R_F() R_DMA(a, 1, DSLOT, 0) R_RD(xb[0][1][1], b_ad[1], (RSLOT) * kTROperand + (16 * (0) + 4) * kTRRowB); R_RD(xa[0][0][0], a_ad[0], (RSLOT) * kTROperand + (16 * (0)) * kTRRowB); R_RD(xa[0][0][1], a_ad[0], (RSLOT) * kTROperand + (16 * (0) + 4) * kTRRowB); R_F() \
    R_PAIR(y, 0, 2) R_F() R_DMA(a, 2, DSLOT, 0) R_RD(xa[0][1][0], a_ad[1], (RSLOT) * kTROperand + (16 * (0)) * kTRRowB); R_RD(xa[0][1][1], a_ad[1], (RSLOT) * kTROperand + (16 * (0) + 4) * kTRRowB); R_RD(xa[0][2][0], a_ad[2], (RSLOT) * kTROperand + (16 * (0)) * kTRRowB); R_F() \
    R_PAIR(y, 0, 3) R_F() R_DMA(a, 3, DSLOT, 0) R_RD(xa[0][2][1], a_ad[2], (RSLOT) * kTROperand + (16 * (0) + 4) * kTRRowB); R_RD(xa[0][3][0], a_ad[3], (RSLOT) * kTROperand + (16 * (0)) * kTRRowB); R_RD(xa[0][3][1], a_ad[3], (RSLOT) * kTROperand + (16 * (0) + 4) * kTRRowB); R_F() \
    R_PAIR(y, 1, 0) R_F() R_DMA(b, 0, DSLOT, 2) R_RD(xb[1][0][0], b_ad[0], (RSLOT) * kTROperand + (16 * (1)) * kTRRowB); R_RD(xb[1][0][1], b_ad[0], (RSLOT) * kTROperand + (16 * (1) + 4) * kTRRowB); R_RD(xb[1][1][0], b_ad[1], (RSLOT) * kTROperand + (16 * (1)) * kTRRowB); R_F() \
    R_PAIR(y, 1, 1) R_F() R_DMA(b, 1, DSLOT, 2) R_RD(xb[1][1][1], b_ad[1], (RSLOT) * kTROperand + (16 * (1) + 4) * kTRRowB); R_RD(xa[1][0][0], a_ad[0], (RSLOT) * kTROperand + (16 * (1)) * kTRRowB); R_RD(xa[1][0][1], a_ad[0], (RSLOT) * kTROperand + (16 * (1) + 4) * kTRRowB); R_F() \
    R_PAIR(y, 1, 2) R_F() R_DMA(b, 2, DSLOT, 2) R_RD(xa[1][1][0], a_ad[1], (RSLOT) * kTROperand + (16 * (1)) * kTRRowB); R_RD(xa[1][1][1], a_ad[1], (RSLOT) * kTROperand + (16 * (1) + 4) * kTRRowB); R_RD(xa[1][2][0], a_ad[2], (RSLOT) * kTROperand + (16 * (1)) * kTRRowB); R_F() \
    R_PAIR(y, 1, 3) R_F() R_DMA(b, 3, DSLOT, 2) R_RD(xa[1][2][1], a_ad[2], (RSLOT) * kTROperand + (16 * (1) + 4) * kTRRowB); R_RD(xa[1][3][0], a_ad[3], (RSLOT) * kTROperand + (16 * (1)) * kTRRowB); R_RD(xa[1][3][1], a_ad[3], (RSLOT) * kTROperand + (16 * (1) + 4) * kTRRowB); R_F() \
    ++ld_slot;                                                                                                          \
    R_BIAS(y)
    // slot s (RB_ = its buffer, WB_ = the other one)
#define R_SLOT(RB_, WB_)                                                                                                \
    {                                                                                                                   \
        R_MFMA_X_READ_Y(RB_)                                                                                            \
        R_LGKM_WAIT(y)                                                                                                  \
        R_SYNC()                        /* own DMA of slot s+1 landed, everybody's after the barrier; slot s fully read */ \
        R_MFMA_Y_ISSUE_READ_X(RB_, WB_) /* slot s+2 requested, first half of slot s+1 read */                           \
        R_LGKM_WAIT(x)                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }

    if (S > 0) {
        R_ISSUE(0)
        R_ISSUE(1)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        R_READ_X(0)
        R_LGKM_WAIT(x)
#pragma unroll 1
        for (int s = 0; s < S; s += 2) {
            R_SLOT(0, 1)
            R_SLOT(1, 0)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped run-ahead DMAs must not outlive the workgroup's LDS
    }
#undef R_SLOT
#undef R_MFMA_Y_ISSUE_READ_X
#undef R_MFMA_X_READ_Y
#undef R_BIAS
#undef R_DMA
#undef R_F
#undef R_PAIR
#undef R_SYNC
#undef R_MFMA
#undef R_BSUM1
#undef R_FRAG
#undef R_LGKM_WAIT
#undef R_W2
#undef R_READ_Y
#undef R_READ_X
#undef R_READ_STEP
#undef R_RD
#undef R_ISSUE

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
    if ((ws_bias != nullptr) && tk == 0 && wn == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float tot = bsum[mt] + __shfl_xor(bsum[mt], 32, 64);
            if (kh == 0) ws_bias[(int64_t)blockIdx.y * N + n0 + wm * 128 + mt * 32 + li] = tot;
        }
    }
}

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_cast_bf16(const float* in, int64_t ld_in, void* out, int64_t rows, int cols, void* stream) {
    if (rows == 0) return VQCPC_OK;
    VQ_REQUIRE(in && out && rows > 0 && cols > 0 && cols % 4 == 0 && ld_in >= cols && ld_in % 4 == 0,
               "cast_bf16: bad arguments (cols, ld_in multiples of 4)");
    VQ_REQUIRE(aligned16(in) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, "cast_bf16: alignment");
    const int blocks = (int)std::min<int64_t>(ceil_div(rows * (cols / 4), 256), 8192);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ld_in, (bf16_t*)out, rows, cols);
    VQ_CHECK_LAUNCH("cast_bf16");
    return VQCPC_OK;
}

// 1: K tiles of 64 by LDS-DMA (gemm_nt_bf16_k64_kernel, K % 128 == 0; the transposed-read weight-gradient kernel); 0: the
// ping-pong kernels, K tiles of 32 (what shapes the DMA kernels do not take still run on).  The A/B SWITCH between them is a
// lab-build facility (round 5): the product library always prefers variant 1.
#if VQCPC_LAB
static std::atomic<int> g_bf16_nt_variant{1};
static inline int bf16_variant() { return g_bf16_nt_variant.load(std::memory_order_relaxed); }
int vqcpc_gemm_bf16_set_variant(int variant) {
    VQ_REQUIRE(variant == 0 || variant == 1, "gemm_bf16_set_variant: 0 (ping-pong kernel, K tiles of 32) or 1 (K tiles of 64 by LDS-DMA)");
    g_bf16_nt_variant.store(variant, std::memory_order_relaxed);
    return VQCPC_OK;
}
#else
static constexpr int bf16_variant() { return 1; }
#endif

int vqcpc_gemm_nt_bf16_supported(int64_t M, int N, int K) { return (M % kB == 0 && N % kB == 0 && K % (2 * kBBK) == 0) ? 1 : 0; }

int vqcpc_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, void* Cb, int64_t ldcb,
                       int64_t M, int N, int K, const float* bias, int act, float drop_p, uint64_t seed, const float* gate,
                       int64_t ldgate, const void* gate_bf16, int64_t ldgate_bf16, float gate_scale, const float* add,
                       int64_t ldadd, const void* add_bf16, int64_t ldadd_bf16, void* stream) {
    if (M == 0) return VQCPC_OK;
    VQ_REQUIRE(A && B && (C || Cb), "gemm_nt_bf16: null pointer");
    VQ_REQUIRE(vqcpc_gemm_nt_bf16_supported(M, N, K), "gemm_nt_bf16: M, N must be multiples of 256 and K of 64 (M=%lld N=%d K=%d)",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && (!C || ldc >= N) && (!Cb || (ldcb >= N && ldcb % 4 == 0)),
               "gemm_nt_bf16: bad leading dimensions");
    VQ_REQUIRE(aligned16(A) && aligned16(B) && (!C || aligned16(C)) && (reinterpret_cast<uintptr_t>(Cb) & 7u) == 0,
               "gemm_nt_bf16: alignment");
    VQ_REQUIRE(!(gate && gate_bf16), "gemm_nt_bf16: one gate operand only");
    VQ_REQUIRE(!(add && add_bf16) && !(add_bf16 && (gate || gate_bf16)), "gemm_nt_bf16: one residual operand, and no gate beside a bf16 one");
    VQ_REQUIRE(!add_bf16 || (ldadd_bf16 >= N && ldadd_bf16 % 4 == 0 && (reinterpret_cast<uintptr_t>(add_bf16) & 7u) == 0),
               "gemm_nt_bf16: bf16 residual operand: alignment / leading dimension");
    VQ_REQUIRE(act == 0 || act == 1, "gemm_nt_bf16: act must be 0 or 1");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm_nt_bf16: bad dropout probability");
    EpiParams ep{bias, act, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, gate, ldgate, gate_scale, add, ldadd, nullptr, 0, 0};
    static const int stagger_per_ktile = lab_env_int("VQCPC_BF16_STAGGER", 0);
    const int tiles_total = (int)((M / kB) * (N / kB));
    Bf16Out o{C, ldc, (bf16_t*)Cb, ldcb, (const bf16_t*)(add_bf16 ? add_bf16 : gate_bf16), add_bf16 ? ldadd_bf16 : ldgate_bf16,
              stagger_per_ktile < 0 ? stagger_per_ktile : (tiles_total >= 4 * kNumCU ? stagger_per_ktile * (K / kBBK) : 0)};
    const bool has_gate = gate || gate_bf16;
    const int flags = (bias ? E_BIAS : 0) | (act == 1 ? E_RELU : 0) | (ep.thr ? E_DROP : 0) | (has_gate ? E_GATE : 0) |
                      ((add || add_bf16) ? E_ADD : 0);
    const int out = (C ? B_OUT_F32 : 0) | (Cb ? B_OUT_BF16 : 0) | (gate_bf16 ? B_GATE_BF16 : 0) | (add_bf16 ? B_ADD_BF16 : 0);
    const int tn = N / kB;
    const int tiles = (int)((M / kB) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kBThreads);
    hipStream_t st = (hipStream_t)stream;
    const bool k64 = bf16_variant() == 1 && K % (2 * kKBK) == 0;
#define BL(EPIV, OUTV)                                                                                                 \
    if (flags == (EPIV) && out == (OUTV) && k64) {                                                                     \
        static bool attr_k64 = false;                                                                                  \
        if (!attr_k64) {                                                                                               \
            (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_k64_kernel<EPIV, OUTV>,                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kKLds);                              \
            attr_k64 = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_nt_bf16_k64_kernel<EPIV, OUTV>), grid, block, kKLds, st, (const bf16_t*)A, lda,       \
                           (const bf16_t*)B, ldb, o, M, N, K, tn, tiles, ep);                                          \
        VQ_CHECK_LAUNCH("gemm_nt_bf16_k64");                                                                           \
        return VQCPC_OK;                                                                                               \
    }                                                                                                                  \
    if (flags == (EPIV) && out == (OUTV)) {                                                                            \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<EPIV, OUTV>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kBLds);                              \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<EPIV, OUTV>), grid, block, kBLds, st, (const bf16_t*)A, lda,           \
                           (const bf16_t*)B, ldb, o, M, N, K, tn, tiles, ep);                                          \
        VQ_CHECK_LAUNCH("gemm_nt_bf16");                                                                               \
        return VQCPC_OK;                                                                                               \
    }
    // the combinations the training step uses
    BL(0, B_OUT_F32)
    BL(E_BIAS, B_OUT_F32)
    BL(E_BIAS, B_OUT_F32 | B_OUT_BF16)
    BL(E_BIAS | E_RELU, B_OUT_BF16)
    BL(E_BIAS | E_RELU | E_DROP, B_OUT_BF16)
    BL(E_BIAS | E_RELU, B_OUT_F32)
    BL(E_BIAS | E_RELU, B_OUT_F32 | B_OUT_BF16)
    BL(E_BIAS | E_RELU | E_DROP, B_OUT_F32 | B_OUT_BF16)
    BL(E_GATE, B_OUT_F32 | B_OUT_BF16 | B_GATE_BF16)
    BL(E_BIAS | E_RELU | E_DROP, B_OUT_F32)
    BL(E_GATE, B_OUT_F32)
    BL(E_GATE, B_OUT_BF16 | B_GATE_BF16)
    BL(E_GATE, B_OUT_F32 | B_GATE_BF16)
    BL(E_ADD, B_OUT_F32)
    BL(E_ADD, B_OUT_F32 | B_ADD_BF16)                // input gradient + the bf16 gradient of the residual branch (round 5)
    BL(E_ADD, B_OUT_BF16 | B_ADD_BF16)               // ... and the sum itself in bf16: the main-stream gradient between sub-layers
    BL(0, B_OUT_BF16)
    BL(E_BIAS, B_OUT_BF16)                           // in_proj output for the all-bf16 attention kernels
    BL(E_BIAS | E_ADD, B_OUT_F32)                    // residual sums for LayerNorm (see gemm.hip)
    BL(E_BIAS | E_DROP | E_ADD, B_OUT_F32)
    BL(E_BIAS | E_ADD, B_OUT_F32 | B_ADD_BF16)       // the same with the residual read from the LayerNorm's bf16 output (round 5)
    BL(E_BIAS | E_DROP | E_ADD, B_OUT_F32 | B_ADD_BF16)
    BL(E_BIAS | E_ADD, B_OUT_BF16)                   // ... and the residual sum itself written in bf16 (the LayerNorm kernels read
    BL(E_BIAS | E_DROP | E_ADD, B_OUT_BF16)          //     it so: vqcpc_layernorm_fwd_xb16 / _bwd_xb16)
    BL(E_BIAS | E_ADD, B_OUT_BF16 | B_ADD_BF16)
    BL(E_BIAS | E_DROP | E_ADD, B_OUT_BF16 | B_ADD_BF16)
#undef BL
    set_error("gemm_nt_bf16: unsupported epilogue / output combination (flags %d, out %d)", flags, out);
    return VQCPC_EINVAL;
}

int vqcpc_gemm_tn_bf16_supported(int64_t M, int N, int K) { return (M % 128 == 0 && N % kB == 0 && K % kB == 0) ? 1 : 0; }

static int tn_bf16_splits(int64_t M, int N, int K) {
    const int64_t tiles = (int64_t)(N / kB) * (K / kB);
    int64_t s = std::max<int64_t>(1, kNumCU / tiles);
    return (int)std::min<int64_t>(s, std::max<int64_t>(1, M / 512));
}

int64_t vqcpc_gemm_tn_bf16_workspace(int64_t M, int N, int K) {
    return (int64_t)tn_bf16_splits(std::max<int64_t>(M, 1), N, K) * ((int64_t)N * K + N) * (int64_t)sizeof(float);
}

int vqcpc_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* dW, float* db, int64_t M, int N, int K,
                       int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(A && B && dW && workspace, "gemm_tn_bf16: null pointer");
    VQ_REQUIRE(vqcpc_gemm_tn_bf16_supported(M, N, K), "gemm_tn_bf16: M must be a multiple of 128, N and K of 256");
    VQ_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= N && ldb >= K && aligned16(A) && aligned16(B),
               "gemm_tn_bf16: bad leading dimensions / alignment");
    if (workspace_bytes < vqcpc_gemm_tn_bf16_workspace(M, N, K)) {
        set_error("gemm_tn_bf16: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const int splits = tn_bf16_splits(M, N, K);
    float* ws = (float*)workspace;
    float* ws_bias = db ? ws + (int64_t)splits * N * K : nullptr;
    hipStream_t s = (hipStream_t)stream;
    // transposed-read kernel: splits of whole 128-row pairs of slots.  Its DMA offsets (row within the split) * ld * 2 are
    // 32-bit against a 2 GB buffer resource based at the split's first row: taken only when a split's rows fit that range
    const int64_t rps = round_up(ceil_div(M, splits), 2 * kTRRows);
    if (bf16_variant() == 1 && lda < (1 << 20) && ldb < (1 << 20) &&
        (rps + 2 * kTRRows) * std::max(lda, ldb) * 2 < ((int64_t)1 << 31)) {
        static bool attr_tr = false;
        if (!attr_tr) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kTRLds);
            attr_tr = true;
        }
        const int tkr = K / kB;
        hipLaunchKernelGGL(gemm_tn_bf16_tr_kernel, dim3((N / kB) * tkr, splits), dim3(kBThreads), kTRLds, s, (const bf16_t*)A, lda,
                           (const bf16_t*)B, ldb, M, N, K, tkr, rps, ws, ws_bias);
        VQ_CHECK_LAUNCH("gemm_tn_bf16_tr");
        if (accumulate == 2) return VQCPC_OK;       // deferred reduction: see vqcpc_gemm_tn_bf16_deferred_splits
        return launch_reduce_splits2(ws, (int64_t)N * K, splits, dW, (int64_t)N * K, ws_bias, N, db, db ? N : 0, accumulate, s);
    }
    const int64_t rows_per_split = round_up(ceil_div(M, splits), 2 * kTBM);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kTBBuf);
        attr_done = true;
    }
    const int tk2 = K / kB;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((N / kB) * tk2, splits), dim3(kBThreads), 2 * kTBBuf, s, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, M, N, K, tk2, rows_per_split, ws, ws_bias);
    VQ_CHECK_LAUNCH("gemm_tn_bf16");
    if (accumulate == 2) return VQCPC_OK;
    return launch_reduce_splits2(ws, (int64_t)N * K, splits, dW, (int64_t)N * K, ws_bias, N, db, db ? N : 0, accumulate, s);
}

// as vqcpc_gemm_tn_deferred_splits, for vqcpc_gemm_tn_bf16 (accumulate == 2 leaves the partial sums in the workspace)
int vqcpc_gemm_tn_bf16_deferred_splits(int64_t M, int N, int K) {
    if (!vqcpc_gemm_tn_bf16_supported(M, N, K)) return 0;
    const int splits = tn_bf16_splits(M, N, K);
    return ((int64_t)N * K >= (1 << 16) && splits <= 64) ? splits : 0;
}

}  // extern "C"
