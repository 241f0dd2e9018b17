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
enum { B_OUT_F32 = 1, B_OUT_BF16 = 2, B_GATE_BF16 = 4 };

struct Bf16Out {
    float* c;            // fp32 output (B_OUT_F32)
    int64_t ldc;
    bf16_t* cb;          // bf16 output (B_OUT_BF16), dense or strided
    int64_t ldcb;
    const bf16_t* gate_b;   // bf16 gate operand (B_GATE_BF16): only its sign is used
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
    constexpr bool AUX_B16 = (OUT & B_GATE_BF16) != 0;             /* gate operand is bf16 (sign only) */              \
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
                if (EPI & E_ADD) v += AUX.f[j][c];                                                                     \
                ov[c] = v;                                                                                             \
            }                                                                                                          \
            if ((OUT & B_OUT_F32) && o.stagger != -1)                                                                  \
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(ov), "v"(voff_c), \
                             "s"(rc), "s"(((MT * 32 + 8 * j) * ldci + NT * 32) * 4) : "memory");                       \
            if ((OUT & B_OUT_BF16) && o.stagger != -1) {                                                               \
                u32x2 pk;                                                                                              \
                pk[0] = cvt_pk_bf16(ov[0], ov[1]);                                                                     \
                pk[1] = cvt_pk_bf16(ov[2], ov[3]);                                                                     \
                asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(pk), "v"(voff_cb), \
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
#define B_EPILOGUE()                                                                                                   \
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
        if (HAS_AUX && !AUX_B16) { B_AUX_LOAD(aux0, 0) B_AUX_LOAD(aux1, 1) B_AUX_LOAD(aux2, 2) }                        \
        fx4 va[4], vb[4];                                                                                              \
        B_SCR_WRITE(0, 0)                                                                                              \
        B_SCR_READ(va)                                                                                                 \
        B_EPI_STEP(0, va, vb, aux0) B_EPI_STEP(1, vb, va, aux1) B_EPI_STEP(2, va, vb, aux2) B_EPI_STEP(3, vb, va, aux0) \
        B_EPI_STEP(4, va, vb, aux1) B_EPI_STEP(5, vb, va, aux2) B_EPI_STEP(6, va, vb, aux0) B_EPI_STEP(7, vb, va, aux1) \
        ep_tile += gridDim.x;                                                                                          \
    }

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


// =====================================================================================================================
// Epilogue WITHOUT LDS, for kernels that accumulate the TRANSPOSED product (the B fragment is the MFMA's first operand):
// the 32 x 32 accumulator layout then gives a lane ONE output row (mt * 32 + li) and, per register quad q, FOUR CONSECUTIVE
// columns nt * 32 + 8 q + 4 kh + 0..3 -- a dwordx4 store of fp32 / a dwordx2 store of bf16 straight from the accumulator,
// no transposition.  The LDS-transposed form above costs 16 ds_write_b32 + 4 ds_read_b128 per 32 x 32 tile and wave:
// 4 096 + 1 024 LDS-array cycles per 256 x 256 output next to the 6 144 of a K = 512 main loop's fragment reads and the DMA
// writes, i.e. the LDS -- not the stores -- paced the K = 512 products (two independent workgroups per CU did not overlap
// their epilogues for that reason: profiles/r04_gemm_bf16.md).  A store instruction covers 32 rows x 32 bytes (fp32); the
// four instructions of a tile complete each row's 128-byte line in the L2 before it is written back.
// Bias is per column: 8 float4 per lane (columns of its quads), loaded once.  Gate / residual operands are fetched in the
// shape of the stores, two tiles ahead.  Uses the kernel's locals as B_EPI_DECLS' macros do (wm, wn, li, kh, o, ep, acc ..).
#define T_EPI_DECLS()                                                                                                  \
    int ep_tile = blockIdx.x;                                                                                          \
    constexpr bool HAS_AUX = (EPI & (E_GATE | E_ADD)) != 0;                                                            \
    constexpr bool AUX_B16 = (OUT & B_GATE_BF16) != 0;             /* gate operand is bf16 (sign only) */              \
    const int t_row = wm * 128 + li, t_col = wn * 64 + 4 * kh;     /* + 32 mt, + 32 nt + 8 q */                        \
    const int ldci = (int)o.ldc, ldcbi = (int)o.ldcb;                                                                  \
    const float* xsrc = (EPI & E_GATE) ? ep.gate : ep.add;                                                             \
    const int ldxi = (int)(AUX_B16 ? o.ldgate_b : ((EPI & E_GATE) ? ep.ldgate : ep.ldadd));                            \
    union AuxT {                                                                                                       \
        fx4 f[4];                                                                                                      \
        u32x2 h[4];                                                                                                    \
    };                                                                                                                 \
    AuxT aux0, aux1;
#define T_EPI_DESC()                                                                                                   \
    const int t_ = xcd_swizzle(ep_tile, tiles);                                                                        \
    const int64_t m0 = (int64_t)(t_ / tiles_n) * kTileRows;                                                            \
    const int n0 = (t_ % tiles_n) * kB;                                                                                \
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(                                               \
        (void*)(!HAS_AUX ? (const void*)smem                                                                           \
                         : AUX_B16 ? (const void*)(o.gate_b + m0 * (int64_t)ldxi + n0)                                 \
                                   : (const void*)(xsrc + m0 * (int64_t)ldxi + n0)),                                   \
        0, 0x7FFFFFFF, 0x00020000);                                                                                    \
    const int voff_x = (t_row * ldxi + t_col) * (AUX_B16 ? 2 : 4);
#define T_AUX_LOAD(DST, TILE)                                                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                    \
        if (AUX_B16)                                                                                                   \
            DST.h[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(                                 \
                rx, voff_x, ((((TILE) >> 1) * 32) * ldxi + ((TILE) & 1) * 32 + 8 * q) * 2, 0));                        \
        else                                                                                                           \
            DST.f[q] = __builtin_bit_cast(fx4, __builtin_amdgcn_raw_buffer_load_b128(                                  \
                rx, voff_x, ((((TILE) >> 1) * 32) * ldxi + ((TILE) & 1) * 32 + 8 * q) * 4, 0));                        \
    }
#define T_AUX_PREFETCH()                                                                                               \
    {                                                                                                                  \
        T_EPI_DESC()                                                                                                   \
        T_AUX_LOAD(aux0, 0) T_AUX_LOAD(aux1, 1)                                                                        \
    }
#define T_EPI_TILE(AUX, TILE)                                                                                          \
    {                                                                                                                  \
        constexpr int MT = (TILE) >> 1, NT = (TILE) & 1;                                                               \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                \
            fx4 ov;                                                                                                    \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                            \
                float v = acc[MT][NT][4 * q + c];                                                                      \
                if (EPI & E_BIAS) v += bias4[NT][q][c];                                                                \
                if (EPI & E_RELU) v = fmaxf(v, 0.0f);                                                                  \
                if (EPI & E_DROP)   /* == drop_scale(ep.seed, (row + ep.row0) * N + col, ..); thr > 0 on this path */  \
                    v *= rng_u24_from_x0(x0_lane + (uint32_t)((MT * 32) * N + NT * 32 + 8 * q + c) * kRngMul, drop_sh) >= ep.thr \
                             ? ep.inv_keep : 0.0f;                                                                     \
                if (EPI & E_GATE) {                                                                                    \
                    bool pos;                                                                                          \
                    if (AUX_B16) {                                                                                     \
                        const unsigned hw = (AUX.h[q][c >> 1] >> (16 * (c & 1))) & 0xFFFFu;       /* bf16 > 0 */       \
                        pos = hw != 0 && hw < 0x8000u;                                                                 \
                    } else {                                                                                           \
                        pos = AUX.f[q][c] > 0.0f;                                                                      \
                    }                                                                                                  \
                    v *= pos ? ep.gate_scale : 0.0f;                                                                   \
                }                                                                                                      \
                if (EPI & E_ADD) v += AUX.f[q][c];                                                                     \
                ov[c] = v;                                                                                             \
            }                                                                                                          \
            if (OUT & B_OUT_F32)                                                                                       \
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 1" ::"v"(ov),    \
                             "v"(voff_c), "s"(rc), "s"((MT * 32) * ldci * 4), "i"((NT * 32 + 8 * q) * 4) : "memory");   \
            if (OUT & B_OUT_BF16) {                                                                                    \
                u32x2 pk;                                                                                              \
                pk[0] = cvt_pk_bf16(ov[0], ov[1]);                                                                     \
                pk[1] = cvt_pk_bf16(ov[2], ov[3]);                                                                     \
                asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, %3 offen offset:%4\n\ts_nop 1" ::"v"(pk),    \
                             "v"(voff_cb), "s"(rcb), "s"((MT * 32) * ldcbi * 2), "i"((NT * 32 + 8 * q) * 2) : "memory"); \
            }                                                                                                          \
        }                                                                                                              \
    }
#define T_EPI_STEP(TILE, AUXC)                                                                                         \
    T_EPI_TILE(AUXC, TILE)                                                                                             \
    if (HAS_AUX && (TILE) + 2 < 8) { T_AUX_LOAD(AUXC, (TILE) + 2) }
// PREFETCHED: the operands of tiles 0 and 1 were requested by T_AUX_PREFETCH
#define T_EPILOGUE(PREFETCHED)                                                                                         \
    {                                                                                                                  \
        T_EPI_DESC()                                                                                                   \
        const uint64_t drop_se = rng_seed_eff(ep.seed);                                                                \
        const uint32_t drop_sh = (uint32_t)(drop_se >> 32);                                                            \
        const uint32_t x0_lane = rng_x0(drop_se, (uint32_t)(m0 + t_row + ep.row0) * (uint32_t)N + (uint32_t)(n0 + t_col)); \
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)((OUT & B_OUT_F32) ? o.c + m0 * o.ldc + n0 : (float*)smem), 0, 0x7FFFFFFF, 0x00020000);             \
        const __amdgpu_buffer_rsrc_t rcb = __builtin_amdgcn_make_buffer_rsrc(                                          \
            (void*)((OUT & B_OUT_BF16) ? o.cb + m0 * o.ldcb + n0 : (bf16_t*)smem), 0, 0x7FFFFFFF, 0x00020000);         \
        const int voff_c = (t_row * ldci + t_col) * 4;                                                                 \
        const int voff_cb = (t_row * ldcbi + t_col) * 2;                                                               \
        fx4 bias4[2][4];                                                                                               \
        if (EPI & E_BIAS) {                                                                                            \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                           \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
                    bias4[nt][q] = *reinterpret_cast<const fx4*>(ep.bias + n0 + t_col + nt * 32 + 8 * q);              \
        }                                                                                                              \
        if (HAS_AUX && !(PREFETCHED)) { T_AUX_LOAD(aux0, 0) T_AUX_LOAD(aux1, 1) }                                      \
        T_EPI_STEP(0, aux0) T_EPI_STEP(1, aux1) T_EPI_STEP(2, aux0) T_EPI_STEP(3, aux1)                                \
        T_EPI_STEP(4, aux0) T_EPI_STEP(5, aux1) T_EPI_STEP(6, aux0) T_EPI_STEP(7, aux1)                                \
    }

// ---------------------------------------------------------------------------------------------------------------------
// gemm_nt_bf16_w4_kernel: the same product on a 128 (rows of A) x 256 (rows of B) output tile per workgroup of FOUR waves
// (1 x 4, wave tile 128 x 64: the accumulator / fragment economy of the 8-wave kernel), TWO workgroups resident per CU,
// operands delivered global -> LDS by DMA (buffer_load_dwordx4 ... lds), no staging registers and no LDS stores.
//
// Why: in the 8-wave ping-pong kernel the two wave groups are coupled by the workgroup barrier of every phase, so the
// epilogue of an output tile (LDS transposition, bias / relu / dropout hash / gate / residual, 32-64 stores per wave) is
// exposed: the partner group can run ONE MFMA phase ahead and then waits.  At K = 512 an output tile is only 16 phase
// pairs long and the exposed epilogues were 22-48 % of the time (none -> bf16: 838, bias+relu+drop -> bf16: 628,
// add -> f32: 553 TFLOP/s against 1020 at K = 2048; profiles/r04_gemm_bf16.md).  Two INDEPENDENT workgroups per CU share
// nothing but the hardware: while one is in its epilogue (VALU, LDS, stores) the other one's waves own the matrix pipes.
//
// K tiles of 32 (64-byte operand rows) in a ring of three 24 KB LDS slots [A 128 rows | B 256 rows]; 16-byte chunk c of row
// r sits at chunk position c ^ ((r >> 2) & 3) (conflict-free ds_read_b128 fragments); a DMA instruction writes 1 KB = 16
// rows lane-linearly, so the swizzle is applied to the per-lane SOURCE address.  Software pipeline of a wave, phase p:
//     s_waitcnt vmcnt(6)     this wave's pieces of K tile p+1 have landed (tile p+2 may be in flight)
//     s_barrier              everybody's have; and every wave has finished reading slot p % 3 (its reads were waited for
//                            at the end of phase p-1)
//     DMA  tile p+3 -> slot p % 3                       (6 instructions per wave: 2 of A, 4 of B)
//     ds_read fragments of tile p+1 -> register set (p+1) & 1   (12 x ds_read_b128)
//     16 MFMAs on register set p & 1
//     s_waitcnt lgkmcnt(0)
// One barrier per 16 MFMAs among four waves; the LDS reads of the next phase and the DMA issue run under this phase's
// MFMAs, the other workgroup's waves fill what is left.  Not persistent: a workgroup owns one output tile, its epilogue
// scratch aliases the operand ring (idle by then), and the CU's other workgroup covers its prologue.
constexpr int kWTM = 128, kWTN = 256;        // output tile
constexpr int kWSlotA = kWTM * kBRowB;       // 8 KB
constexpr int kWSlot = (kWTM + kWTN) * kBRowB;   // 24 KB
constexpr int kWLds = 3 * kWSlot;            // 72 KB: two workgroups per CU
constexpr int kWThreads = 256;

template <int EPI, int OUT>
__global__ __launch_bounds__(kWThreads, 2) void gemm_nt_bf16_w4_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                      const bf16_t* __restrict__ B, int64_t ldb, Bf16Out o,
                                                                      int64_t M, int N, int K, int tiles_n, int tiles,
                                                                      EpiParams ep) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int wm = 0;
    const int wn = wave;
    const int li = lane & 31, kh = lane >> 5;
    const int P = K / kBBK;                                        // K tiles (K % 64 == 0: P even, >= 2)
    const int t_own = xcd_swizzle((int)blockIdx.x, tiles);
    const int64_t m_own = (int64_t)(t_own / tiles_n) * kWTM;
    const int n_own = (t_own % tiles_n) * kWTN;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- DMA: wave w delivers A rows [32 w, 32 w + 32) and B rows [64 w, 64 w + 64) of a K tile ----
    // lane l of an instruction -> row l >> 2 of its 16-row block, chunk position l & 3 <- logical chunk (l & 3) ^ ((l >> 4) & 3)
    const int dl_row = lane >> 2, dl_c = (lane & 3) ^ ((lane >> 4) & 3);
    const int voff_a = (dl_row * (int)lda + dl_c * 8) * 2;
    const int voff_b = (dl_row * (int)ldb + dl_c * 8) * 2;
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)(A + (m_own + 32 * wave) * lda), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)(B + ((int64_t)n_own + 64 * wave) * ldb), 0, 0x7FFFFFFF, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int ldai = (int)lda, ldbi = (int)ldb;
#define W_ISSUE(SOFF, KT)                                                                                               \
    {                                                                                                                   \
        unsigned char* da_ = smem + (SOFF) + wave * 2048;                                                               \
        unsigned char* db_ = smem + (SOFF) + kWSlotA + wave * 4096;                                                     \
        const int k_ = (KT) * kBBK;                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(da_ + j * 1024), 16, voff_a,                     \
                                                     (j * 16 * ldai + k_) * 2, 0, 0);                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(db_ + j * 1024), 16, voff_b,                     \
                                                     (j * 16 * ldbi + k_) * 2, 0, 0);                                   \
    }

    // ---- fragments: lane (row li, k group kh) of k16 step ks reads chunk 2 ks + kh of its row (inline asm: for a C++ LDS
    // load hipcc would wait vmcnt(0) as soon as an LDS-DMA is in flight) ----
    const unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int fsw = (li >> 2) & 3;
    const unsigned a_ad0 = ldsb + li * kBRowB + (((0 + kh) ^ fsw) << 4);
    const unsigned a_ad1 = ldsb + li * kBRowB + (((2 + kh) ^ fsw) << 4);
    const unsigned b_ad0 = a_ad0 + kWSlotA + wn * 64 * kBRowB;
    const unsigned b_ad1 = a_ad1 + kWSlotA + wn * 64 * kBRowB;
    fx4 xb[2][2], xa[2][4], yb[2][2], ya[2][4];                    // two register sets, [ks][tile]
#define W_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
#define W_READ_FRAGS(S_, SOFF)                                                                                          \
    {                                                                                                                   \
        const unsigned a0_ = a_ad0 + (SOFF), a1_ = a_ad1 + (SOFF), b0_ = b_ad0 + (SOFF), b1_ = b_ad1 + (SOFF);          \
        W_RD(S_##b[0][0], b0_, 0);                                                                                      \
        W_RD(S_##b[0][1], b0_, 32 * kBRowB);                                                                            \
        W_RD(S_##a[0][0], a0_, 0);                                                                                      \
        W_RD(S_##a[0][1], a0_, 32 * kBRowB);                                                                            \
        W_RD(S_##a[0][2], a0_, 64 * kBRowB);                                                                            \
        W_RD(S_##a[0][3], a0_, 96 * kBRowB);                                                                            \
        W_RD(S_##b[1][0], b1_, 0);                                                                                      \
        W_RD(S_##b[1][1], b1_, 32 * kBRowB);                                                                            \
        W_RD(S_##a[1][0], a1_, 0);                                                                                      \
        W_RD(S_##a[1][1], a1_, 32 * kBRowB);                                                                            \
        W_RD(S_##a[1][2], a1_, 64 * kBRowB);                                                                            \
        W_RD(S_##a[1][3], a1_, 96 * kBRowB);                                                                            \
    }
    // the wait names every destination read-write: no consumer can be scheduled above it
#define W_LGKM_WAIT(S_)                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                 : "+v"(S_##b[0][0]), "+v"(S_##b[0][1]), "+v"(S_##b[1][0]), "+v"(S_##b[1][1]), "+v"(S_##a[0][0]),       \
                   "+v"(S_##a[0][1]), "+v"(S_##a[0][2]), "+v"(S_##a[0][3]), "+v"(S_##a[1][0]), "+v"(S_##a[1][1]),       \
                   "+v"(S_##a[1][2]), "+v"(S_##a[1][3]));
#define W_MFMA(S_)                                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                    \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                              \
            /* B fragment first: the accumulator holds the TRANSPOSED tile (lane = output row, registers = columns) */  \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, S_##b[ks][0]),              \
                                                                 __builtin_bit_cast(bf16x8, S_##a[ks][mt]), acc[mt][0], 0, 0, 0); \
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, S_##b[ks][1]),              \
                                                                 __builtin_bit_cast(bf16x8, S_##a[ks][mt]), acc[mt][1], 0, 0, 0); \
        }
    // vmcnt(6): this wave's pieces of the NEXT tile have landed, the one after it (6 instructions) may be in flight
#define W_SYNC(VMC)                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #VMC ")\n\ts_barrier" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);

    constexpr int kTileRows = kWTM;
    T_EPI_DECLS()

    // one phase for K tile p: CUR_ = register set holding its fragments, NXT_ = set that receives tile p+1.
    // so0 / so1 / so2 = byte offsets of the slots of tiles p, p+1, p+2 (rotated at the end of the phase)
#define W_PHASE(CUR_, NXT_, VMC, ISSUE_, READ_, PREFETCH_)                                                              \
    {                                                                                                                   \
        W_SYNC(VMC)                                                                                                     \
        if (ISSUE_) W_ISSUE(so0, p + 3)                                                                                 \
        if ((PREFETCH_) && HAS_AUX && AUX_B16) T_AUX_PREFETCH()                                                          \
        if (READ_) W_READ_FRAGS(NXT_, so1)                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        W_MFMA(CUR_)                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        if (READ_) { W_LGKM_WAIT(NXT_) }                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        ++p;                                                                                                            \
        { const int t_ = so0; so0 = so1; so1 = so2; so2 = t_; }                                                         \
    }

    // prologue: tiles 0, 1, 2 requested; tile 0 landed for everybody -> fragments of tile 0 in set x  (P >= 4, host)
    int so0 = 0, so1 = kWSlot, so2 = 2 * kWSlot;
    int p = 0;
    W_ISSUE(0, 0)
    W_ISSUE(kWSlot, 1)
    W_ISSUE(2 * kWSlot, 2)
    W_SYNC(12)
    W_READ_FRAGS(x, 0)
    W_LGKM_WAIT(x)
    // steady state, two phases per trip (the register sets alternate); the last four phases are peeled: they stop issuing
    // (tile p+3 does not exist), wait for everything, request the epilogue's bf16 gate operand and stop reading
#pragma unroll 1
    while (p < P - 4) {
        W_PHASE(x, y, 6, true, true, false)
        W_PHASE(y, x, 6, true, true, false)
    }
    W_PHASE(x, y, 6, true, true, false)          // p = P - 4: requests the last tile
    W_PHASE(y, x, 6, false, true, false)         // p = P - 3
    W_PHASE(x, y, 0, false, true, true)          // p = P - 2: nothing in flight behind tile P - 1
    W_PHASE(y, x, 0, false, false, false)        // p = P - 1
    T_EPILOGUE(HAS_AUX && AUX_B16)
#undef W_PHASE
#undef W_SYNC
#undef W_MFMA
#undef W_LGKM_WAIT
#undef W_READ_FRAGS
#undef W_RD
#undef W_ISSUE
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

static std::atomic<int> g_bf16_nt_variant{1};      // 1: four-wave workgroups, two per CU (gemm_nt_bf16_w4_kernel); 0: 8-wave ping-pong

int vqcpc_gemm_bf16_set_variant(int variant) {
    VQ_REQUIRE(variant == 0 || variant == 1, "gemm_bf16_set_variant: 0 (8-wave ping-pong kernel) or 1 (two 4-wave workgroups per CU)");
    g_bf16_nt_variant.store(variant, std::memory_order_relaxed);
    return VQCPC_OK;
}

int vqcpc_gemm_nt_bf16_supported(int64_t M, int N, int K) { return (M % kB == 0 && N % kB == 0 && K % (2 * kBBK) == 0) ? 1 : 0; }

int vqcpc_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, void* Cb, int64_t ldcb,
                       int64_t M, int N, int K, const float* bias, int act, float drop_p, uint64_t seed, const float* gate,
                       int64_t ldgate, const void* gate_bf16, int64_t ldgate_bf16, float gate_scale, const float* add,
                       int64_t ldadd, void* stream) {
    if (M == 0) return VQCPC_OK;
    VQ_REQUIRE(A && B && (C || Cb), "gemm_nt_bf16: null pointer");
    VQ_REQUIRE(vqcpc_gemm_nt_bf16_supported(M, N, K), "gemm_nt_bf16: M, N must be multiples of 256 and K of 64 (M=%lld N=%d K=%d)",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && (!C || ldc >= N) && (!Cb || (ldcb >= N && ldcb % 4 == 0)),
               "gemm_nt_bf16: bad leading dimensions");
    VQ_REQUIRE(aligned16(A) && aligned16(B) && (!C || aligned16(C)) && (reinterpret_cast<uintptr_t>(Cb) & 7u) == 0,
               "gemm_nt_bf16: alignment");
    VQ_REQUIRE(!(gate && gate_bf16), "gemm_nt_bf16: one gate operand only");
    VQ_REQUIRE(act == 0 || act == 1, "gemm_nt_bf16: act must be 0 or 1");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm_nt_bf16: bad dropout probability");
    EpiParams ep{bias, act, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, gate, ldgate, gate_scale, add, ldadd, nullptr, 0, 0};
    static const int stagger_per_ktile = getenv("VQCPC_BF16_STAGGER") ? atoi(getenv("VQCPC_BF16_STAGGER")) : 0;
    const int tiles_total = (int)((M / kB) * (N / kB));
    Bf16Out o{C, ldc, (bf16_t*)Cb, ldcb, (const bf16_t*)gate_bf16, ldgate_bf16,
              stagger_per_ktile < 0 ? stagger_per_ktile : (tiles_total >= 4 * kNumCU ? stagger_per_ktile * (K / kBBK) : 0)};
    const bool has_gate = gate || gate_bf16;
    const int flags = (bias ? E_BIAS : 0) | (act == 1 ? E_RELU : 0) | (ep.thr ? E_DROP : 0) | (has_gate ? E_GATE : 0) |
                      (add ? E_ADD : 0);
    const int out = (C ? B_OUT_F32 : 0) | (Cb ? B_OUT_BF16 : 0) | (gate_bf16 ? B_GATE_BF16 : 0);
    const int tn = N / kB;
    const int tiles = (int)((M / kB) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kBThreads);
    hipStream_t st = (hipStream_t)stream;
    const bool w4 = g_bf16_nt_variant.load(std::memory_order_relaxed) == 1 && K >= 4 * kBBK;
    const int tiles_w4 = (int)((M / kWTM) * tn);
#define BL(EPIV, OUTV)                                                                                                 \
    if (flags == (EPIV) && out == (OUTV) && w4) {                                                                      \
        static bool attr_w4 = false;                                                                                   \
        if (!attr_w4) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_w4_kernel<EPIV, OUTV>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kWLds);                              \
            attr_w4 = true;                                                                                            \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_nt_bf16_w4_kernel<EPIV, OUTV>), dim3((unsigned)tiles_w4), dim3(kWThreads), kWLds, st, \
                           (const bf16_t*)A, lda, (const bf16_t*)B, ldb, o, M, N, K, tn, tiles_w4, ep);                \
        VQ_CHECK_LAUNCH("gemm_nt_bf16_w4");                                                                            \
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
    BL(0, B_OUT_BF16)
    BL(E_BIAS, B_OUT_BF16)                           // in_proj output for the all-bf16 attention kernels
    BL(E_BIAS | E_ADD, B_OUT_F32)                    // residual sums for LayerNorm (see gemm.hip)
    BL(E_BIAS | E_DROP | E_ADD, B_OUT_F32)
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
    const int64_t rows_per_split = round_up(ceil_div(M, splits), 2 * kTBM);
    float* ws = (float*)workspace;
    float* ws_bias = db ? ws + (int64_t)splits * N * K : nullptr;
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kTBBuf);
        attr_done = true;
    }
    const int tk2 = K / kB;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((N / kB) * tk2, splits), dim3(kBThreads), 2 * kTBBuf, s, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, M, N, K, tk2, rows_per_split, ws, ws_bias);
    VQ_CHECK_LAUNCH("gemm_tn_bf16");
    return launch_reduce_splits2(ws, (int64_t)N * K, splits, dW, (int64_t)N * K, ws_bias, N, db, db ? N : 0, accumulate, s);
}

}  // extern "C"
