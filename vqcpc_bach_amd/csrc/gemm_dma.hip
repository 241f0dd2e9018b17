// bf16x6 NT GEMM with LDS-DMA operand delivery (gfx950: buffer_load_dwordx4 ... lds).
//
// C[M,N] = epi(A[M,K] . B[N,K]^T), fp32 in / fp32 out, every product evaluated as six bf16 MFMAs on the exact 3-way split
// of its operands (gemm_common.h).  256 x 256 output tile, 8 waves (2 x 4, wave tile 128 x 64), persistent over tiles.
//
// What is different from gemm_nt_x6_pp_kernel (gemm.hip): there the fp32 operands travel global -> VGPR -> split -> LDS
// planes, so the prefetch distance is bounded by the staging registers (one raw set = ~0.8 us of latency cover, and the
// MFMA pipe was busy 63-67 % of the time waiting for it).  Here the RAW fp32 K tiles go global -> LDS by DMA, no VGPRs:
// four 32 KB stages (256 rows x 16 k of A and of B) are in flight / resident, i.e. a K tile is requested three phase
// pairs (~4 us) before it is read.  Each wave then reads its own fragment rows as fp32 (12 ds_read_b128 per K tile, no LDS
// writes at all) and splits them in registers right before its MFMA phase.  The split work per wave is 3x that of the
// staged kernel (a fragment row is split once per wave that uses it instead of once per workgroup), but it sits in the
// memory phase of the ping-pong, which is otherwise idle once the global-load wait is gone.
//
// LDS image of a stage: A rows 0..255 then B rows 0..255, 64 bytes (16 floats) per row.  A DMA instruction writes
// 1 KB = 16 rows lane-linearly (lane l -> row l >> 2, 16-byte chunk l & 3), so the bank swizzle is applied to the SOURCE
// address: physical chunk p of row r holds logical chunk p ^ ((r >> 2) & 3).  A fragment read (lane = row li, 8 floats at
// k = 8 kh ..) is two ds_read_b128 at chunks (2 kh) ^ f and (2 kh + 1) ^ f, f = (li >> 2) & 3: every 16-lane group of a
// b128 read touches 16 distinct 16-byte slots (conflict free).
//
// Schedule (ping-pong wave groups, as gemm_nt_x6_pp_kernel): the K-tile stream of the workgroup is cut into a MEMORY
// phase (issue the DMA of stage s+3, read + split the fragments of stage s [+ epilogue of the finished output tile]) and
// an MFMA phase (48 MFMAs), each closed by a workgroup barrier; wave group 1 (rows 128..255) runs one phase behind group 0,
// so that on every SIMD one wave issues MFMAs while the other one does its memory phase.
//   stage s+1 is complete before anyone reads it: every wave waits for its own share (s_waitcnt vmcnt) before the barrier
//   that ends its memory phase s, and both groups pass such a barrier before group 0 starts memory phase s+1;
//   the slot that receives stage s+3 held stage s-1, whose last readers (group 1, memory phase s-1) finished one barrier
//   before group 0 issues into it.
#include <stdlib.h>

#include "gemm_common.h"

namespace vq {

constexpr int kD = 256;                      // tile edge
constexpr int kDBK = 16;                     // k per stage
constexpr int kDStages = 4;
constexpr int kDRowB = kDBK * 4;             // 64 bytes per operand row per stage
constexpr int kDOperand = kD * kDRowB;       // 16 KB
constexpr int kDStage = 2 * kDOperand;       // 32 KB
constexpr int kDThreads = 512;
constexpr int kDLds = kDStages * kDStage + 8 * 4096;    // 128 KB of stages + a 4 KB epilogue scratch per wave = all 160 KB

typedef __attribute__((address_space(3))) void* lds_ptr_t;

typedef float fx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const fx4& lo, const fx4& hi, bf16x8& h, bf16x8& m, bf16x8& l) {
    uint2 h0, m0, l0, h1, m1, l1;
    split3x4(make_float4(lo.x, lo.y, lo.z, lo.w), h0, m0, l0);
    split3x4(make_float4(hi.x, hi.y, hi.z, hi.w), h1, m1, l1);
    h = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    m = __builtin_bit_cast(bf16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
    l = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

template <int EPI, int ABL = 0>
__global__ __launch_bounds__(kDThreads, 2) void gemm_nt_x6_dma_kernel(const float* __restrict__ A, int64_t lda,
                                                                     const float* __restrict__ B, int64_t ldb,
                                                                     float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                     int K, int tiles_n, int tiles, EpiParams ep) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: descriptors stay in SGPRs
    const int wm = wave >> 2, wn = wave & 3;                       // wm = wave group: 0 leads, 1 runs one phase behind
    const int li = lane & 31, kh = lane >> 5;
    const int T = K / kDBK;                                        // K tiles per output tile (K % 64 == 0: T % 4 == 0)
    const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * T;                                    // length of this workgroup's K-tile stream

    // measurement (VQCPC_GEMM_STAGGER, carried in ep.split_plane which this kernel does not use): the persistent workgroups
    // start 8 phases apart -- phase (blockIdx.x >> 3) & 7, `stagger` x 1024 cycles each -- so that their tile boundaries (the
    // output-store bursts) are spread over a tile period instead of coinciding chip-wide
    if (ep.split_plane > 0) {
        const int n_ = (((int)blockIdx.x >> 3) & 7) * (int)ep.split_plane;
        for (int i = 0; i < n_; ++i) __builtin_amdgcn_s_sleep(16);
    }
    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- DMA cursor: waves 0-3 deliver the A rows [64 w, 64 w + 64) of a stage, waves 4-7 the B rows ----
    const bool loads_a = wave < 4;
    const float* const src_p = loads_a ? A : B;
    const int src_ld = (int)(loads_a ? lda : ldb);
    const int src_r0 = (wave & 3) * 64;
    // per-lane byte offset inside a 16-row block: row l >> 2, logical chunk (l & 3) ^ ((l >> 4) & 3)
    const int voff = ((lane >> 2) * src_ld + (((lane & 3) ^ ((lane >> 4) & 3)) << 2)) * 4;
    int ld_tile = blockIdx.x, ld_k = 0, s_issue_ = 0;
    __amdgpu_buffer_rsrc_t src_rs;
#define D_SET_SRC()                                                                                          \
    {                                                                                                        \
        const int t_ = xcd_swizzle(min(ld_tile, tiles - 1), tiles);                                          \
        const int64_t row0_ = (int64_t)(loads_a ? t_ / tiles_n : t_ % tiles_n) * kD + src_r0;                \
        src_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_p + row0_ * src_ld), 0, 0x7FFFFFFF, 0x00020000); \
    }
    D_SET_SRC()
#define D_ISSUE(SLOT)                                                                                        \
    {                                                                                                        \
        unsigned char* dst_ = smem + (SLOT) * kDStage + wave * 4096;                                         \
        if (ABL != 2 || s_issue_ < 3)                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (lds_ptr_t)(dst_ + j * 1024), 16, voff,         \
                                                     (j * 16 * src_ld + ld_k) * 4, 0, 0);                    \
        ld_k += kDBK; ++s_issue_;                                                                            \
        if (ld_k == K) {                                                                                     \
            ld_k = 0;                                                                                        \
            ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */          \
            D_SET_SRC()                                                                                      \
        }                                                                                                    \
    }

    // ---- fragment addressing ----
    // The fragment reads are inline asm: for a C++ load from LDS hipcc inserts `s_waitcnt vmcnt(0)` in front of it as soon as
    // an LDS-DMA is in flight (it cannot prove that the DMA destination does not alias), which would drain the three
    // stages of run-ahead every phase.  Ordering against the DMA is by the protocol in the header (counted vmcnt + barriers).
    const int fsw = (li >> 2) & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned f_off0 = li * kDRowB + (((2 * kh) ^ fsw) << 4);  // first 4 floats of the lane's 8 (k = 8 kh .. 8 kh + 3)
    const unsigned a_addr = lds0 + (wm * 128) * kDRowB + f_off0;    // + 16 (xor) for the next 4: f_off0 ^ 16
    const unsigned b_addr = lds0 + kDOperand + (wn * 64) * kDRowB + f_off0;
    const unsigned a_addr1 = lds0 + (wm * 128) * kDRowB + (f_off0 ^ 16u);
    const unsigned b_addr1 = lds0 + kDOperand + (wn * 64) * kDRowB + (f_off0 ^ 16u);
    bf16x8 fb[3][2], fa[4][3];
#define D_FAKE(LO, HI, H, M, L) { H = __builtin_bit_cast(bf16x8, LO); M = __builtin_bit_cast(bf16x8, HI); L = H; }
#define D_LDS_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
#define D_READ_FRAGS(SLOT)                                                                                   \
    {                                                                                                        \
        const unsigned a0_ = a_addr + (SLOT) * kDStage, a1_ = a_addr1 + (SLOT) * kDStage;                    \
        const unsigned b0_ = b_addr + (SLOT) * kDStage, b1_ = b_addr1 + (SLOT) * kDStage;                    \
        fx4 rb0l, rb0h, rb1l, rb1h, ra0l, ra0h, ra1l, ra1h, ra2l, ra2h, ra3l, ra3h;                       \
        D_LDS_READ(rb0l, b0_, 0);                                                                            \
        D_LDS_READ(rb0h, b1_, 0);                                                                            \
        D_LDS_READ(rb1l, b0_, 32 * kDRowB);                                                                  \
        D_LDS_READ(rb1h, b1_, 32 * kDRowB);                                                                  \
        D_LDS_READ(ra0l, a0_, 0);                                                                            \
        D_LDS_READ(ra0h, a1_, 0);                                                                            \
        D_LDS_READ(ra1l, a0_, 32 * kDRowB);                                                                  \
        D_LDS_READ(ra1h, a1_, 32 * kDRowB);                                                                  \
        D_LDS_READ(ra2l, a0_, 64 * kDRowB);                                                                  \
        D_LDS_READ(ra2h, a1_, 64 * kDRowB);                                                                  \
        D_LDS_READ(ra3l, a0_, 96 * kDRowB);                                                                  \
        D_LDS_READ(ra3h, a1_, 96 * kDRowB);                                                                  \
        /* the wait names every destination read-write: no consumer can be scheduled above it */            \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                  \
                     : "+v"(rb0l), "+v"(rb0h), "+v"(rb1l), "+v"(rb1h), "+v"(ra0l), "+v"(ra0h), "+v"(ra1l),    \
                       "+v"(ra1h), "+v"(ra2l), "+v"(ra2h), "+v"(ra3l), "+v"(ra3h));                          \
        if (ABL == 1 || ABL == 2) {   /* ablation (tools only): no split arithmetic, garbage planes */                   \
            D_FAKE(rb0l, rb0h, fb[0][0], fb[1][0], fb[2][0]) D_FAKE(rb1l, rb1h, fb[0][1], fb[1][1], fb[2][1])     \
            D_FAKE(ra0l, ra0h, fa[0][0], fa[0][1], fa[0][2]) D_FAKE(ra1l, ra1h, fa[1][0], fa[1][1], fa[1][2])     \
            D_FAKE(ra2l, ra2h, fa[2][0], fa[2][1], fa[2][2]) D_FAKE(ra3l, ra3h, fa[3][0], fa[3][1], fa[3][2])     \
        } else {                                                                                             \
        split8(rb0l, rb0h, fb[0][0], fb[1][0], fb[2][0]);                                                    \
        split8(rb1l, rb1h, fb[0][1], fb[1][1], fb[2][1]);                                                    \
        split8(ra0l, ra0h, fa[0][0], fa[0][1], fa[0][2]);                                                    \
        split8(ra1l, ra1h, fa[1][0], fa[1][1], fa[1][2]);                                                    \
        split8(ra2l, ra2h, fa[2][0], fa[2][1], fa[2][2]);                                                    \
        split8(ra3l, ra3h, fa[3][0], fa[3][1], fa[3][2]);                                                    \
        }                                                                                                    \
    }
#define D_TERM(PA, PB)                                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                       \
        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][PA], fb[PB][0], acc[mt][0], 0, 0, 0);    \
        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][PA], fb[PB][1], acc[mt][1], 0, 0, 0);    \
    }
#define D_MFMA() D_TERM(2, 0) D_TERM(0, 2) D_TERM(1, 1) D_TERM(1, 0) D_TERM(0, 1) D_TERM(0, 0)
#define D_BARRIER()                                                  \
    __builtin_amdgcn_sched_barrier(0);                               \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    __builtin_amdgcn_sched_barrier(0);
    // this wave's share of stage s+1 has landed when at most N of its younger DMA instructions are outstanding
    // (N = 8: stages s+2, s+3 requested; N = 4: only s+2)
#define D_WAIT(N)                                                    \
    __builtin_amdgcn_sched_barrier(0);                               \
    if (ABL != 2) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue of the output tile with linear index `ep_tile` ----
    // The MFMA accumulator layout gives a lane ONE column and 16 rows of a 32x32 tile: storing it directly is 16 dword
    // stores per tile and lane (128 per wave and output tile), and the epilogue -- not the MFMAs -- paced the K = 256 GEMMs
    // (23 % of their time with everything else ablated).  Each 32x32 tile therefore goes through a 4 KB LDS scratch of
    // its wave (16 ds_write_b32, 4 ds_read_b128: lane -> row lane >> 3 (+ 8 j), columns 4 (lane & 7) ..) and leaves as
    // 4 dwordx4 stores (8 row segments of 128 bytes each); bias / gate / residual operands are fetched in the same shape.
    // LDS accesses are inline asm for the same reason as the fragment reads (no compiler-inserted vmcnt(0)); the LDS
    // executes a wave's instructions in order, so write -> read -> next write need no waits among themselves.
    int ep_tile = blockIdx.x;
    const float zero_rt = ep.gate_scale * 0.0f;      // debug builds only (ABL >= 3)
    constexpr int EPV = (ABL == 3 || ABL == 4) ? 1 : 0;     // debug variants of the epilogue (tools/_dbg.sh)
    constexpr bool HAS_AUX = (EPI & (E_GATE | E_ADD)) != 0;
    const int ldci = (int)ldc;
    const float* xsrc = (EPI & E_GATE) ? ep.gate : ep.add;
    const int ldxi = (int)((EPI & E_GATE) ? ep.ldgate : ep.ldadd);
    const unsigned scr = lds0 + kDStages * kDStage + wave * 4096;
    const unsigned scr_w = scr + ((4 * kh) * 32 + li) * 4;                 // element (row 4 kh + .., col li)
    const unsigned scr_r = scr + ((lane >> 3) * 32 + (lane & 7) * 4) * 4;  // row lane >> 3 (+ 8 j), 4 columns
    const int e_row = wm * 128 + (lane >> 3), e_col = wn * 64 + 4 * (lane & 7);
#define D_SCR_WRITE(MT, NT)                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                     \
        asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(scr_w), "v"(acc[MT][NT][r]), "i"(((r & 3) + 8 * (r >> 2)) * 128));
#define D_SCR_READ(V)                                                                                                  \
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(V[0]) : "v"(scr_r));                                            \
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(V[1]) : "v"(scr_r));                                         \
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(V[2]) : "v"(scr_r));                                         \
    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(V[3]) : "v"(scr_r));
#define D_AUX_LOAD(DST, MT, NT)                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) DST[j] = __builtin_bit_cast(fx4, __builtin_amdgcn_raw_buffer_load_b128( \
        rx, voff_x, (((MT) * 32 + 8 * j) * ldxi + (NT) * 32) * 4, 0));
#define D_EPI_TILE(V, AUX, MT, NT)                                                                                     \
    {                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
            const int64_t row = m0 + e_row + (MT) * 32 + 8 * j;                                                        \
            const int col = n0 + e_col + (NT) * 32;                                                                    \
            fx4 o;                                                                                                     \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                            \
                float v = V[j][c];                                                                                     \
                if (ABL >= 3) v += zero_rt;             /* debug: the scheduling that exposed the race */             \
                if (EPI & E_BIAS) v += bias4[NT][c];                                            \
                if (EPI & E_RELU) v = fmaxf(v, 0.0f);                                                                  \
                if (EPI & E_DROP) v *= drop_scale(ep.seed, (uint64_t)(row + ep.row0) * N + col + c, ep.thr, ep.inv_keep); \
                if (EPI & E_GATE) v *= (AUX[j][c] > 0.0f ? ep.gate_scale : 0.0f);                                      \
                if (EPI & E_ADD) v += AUX[j][c];                                                                       \
                o[c] = v;                                                                                              \
            }                                                                                                          \
            /* inline asm with its own wait states: hipcc (ROCm 7.2) put a v_pk_add_f32 that overwrites the data */    \
            /* registers DIRECTLY behind a buffer_store_dwordx4 (0 wait states; the ISA asks for 1-2): the last dword */ \
            /* of a quarter of the lanes was then stored from the NEXT row group's value (found by the bitwise test). */ \
            /* Leading s_nop 4: an SGPR operand may come straight from a v_readlane (SGPR spill reload), which needs 5 */ \
            /* wait states before a VMEM instruction reads it, and nothing is padded inside or in front of asm. */      \
            asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(o), "v"(voff_c), "s"(rc), \
                         "s"((((MT) * 32 + 8 * j) * ldci + (NT) * 32) * 4) : "memory");                               \
        }                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[MT][NT][r] = 0.0f;                                          \
    }
#define D_EPILOGUE()                                                                                                   \
    {                                                                                                                  \
        const int t_ = xcd_swizzle(ep_tile, tiles);                                                                    \
        const int64_t m0 = (int64_t)(t_ / tiles_n) * kD;                                                               \
        const int n0 = (t_ % tiles_n) * kD;                                                                            \
        const __amdgpu_buffer_rsrc_t rc =                                                                              \
            __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * ldc + n0), 0, 0x7FFFFFFF, 0x00020000);                  \
        const int voff_c = (e_row * ldci + e_col) * 4;                                                                 \
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)(HAS_AUX ? xsrc + m0 * (int64_t)ldxi + n0 : C), 0, 0x7FFFFFFF, 0x00020000);                         \
        const int voff_x = (e_row * ldxi + e_col) * 4;                                                                 \
        fx4 bias4[2];                                                                                                  \
        if (EPI & E_BIAS) {                                                                                            \
            bias4[0] = *reinterpret_cast<const fx4*>(ep.bias + n0 + e_col);                                            \
            bias4[1] = *reinterpret_cast<const fx4*>(ep.bias + n0 + e_col + 32);                                       \
        }                                                                                                              \
        fx4 va[4], vb[4], xa[4], xb[4];                                                                                \
        if (HAS_AUX) { D_AUX_LOAD(xa, 0, 0) }                                                                          \
        if (EPV == 1) {        /* debug variant: every LDS step of the transposition waited for */                     \
            _Pragma("unroll") for (int tile = 0; tile < 8; ++tile) {                                                   \
                const int mt = tile >> 1, nt = tile & 1;                                                               \
                D_SCR_WRITE(mt, nt)                                                                                    \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
                D_SCR_READ(va)                                                                                         \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]));             \
                if (HAS_AUX) { D_AUX_LOAD(xa, mt, nt) }                                                                \
                D_EPI_TILE(va, xa, mt, nt)                                                                             \
                if (ABL == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
            }                                                                                                          \
        } else {                                                                                                       \
        D_SCR_WRITE(0, 0)                                                                                              \
        D_SCR_READ(va)                                                                                                 \
        _Pragma("unroll") for (int tp = 0; tp < 4; ++tp) {           /* two tiles per iteration: (tp, 0) then (tp, 1) */ \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]));                 \
            D_SCR_WRITE(tp, 1)                                                                                         \
            D_SCR_READ(vb)                                                                                             \
            if (HAS_AUX) { D_AUX_LOAD(xb, tp, 1) }                                                                     \
            D_EPI_TILE(va, xa, tp, 0)                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]));                 \
            if (tp + 1 < 4) {                                                                                          \
                D_SCR_WRITE(tp + 1, 0)                                                                                 \
                D_SCR_READ(va)                                                                                         \
                if (HAS_AUX) { D_AUX_LOAD(xa, tp + 1, 0) }                                                             \
            }                                                                                                          \
            D_EPI_TILE(vb, xb, tp, 1)                                                                                  \
            if (ABL == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        }                                                                                                              \
        }                                                                                                              \
        ep_tile += gridDim.x;                                                                                          \
    }

    // one phase pair for stream position s; SLOT = s % 4 holds K tile s, stage s+3 goes to slot (s + 3) % 4
#define D_PHASES(SLOT)                                                            \
    {                                                                             \
        if (kt == 0 && s > 0) {                                                   \
            /* tile boundary: stage s+1 first (only s+1, s+2 are outstanding), then the epilogue -- its operand loads */ \
            /* then wait for nothing younger than stage s+2 and its stores drain under the next MFMA phase --, then s+3 */ \
            D_WAIT(4)                                                             \
            D_EPILOGUE()                                                          \
            D_ISSUE(((SLOT) + 3) & 3)                                             \
        } else {                                                                  \
            D_ISSUE(((SLOT) + 3) & 3)                                             \
            D_WAIT(8)                                                             \
        }                                                                         \
        D_READ_FRAGS(SLOT)                                                        \
        D_BARRIER()                                                               \
        __builtin_amdgcn_s_setprio(1);                                            \
        D_MFMA()                                                                  \
        __builtin_amdgcn_s_setprio(0);                                            \
        D_BARRIER()                                                               \
        ++s;                                                                      \
        kt = (kt + 1 == T) ? 0 : kt + 1;                                          \
    }

    // prologue: stages 0, 1, 2 requested; stage 0 complete for everybody after the first barrier
    D_ISSUE(0)
    D_ISSUE(1)
    D_ISSUE(2)
    D_WAIT(8)
    D_BARRIER()
    if (wm == 1) { D_BARRIER() }                         // group 1 falls one phase behind
    int s = 0, kt = 0;
#pragma unroll 1
    while (s < S) {
        D_PHASES(0)
        D_PHASES(1)
        D_PHASES(2)
        D_PHASES(3)
    }
    if (wm == 0) { D_BARRIER() }                         // pairs with group 1's last barrier
    D_EPILOGUE()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead DMAs must not outlive the workgroup's LDS
#undef D_PHASES
#undef D_EPILOGUE
#undef D_EPI_TILE
#undef D_AUX_LOAD
#undef D_SCR_READ
#undef D_SCR_WRITE
#undef D_WAIT
#undef D_BARRIER
#undef D_MFMA
#undef D_TERM
#undef D_READ_FRAGS
#undef D_LDS_READ
#undef D_FAKE
#undef D_ISSUE
#undef D_SET_SRC
}

// ---------------------------------------------------------------------------------------------------------------------
bool gemm_nt_dma_ok(int64_t M, int N, int K, int flags) {
    if (M % kD || N % kD || K % (4 * kDBK)) return false;
    switch (flags) {
        case 0:
        case E_BIAS:
        case E_BIAS | E_RELU:
        case E_BIAS | E_RELU | E_DROP:
        case E_GATE:
        case E_ADD: return true;
        default: return false;
    }
}

int gemm_nt_dma_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                       int K, int flags, const EpiParams& ep, hipStream_t st) {
    const int tn = N / kD;
    const int tiles = (int)((M / kD) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kDThreads);
    static const int abl = lab_env_int("VQCPC_GEMM_ABL", 0);   // ablation builds (tools only)
    static const int stagger = lab_env_int("VQCPC_GEMM_STAGGER", 0);
    EpiParams ep_st = ep;
    ep_st.split_plane = stagger;
#define ep ep_st
#define D_LAUNCH(EPIV)                                                                                                \
    {                                                                                                                 \
        static bool attr_done = false;                                                                                \
        if (!attr_done) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_dma_kernel<EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      kDLds);                                                                         \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        if (abl == 1) {                                                                                               \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_dma_kernel<EPIV, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kDLds); \
            hipLaunchKernelGGL((gemm_nt_x6_dma_kernel<EPIV, 1>), grid, block, kDLds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        } else if (abl == 3 && (EPIV) == 0) {                                                                         \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_dma_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kDLds); \
            hipLaunchKernelGGL((gemm_nt_x6_dma_kernel<0, 3>), grid, block, kDLds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        } else if (abl == 4 && (EPIV) == 0) {                                                                         \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_dma_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kDLds); \
            hipLaunchKernelGGL((gemm_nt_x6_dma_kernel<0, 4>), grid, block, kDLds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        } else if (abl == 5 && (EPIV) == 0) {                                                                         \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_dma_kernel<0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, kDLds); \
            hipLaunchKernelGGL((gemm_nt_x6_dma_kernel<0, 5>), grid, block, kDLds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        } else if (abl == 6 && (EPIV) == 0) {                                                                         \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_dma_kernel<0, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, kDLds); \
            hipLaunchKernelGGL((gemm_nt_x6_dma_kernel<0, 6>), grid, block, kDLds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        } else if (abl == 2) {                                                                                        \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_dma_kernel<EPIV, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kDLds); \
            hipLaunchKernelGGL((gemm_nt_x6_dma_kernel<EPIV, 2>), grid, block, kDLds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        } else                                                                                                        \
        hipLaunchKernelGGL((gemm_nt_x6_dma_kernel<EPIV>), grid, block, kDLds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep); \
        VQ_CHECK_LAUNCH("gemm_nt_x6_dma");                                                                            \
        return VQCPC_OK;                                                                                              \
    }
    switch (flags) {
        case 0: D_LAUNCH(0)
        case E_BIAS: D_LAUNCH(E_BIAS)
        case E_BIAS | E_RELU: D_LAUNCH(E_BIAS | E_RELU)
        case E_BIAS | E_RELU | E_DROP: D_LAUNCH(E_BIAS | E_RELU | E_DROP)
        case E_GATE: D_LAUNCH(E_GATE)
        case E_ADD: D_LAUNCH(E_ADD)
        default: break;
    }
#undef D_LAUNCH
#undef ep
    set_error("gemm_nt_dma: unsupported epilogue %d", flags);
    return VQCPC_EINVAL;
}

}  // namespace vq
