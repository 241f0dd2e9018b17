// bf16x6 NT GEMM on PRE-SPLIT operands ("tiled planes").
//
// The fp32-in kernels (gemm.hip) split every operand element into its three bf16 pieces while it is staged: 5.5 VALU
// instructions per element, repeated by every workgroup that stages the element (a weight row once per 256-row M tile, an
// activation row once per 256-column N tile), with ONE raw-operand register set of latency cover (the split temporaries
// and the 72 fragment registers leave no room for a second one).  Ablations of gemm_nt_x6_pp_kernel on the C1 shapes
// (tools/ablate_pp_gemm.py): no split arithmetic +10-12 %, no global loads +14-19 %, neither +25-28 %.
//
// Here the split happens ONCE, where the tensor is produced (vqcpc_split3_planes for weights; the LayerNorm / attention /
// GEMM-epilogue producers for activations), into the format below, and the GEMM moves bytes only:
//
//   P3 format of X[R][K] (fp32, K % 16 == 0): three planes p = 0 (high), 1 (mid), 2 (low) of bf16 with
//   x == high + mid + low exactly (gemm_common.h split3), stored K-TILE-MAJOR:
//       element (p, r, k)  ->  ((p * K/16 + k/16) * R + r) * 16 + k % 16          (in bf16 units)
//   so the 16-k slice of 256 consecutive rows that one workgroup stages per K tile and plane is ONE contiguous 8 KB
//   stretch (row-major fp32 gave 64-byte pieces at a K * 4-byte stride).
//
// Kernel = gemm_nt_x6_pp_kernel's schedule (256 x 256 x 16 tile, 8 waves, two wave groups one phase apart, persistent over
// tiles, same LDS image and fragment reads, same MFMA order => bit-identical results) with the operands delivered by
// LDS-DMA straight into that image: no staging registers (a second raw register set next to the 72 fragment and 128
// accumulator registers spilled: 256 VGPRs + 76-108 bytes of scratch), no VALU, no LDS writes by the waves, and a K tile
// is requested two phase pairs before it is read.
//
// MEASURED (tools/bench_gemm_planes.py, C1 shapes, isolated): +4-11 % over the fp32-in kernel at K = 256 (195 -> 216,
// 180 -> 196, 173 -> 185 TFLOP/s), +2-7 % at K = 768 / 1024 (213 -> 222-227); with the A panel L2-resident (same rows for
// every M tile) 227 -> 251 at K = 1024.  The schedule alone (no operand delivery at all) runs at 246 / 304 TFLOP/s
// (K = 256 / 1024), so what is left is the cost of moving 48 KB per K tile into the LDS of a CU that is issuing MFMAs,
// whichever way it is moved.  An activation in P3 costs its producer 6 instead of 4 bytes per element in an HBM-bound
// kernel -- more than these GEMM gains return at the C1 sizes -- so the training step does NOT use this path; it stays
// as a tested, opt-in building block.  (Pre-splitting ONLY the weight operand of the fp32-in kernel was also measured: no
// gain -- gemm.hip, ablation notes.)
#include "gemm_common.h"

namespace vq {

constexpr int kP = 256;                 // tile edge
constexpr int kPBK = 16;
constexpr int kPThreads = 512;
constexpr int kPPlane = kP * 32;        // 8 KB: 256 rows x 16 bf16
constexpr int kPBuf = 6 * kPPlane;      // 48 KB per buffer (A h/m/l, B h/m/l)

// ---------------------------------------------------------------------------------------------------------------------
// fp32 [R][K] (row stride ld) -> P3.  One wavefront per (16 rows, one K tile): lane -> (row = lane >> 2, 4 k's), so the
// three stores of a wave are 512 contiguous bytes each.
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int K,
                                                            unsigned short* __restrict__ out) {
    const int KT = K >> 4;
    const int lane = threadIdx.x & 63;
    const int64_t nrb = (R + 15) >> 4;
    const int64_t nw = nrb * KT;
    const int64_t plane = (int64_t)KT * R * 16;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += (int64_t)gridDim.x * 4) {
        const int kt = (int)(w % KT);
        const int64_t r = (w / KT) * 16 + (lane >> 2);
        if (r >= R) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + kt * 16 + (lane & 3) * 4);
        uint2 h, m, l;
        split3x4(v, h, m, l);
        unsigned short* o = out + ((int64_t)kt * R + r) * 16 + (lane & 3) * 4;
        *reinterpret_cast<uint2*>(o) = h;
        *reinterpret_cast<uint2*>(o + plane) = m;
        *reinterpret_cast<uint2*>(o + 2 * plane) = l;
    }
}

// P3 -> fp32 (tests, debugging): x = (h + m) + l is exact (the three pieces have disjoint mantissa ranges)
__global__ __launch_bounds__(256) void join3_planes_kernel(const unsigned short* __restrict__ in, int64_t R, int K,
                                                           float* __restrict__ x, int64_t ld) {
    const int KT = K >> 4;
    const int64_t plane = (int64_t)KT * R * 16;
    const int64_t total = R * K;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / K;
        const int k = (int)(e - r * K);
        const int64_t o = ((int64_t)(k >> 4) * R + r) * 16 + (k & 15);
        const float h = __uint_as_float((uint32_t)in[o] << 16), m = __uint_as_float((uint32_t)in[o + plane] << 16),
                    l = __uint_as_float((uint32_t)in[o + 2 * plane] << 16);
        x[r * ld + k] = (h + m) + l;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS: three stages of 48 KB (A h/m/l, B h/m/l planes of one K tile: 256 rows x 32 bytes each) = 144 KB.  A stage is
// delivered by LDS-DMA (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes = 1 KB per instruction, written lane-linearly):
// wave w delivers rows [64 (w & 3), + 64) of the three planes of operand w >> 2, two instructions per plane.  The bank
// swizzle of the image (16-byte chunk of a row XOR bit 3 of the row, as in gemm_nt_x6_pp_kernel) is applied to the SOURCE
// address: lane l writes physical chunk l & 1 of row l >> 1 and therefore fetches logical chunk (l & 1) ^ ((l >> 4) & 1).
//
// Schedule = the ping-pong of gemm_nt_x6_pp_kernel / gemm_nt_x6_dma_kernel: memory phase s = issue the DMA of stage s+2,
// wait for this wave's share of stage s+1 (counted vmcnt), read the 18 fragments of stage s [+ the epilogue of a finished
// output tile]; MFMA phase = 48 MFMAs; a workgroup barrier after each; wave group 1 (rows 128..255) one phase behind.
//   stage s+1 is complete before anyone reads it: every wave waits for its own share before the barrier that ends its
//   memory phase s, and both groups pass such a barrier before group 0 starts memory phase s+1;
//   the slot that receives stage s+2 held stage s-1, whose last readers (group 1, memory phase s-1) finished one barrier
//   before group 0 issues into it.
// LDS reads are inline asm: for a C++ LDS load hipcc inserts s_waitcnt vmcnt(0) as soon as an LDS-DMA is in flight.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int kPStages = 3;
constexpr int kPStage = kPBuf;                       // 48 KB
constexpr int kPLds = kPStages * kPStage;            // 144 KB

template <int EPI>
__global__ __launch_bounds__(kPThreads, 2) void gemm_nt_x6_planes_kernel(const unsigned short* __restrict__ Ap,
                                                                         const unsigned short* __restrict__ Bp,
                                                                         float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                         int K, int tiles_n, int tiles, EpiParams ep) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_p[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: descriptors stay in SGPRs
    const int wm = wave >> 2, wn = wave & 3;                       // wm = wave group: 0 leads, 1 runs one phase behind
    const int li = lane & 31, kh = lane >> 5;
    const int T = K / kPBK;                                        // K tiles per output tile
    const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * T;                                    // length of this workgroup's K-tile stream

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- DMA cursor ----
    const bool loads_a = wave < 4;
    const unsigned short* const src_p = loads_a ? Ap : Bp;
    const int64_t src_rows = loads_a ? M : (int64_t)N;                         // rows of the operand matrix
    const unsigned kt_bytes = (unsigned)(src_rows * 32);                       // one K tile of one plane
    const unsigned plane_bytes = (unsigned)T * kt_bytes;                       // 3 planes < 4 GB: checked on the host
    const int src_r0 = (wave & 3) * 64;
    const int voff = (lane >> 1) * 32 + ((((lane & 1) ^ ((lane >> 4) & 1))) << 4);
    int ld_tile = blockIdx.x, ld_k = 0;
    __amdgpu_buffer_rsrc_t src_rs;
#define PL_SET_SRC()                                                                                              \
    {                                                                                                             \
        const int t_ = xcd_swizzle(min(ld_tile, tiles - 1), tiles);                                               \
        const int64_t row0_ = (int64_t)(loads_a ? t_ / tiles_n : t_ % tiles_n) * kP + src_r0;                     \
        src_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_p + row0_ * 16), 0, 0xFFFFFFFF, 0x00020000);       \
    }
    PL_SET_SRC()
#define PL_ISSUE(SLOT)                                                                                            \
    {                                                                                                             \
        unsigned char* dst_ = smem_p + (SLOT) * kPStage + (loads_a ? 0 : 3 * kPPlane) + src_r0 * 32;              \
        const unsigned so_ = (unsigned)ld_k * kt_bytes;                                                           \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (lds_ptr_t)(dst_ + p * kPPlane), 16, voff,           \
                                                     so_ + p * plane_bytes, 0, 0);                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (lds_ptr_t)(dst_ + p * kPPlane + 1024), 16, voff,    \
                                                     so_ + p * plane_bytes + 1024, 0, 0);                         \
        }                                                                                                         \
        ++ld_k;                                                                                                   \
        if (ld_k == T) {                                                                                          \
            ld_k = 0;                                                                                             \
            ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */               \
            PL_SET_SRC()                                                                                          \
        }                                                                                                         \
    }

    // ---- fragment addressing (image of gemm_nt_x6_pp_kernel) ----
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_p;
    const unsigned swz = ((kh ^ (li >> 3)) & 1) << 4;              // every fragment row is base + li with base % 16 == 0
    const unsigned a_addr = lds0 + (wm * 128 + li) * 32 + swz;
    const unsigned b_addr = lds0 + 3 * kPPlane + (wn * 64 + li) * 32 + swz;
    bf16x8 fb[3][2], fa[4][3];
#define PL_LDS_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
#define PL_READ_FRAGS(SLOT)                                                                                       \
    {                                                                                                             \
        const unsigned a_ = a_addr + (SLOT) * kPStage, b_ = b_addr + (SLOT) * kPStage;                            \
        u32x4 rb[3][2], ra[4][3];                                                                                 \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                                        \
            PL_LDS_READ(rb[pc][0], b_, pc * kPPlane);                                                             \
            PL_LDS_READ(rb[pc][1], b_, pc * kPPlane + 32 * 32);                                                   \
            PL_LDS_READ(ra[0][pc], a_, pc * kPPlane);                                                             \
            PL_LDS_READ(ra[1][pc], a_, pc * kPPlane + 32 * 32);                                                   \
            PL_LDS_READ(ra[2][pc], a_, pc * kPPlane + 64 * 32);                                                   \
            PL_LDS_READ(ra[3][pc], a_, pc * kPPlane + 96 * 32);                                                   \
        }                                                                                                         \
        /* the wait names every destination read-write: no consumer can be scheduled above it */                 \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                       \
                     : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]), \
                       "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[0][2]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[1][2]), \
                       "+v"(ra[2][0]), "+v"(ra[2][1]), "+v"(ra[2][2]), "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(ra[3][2])); \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) {                                                        \
            fb[pc][0] = __builtin_bit_cast(bf16x8, rb[pc][0]);                                                    \
            fb[pc][1] = __builtin_bit_cast(bf16x8, rb[pc][1]);                                                    \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) fa[mt][pc] = __builtin_bit_cast(bf16x8, ra[mt][pc]); \
        }                                                                                                         \
    }
#define PL_TERM(PA, PB)                                                                                              \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                               \
        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][PA], fb[PB][0], acc[mt][0], 0, 0, 0);            \
        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][PA], fb[PB][1], acc[mt][1], 0, 0, 0);            \
    }
#define PL_MFMA() PL_TERM(2, 0) PL_TERM(0, 2) PL_TERM(1, 1) PL_TERM(1, 0) PL_TERM(0, 1) PL_TERM(0, 0)
#define PL_BARRIER()                          \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);
#define PL_WAIT(N)                                                   \
    __builtin_amdgcn_sched_barrier(0);                               \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue of the output tile with linear index `ep_tile` (as gemm_nt_x6_pp_kernel: buffer addressing, the
    // bias values and the gate / add operands requested ahead) ----
    int ep_tile = blockIdx.x;
    constexpr bool HAS_AUX = (EPI & (E_GATE | E_ADD)) != 0;
    const int ldci = (int)ldc;
    const float* xsrc = (EPI & E_GATE) ? ep.gate : ep.add;
    const int ldxi = (int)((EPI & E_GATE) ? ep.ldgate : ep.ldadd);
    float bias_nx0 = 0.0f, bias_nx1 = 0.0f;
#define PL_BIAS_REQUEST()                                                                                             \
    if (EPI & E_BIAS) {                                                                                                \
        const int tb_ = xcd_swizzle(min(ep_tile, tiles - 1), tiles);                                                   \
        const float* bp_ = ep.bias + (tb_ % tiles_n) * kP + wn * 64 + li;                                              \
        bias_nx0 = bp_[0];                                                                                             \
        bias_nx1 = bp_[32];                                                                                            \
    }
#define PL_EPILOGUE()                                                                                                  \
    {                                                                                                                  \
        const int t_ = xcd_swizzle(ep_tile, tiles);                                                                    \
        const int64_t m0 = (int64_t)(t_ / tiles_n) * kP;                                                               \
        const int n0 = (t_ % tiles_n) * kP;                                                                            \
        const __amdgpu_buffer_rsrc_t rc =                                                                              \
            __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * ldc + n0), 0, 0x7FFFFFFF, 0x00020000);                  \
        const int voff_c = ((wm * 128 + 4 * kh) * ldci + wn * 64 + li) * 4;                                            \
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)(HAS_AUX ? xsrc + m0 * (int64_t)ldxi + n0 : C), 0, 0x7FFFFFFF, 0x00020000);                         \
        const int voff_x = ((wm * 128 + 4 * kh) * ldxi + wn * 64 + li) * 4;                                            \
        float aux[2][16];                                                                                              \
        if (HAS_AUX) {                                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) aux[0][r] = __builtin_bit_cast(                             \
                float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff_x, (((r & 3) + 8 * (r >> 2)) * ldxi) * 4, 0));    \
        }                                                                                                              \
        const int64_t row_base = m0 + wm * 128 + 4 * kh;                                                               \
        const int col_base = n0 + wn * 64 + li;                                                                        \
        const float bv_cur0 = bias_nx0, bv_cur1 = bias_nx1;                                                            \
        ep_tile += gridDim.x;                                                                                          \
        PL_BIAS_REQUEST()                                                                                              \
        _Pragma("unroll") for (int tile = 0; tile < 8; ++tile) {                                                       \
            const int mt = tile >> 1, nt = tile & 1;                                                                   \
            if (HAS_AUX && tile + 1 < 8) {                                                                             \
                const int mt2 = (tile + 1) >> 1, nt2 = (tile + 1) & 1;                                                 \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) aux[(tile + 1) & 1][r] = __builtin_bit_cast(            \
                    float, __builtin_amdgcn_raw_buffer_load_b32(                                                       \
                               rx, voff_x, ((mt2 * 32 + (r & 3) + 8 * (r >> 2)) * ldxi + nt2 * 32) * 4, 0));           \
            }                                                                                                          \
            const int col = col_base + nt * 32;                                                                        \
            const float bv = nt ? bv_cur1 : bv_cur0;                                                                   \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                           \
                const int64_t row = row_base + mt * 32 + (r & 3) + 8 * (r >> 2);                                       \
                float v = acc[mt][nt][r] + bv;                                                                         \
                if (EPI & E_RELU) v = fmaxf(v, 0.0f);                                                                  \
                if (EPI & E_DROP) v *= drop_scale(ep.seed, (uint64_t)(row + ep.row0) * N + col, ep.thr, ep.inv_keep);  \
                if (EPI & E_GATE) v *= (aux[tile & 1][r] > 0.0f ? ep.gate_scale : 0.0f);                               \
                if (EPI & E_ADD) v += aux[tile & 1][r];                                                                \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff_c,                     \
                                                      ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldci + nt * 32) * 4, 0);   \
                acc[mt][nt][r] = 0.0f;                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
    }

    // one phase pair for stream position s; SLOT = s % 3 holds K tile s, stage s+2 goes to slot (s + 2) % 3
#define PL_PHASES(SLOT)                                                           \
    {                                                                             \
        if (kt == 0 && s > 0) {                                                   \
            /* tile boundary: stage s+1 is the only request outstanding: wait for it, run the epilogue (its stores */ \
            /* drain under the next MFMA phases), then request stage s+2 */       \
            PL_WAIT(0)                                                            \
            PL_EPILOGUE()                                                         \
            PL_ISSUE(((SLOT) + 2) % 3)                                            \
        } else {                                                                  \
            PL_ISSUE(((SLOT) + 2) % 3)                                            \
            PL_WAIT(6)                                                            \
        }                                                                         \
        PL_READ_FRAGS(SLOT)                                                       \
        PL_BARRIER()                                                              \
        __builtin_amdgcn_s_setprio(1);                                            \
        PL_MFMA()                                                                 \
        __builtin_amdgcn_s_setprio(0);                                            \
        PL_BARRIER()                                                              \
        ++s;                                                                      \
        kt = (kt + 1 == T) ? 0 : kt + 1;                                          \
    }

    // prologue: stages 0, 1 requested; stage 0 complete for everybody after the first barrier
    PL_BIAS_REQUEST()
    PL_ISSUE(0)
    PL_ISSUE(1)
    PL_WAIT(6)
    PL_BARRIER()
    if (wm == 1) { PL_BARRIER() }                        // group 1 falls one phase behind
    int s = 0, kt = 0;
#pragma unroll 1
    for (;;) {
        PL_PHASES(0)
        if (s == S) break;
        PL_PHASES(1)
        if (s == S) break;
        PL_PHASES(2)
        if (s == S) break;
    }
    if (wm == 0) { PL_BARRIER() }                        // pairs with group 1's last barrier
    PL_EPILOGUE()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead DMAs must not outlive the workgroup's LDS
#undef PL_PHASES
#undef PL_EPILOGUE
#undef PL_BIAS_REQUEST
#undef PL_WAIT
#undef PL_BARRIER
#undef PL_MFMA
#undef PL_TERM
#undef PL_READ_FRAGS
#undef PL_LDS_READ
#undef PL_ISSUE
#undef PL_SET_SRC
}

}  // namespace vq

using namespace vq;

extern "C" {

int64_t vqcpc_planes_bytes(int64_t rows, int cols) { return 3 * rows * (int64_t)cols * 2; }

int vqcpc_split3_planes(const float* x, int64_t ld, int64_t rows, int cols, void* planes, void* stream) {
    if (rows == 0) return VQCPC_OK;
    VQ_REQUIRE(x && planes && rows > 0 && cols >= 16 && cols % 16 == 0 && ld >= cols && ld % 4 == 0,
               "split3_planes: bad arguments (cols must be a multiple of 16)");
    VQ_REQUIRE(aligned16(x) && aligned16(planes), "split3_planes: buffers must be 16-byte aligned");
    const int64_t nw = ((rows + 15) / 16) * (cols / 16);
    const int blocks = (int)std::min<int64_t>(ceil_div(nw, 4), 16384);
    hipLaunchKernelGGL(split3_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols,
                       (unsigned short*)planes);
    VQ_CHECK_LAUNCH("split3_planes");
    return VQCPC_OK;
}

int vqcpc_join3_planes(const void* planes, int64_t rows, int cols, float* x, int64_t ld, void* stream) {
    if (rows == 0) return VQCPC_OK;
    VQ_REQUIRE(x && planes && rows > 0 && cols >= 16 && cols % 16 == 0 && ld >= cols, "join3_planes: bad arguments");
    const int blocks = (int)std::min<int64_t>(ceil_div(rows * cols, 256), 16384);
    hipLaunchKernelGGL(join3_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)planes,
                       rows, cols, x, ld);
    VQ_CHECK_LAUNCH("join3_planes");
    return VQCPC_OK;
}

int vqcpc_gemm_nt_planes_supported(int64_t M, int N, int K) {
    return (M >= kP && M % kP == 0 && N % kP == 0 && K % 32 == 0 && K >= 64) ? 1 : 0;
}

int vqcpc_gemm_nt_planes(const void* a_planes, const void* b_planes, float* C, int64_t ldc, int64_t M, int N, int K,
                         const float* bias, int act, float drop_p, uint64_t seed, const float* gate, int64_t ldgate,
                         float gate_scale, const float* add, int64_t ldadd, void* stream) {
    VQ_REQUIRE(a_planes && b_planes && C, "gemm_nt_planes: null pointer");
    VQ_REQUIRE(vqcpc_gemm_nt_planes_supported(M, N, K), "gemm_nt_planes: shape M=%lld N=%d K=%d not supported "
               "(M, N multiples of 256, K a multiple of 32)", (long long)M, N, K);
    VQ_REQUIRE(ldc >= N && aligned16(a_planes) && aligned16(b_planes), "gemm_nt_planes: bad leading dimension / alignment");
    VQ_REQUIRE(!(gate && add), "gemm_nt_planes: gate and add are exclusive");
    VQ_REQUIRE(3 * M * (int64_t)K * 2 < ((int64_t)1 << 32) && 3 * (int64_t)N * K * 2 < ((int64_t)1 << 32),
               "gemm_nt_planes: an operand's planes must stay below 4 GB (32-bit buffer offsets)");
    VQ_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "gemm_nt_planes: drop_p out of range");
    EpiParams ep{};
    ep.bias = bias;
    ep.act = act;
    ep.thr = drop_threshold(drop_p);
    ep.inv_keep = 1.0f / (1.0f - drop_p);
    ep.seed = seed;
    ep.gate = gate;
    ep.ldgate = ldgate;
    ep.gate_scale = gate_scale;
    ep.add = add;
    ep.ldadd = ldadd;
    int flags = 0;
    if (bias) flags |= E_BIAS;
    if (act == 1) flags |= E_RELU;
    if (drop_p > 0.0f) flags |= E_DROP;
    if (gate) flags |= E_GATE;
    if (add) flags |= E_ADD;
    const int tn = N / kP;
    const int tiles = (int)((M / kP) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kPThreads);
    const size_t lds = kPLds;
    hipStream_t st = (hipStream_t)stream;
#define PL_LAUNCH(EPIV)                                                                                                \
    {                                                                                                                  \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)gemm_nt_x6_planes_kernel<EPIV>,                                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_nt_x6_planes_kernel<EPIV>), grid, block, lds, st, (const unsigned short*)a_planes,    \
                           (const unsigned short*)b_planes, C, ldc, M, N, K, tn, tiles, ep);                           \
        VQ_CHECK_LAUNCH("gemm_nt_x6_planes");                                                                          \
        return VQCPC_OK;                                                                                               \
    }
    switch (flags) {
        case 0: PL_LAUNCH(0)
        case E_BIAS: PL_LAUNCH(E_BIAS)
        case E_BIAS | E_RELU: PL_LAUNCH(E_BIAS | E_RELU)
        case E_BIAS | E_RELU | E_DROP: PL_LAUNCH(E_BIAS | E_RELU | E_DROP)
        case E_GATE: PL_LAUNCH(E_GATE)
        case E_ADD: PL_LAUNCH(E_ADD)
        default: break;
    }
#undef PL_LAUNCH
    set_error("gemm_nt_planes: epilogue combination %d not instantiated", flags);
    return VQCPC_EINVAL;
}

}  // extern "C"
