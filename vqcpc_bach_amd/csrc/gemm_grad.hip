// Gradient GEMMs of the backward pass on THREE fp16 MFMAs per product (round 5): dgrad (NT) and wgrad (TN) of every large
// linear layer -- vqcpc_encoder_trainer.py:311-313 (`loss.backward()`) through transformer_custom.py:279-289 and
// multihead_attention_custom.py:171,346 -- at fp32-class accuracy with half the matrix instructions of the six-product
// bf16 split the forward pass uses (gemm.hip).
//
// Arithmetic.  Each fp32 operand element x of a tensor with scale s = 2^e is carried as two fp16 planes
//      h = rtz_f16(x s)             (v_cvt_pkrtz_f16_f32: 11 significant bits, saturating at +-65504, never inf)
//      m = rn_f16(x s - h)          (v_fma_mixlo/hi_f16: the fp32 FMA x s - h is exact, one rounding to fp16)
// so |x s - h - m| <= 2^-21 |x s| (fp16 subnormals are honoured by the converts and by the matrix pipe:
// tools/micro/f16_probe.hip) and a product is  hh + hm + mh  on v_mfma_f32_32x32x16_f16 with fp32 accumulation; the dropped mm
// term is < 2^-20 |ab| with mean 2^-22 |ab| (a relative bias of the RESULT, not of sum |ab|).  Measured against fp64:
// rms 3-5e-7 of the result's rms, i.e. the class of an fp32 GEMM (fp32-MFMA kernel: 3-8e-7; bf16 pair planes: 4.4e-6).
// Since late round 5 the whole-round forward products of a TRAINING step use it too (vqcpc_gemm_nt_f16x3: `train_model()`'s
// default with the gradient GEMMs); evaluation, encode_indices and inference stay on the six-product split (no step, no previous
// amax).  Normwise fp32-class, componentwise 22-bit operands: tests/test_benchmarked_path_gpu.py states and checks the bound.
//
// Scales.  e is chosen per operand TENSOR so that its largest magnitude of the PREVIOUS step lands in [2^11, 2^12): four
// binades of head-room (a tensor may grow 16 x from one step to the next and stay exact; beyond that its largest elements
// saturate at 65504 / s, finite), 22-bit operands down to 2^-14 of the largest element, an absolute floor of 2^-36 of it below.  The kernels compute the operand's amax of THIS
// step while they stage it (v_max3 on the raw registers, one atomic max per workgroup and operand: g3_amax_publish_wg) and leave it for the next step: the caller
// owns a 4-float state per GEMM call site {amax A, amax B (read), amax A, amax B (written)} and rolls it once per step
// (vqcpc_grad_scale_roll); vqcpc_grad_amax primes a site on its first use.  Power-of-two scales are exact; the result is
// multiplied by 2^-(eA + eB) in the epilogue.
//
// Schedule (both kernels): 256 x 256 output tile, 8 waves of 128 x 64, K steps of 16, 24 MFMAs per wave and step.  No phases:
// every wave runs the same software pipeline -- MFMAs of step j, fragment reads of step j+1 (a ring of THREE 32 KB LDS slots),
// split + LDS write of step j+2 from raw registers requested two steps earlier, request of step j+4 -- with ONE workgroup
// barrier per step placed between the last writes of step j+2 and their first read.  The six-product kernels' ping-pong of a
// "memory phase" and an "MFMA phase" is latency-bound once the MFMA phase is 768 cycles (profiles/r05_pmc_grad3_vs_six.txt:
// a K step takes 2 990 cycles per wave for 768 of matrix work, 980 of them parked in s_waitcnt / s_barrier).
#include <algorithm>
#include <atomic>

#include "gemm_common.h"

namespace vq {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

constexpr int kG = 256;                      // tile edge
constexpr int kGBK = 16;                     // contraction elements per step
constexpr int kGThreads = 512;
constexpr int kGPlane = kG * 32;             // NT: 256 rows x 16 fp16 = 8 KB per plane
constexpr int kGSlot = 4 * kGPlane;          // A.h | A.m | B.h | B.m
constexpr int kGSlots = 3;
constexpr int kGTarget = 11;                 // amax of the previous step -> [2^11, 2^12): saturation 16-32 x above it

// scale exponent of a tensor whose previous-step amax is `amax` (0 / denormal: the clamp; inf / nan: the other clamp)
__device__ __forceinline__ int g3_scale_exp(float amax) {
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    // amax == 0 (the tensor was all zeros when the site was primed: nothing is known about its magnitude): the neutral scale 1 --
    // 2^60 would clamp whatever the tensor holds one step later to 65504 * 2^-60, i.e. zero the operand for that step
    if (E == 0) return 0;
    return max(-60, min(60, (kGTarget + 127) - E));
}
__device__ __forceinline__ float g3_pow2(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }

// two fp32 -> (h, m) fp16 pairs, element 0 in the low half.  h = rtz_f16(x s) saturates at +-65504 by itself; the residual is
// taken from x s clamped to the fp16 range (v_med3_f32), so an element beyond the head-room of a stale scale becomes +-65504
// (h) + 0 (m) instead of an infinite m -- bounded, finite, and gone one step later when the scale has followed.  A NaN input
// stays a NaN (through h: the clamp only feeds m).
__device__ __forceinline__ void g3_split_pair(float a, float b, float s, uint32_t& h, uint32_t& m) {
    const float xa = a * s, xb = b * s;
    h = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(xa, xb));
    const float ta = __builtin_amdgcn_fmed3f(xa, -65504.0f, 65504.0f), tb = __builtin_amdgcn_fmed3f(xb, -65504.0f, 65504.0f);
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(m) : "v"(ta), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(tb), "v"(h));
}
__device__ __forceinline__ void g3_amax4(const float4& v, float& amax) {
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v.x), "v"(v.y));
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v.z), "v"(v.w));
}
__device__ __forceinline__ void g3_split4(const float4& v, float s, uint2& h, uint2& m) {
    g3_split_pair(v.x, v.y, s, h.x, m.x);
    g3_split_pair(v.z, v.w, s, h.y, m.y);
}
// amax of a wave -> one atomic max on the non-negative float's bit pattern (order independent: deterministic)
__device__ __forceinline__ void g3_amax_publish(float amax, float* slot) {
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0 && amax > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(amax));
}

// amax of a WORKGROUP -> one atomic max per operand.  Atomics on one address retire one by one at the memory side (measured on the
// tail-row kernel below: ~11.5 ns each -- 256 workgroups x 8 waves x 2 operands = 4 096 of them took 47 us, most of that launch):
// the eight waves' values meet in LDS first.  `lds`: 64 bytes of the workgroup's LDS that no wave still reads after the barrier.
__device__ __forceinline__ void g3_amax_publish_wg(float amax_a, float amax_b, float* state, unsigned char* lds) {
    amax_a = wave_max(amax_a);
    amax_b = wave_max(amax_b);
    __syncthreads();                                     // every wave is done with the operand slots
    float* red = reinterpret_cast<float*>(lds);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = amax_a;
        red[8 + (threadIdx.x >> 6)] = amax_b;
    }
    __syncthreads();
    if (threadIdx.x < 16) {                              // lanes 0-7: operand A, 8-15: operand B
        float v = red[threadIdx.x];
        v = fmaxf(v, __shfl_xor(v, 4, 64));
        v = fmaxf(v, __shfl_xor(v, 2, 64));
        v = fmaxf(v, __shfl_xor(v, 1, 64));
        // ... and a workgroup whose value does not exceed what is already there (a plain load: loads pipeline, atomics on one address
        // do not; a stale smaller value only costs the atomic) skips it: the B operand of a one-column-tile product is the same
        // matrix for every workgroup
        if ((threadIdx.x & 7) == 0 && v > 0.0f) {
            unsigned int* slot = reinterpret_cast<unsigned int*>(state + 2 + (threadIdx.x >> 3));
            if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < __float_as_uint(v))
                atomicMax(slot, __float_as_uint(v));
        }
    }
}

// =====================================================================================================================
// dgrad: C[M, N] = epilogue(A[M, K] . B[N, K]^T), A = output gradient rows, B = W^T rows (ops.transpose)
// EPI in {0, E_ADD, E_ADD | E_ADD2, E_GATEBITS, G_ACCUM} (gradient products) and the forward forms of vqcpc_gemm_nt_f16x3:
// E_BIAS, E_BIAS | E_ADD, E_BIAS | E_DROP | E_ADD, E_BIAS | E_RELU | E_MASKOUT, E_BIAS | E_RELU | E_DROP | E_MASKOUT
// G_ACCUM: C += A . B^T by buffer_atomic_add_f32 (no return): the "+ residual" form when the residual already sits in C.  One
// fp32 add per element at the L2, the same value load-add-store gives, and no operand load for the epilogue to wait for (with
// E_ADD all eight waves stall on those loads together at every tile boundary: the slowest epilogue of this kernel).
constexpr int G_ACCUM = 512;
// =====================================================================================================================
// ABL (lab builds, VQCPC_G3_ABL; results are wrong by construction): 1 = no operand requests after the prologue, 2 = no split /
// LDS stores after the prologue, 4 = no MFMAs, 8 = no output stores, 16 = no fragment reads after the prologue
// PL (round 6): bit 0 / bit 1 = operand A / B arrives PRE-SPLIT in the "P4" format -- same shape, leading dimension and bytes as the
// fp32 matrix, every aligned group of four consecutive contraction elements (16 bytes) holding {h0 h1 | h2 h3 | m0 m1 | m2 m3} (fp16
// pairs, element 0 in the low half) under the power-of-two scale of ep.pl_amax_a / _b.  The staging thread that would split its
// float4 writes the two halves of the 16 bytes it loaded straight into the h / m planes: same loads, same LDS image, same products in
// the same order (bit-identical to the fp32-operand kernel under the same scale), no split, no amax in the K loop.  Weights are split
// ONCE per step (vqcpc_weight_planes_many) instead of once per output tile; an activation whose only readers are GEMMs can be
// written in P4 by its producer's epilogue.
template <int EPI, int ABL = 0, int PL = 0>
__global__ __launch_bounds__(kGThreads, 2) void gemm_nt_g3_kernel(const float* __restrict__ A, int64_t lda,
                                                                  const float* __restrict__ B, int64_t ldb,
                                                                  float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                                                                  int tiles_n, int tiles, EpiParams ep,
                                                                  float* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemg[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    // split-K launches (vqcpc_gemm_nt_grad_splitk, EPI == 0): blockIdx.y = K slice -- `K` is the slice length, the operands start K
    // columns further per slice, the partial product goes to plane blockIdx.y of C (ep.split_plane floats apart)
    A += (int64_t)blockIdx.y * K;
    B += (int64_t)blockIdx.y * K;
    C += (int64_t)blockIdx.y * ep.split_plane;
#if VQCPC_LAB
    // measurement (VQCPC_G3_STAGGER = n, lab build): workgroup b starts (b & 7) * n * 512 clocks late, so that the output tiles of the 256
    // persistent workgroups -- equal work, lockstep -- are not all stored at the same moment
    if (ep.act > 0) {
        const int n_ = ((int)blockIdx.x & 7) * ep.act;
        for (int i = 0; i < n_; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const int T = K / kGBK;                              // steps per output tile (even: K % 32 == 0)
    const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my_tiles * T;                          // this workgroup's stream of steps (even)

    const int ea = g3_scale_exp((PL & 1) ? *ep.pl_amax_a : state[0]), eb = g3_scale_exp((PL & 2) ? *ep.pl_amax_b : state[1]);
    const float sa = g3_pow2(ea), sb = g3_pow2(eb), inv = g3_pow2(-(ea + eb));
    float amax_a = 0.0f, amax_b = 0.0f;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // ---- load cursor: one call = the four 16-byte pieces of the NEXT step of the stream, in stream order ----
    const int ld_row = (tid >> 8) * 128 + ((tid & 255) >> 2), ld_c4 = (tid & 3) * 4;
    int ld_tile = blockIdx.x, ld_k = 0;
    const float* a_src;
    const float* b_src;
#define G_SET_SRC()                                                         \
    {                                                                       \
        const int t_ = xcd_swizzle(min(ld_tile, tiles - 1), tiles);         \
        a_src = A + ((int64_t)(t_ / tiles_n) * kG + ld_row) * lda + ld_c4;  \
        b_src = B + ((int64_t)(t_ % tiles_n) * kG + ld_row) * ldb + ld_c4;  \
        if (ABL & 32) {     /* timing only: whole 128-byte lines per request (8 lanes per row), same bytes per two steps */ \
            a_src = A + ((int64_t)(t_ / tiles_n) * kG + (tid >> 3)) * lda + (tid & 7) * 4;  \
            b_src = B + ((int64_t)(t_ % tiles_n) * kG + (tid >> 3)) * ldb + (tid & 7) * 4;  \
        }                                                                   \
    }
    G_SET_SRC()
    // the four pieces of a step are re-requested one by one (each right after it has been split), the cursor moves after the last
#define G_LDK ((ABL & 32) ? (ld_k & ~16) : ld_k)
#define G_OPAQUE(V) asm volatile("" : "+v"(V.x), "+v"(V.y), "+v"(V.z), "+v"(V.w));
#define G_LD_A0(S_) if (!(ABL & 1) || !in_loop) { S_##a0 = *reinterpret_cast<const float4*>(a_src + G_LDK); } else { G_OPAQUE(S_##a0) }
#define G_LD_A1(S_) if (!(ABL & 1) || !in_loop) { S_##a1 = *reinterpret_cast<const float4*>(a_src + (int64_t)64 * lda + G_LDK); } else { G_OPAQUE(S_##a1) }
#define G_LD_B0(S_) if (!(ABL & 1) || !in_loop) { S_##b0 = *reinterpret_cast<const float4*>(b_src + G_LDK); } else { G_OPAQUE(S_##b0) }
#define G_LD_B1(S_) if (!(ABL & 1) || !in_loop) { S_##b1 = *reinterpret_cast<const float4*>(b_src + (int64_t)64 * ldb + G_LDK); } else { G_OPAQUE(S_##b1) }
#define G_ADVANCE()                                                         \
    if (ABL & 32) {                                                         \
        if (ld_k & 16) { a_src -= 128 * lda; b_src -= 128 * ldb; } else { a_src += 128 * lda; b_src += 128 * ldb; } \
    }                                                                       \
    ld_k += kGBK;                                                           \
    if (ld_k == K) {                                                        \
        ld_k = 0;                                                           \
        ld_tile += gridDim.x;            /* past the end: re-reads the last tile, never used */ \
        G_SET_SRC()                                                         \
    }
    float4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;      // two raw sets: steps j+2 (being split) and j+3

    // LDS image of a plane: unpadded 32-byte rows, the two 16-byte chunks of a row XOR-swizzled by bit 3 of the row (the image of
    // gemm_nt_x6_pp_kernel: fragment reads and the staging stores are conflict free)
#define G_ST(R, PLANE0, ROW, SC, AM, WB)                                                                     \
    if (!(ABL & 2) || !in_loop) {                                                                            \
        uint2 h_, m_;                                                                                        \
        if (PL & ((PLANE0) ? 2 : 1)) {           /* P4 operand: the 16 bytes ARE {h pair, h pair, m pair, m pair} */ \
            h_ = make_uint2(__float_as_uint(R.x), __float_as_uint(R.y));                                     \
            m_ = make_uint2(__float_as_uint(R.z), __float_as_uint(R.w));                                     \
        } else {                                                                                             \
            g3_amax4(R, AM);                                                                                 \
            g3_split4(R, SC, h_, m_);                                                                        \
        }                                                                                                    \
        const int o_ = (ROW) * 32 + ((((ld_c4 >> 3) ^ ((ROW) >> 3)) & 1) << 4) + (ld_c4 & 7) * 2;            \
        *reinterpret_cast<uint2*>((WB) + ((PLANE0) + 0) * kGPlane + o_) = h_;                                \
        *reinterpret_cast<uint2*>((WB) + ((PLANE0) + 1) * kGPlane + o_) = m_;                                \
    }
    const int swz = ((kh ^ (li >> 3)) & 1) << 4;
    const int a_off = (wm * 128 + li) * 32 + swz;
    const int b_off = 2 * kGPlane + (wn * 64 + li) * 32 + swz;
    half8 fa[2][2];                                      // [buffer][h | m]: row tile g of the step lives in buffer g & 1
    half8 fb[2][2];                                      // [h | m][column tile]
#define G_MFMA1(MT, NT, PA, PB) \
    if (!(ABL & 4)) acc[MT][NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[(MT) & 1][PA], fb[PB][NT], acc[MT][NT], 0, 0, 0); \
    else { asm volatile("" : "+v"(fa[(MT) & 1][PA]), "+v"(fb[PB][NT])); }
#define G_FENCE() __builtin_amdgcn_sched_barrier(0);
#define G_BARRIER()                                                   \
    __builtin_amdgcn_sched_barrier(0);                                \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue of the output tile with linear index `ep_tile` (buffer addressing; the add operand one 32 x 32 tile ahead) ----
    int ep_tile = blockIdx.x;
    constexpr bool HAS_AUX = (EPI & E_ADD) != 0;
    const int ldci = (int)ldc;
    const int ldxi = (int)ep.ldadd;
    // forward epilogues (round 5: vqcpc_gemm_nt_f16x3): bias, relu, dropout and the bit mask of the positive outputs, exactly as in
    // gemm_nt_x6_pp_kernel (same dropout element index, same mask layout); the tile's two bias values are requested one output tile ahead
    float bias_nx0 = 0.0f, bias_nx1 = 0.0f;
    const uint64_t drop_se = rng_seed_eff(ep.seed);
    const uint32_t drop_sh = (uint32_t)(drop_se >> 32), nc1 = (uint32_t)N * kRngMul;
#define G_BIAS_REQUEST(TILE)                                                                                           \
    if (EPI & E_BIAS) {                                                                                                \
        const int tb_ = xcd_swizzle(min((TILE), tiles - 1), tiles);                                                    \
        const float* bp_ = ep.bias + (tb_ % tiles_n) * kG + wn * 64 + li;                                              \
        bias_nx0 = bp_[0];                                                                                             \
        bias_nx1 = bp_[32];                                                                                            \
    }
    G_BIAS_REQUEST(ep_tile)
#define G_EPILOGUE()                                                                                                   \
    {                                                                                                                  \
        const int t_ = xcd_swizzle(ep_tile, tiles);                                                                    \
        const int64_t m0 = (int64_t)(t_ / tiles_n) * kG;                                                               \
        const int n0 = (t_ % tiles_n) * kG;                                                                            \
        const int64_t row_base = m0 + wm * 128 + 4 * kh;                                                               \
        const int col_base = n0 + wn * 64 + li;                                                                        \
        const float bv_cur0 = bias_nx0, bv_cur1 = bias_nx1;                                                            \
        G_BIAS_REQUEST(ep_tile + (int)gridDim.x)                                                                       \
        const __amdgpu_buffer_rsrc_t rc =                                                                              \
            __builtin_amdgcn_make_buffer_rsrc((void*)(C + m0 * ldc + n0), 0, 0x7FFFFFFF, 0x00020000);                  \
        const int voff_c = ((wm * 128 + 4 * kh) * ldci + wn * 64 + li) * 4;                                            \
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)(HAS_AUX ? ep.add + m0 * (int64_t)ldxi + n0 : C), 0, 0x7FFFFFFF, 0x00020000);                       \
        const int voff_x = ((wm * 128 + 4 * kh) * ldxi + wn * 64 + li) * 4;                                            \
        const int ldx2i = (int)ep.ldadd2;                                                                              \
        const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(                                          \
            (void*)((EPI & E_ADD2) ? ep.add2 + m0 * (int64_t)ldx2i + n0 : C), 0, 0x7FFFFFFF, 0x00020000);              \
        const int voff_x2 = ((wm * 128 + 4 * kh) * ldx2i + wn * 64 + li) * 4;                                          \
        float aux[2][16];                                                                                              \
        if (HAS_AUX) {                                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) aux[0][r] = __builtin_bit_cast(                             \
                float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff_x, (((r & 3) + 8 * (r >> 2)) * ldxi) * 4, 0));    \
        }                                                                                                              \
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(                                           \
            (void*)((EPI & (E_GATEBITS | E_MASKOUT)) ? (void*)ep.mask : (void*)C), 0, 0x7FFFFFFF, 0x00020000);         \
        const int nw16 = (N >> 5) * 16;                                   /* bytes of one 4-row group of mask words */  \
        const int mrow4 = (int)((m0 + wm * 128) >> 2), mcb = (n0 >> 5) + wn * 2;                                       \
        u32x4 gb[2][4];                                                                                                \
        if (EPI & E_GATEBITS) {                                                                                        \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) gb[0][jj] = __builtin_bit_cast(u32x4,                     \
                __builtin_amdgcn_raw_buffer_load_b128(rm, kh * nw16, ((mrow4 + 2 * jj) * (N >> 5) + mcb) * 16, 0));    \
        }                                                                                                              \
        const float gsc = (EPI & E_GATEBITS) ? inv * ep.gate_scale : inv;                                              \
        _Pragma("unroll") for (int tile = 0; tile < 8; ++tile) {                                                       \
            const int mt = tile >> 1, nt = tile & 1;                                                                   \
            /* dropout hash input of this lane's first row of the tile; the other 15 rows are multiples of nc1 away */ \
            const uint32_t x0t = (EPI & E_DROP) ? rng_x0(drop_se, (uint32_t)(row_base + mt * 32 + ep.row0) * (uint32_t)N + \
                                                                      (uint32_t)(col_base + nt * 32)) : 0u;            \
            const float bv = nt ? bv_cur1 : bv_cur0;                                                                   \
            uint32_t mword = 0;                                                                                        \
            if ((EPI & E_GATEBITS) && tile + 1 < 8) {                                                                  \
                const int mt2 = (tile + 1) >> 1, nt2 = (tile + 1) & 1;                                                 \
                _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) gb[(tile + 1) & 1][jj] = __builtin_bit_cast(u32x4,    \
                    __builtin_amdgcn_raw_buffer_load_b128(rm, kh * nw16,                                               \
                        ((mrow4 + mt2 * 8 + 2 * jj) * (N >> 5) + mcb + nt2) * 16, 0));                                 \
            }                                                                                                          \
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
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                           \
                float v = acc[mt][nt][r] * gsc;              /* exact: a power of two (times the gate's 1 / (1 - p)) */ \
                if (EPI & E_BIAS) v += bv;                                                                             \
                if (EPI & E_RELU) v = fmaxf(v, 0.0f);                                                                  \
                if (EPI & E_DROP)   /* == drop_scale(ep.seed, (row + ep.row0) * N + col, ..): thr > 0 on this path */   \
                    v *= rng_u24_from_x0(x0t + (uint32_t)((r & 3) + 8 * (r >> 2)) * nc1, drop_sh) >= ep.thr ? ep.inv_keep : 0.0f; \
                if (EPI & E_GATEBITS) v = ((gb[tile & 1][r >> 2][r & 3] >> li) & 1u) ? v : 0.0f;                      \
                if (EPI & E_ADD) v += aux[tile & 1][r];                                                                \
                if (EPI & E_ADD2) v += a2[r];                                                                          \
                if (EPI & E_MASKOUT) {                                                                                 \
                    const uint64_t bal = __ballot(v > 0.0f);      /* low word: this row for kh = 0, high word: kh = 1 */ \
                    asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"                 \
                                 : "+v"(mword) : "s"((uint32_t)bal), "n"(r), "s"((uint32_t)(bal >> 32)), "n"(16 + r));   \
                }                                                                                                      \
                if (ABL & 8) { asm volatile("" :: "v"(v)); } else if (EPI & G_ACCUM)                                   \
                (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rc, voff_c,                                   \
                                                      ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldci + nt * 32) * 4, 0);   \
                else                                                                                                   \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff_c,                     \
                                                      ((mt * 32 + (r & 3) + 8 * (r >> 2)) * ldci + nt * 32) * 4, 0);   \
                acc[mt][nt][r] = 0.0f;                                                                                 \
            }                                                                                                          \
            if ((EPI & E_MASKOUT) && lane < 32) {    /* lane = r + 16 kh holds the word of row (r & 3) + 8 (r >> 2) + 4 kh */ \
                const int jj = (lane >> 2) & 3, khh = lane >> 4;                                                       \
                __builtin_amdgcn_raw_buffer_store_b32(mword, rm, (2 * jj + khh) * nw16 + (lane & 3) * 4,               \
                                                      ((mrow4 + mt * 8) * (N >> 5) + mcb + nt) * 16, 0);               \
            }                                                                                                          \
        }                                                                                                              \
        ep_tile += gridDim.x;                                                                                          \
    }

    // ---- one step j of the stream (RS = raw set holding step j+2) ----
    // LDS ring of three slots: `cur` holds step j, `rd` step j+1, `wr` receives step j+2.  Registers: ONE set of B fragments
    // and TWO A-fragment buffers (row tile g of the step in buffer g & 1) -- 32 fragment registers instead of the 72 of the
    // six-product kernel -- refilled just in time: A tile g+1 while tile g multiplies (tile 0 of step j+1 during tile 3), the B
    // column-tile-0 fragments of step j+1 after the step's last column-0 MFMA (the 21st), column tile 1 after the last MFMA.
    // The two waves of a SIMD run this stream in near lockstep, so matrix work and everything else only overlap if they
    // alternate INSIDE a wave (with whole 6-MFMA groups followed by the group's staging work the kernel took the SUM of its
    // matrix time and its memory-side time, profiles/r05_g3_ablation_v1.log): every MFMA is followed by ONE small chunk of the
    // step's other work, fenced so that hipcc keeps the placement:
    //   behind MFMAs 0 1 | 3 4 | 6 7 | 9 10: split of elements 0-1 | 2-3 of the four 16-byte pieces of step j+2; behind 2 | 5 | 8 |
    //   11: the piece's two ds_write_b64 and its re-request (step j+4);
    //   behind 0 1 | 6 7 | 12 13: the A fragments of row tiles 1, 2, 3 of step j (plane m first: the next tile's first MFMA reads
    //   it), four to six MFMAs before their first use; behind 18 19: row tile 0 of step j+1; behind 20 / 23 (the last MFMA of
    //   column tile 0 / 1): the B fragments of step j+1;
    //   behind MFMA 16: THE barrier of the step -- after the last store of step j+2 (11) and the last read of step j (13), before
    //   the first read of step j+1 (18; it was stored during step j-1) and before the next step's first store, which lands in the
    //   slot step j occupied.
    uint2 ph, pm;
#define G_P_XY(R, SC, AM, PLF)                                                                               \
    if (PLF) {                                                                                               \
        ph = make_uint2(__float_as_uint(R.x), __float_as_uint(R.y));                                         \
    } else if (!(ABL & 2) || !in_loop) {                                                                     \
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(AM) : "v"(R.x), "v"(R.y));                                \
        g3_split_pair(R.x, R.y, SC, ph.x, pm.x);                                                             \
    }
#define G_P_ZW(R, SC, AM, PLF)                                                                               \
    if (PLF) {                                                                                               \
        pm = make_uint2(__float_as_uint(R.z), __float_as_uint(R.w));                                         \
    } else if (!(ABL & 2) || !in_loop) {                                                                     \
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(AM) : "v"(R.z), "v"(R.w));                                \
        g3_split_pair(R.z, R.w, SC, ph.y, pm.y);                                                             \
    }
#define G_P_WR(PLANE0, ROW, WB)                                                                              \
    if (!(ABL & 2) || !in_loop) {                                                                            \
        const int o_ = (ROW) * 32 + ((((ld_c4 >> 3) ^ ((ROW) >> 3)) & 1) << 4) + (ld_c4 & 7) * 2;            \
        *reinterpret_cast<uint2*>((WB) + ((PLANE0) + 0) * kGPlane + o_) = ph;                                \
        *reinterpret_cast<uint2*>((WB) + ((PLANE0) + 1) * kGPlane + o_) = pm;                                \
    }
    // A fragment (plane PC) of row tile MT -> buffer MT & 1
#define G_RA(MT, PC, RB)                                                                                     \
    if (!(ABL & 16) || !in_loop) fa[(MT) & 1][PC] = *reinterpret_cast<const half8*>((RB) + a_off + (PC) * kGPlane + (MT) * 32 * 32);
    // both planes of the B fragment of column tile NT
#define G_RB(NT, RB)                                                                                         \
    if (!(ABL & 16) || !in_loop) {                                                                           \
        fb[0][NT] = *reinterpret_cast<const half8*>((RB) + b_off + (NT) * 32 * 32);                          \
        fb[1][NT] = *reinterpret_cast<const half8*>((RB) + b_off + kGPlane + (NT) * 32 * 32);                \
    }
#define G_S(MT, NT, PA, PB, ...) G_MFMA1(MT, NT, PA, PB) G_FENCE() __VA_ARGS__ G_FENCE()
#define G_STEP(RS)                                                                                \
    {                                                                                             \
        unsigned char* const cur = smemg + sc * kGSlot;                                           \
        unsigned char* const rd = smemg + sr * kGSlot;                                            \
        unsigned char* const wr = smemg + sw * kGSlot;                                            \
        G_S(0, 0, 1, 0, G_P_XY(RS##a0, sa, amax_a, PL & 1) G_RA(1, 1, cur))                               \
        G_S(0, 0, 0, 1, G_P_ZW(RS##a0, sa, amax_a, PL & 1) G_RA(1, 0, cur))                               \
        G_S(0, 0, 0, 0, G_P_WR(0, ld_row, wr) G_LD_A0(RS))                                        \
        G_S(0, 1, 1, 0, G_P_XY(RS##a1, sa, amax_a, PL & 1))                                               \
        G_S(0, 1, 0, 1, G_P_ZW(RS##a1, sa, amax_a, PL & 1))                                               \
        G_S(0, 1, 0, 0, G_P_WR(0, ld_row + 64, wr) G_LD_A1(RS))                                   \
        G_S(1, 0, 1, 0, G_P_XY(RS##b0, sb, amax_b, PL & 2) G_RA(2, 1, cur))                               \
        G_S(1, 0, 0, 1, G_P_ZW(RS##b0, sb, amax_b, PL & 2) G_RA(2, 0, cur))                               \
        G_S(1, 0, 0, 0, G_P_WR(2, ld_row, wr) G_LD_B0(RS))                                        \
        G_S(1, 1, 1, 0, G_P_XY(RS##b1, sb, amax_b, PL & 2))                                               \
        G_S(1, 1, 0, 1, G_P_ZW(RS##b1, sb, amax_b, PL & 2))                                               \
        G_S(1, 1, 0, 0, G_P_WR(2, ld_row + 64, wr) G_LD_B1(RS) G_ADVANCE())                       \
        G_S(2, 0, 1, 0, G_RA(3, 1, cur))                                                          \
        G_S(2, 0, 0, 1, G_RA(3, 0, cur))                                                          \
        G_S(2, 0, 0, 0, )                                                                         \
        G_S(2, 1, 1, 0, )                                                                         \
        G_MFMA1(2, 1, 0, 1)                                                                       \
        G_BARRIER()                                                                               \
        G_S(2, 1, 0, 0, )                                                                         \
        G_S(3, 0, 1, 0, G_RA(0, 1, rd))                                                           \
        G_S(3, 0, 0, 1, G_RA(0, 0, rd))                                                           \
        G_S(3, 0, 0, 0, G_RB(0, rd))                                                              \
        G_S(3, 1, 1, 0, )                                                                         \
        G_S(3, 1, 0, 1, )                                                                         \
        G_S(3, 1, 0, 0, G_RB(1, rd))                                                              \
        sc = sr;                                                                                  \
        sr = sw;                                                                                  \
        sw = (sw == kGSlots - 1) ? 0 : sw + 1;                                                    \
        if (++kt == T) {                     /* the output tile is complete */                    \
            kt = 0;                                                                               \
            G_EPILOGUE()                                                                          \
            G_FENCE()                                                                             \
        }                                                                                         \
    }

    // prologue: steps 0 and 1 split into slots 0 and 1, steps 2 and 3 requested, first fragments of step 0 read
    int kt = 0, sc = 0, sr = 1, sw = 2;
    bool in_loop = false;
    G_LD_A0(x) G_LD_A1(x) G_LD_B0(x) G_LD_B1(x) G_ADVANCE()
    G_LD_A0(y) G_LD_A1(y) G_LD_B0(y) G_LD_B1(y) G_ADVANCE()
    {
        unsigned char* const w0 = smemg;
        unsigned char* const w1 = smemg + kGSlot;
        G_ST(xa0, 0, ld_row, sa, amax_a, w0) G_ST(xa1, 0, ld_row + 64, sa, amax_a, w0)
        G_ST(xb0, 2, ld_row, sb, amax_b, w0) G_ST(xb1, 2, ld_row + 64, sb, amax_b, w0)
        G_LD_A0(x) G_LD_A1(x) G_LD_B0(x) G_LD_B1(x) G_ADVANCE()
        G_ST(ya0, 0, ld_row, sa, amax_a, w1) G_ST(ya1, 0, ld_row + 64, sa, amax_a, w1)
        G_ST(yb0, 2, ld_row, sb, amax_b, w1) G_ST(yb1, 2, ld_row + 64, sb, amax_b, w1)
        G_LD_A0(y) G_LD_A1(y) G_LD_B0(y) G_LD_B1(y) G_ADVANCE()
        G_BARRIER()
        G_RA(0, 0, w0) G_RA(0, 1, w0) G_RB(0, w0) G_RB(1, w0)
    }
    in_loop = true;
#pragma unroll 1
    for (int s = 0; s < S; s += 2) {
        G_STEP(x)
        G_STEP(y)
    }
    // the clamped run-ahead requests re-read the last tile: amax of real data only
    g3_amax_publish_wg(amax_a, amax_b, state, smemg);
#undef G_STEP
#undef G_S
#undef G_RB
#undef G_RA
#undef G_P_WR
#undef G_P_ZW
#undef G_P_XY
#undef G_EPILOGUE
#undef G_BIAS_REQUEST
#undef G_BARRIER
#undef G_FENCE
#undef G_MFMA1
#undef G_ST
#undef G_ADVANCE
#undef G_LD_B1
#undef G_LD_B0
#undef G_LD_A1
#undef G_LD_A0
#undef G_OPAQUE
#undef G_LDK
#undef G_SET_SRC
}

// =====================================================================================================================
// wgrad: dW[N, K] = A[M, N]^T . B[M, K] (+ column sums of A), contraction over the M rows split across blockIdx.y
// LDS image of gemm_tn_x6_pq_kernel (a staging thread owns four consecutive rows of four columns of one operand; a fragment
// is two conflict-free ds_read_b64), partial sums into the workspace, deterministic reduction by the caller.
// =====================================================================================================================
constexpr int kQES = 2048 + 64;                      // bytes per (col & 3) block: [m >> 3][(m >> 2) & 1][64 slots] x 8 B, + 64 B
constexpr int kQPlane = 4 * kQES;                    // 8448 B
constexpr int kQSlot = 4 * kQPlane;                  // A.h | A.m | B.h | B.m
constexpr int kQTM = 16;                             // rows of the contraction per step

__global__ __launch_bounds__(kGThreads, 2) void gemm_tn_g3_kernel(const float* __restrict__ A, int64_t lda,
                                                                  const float* __restrict__ B, int64_t ldb, int64_t M, int N,
                                                                  int K, int tiles_k, int64_t rows_per_split,
                                                                  float* __restrict__ ws, float* __restrict__ ws_bias,
                                                                  float* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemq[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    // (tile, split) of this workgroup: the tiles of ONE split are neighbours on one XCD (see gemm_tn_x6_pq_kernel)
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.x > 1 && gridDim.y % 8 == 0) {
        const int L = by * (int)gridDim.x + bx;
        const int j = L >> 3;
        bx = j % (int)gridDim.x;
        by = (j / (int)gridDim.x) * 8 + (L & 7);
    }
    const int tn = bx / tiles_k, tk = bx % tiles_k;
    const int n0 = tn * kG, k0 = tk * kG;
    const int64_t m_begin = (int64_t)by * rows_per_split;
    const int64_t m_end = min(m_begin + rows_per_split, M);          // (m_end - m_begin) % 32 == 0 (host)
    const bool want_bias = (ws_bias != nullptr) && tk == 0;

    const int ea = g3_scale_exp(state[0]), eb = g3_scale_exp(state[1]);
    const float inv = g3_pow2(-(ea + eb));
    float amax = 0.0f;

    floatx16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging: thread = (operand, row quad rq of 4, column quad c4 of 64): rows 4 rq .. 4 rq + 3 of the 16-row step
    const int c4 = (tid & 63) * 4, rq = (tid >> 6) & 3, opnd = tid >> 8;          // opnd is uniform per wave
    const float sc = g3_pow2(opnd ? eb : ea);
    const int64_t ld_s = opnd ? ldb : lda;
    const float* s_src = (opnd ? B + k0 : A + n0) + (m_begin + 4 * rq) * ld_s + c4;
    const int st_off = opnd * 2 * kQPlane + (rq >> 1) * 1024 + (rq & 1) * 512 + (c4 >> 2) * 8;
    const int64_t nsteps = (m_end - m_begin) / kQTM;                              // even
    float4 x0, x1, x2, x3, y0, y1, y2, y3;                                        // raw sets: steps j+2 and j+3
    int64_t ld_step = 0;                                                          // next step to request (clamped to the last)
#define Q_LD(S_, R)                                                                                          \
    S_##R = *reinterpret_cast<const float4*>(s_src + (min(ld_step, nsteps - 1) * kQTM + (R)) * ld_s);
    // amax and the bias sums read the raw registers; `counts_` is false for the clamped run-ahead steps past the end
#define Q_STATS(S_, counts_)                                                                                 \
    if (counts_) {                                                                                           \
        g3_amax4(S_##0, amax); g3_amax4(S_##1, amax); g3_amax4(S_##2, amax); g3_amax4(S_##3, amax);          \
        if (want_bias && opnd == 0) {                                                                        \
            bsum.x += ((S_##0).x + (S_##1).x) + ((S_##2).x + (S_##3).x);                                     \
            bsum.y += ((S_##0).y + (S_##1).y) + ((S_##2).y + (S_##3).y);                                     \
            bsum.z += ((S_##0).z + (S_##1).z) + ((S_##2).z + (S_##3).z);                                     \
            bsum.w += ((S_##0).w + (S_##1).w) + ((S_##2).w + (S_##3).w);                                     \
        }                                                                                                    \
    }
    half8 fa[2][2], fb[2][2];                            // as in gemm_nt_g3_kernel: two A buffers, one B set
    const int fr_off = (li & 3) * kQES + kh * 1024 + (li >> 2) * 8;
#define Q_FRAG(DST, P)                                                                                       \
    {                                                                                                        \
        const uint2 p0_ = *reinterpret_cast<const uint2*>(P);                                                \
        const uint2 p1_ = *reinterpret_cast<const uint2*>((P) + 512);                                        \
        const u32x4 u_ = {p0_.x, p0_.y, p1_.x, p1_.y};                                                       \
        DST = __builtin_bit_cast(half8, u_);                                                                 \
    }
#define Q_RA(MT, PC, RB) Q_FRAG(fa[(MT) & 1][PC], (RB) + fr_off + (PC) * kQPlane + wm * 256 + (MT) * 64)
#define Q_RB(NT, RB)                                                                                         \
    Q_FRAG(fb[0][NT], (RB) + fr_off + 2 * kQPlane + wn * 128 + (NT) * 64)                                    \
    Q_FRAG(fb[1][NT], (RB) + fr_off + 3 * kQPlane + wn * 128 + (NT) * 64)
#define Q_MFMA1(MT, NT, PA, PB) \
    acc[MT][NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[(MT) & 1][PA], fb[PB][NT], acc[MT][NT], 0, 0, 0);
#define Q_FENCE() __builtin_amdgcn_sched_barrier(0);
#define Q_BARRIER()                                                   \
    __builtin_amdgcn_sched_barrier(0);                                \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);
    // column E of the raw set: rows 0-1 | rows 2-3 | the two stores
    uint2 ph, pm;
#define Q_C01(S_, E) g3_split_pair((S_##0).E, (S_##1).E, sc, ph.x, pm.x);
#define Q_C23(S_, E) g3_split_pair((S_##2).E, (S_##3).E, sc, ph.y, pm.y);
#define Q_CWR(EI, WB)                                                                                        \
    {                                                                                                        \
        unsigned char* d_ = (WB) + st_off + (EI) * kQES;                                                     \
        *reinterpret_cast<uint2*>(d_) = ph;                                                                  \
        *reinterpret_cast<uint2*>(d_ + kQPlane) = pm;                                                        \
    }
#define Q_COL(S_, E, EI, WB) Q_C01(S_, E) Q_C23(S_, E) Q_CWR(EI, WB)
#define Q_S(MT, NT, PA, PB, ...) Q_MFMA1(MT, NT, PA, PB) Q_FENCE() __VA_ARGS__ Q_FENCE()
    // step j: the pipeline and chunk placement of gemm_nt_g3_kernel (`cur` / `rd` / `wr` = slots of steps j / j+1 / j+2); the four
    // rows of the raw set are re-requested together once its fourth column has been split
#define Q_STEP(RS)                                                                                \
    {                                                                                             \
        unsigned char* const cur = smemq + sc_ * kQSlot;                                          \
        unsigned char* const rd = smemq + sr * kQSlot;                                            \
        unsigned char* const wr = smemq + sw * kQSlot;                                            \
        const bool counts_ = st + 2 < nsteps;                                                     \
        Q_S(0, 0, 1, 0, Q_C01(RS, x) Q_RA(1, 1, cur))                                             \
        Q_S(0, 0, 0, 1, Q_C23(RS, x) Q_RA(1, 0, cur))                                             \
        Q_S(0, 0, 0, 0, Q_CWR(0, wr))                                                             \
        Q_S(0, 1, 1, 0, Q_C01(RS, y))                                                             \
        Q_S(0, 1, 0, 1, Q_C23(RS, y))                                                             \
        Q_S(0, 1, 0, 0, Q_CWR(1, wr))                                                             \
        Q_S(1, 0, 1, 0, Q_C01(RS, z) Q_RA(2, 1, cur))                                             \
        Q_S(1, 0, 0, 1, Q_C23(RS, z) Q_RA(2, 0, cur))                                             \
        Q_S(1, 0, 0, 0, Q_CWR(2, wr))                                                             \
        Q_S(1, 1, 1, 0, Q_C01(RS, w))                                                             \
        Q_S(1, 1, 0, 1, Q_C23(RS, w))                                                             \
        Q_S(1, 1, 0, 0, Q_CWR(3, wr))                                                             \
        Q_S(2, 0, 1, 0, Q_STATS(RS, counts_) Q_RA(3, 1, cur))                                     \
        Q_S(2, 0, 0, 1, Q_LD(RS, 0) Q_LD(RS, 1) Q_LD(RS, 2) Q_LD(RS, 3) ++ld_step; Q_RA(3, 0, cur)) \
        Q_S(2, 0, 0, 0, )                                                                         \
        Q_S(2, 1, 1, 0, )                                                                         \
        Q_MFMA1(2, 1, 0, 1)                                                                       \
        Q_BARRIER()                                                                               \
        Q_S(2, 1, 0, 0, )                                                                         \
        Q_S(3, 0, 1, 0, Q_RA(0, 1, rd))                                                           \
        Q_S(3, 0, 0, 1, Q_RA(0, 0, rd))                                                           \
        Q_S(3, 0, 0, 0, Q_RB(0, rd))                                                              \
        Q_S(3, 1, 1, 0, )                                                                         \
        Q_S(3, 1, 0, 1, )                                                                         \
        Q_S(3, 1, 0, 0, Q_RB(1, rd))                                                              \
        sc_ = sr;                                                                                 \
        sr = sw;                                                                                  \
        sw = (sw == kGSlots - 1) ? 0 : sw + 1;                                                    \
        ++st;                                                                                     \
    }

    if (nsteps > 0) {
        int sc_ = 0, sr = 1, sw = 2;
        int64_t st = 0;
        Q_LD(x, 0) Q_LD(x, 1) Q_LD(x, 2) Q_LD(x, 3) ++ld_step;
        Q_LD(y, 0) Q_LD(y, 1) Q_LD(y, 2) Q_LD(y, 3) ++ld_step;
        {
            unsigned char* const w0 = smemq;
            unsigned char* const w1 = smemq + kQSlot;
            Q_STATS(x, true)
            Q_COL(x, x, 0, w0) Q_COL(x, y, 1, w0) Q_COL(x, z, 2, w0) Q_COL(x, w, 3, w0)
            Q_LD(x, 0) Q_LD(x, 1) Q_LD(x, 2) Q_LD(x, 3) ++ld_step;
            Q_STATS(y, true)                                   /* nsteps >= 2 (even) */
            Q_COL(y, x, 0, w1) Q_COL(y, y, 1, w1) Q_COL(y, z, 2, w1) Q_COL(y, w, 3, w1)
            Q_LD(y, 0) Q_LD(y, 1) Q_LD(y, 2) Q_LD(y, 3) ++ld_step;
            Q_BARRIER()
            Q_RA(0, 0, w0) Q_RA(0, 1, w0) Q_RB(0, w0) Q_RB(1, w0)
        }
#pragma unroll 1
        while (st < nsteps) {
            Q_STEP(x)
            Q_STEP(y)
        }
    }
#undef Q_STEP
#undef Q_S
#undef Q_COL
#undef Q_CWR
#undef Q_C23
#undef Q_C01
#undef Q_BARRIER
#undef Q_FENCE
#undef Q_MFMA1
#undef Q_RB
#undef Q_RA
#undef Q_FRAG
#undef Q_STATS
#undef Q_LD

    float* out = ws + (int64_t)by * N * K;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = k0 + wn * 64 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                out[(int64_t)row * K + col] = acc[mt][nt][r] * inv;
            }
        }
    }
    // each operand's amax goes to its own slot (the waves of one operand are uniform)
    g3_amax_publish_wg(opnd ? 0.0f : amax, opnd ? amax : 0.0f, state, smemq);
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smemq);          // [4 row quads][256]
        if (opnd == 0) *reinterpret_cast<float4*>(red + rq * kG + c4) = bsum;
        __syncthreads();
        if (tid < kG) {
            float tot = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) tot += red[g * kG + tid];
            ws_bias[(int64_t)by * N + n0 + tid] = tot;
        }
    }
}

// Sum of the K-slice planes of a split-K launch (ascending slice order: deterministic) + the epilogue of the product: bias, dropout
// (element index (row0 + row) * N + col, as everywhere), residuals.  One float4 per thread.
__global__ __launch_bounds__(256) void g3_splitk_epilogue_kernel(const float* __restrict__ ws, int64_t plane, int splits,
                                                                 float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                 const float* __restrict__ bias, uint32_t thr, float inv_keep,
                                                                 uint64_t seed, int64_t row0, const float* add, int64_t ldadd,
                                                                 const float* add2, int64_t ldadd2) {
    const int n4 = N >> 2;
    const int64_t total = M * n4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / n4;
        const int col = (int)(i - row * n4) * 4;
        float4 v = *reinterpret_cast<const float4*>(ws + row * N + col);
        for (int z = 1; z < splits; ++z) {
            const float4 p = *reinterpret_cast<const float4*>(ws + z * plane + row * N + col);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (thr) {
            const uint64_t e = (uint64_t)(row0 + row) * N + col;
            v.x *= drop_scale(seed, e + 0, thr, inv_keep);
            v.y *= drop_scale(seed, e + 1, thr, inv_keep);
            v.z *= drop_scale(seed, e + 2, thr, inv_keep);
            v.w *= drop_scale(seed, e + 3, thr, inv_keep);
        }
        if (add) {       // may be C itself (the residual accumulated in place): read before the store below, same thread
            const float4 a = *reinterpret_cast<const float4*>(add + row * ldadd + col);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (add2) {
            const float4 a = *reinterpret_cast<const float4*>(add2 + row * ldadd2 + col);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(C + row * ldc + col) = v;
    }
}

// =====================================================================================================================
// The TAIL rows of a ragged launch (139 264 x 256 = 2.125 rounds of 256-tiles on 256 persistent workgroups: two whole rounds on
// gemm_nt_g3_kernel, the last 8 192 rows here): 64 x 128 output tiles, one per workgroup, so that the few rows left still occupy
// every CU (8 192 x 256 -> 256 workgroups) without partial planes (the split-K remainder parks 64 MB per launch) and without a
// third round of 256-tiles.  Same arithmetic, same order of the three products per 16 contraction elements and the same epilogue
// expression as gemm_nt_g3_kernel: a row computed here carries the bits the 256-tile kernel gives it.
// 8 waves of 32 x 32 (wm = row half, wn = column quarter), K steps of 32 = two 16-element halves in the plane image of the
// 256-tile kernel (32-byte rows, 16-byte chunks XOR-swizzled by bit 3 of the row), two LDS slots, one barrier per step: fragments
// of step j, split + LDS write of step j+1 (requested during step j-1), request of step j+2, six MFMAs.  Not a tuned schedule:
// this kernel runs 1/17 of a ragged launch's rows.
// =====================================================================================================================
constexpr int kRM = 64, kRN = 128, kRBK = 32;
constexpr int kRPA = kRM * 32, kRPB = kRN * 32;          // bytes per A / B plane of one 16-element half
constexpr int kRHalf = 2 * kRPA + 2 * kRPB + 128;        // A.h | A.m | B.h | B.m (+ 128 B: the two halves of a row on different banks)
constexpr int kRSlot = 2 * kRHalf;

__global__ __launch_bounds__(kGThreads) void gemm_nt_g3_tail_kernel(const float* __restrict__ A, int64_t lda,
                                                                   const float* __restrict__ B, int64_t ldb,
                                                                   float* C, int64_t ldc, int N, int K, int tiles_n,     // C may be ep.add
                                                                   EpiParams ep, float* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemr[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, kh = lane >> 5;
    const int t = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int64_t m0 = (int64_t)(t / tiles_n) * kRM;
    const int n0 = (t % tiles_n) * kRN;
    const int T = K / kRBK;

    const bool pla = ep.pl_amax_a != nullptr, plb = ep.pl_amax_b != nullptr;        // P4 operands (see gemm_nt_g3_kernel)
    const int ea = g3_scale_exp(pla ? *ep.pl_amax_a : state[0]), eb = g3_scale_exp(plb ? *ep.pl_amax_b : state[1]);
    const float sa = g3_pow2(ea), sb = g3_pow2(eb), inv = g3_pow2(-(ea + eb));
    float amax_a = 0.0f, amax_b = 0.0f;

    // staging: 8 lanes cover the 128 bytes of one row's K step; q = 16-byte piece, q >> 2 = half, (q & 3) * 4 = first element
    const int s_row = tid >> 3, q = tid & 7, c4 = (q & 3) * 4;
    const float* a_src = A + (m0 + s_row) * lda + q * 4;
    const float* b_src = B + ((int64_t)n0 + s_row) * ldb + q * 4;
    const int64_t b_src1 = (int64_t)64 * ldb;
    const int st_half = (q >> 2) * kRHalf;
    const int o_a = s_row * 32 + ((((c4 >> 3) ^ (s_row >> 3)) & 1) << 4) + (c4 & 7) * 2;      // rows s_row and s_row + 64: same swizzle bit
    const int st_a = st_half + o_a, st_b0 = st_half + 2 * kRPA + o_a, st_b1 = st_b0 + 64 * 32;
    // raw operand registers: a ring of THREE sets, requested three steps ahead (round 6).  A launch of this kernel has one to three
    // workgroups per CU (3 072 x 512: 192 tiles on 256 CUs): with the one-step run-ahead of round 5 every K step waited out most of
    // an HBM round trip (0.75 us per step for 192 cycles of matrix work per wave, tools/bench_small_f16x3.py).
    struct Raw { float4 a, b0, b1; };
    Raw r0, r1, r2;
#define R_LOAD(S, KT)                                                                      \
    {                                                                                      \
        S.a = *reinterpret_cast<const float4*>(a_src + (int64_t)(KT) * kRBK);              \
        S.b0 = *reinterpret_cast<const float4*>(b_src + (int64_t)(KT) * kRBK);             \
        S.b1 = *reinterpret_cast<const float4*>(b_src + b_src1 + (int64_t)(KT) * kRBK);    \
    }
#define R_SPLIT(R, SC, AM, ISPL)                                                           \
    if (ISPL) {                                                                            \
        h_ = make_uint2(__float_as_uint(R.x), __float_as_uint(R.y));                       \
        m_ = make_uint2(__float_as_uint(R.z), __float_as_uint(R.w));                       \
    } else {                                                                               \
        g3_amax4(R, AM);                                                                   \
        g3_split4(R, SC, h_, m_);                                                          \
    }
#define R_STORE(WB, S)                                                                     \
    {                                                                                      \
        uint2 h_, m_;                                                                      \
        R_SPLIT(S.a, sa, amax_a, pla)                                                      \
        *reinterpret_cast<uint2*>((WB) + st_a) = h_;                                       \
        *reinterpret_cast<uint2*>((WB) + st_a + kRPA) = m_;                                \
        R_SPLIT(S.b0, sb, amax_b, plb)                                                     \
        *reinterpret_cast<uint2*>((WB) + st_b0) = h_;                                      \
        *reinterpret_cast<uint2*>((WB) + st_b0 + kRPB) = m_;                               \
        R_SPLIT(S.b1, sb, amax_b, plb)                                                     \
        *reinterpret_cast<uint2*>((WB) + st_b1) = h_;                                      \
        *reinterpret_cast<uint2*>((WB) + st_b1 + kRPB) = m_;                               \
    }
    const int swz = ((kh ^ (li >> 3)) & 1) << 4;
    const int fa_off = (wm * 32 + li) * 32 + swz;
    const int fb_off = 2 * kRPA + (wn * 32 + li) * 32 + swz;

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    // step s lives in register set s % 3: step J multiplies slot J & 1, splits + stores step J + 1 (set SN) into the other slot and
    // re-requests step J + 4 into the same set
#define R_STEP(J, SN)                                                                      \
    {                                                                                      \
        const unsigned char* cur = smemr + ((J) & 1) * kRSlot;                             \
        unsigned char* nxt = smemr + (((J) + 1) & 1) * kRSlot;                             \
        half8 fa[2][2], fb[2][2];                        /* [half][h | m] */               \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                 \
            fa[hf][0] = *reinterpret_cast<const half8*>(cur + hf * kRHalf + fa_off);       \
            fa[hf][1] = *reinterpret_cast<const half8*>(cur + hf * kRHalf + fa_off + kRPA); \
            fb[hf][0] = *reinterpret_cast<const half8*>(cur + hf * kRHalf + fb_off);       \
            fb[hf][1] = *reinterpret_cast<const half8*>(cur + hf * kRHalf + fb_off + kRPB); \
        }                                                                                  \
        if ((J) + 1 < T) {                                                                 \
            R_STORE(nxt, SN)                                                               \
            if ((J) + 4 < T) R_LOAD(SN, (J) + 4)                                           \
        }                                                                                  \
        _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {   /* m.h, h.m, h.h: the order of gemm_nt_g3_kernel */ \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[hf][1], fb[hf][0], acc, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[hf][0], fb[hf][1], acc, 0, 0, 0); \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[hf][0], fb[hf][0], acc, 0, 0, 0); \
        }                                                                                  \
        __syncthreads();                                                                   \
    }

    R_LOAD(r0, 0)
    if (T > 1) R_LOAD(r1, 1)
    if (T > 2) R_LOAD(r2, 2)
    R_STORE(smemr, r0)
    if (T > 3) R_LOAD(r0, 3)
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < T; j += 3) {
        R_STEP(j, r1)
        if (j + 1 < T) R_STEP(j + 1, r2)
        if (j + 2 < T) R_STEP(j + 2, r0)
    }
#undef R_STEP
#undef R_STORE
#undef R_SPLIT
#undef R_LOAD
    g3_amax_publish_wg(amax_a, amax_b, state, smemr);

    // epilogue: lane (li, kh) holds column li of rows 4 kh + (r & 3) + 8 (r >> 2) of the wave's 32 x 32 block
    const int col = n0 + wn * 32 + li;
    const int64_t row_base = m0 + wm * 32 + 4 * kh;
    const float bv = ep.bias ? ep.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row_base + (r & 3) + 8 * (r >> 2);
        float v = acc[r] * inv;                          // exact: a power of two
        if (ep.bias) v += bv;
        if (ep.act == 1) v = fmaxf(v, 0.0f);             // (round 6: the epilogue order of gemm_nt_kernel, gemm.hip)
        if (ep.thr) v *= drop_scale(ep.seed, (uint64_t)(row + ep.row0) * (uint64_t)N + (uint64_t)col, ep.thr, ep.inv_keep);
        if (ep.gate) v *= (ep.gate[row * ep.ldgate + col] > 0.0f ? ep.gate_scale : 0.0f);
        if (ep.add) v += ep.add[row * ep.ldadd + col];   // may be C itself (in place): read and written by this lane only
        if (ep.add2) v += ep.add2[row * ep.ldadd2 + col];
        C[row * ldc + col] = v;
    }
}

// amax of a (rows, cols) fp32 matrix with row stride ld -> atomic max into *slot (priming of a call site's state)
__global__ __launch_bounds__(256) void grad_amax_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int cols,
                                                        float* __restrict__ slot) {
    const int c4n = cols >> 2;
    const int64_t total = rows * c4n;
    float amax = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        g3_amax4(v, amax);
    }
    g3_amax_publish(amax, slot);
}

// per call site: {read A, read B, written A, written B}: read <- written (a site that did not run keeps its value), written <- 0
// `saturated` (may be NULL): counts the (site, operand) pairs whose amax of THIS step lay beyond the fp16 range under the scale the
// step used (the elements above 65504 / scale were clamped there): the caller's monitor of a scale that lagged too far
__global__ __launch_bounds__(256) void grad_scale_roll_kernel(float* __restrict__ state, int nsites,
                                                              unsigned int* __restrict__ saturated) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsites * 2) return;
    float* s = state + (i >> 1) * 4 + (i & 1);
    const float w = s[2];
    if (saturated && w * g3_pow2(g3_scale_exp(s[0])) > 65504.0f) atomicAdd(saturated, 1u);
    if (w > 0.0f) s[0] = w;
    s[2] = 0.0f;
}

// The same roll for up to 512 sites in ONE workgroup, with a log: monitor[0] += saturated (site, operand) pairs (as above), monitor[1]
// = rolls of this table so far (its step index), monitor[2] = steps with at least one saturated pair, monitor[3 + k] = step index of
// the k-th such step (the first `log_capacity` of them): what the trainers report at the end of an epoch.
__global__ __launch_bounds__(1024) void grad_scale_roll_logged_kernel(float* __restrict__ state, int nsites, int* __restrict__ monitor,
                                                                      int log_capacity) {
    __shared__ int any_sat;
    if (threadIdx.x == 0) any_sat = 0;
    __syncthreads();
    const int i = threadIdx.x;
    if (i < nsites * 2) {
        float* s = state + (i >> 1) * 4 + (i & 1);
        const float w = s[2];
        if (w * g3_pow2(g3_scale_exp(s[0])) > 65504.0f) {
            atomicAdd(monitor, 1);
            any_sat = 1;
        }
        if (w > 0.0f) s[0] = w;
        s[2] = 0.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int step = monitor[1];
        if (any_sat) {
            const int k = monitor[2];
            if (k < log_capacity) monitor[3 + k] = step;
            monitor[2] = k + 1;
        }
        monitor[1] = step + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weights as P4 planes, ONCE per step (round 6).  Matrix i lives at base + desc[4 i] (rows desc[4 i + 1], cols desc[4 i + 2], index of
// its first 32 x 32 tile in the launch desc[4 i + 3]: the descriptor table of vqcpc_transpose_many).  Pass 1: amax[i] = max |W_i| of
// THIS step (atomic max of bit patterns; the caller zeroes `amax` first).  Pass 2: the P4 image of W_i (groups of four along the columns
// = the contraction of the FORWARD product x W^T) at the same offset of `planes`, and the P4 image of W_i^T (groups along the rows = the
// contraction of the INPUT-GRADIENT product dy W) at the same offset of `planes_t` (the layout of the transposed-weight arena), both
// under the power-of-two scale of amax[i].  The weights of a step are final when its forward begins: the scale is exact, never stale.
__global__ __launch_bounds__(256) void weight_amax_many_kernel(const float* __restrict__ base, const int64_t* __restrict__ desc, int n,
                                                               float* __restrict__ tile_max) {
    int lo = 0, hi = n - 1;
    const int64_t t = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[4 * mid + 3] <= t) lo = mid; else hi = mid - 1;
    }
    const int R = (int)desc[4 * lo + 1], C = (int)desc[4 * lo + 2];
    const int local = (int)(t - desc[4 * lo + 3]), tiles_c = (C + 31) / 32;
    const float* in = base + desc[4 * lo];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = (local % tiles_c) * 32 + tx, r0 = (local / tiles_c) * 32;
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + k * 8;
        if (r < R && c < C) a = fmaxf(a, fabsf(in[(int64_t)r * C + c]));
    }
    // no atomics: one value per 32 x 32 tile; the plane pass reduces its matrix's tiles itself (an atomic max per workgroup on one
    // address per matrix took 67-210 us per step here: atomics on one address retire one by one, profiles/r06_perf_log.md)
    __shared__ float red[4];
    a = wave_max(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) tile_max[t] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void weight_planes_many_kernel(const float* __restrict__ base, const int64_t* __restrict__ desc, int n,
                                                                 const float* __restrict__ tile_max, int64_t total_tiles,
                                                                 float* __restrict__ amax, uint4* __restrict__ planes,
                                                                 uint4* __restrict__ planes_t) {
    __shared__ float tile[32][33];
    __shared__ float red[4];
    int lo = 0, hi = n - 1;
    const int64_t t = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[4 * mid + 3] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t off = desc[4 * lo];
    const int R = (int)desc[4 * lo + 1], C = (int)desc[4 * lo + 2];
    const int local = (int)(t - desc[4 * lo + 3]), tiles_c = (C + 31) / 32;
    // amax of the matrix = max over its tiles' maxima (the previous pass), the same value in every workgroup of the matrix
    float am = 0.0f;
    {
        const int64_t t0 = desc[4 * lo + 3], t1 = (lo + 1 < n) ? desc[4 * (lo + 1) + 3] : total_tiles;
        for (int64_t i = t0 + threadIdx.x; i < t1; i += 256) am = fmaxf(am, tile_max[i]);
        am = wave_max(am);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
        __syncthreads();
        am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (local == 0 && threadIdx.x == 0) amax[lo] = am;              // what the GEMM kernels derive the same scale from
    }
    const float sc = g3_pow2(g3_scale_exp(am));
    const float* in = base + off;
    const int c0 = (local % tiles_c) * 32, r0 = (local / tiles_c) * 32;
    const int g = threadIdx.x & 7, row = threadIdx.x >> 3;            // 32 rows x 8 groups of four columns
    const bool fwd_ok = (C & 3) == 0, bwd_ok = (R & 3) == 0;
    {
        const int r = r0 + row, c = c0 + 4 * g;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fwd_ok && r < R && c < C) v = *reinterpret_cast<const float4*>(in + (int64_t)r * C + c);      // (off % 4 == 0: host)
        else if (r < R) {
            if (c < C) v.x = in[(int64_t)r * C + c];
            if (c + 1 < C) v.y = in[(int64_t)r * C + c + 1];
            if (c + 2 < C) v.z = in[(int64_t)r * C + c + 2];
            if (c + 3 < C) v.w = in[(int64_t)r * C + c + 3];
        }
        tile[row][4 * g] = v.x; tile[row][4 * g + 1] = v.y; tile[row][4 * g + 2] = v.z; tile[row][4 * g + 3] = v.w;
        if (fwd_ok && r < R && c < C) {
            uint2 h, m;
            g3_split4(v, sc, h, m);
            planes[(off + (int64_t)r * C + c) >> 2] = make_uint4(h.x, h.y, m.x, m.y);
        }
    }
    __syncthreads();
    if (bwd_ok) {                                   // W^T: row c0 + col of the transpose, group of four ORIGINAL rows r0 + 4 g ..
        const int col = threadIdx.x >> 3;
        const int c = c0 + col, r = r0 + 4 * g;
        if (c < C && r < R) {
            const float4 v = make_float4(tile[4 * g][col], tile[4 * g + 1][col], tile[4 * g + 2][col], tile[4 * g + 3][col]);
            uint2 h, m;
            g3_split4(v, sc, h, m);
            planes_t[(off + (int64_t)c * R + r) >> 2] = make_uint4(h.x, h.y, m.x, m.y);
        }
    }
}

static int tn_g3_splits(int64_t M, int N, int K) {
    const int64_t tiles = (int64_t)(N / kG) * (K / kG);
    int64_t s = std::max<int64_t>(1, kNumCU / tiles);
    s = std::min<int64_t>(s, std::max<int64_t>(1, M / 256));
    return (int)s;
}

}  // namespace vq

using namespace vq;

extern "C" {

// dgrad shapes the three-product kernel takes: whole 256 x 256 tiles, K a multiple of 32.  (Whether the tiles fill the 256
// persistent workgroups well enough is the caller's policy: ops.py cuts ragged launches by rows.)
int vqcpc_gemm_nt_grad_supported(int64_t M, int N, int K) {
    return (M >= kG && (M % kG) == 0 && N >= kG && (N % kG) == 0 && K >= 32 && (K % 32) == 0 && (M / kG) * (N / kG) < (1ll << 30)) ? 1 : 0;
}

int vqcpc_gemm_nt_grad(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                       const float* add, int64_t ldadd, const float* add2, int64_t ldadd2, const void* gate_mask,
                       float gate_scale, float* scale_state, void* stream) {
    VQ_REQUIRE(A && B && C && scale_state, "gemm_nt_grad: null pointer");
    VQ_REQUIRE(vqcpc_gemm_nt_grad_supported(M, N, K), "gemm_nt_grad: M, N multiples of 256 and K of 32, got M=%lld N=%d K=%d",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && aligned16(A) && aligned16(B),
               "gemm_nt_grad: bad leading dimensions / alignment");
    VQ_REQUIRE(!(add2 && !add) && !(gate_mask && add), "gemm_nt_grad: epilogue is one of none / add / add + add2 / gate mask");
    VQ_REQUIRE(!gate_mask || aligned16(gate_mask), "gemm_nt_grad: gate mask must be 16-byte aligned");
    EpiParams ep{};
    ep.add = add;
    ep.ldadd = ldadd;
    ep.add2 = add2;
    ep.ldadd2 = ldadd2;
    ep.mask = (uint32_t*)const_cast<void*>(gate_mask);
    ep.gate_scale = gate_scale;
#if VQCPC_LAB
    {
        static const int stagger = lab_env_int("VQCPC_G3_STAGGER", 0);
        ep.act = stagger;
    }
#endif
    const int tn = N / kG;
    const int tiles = (int)((M / kG) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kGThreads);
    const size_t lds = (size_t)kGSlots * kGSlot;
    hipStream_t st = (hipStream_t)stream;
#define G3_LAUNCH(EPIV)                                                                                                  \
    {                                                                                                                    \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            (void)hipFuncSetAttribute((const void*)gemm_nt_g3_kernel<EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                      (int)lds);                                                                         \
            attr = true;                                                                                                 \
        }                                                                                                                \
        hipLaunchKernelGGL((gemm_nt_g3_kernel<EPIV>), grid, block, lds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles,  \
                           ep, scale_state);                                                                             \
    }
#if VQCPC_LAB
    {
        static const int abl = lab_env_int("VQCPC_G3_ABL", 0);
        if (abl && !gate_mask && !add) {
#define G3_ABL_CASE(V)                                                                                                   \
    if (abl == V) {                                                                                                      \
        (void)hipFuncSetAttribute((const void*)gemm_nt_g3_kernel<0, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_nt_g3_kernel<0, V>), grid, block, lds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, ep, \
                           scale_state);                                                                                 \
    }
            G3_ABL_CASE(1) G3_ABL_CASE(2) G3_ABL_CASE(4) G3_ABL_CASE(8) G3_ABL_CASE(16) G3_ABL_CASE(19) G3_ABL_CASE(32)
#undef G3_ABL_CASE
            VQ_CHECK_LAUNCH("gemm_nt_g3 (ablation)");
            return VQCPC_OK;
        }
    }
#endif
    if (gate_mask) G3_LAUNCH(E_GATEBITS)
    else if (add && !add2 && add == C && ldadd == ldc) G3_LAUNCH(G_ACCUM)        // the residual already sits in C: accumulate in place
    else if (add2 && add2 == C && ldadd2 == ldc) G3_LAUNCH(E_ADD | G_ACCUM)      // C += A . B^T + add: the second residual already sits in C
    else if (add2) G3_LAUNCH(E_ADD | E_ADD2)
    else if (add) G3_LAUNCH(E_ADD)
    else G3_LAUNCH(0)
#undef G3_LAUNCH
    VQ_CHECK_LAUNCH("gemm_nt_g3");
    return VQCPC_OK;
}

// Few output tiles, long K: the remainder rows of a launch whose 256-tiles do not fill whole rounds (139 264 x 256: 2 rounds + 32
// tiles).  `splits` K slices as one launch of the same kernel (grid = tiles x splits, partial products into `workspace` planes),
// then one pass that sums the planes in ascending order and applies the epilogue: bias, dropout (row0 = global row of the first
// row, for the element index), residual(s).  Deterministic; differs from the unsplit product by fp32 summation order only.
int64_t vqcpc_gemm_nt_grad_splitk_workspace(int64_t M, int N, int splits) {
    return (int64_t)std::max(splits, 1) * std::max<int64_t>(M, 1) * N * (int64_t)sizeof(float);
}

int vqcpc_gemm_nt_grad_splitk(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                              int splits, const float* bias, float drop_p, uint64_t seed, int64_t row0, const float* add,
                              int64_t ldadd, const float* add2, int64_t ldadd2, void* workspace, int64_t workspace_bytes,
                              float* scale_state, void* stream) {
    VQ_REQUIRE(A && B && C && scale_state && workspace, "gemm_nt_grad_splitk: null pointer");
    VQ_REQUIRE(splits >= 1 && splits <= 64 && K % splits == 0 && vqcpc_gemm_nt_grad_supported(M, N, K / splits),
               "gemm_nt_grad_splitk: M, N multiples of 256, K / splits a multiple of 32, got M=%lld N=%d K=%d splits=%d", (long long)M,
               N, K, splits);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && ldc % 4 == 0 && aligned16(A) && aligned16(B) &&
                   aligned16(C) && aligned16(workspace),
               "gemm_nt_grad_splitk: bad leading dimensions / alignment");
    VQ_REQUIRE((!add || (aligned16(add) && ldadd % 4 == 0 && ldadd >= N)) && (!add2 || (add && aligned16(add2) && ldadd2 % 4 == 0)) &&
                   (!bias || aligned16(bias)) && drop_p >= 0.f && drop_p < 1.f,
               "gemm_nt_grad_splitk: bad epilogue operands");
    if (workspace_bytes < vqcpc_gemm_nt_grad_splitk_workspace(M, N, splits)) {
        set_error("gemm_nt_grad_splitk: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    EpiParams ep{};
    ep.gate_scale = 1.0f;
    ep.split_plane = M * (int64_t)N;
    const int tn = N / kG;
    const int tiles = (int)((M / kG) * tn);
    const size_t lds = (size_t)kGSlots * kGSlot;
    hipStream_t st = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_g3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_nt_g3_kernel<0>), dim3((unsigned)std::min(tiles, kNumCU), (unsigned)splits), dim3(kGThreads), lds, st, A,
                       lda, B, ldb, (float*)workspace, (int64_t)N, M, N, K / splits, tn, tiles, ep, scale_state);
    VQ_CHECK_LAUNCH("gemm_nt_g3 (split-K)");
    const int64_t total = M * (N / 4);
    hipLaunchKernelGGL(g3_splitk_epilogue_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 8 * kNumCU)), dim3(256), 0, st,
                       (const float*)workspace, ep.split_plane, splits, C, ldc, M, N, bias, drop_threshold(drop_p),
                       1.0f / (1.0f - drop_p), seed, row0, add, ldadd, add2, ldadd2);
    VQ_CHECK_LAUNCH("gemm_nt_g3 split-K epilogue");
    return VQCPC_OK;
}

// The tail rows of a ragged launch on 64 x 128 tiles (gemm_nt_g3_tail_kernel): every epilogue the ragged launches of a training
// step use -- none | + add | + add + add2 (add may be C: in place) for the input gradients, + bias | + bias + add | + bias +
// dropout + add for the forward -- with row0 = global row of the first row of this call (dropout element index).
int vqcpc_gemm_nt_grad_tail_supported(int64_t M, int N, int K) {
    return (M >= kRM && (M % kRM) == 0 && N >= kRN && (N % kRN) == 0 && K >= kRBK && (K % kRBK) == 0 &&
            (M / kRM) * (N / kRN) < (1ll << 30)) ? 1 : 0;
}

int vqcpc_gemm_nt_grad_tail(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                            const float* bias, float drop_p, uint64_t seed, int64_t row0, const float* add, int64_t ldadd,
                            const float* add2, int64_t ldadd2, float* scale_state, void* stream) {
    VQ_REQUIRE(A && B && C && scale_state, "gemm_nt_grad_tail: null pointer");
    VQ_REQUIRE(vqcpc_gemm_nt_grad_tail_supported(M, N, K), "gemm_nt_grad_tail: M a multiple of 64, N of 128, K of 32, got M=%lld N=%d K=%d",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && aligned16(A) && aligned16(B),
               "gemm_nt_grad_tail: bad leading dimensions / alignment");
    VQ_REQUIRE(!(add2 && !add) && !(add && ldadd < N) && !(add2 && ldadd2 < N) && drop_p >= 0.f && drop_p < 1.f,
               "gemm_nt_grad_tail: bad epilogue operands");
    EpiParams ep{};
    ep.bias = bias;
    ep.thr = drop_threshold(drop_p);
    ep.inv_keep = 1.0f / (1.0f - drop_p);
    ep.seed = seed;
    ep.row0 = row0;
    ep.add = add;
    ep.ldadd = ldadd;
    ep.add2 = add2;
    ep.ldadd2 = ldadd2;
    ep.gate_scale = 1.0f;
    const int tn = N / kRN;
    const size_t lds = (size_t)2 * kRSlot;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_g3_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(gemm_nt_g3_tail_kernel, dim3((unsigned)((M / kRM) * tn)), dim3(kGThreads), lds, (hipStream_t)stream, A, lda, B,
                       ldb, C, ldc, N, K, tn, ep, scale_state);
    VQ_CHECK_LAUNCH("gemm_nt_g3 (tail rows)");
    return VQCPC_OK;
}

// The forward products of a TRAINING step on the same kernel (round 5): C = epilogue(A . B^T) with the epilogues of vqcpc_gemm_nt that
// the 256-tile forward launches use -- bias; bias + residual; bias + dropout + residual (the residual sums of LayerNorm); bias +
// relu (+ dropout) with the bit mask of the positive outputs (vqcpc_gemm_nt_relu_mask).  Same dropout element index and mask layout
// as vqcpc_gemm_nt, so the backward is unchanged.  Scale state as for vqcpc_gemm_nt_grad.
int vqcpc_gemm_nt_f16x3(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                        const float* bias, int act, float drop_p, uint64_t seed, const float* add, int64_t ldadd, void* mask_out,
                        float* scale_state, void* stream) {
    VQ_REQUIRE(A && B && C && scale_state && bias, "gemm_nt_f16x3: null pointer (a bias is part of every forward form)");
    VQ_REQUIRE(vqcpc_gemm_nt_grad_supported(M, N, K), "gemm_nt_f16x3: M, N multiples of 256 and K of 32, got M=%lld N=%d K=%d",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && aligned16(A) && aligned16(B),
               "gemm_nt_f16x3: bad leading dimensions / alignment");
    VQ_REQUIRE(act == 0 || act == 1, "gemm_nt_f16x3: act must be 0 or 1");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm_nt_f16x3: bad dropout probability");
    VQ_REQUIRE((act == 1) == (mask_out != nullptr) && !(act == 1 && add) && !(add && ldadd < N),
               "gemm_nt_f16x3: epilogue is one of bias / bias + add / bias + dropout + add / bias + relu (+ dropout) + mask_out");
    VQ_REQUIRE(!mask_out || aligned16(mask_out), "gemm_nt_f16x3: mask must be 16-byte aligned");
    EpiParams ep{};
    ep.bias = bias;
    ep.act = act;
    ep.thr = drop_threshold(drop_p);
    ep.inv_keep = 1.0f / (1.0f - drop_p);
    ep.seed = seed;
    ep.add = add;
    ep.ldadd = ldadd;
    ep.mask = (uint32_t*)mask_out;
    ep.gate_scale = 1.0f;
    const int tn = N / kG;
    const int tiles = (int)((M / kG) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kGThreads);
    const size_t lds = (size_t)kGSlots * kGSlot;
    hipStream_t st = (hipStream_t)stream;
#define G3_LAUNCH(EPIV)                                                                                                  \
    {                                                                                                                    \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            (void)hipFuncSetAttribute((const void*)gemm_nt_g3_kernel<EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                      (int)lds);                                                                         \
            attr = true;                                                                                                 \
        }                                                                                                                \
        hipLaunchKernelGGL((gemm_nt_g3_kernel<EPIV>), grid, block, lds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles,  \
                           ep, scale_state);                                                                             \
    }
    const bool drop = ep.thr != 0;
    if (act == 1) {
        if (drop) G3_LAUNCH(E_BIAS | E_RELU | E_DROP | E_MASKOUT)
        else G3_LAUNCH(E_BIAS | E_RELU | E_MASKOUT)
    } else if (add) {
        if (drop) G3_LAUNCH(E_BIAS | E_DROP | E_ADD)
        else G3_LAUNCH(E_BIAS | E_ADD)
    } else {
        VQ_REQUIRE(!drop, "gemm_nt_f16x3: dropout without a residual or relu is not a forward form of the step");
        G3_LAUNCH(E_BIAS)
    }
#undef G3_LAUNCH
    VQ_CHECK_LAUNCH("gemm_nt_g3 (forward)");
    return VQCPC_OK;
}

// The same products on PRE-SPLIT operands (round 6; "P4" format, see gemm_nt_g3_kernel): `pl_amax_b` (required) = device scalar, the
// amax B's planes were scaled with -- B then points at the P4 image of the (N, K) operand, same leading dimension; `pl_amax_a`
// (optional) likewise for A.  One entry point for every epilogue form of vqcpc_gemm_nt_grad (bias == NULL: none | + add | + add + add2
// | add == C in place | gate-bit mask) and of vqcpc_gemm_nt_f16x3 (bias != NULL: bias | + add | + dropout + add | relu (+ dropout) +
// mask_out).  Results are bit-identical to those entry points when their scale state holds the same amax values.
int vqcpc_gemm_nt_g3_pl(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                        const float* bias, int act, float drop_p, uint64_t seed, const float* add, int64_t ldadd, const float* add2,
                        int64_t ldadd2, const void* gate_mask, float gate_scale, void* mask_out, float* scale_state,
                        const float* pl_amax_a, const float* pl_amax_b, void* stream) {
    VQ_REQUIRE(A && B && C && scale_state && pl_amax_b, "gemm_nt_g3_pl: null pointer (B must be a P4 operand)");
    VQ_REQUIRE(vqcpc_gemm_nt_grad_supported(M, N, K), "gemm_nt_g3_pl: M, N multiples of 256 and K of 32, got M=%lld N=%d K=%d",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && aligned16(A) && aligned16(B),
               "gemm_nt_g3_pl: bad leading dimensions / alignment");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (act == 0 || act == 1), "gemm_nt_g3_pl: bad dropout probability / act");
    EpiParams ep{};
    ep.bias = bias;
    ep.act = act;
    ep.thr = drop_threshold(drop_p);
    ep.inv_keep = 1.0f / (1.0f - drop_p);
    ep.seed = seed;
    ep.add = add;
    ep.ldadd = ldadd;
    ep.add2 = add2;
    ep.ldadd2 = ldadd2;
    ep.gate_scale = bias ? 1.0f : gate_scale;
    ep.pl_amax_a = pl_amax_a;
    ep.pl_amax_b = pl_amax_b;
    const int tn = N / kG;
    const int tiles = (int)((M / kG) * tn);
    const dim3 grid((unsigned)std::min(tiles, kNumCU)), block(kGThreads);
    const size_t lds = (size_t)kGSlots * kGSlot;
    hipStream_t st = (hipStream_t)stream;
    const bool drop = ep.thr != 0;
#define G3PL_LAUNCH(EPIV, PLV)                                                                                           \
    {                                                                                                                    \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            (void)hipFuncSetAttribute((const void*)gemm_nt_g3_kernel<EPIV, 0, PLV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                         \
            attr = true;                                                                                                 \
        }                                                                                                                \
        hipLaunchKernelGGL((gemm_nt_g3_kernel<EPIV, 0, PLV>), grid, block, lds, st, A, lda, B, ldb, C, ldc, M, N, K, tn, tiles, \
                           ep, scale_state);                                                                             \
    }
#define G3PL_BOTH(EPIV) { if (pl_amax_a) G3PL_LAUNCH(EPIV, 3) else G3PL_LAUNCH(EPIV, 2) }
    if (bias) {                                  // the forward forms
        VQ_REQUIRE(!gate_mask && !add2 && (act == 1) == (mask_out != nullptr) && !(act == 1 && add) && !(add && ldadd < N),
                   "gemm_nt_g3_pl: forward epilogue is one of bias / bias + add / bias + dropout + add / bias + relu (+ dropout) + mask_out");
        VQ_REQUIRE(!mask_out || aligned16(mask_out), "gemm_nt_g3_pl: mask must be 16-byte aligned");
        ep.mask = (uint32_t*)mask_out;
        if (act == 1) {
            VQ_REQUIRE(!pl_amax_a, "gemm_nt_g3_pl: the relu + mask form takes an fp32 A operand");
            if (drop) G3PL_LAUNCH(E_BIAS | E_RELU | E_DROP | E_MASKOUT, 2)
            else G3PL_LAUNCH(E_BIAS | E_RELU | E_MASKOUT, 2)
        } else if (add) {
            if (drop) G3PL_BOTH(E_BIAS | E_DROP | E_ADD)
            else G3PL_BOTH(E_BIAS | E_ADD)
        } else {
            VQ_REQUIRE(!drop, "gemm_nt_g3_pl: dropout without a residual or relu is not a forward form of the step");
            G3PL_BOTH(E_BIAS)
        }
    } else {                                     // the input-gradient forms
        VQ_REQUIRE(!mask_out && !drop && !(add2 && !add) && !(gate_mask && add), "gemm_nt_g3_pl: gradient epilogue is one of none / add / add + add2 / gate mask");
        VQ_REQUIRE(!gate_mask || aligned16(gate_mask), "gemm_nt_g3_pl: gate mask must be 16-byte aligned");
        ep.mask = (uint32_t*)const_cast<void*>(gate_mask);
        if (gate_mask) {
            VQ_REQUIRE(!pl_amax_a, "gemm_nt_g3_pl: the gate-mask form takes an fp32 A operand");
            G3PL_LAUNCH(E_GATEBITS, 2)
        } else if (add && !add2 && add == C && ldadd == ldc) G3PL_BOTH(G_ACCUM)
        else if (add2 && add2 == C && ldadd2 == ldc) G3PL_LAUNCH(E_ADD | G_ACCUM, 2)
        else if (add2) G3PL_LAUNCH(E_ADD | E_ADD2, 2)
        else if (add) G3PL_BOTH(E_ADD)
        else G3PL_BOTH(0)
    }
#undef G3PL_BOTH
#undef G3PL_LAUNCH
    VQ_CHECK_LAUNCH("gemm_nt_g3 (P4 operands)");
    return VQCPC_OK;
}

// The 64 x 128-tile kernel as a GENERAL entry point (round 6): the tail rows of a ragged launch on P4 operands, and whole products
// whose 256-tiles would not fill the chip (the 3 072 - 12 288-row products of the student / decoder steps: x 1.1-1.36 the six-product
// 128-tile / split-K path, tools/bench_small_f16x3.py).  Operands fp32 or P4 (pl_amax_a / pl_amax_b NULL: fp32, split in the kernel).
// Epilogue, in the order of vqcpc_gemm_nt: + bias, relu (act == 1), dropout (element index (row0 + row) * N + col), * (gate > 0 ?
// gate_scale : 0), + add (may be C), + add2.
int vqcpc_gemm_nt_g3_small(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                           const float* bias, int act, float drop_p, uint64_t seed, int64_t row0, const float* gate, int64_t ldgate,
                           float gate_scale, const float* add, int64_t ldadd, const float* add2, int64_t ldadd2, float* scale_state,
                           const float* pl_amax_a, const float* pl_amax_b, void* stream) {
    VQ_REQUIRE(A && B && C && scale_state, "gemm_nt_g3_small: null pointer");
    VQ_REQUIRE(vqcpc_gemm_nt_grad_tail_supported(M, N, K), "gemm_nt_g3_small: M a multiple of 64, N of 128, K of 32, got M=%lld N=%d K=%d",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && aligned16(A) && aligned16(B),
               "gemm_nt_g3_small: bad leading dimensions / alignment");
    VQ_REQUIRE(!(add2 && !add) && !(add && ldadd < N) && !(add2 && ldadd2 < N) && !(gate && ldgate < N) && drop_p >= 0.f && drop_p < 1.f &&
                   (act == 0 || act == 1),
               "gemm_nt_g3_small: bad epilogue operands");
    EpiParams ep{};
    ep.bias = bias;
    ep.act = act;
    ep.thr = drop_threshold(drop_p);
    ep.inv_keep = 1.0f / (1.0f - drop_p);
    ep.seed = seed;
    ep.row0 = row0;
    ep.gate = gate;
    ep.ldgate = ldgate;
    ep.gate_scale = gate_scale;
    ep.add = add;
    ep.ldadd = ldadd;
    ep.add2 = add2;
    ep.ldadd2 = ldadd2;
    ep.pl_amax_a = pl_amax_a;
    ep.pl_amax_b = pl_amax_b;
    const int tn = N / kRN;
    const size_t lds = (size_t)2 * kRSlot;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_g3_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(gemm_nt_g3_tail_kernel, dim3((unsigned)((M / kRM) * tn)), dim3(kGThreads), lds, (hipStream_t)stream, A, lda, B,
                       ldb, C, ldc, N, K, tn, ep, scale_state);
    VQ_CHECK_LAUNCH("gemm_nt_g3 (64 x 128 tiles)");
    return VQCPC_OK;
}

// wgrad shapes: whole 256 x 256 output tiles, M a multiple of 32 (an even number of 16-row steps per split)
int vqcpc_gemm_tn_grad_supported(int64_t M, int N, int K) {
    return (N >= kG && (N % kG) == 0 && K >= kG && (K % kG) == 0 && M >= 32 && (M % 32) == 0) ? 1 : 0;
}

int64_t vqcpc_gemm_tn_grad_workspace(int64_t M, int N, int K) {
    return (int64_t)tn_g3_splits(std::max<int64_t>(M, 1), N, K) * ((int64_t)N * K + N) * (int64_t)sizeof(float);
}

int vqcpc_gemm_tn_grad(const float* A, int64_t lda, const float* B, int64_t ldb, float* dW, float* db, int64_t M, int N, int K,
                       int accumulate, void* workspace, int64_t workspace_bytes, float* scale_state, void* stream) {
    VQ_REQUIRE(A && B && dW && workspace && scale_state, "gemm_tn_grad: null pointer");
    VQ_REQUIRE(vqcpc_gemm_tn_grad_supported(M, N, K), "gemm_tn_grad: shape M=%lld N=%d K=%d not supported", (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= N && ldb >= K && aligned16(A) && aligned16(B),
               "gemm_tn_grad: bad leading dimensions / alignment");
    if (workspace_bytes < vqcpc_gemm_tn_grad_workspace(M, N, K)) {
        set_error("gemm_tn_grad: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const int splits = tn_g3_splits(M, N, K);
    const int64_t rows_per_split = round_up(ceil_div(M, splits), 32);
    float* ws = (float*)workspace;
    float* ws_bias = db ? ws + (int64_t)splits * N * K : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)kGSlots * kQSlot;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_g3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int tk = K / kG;
    hipLaunchKernelGGL(gemm_tn_g3_kernel, dim3((N / kG) * tk, splits), dim3(kGThreads), lds, s, A, lda, B, ldb, M, N, K, tk,
                       rows_per_split, ws, ws_bias, scale_state);
    VQ_CHECK_LAUNCH("gemm_tn_g3");
    return launch_reduce_splits2(ws, (int64_t)N * K, splits, dW, (int64_t)N * K, ws_bias, N, db, db ? N : 0, accumulate, s);
}

int vqcpc_grad_amax(const float* x, int64_t ld, int64_t rows, int cols, float* amax_slot, void* stream) {
    VQ_REQUIRE(x && amax_slot && rows >= 1 && cols >= 4 && cols % 4 == 0 && ld % 4 == 0 && ld >= cols && aligned16(x),
               "grad_amax: bad arguments");
    const int64_t total = rows * (cols / 4);
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(total, 256), 4 * kNumCU);
    hipLaunchKernelGGL(grad_amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, amax_slot);
    VQ_CHECK_LAUNCH("grad_amax");
    return VQCPC_OK;
}

int vqcpc_grad_scale_roll(float* state, int nsites, void* stream) {
    VQ_REQUIRE(state && nsites >= 0, "grad_scale_roll: bad arguments");
    if (nsites == 0) return VQCPC_OK;
    hipLaunchKernelGGL(grad_scale_roll_kernel, dim3((unsigned)ceil_div(2 * nsites, 256)), dim3(256), 0, (hipStream_t)stream,
                       state, nsites, (unsigned int*)nullptr);
    VQ_CHECK_LAUNCH("grad_scale_roll");
    return VQCPC_OK;
}

int vqcpc_grad_scale_roll_counted(float* state, int nsites, void* saturated_count, void* stream) {
    VQ_REQUIRE(state && nsites >= 0 && saturated_count, "grad_scale_roll_counted: bad arguments");
    if (nsites == 0) return VQCPC_OK;
    hipLaunchKernelGGL(grad_scale_roll_kernel, dim3((unsigned)ceil_div(2 * nsites, 256)), dim3(256), 0, (hipStream_t)stream,
                       state, nsites, (unsigned int*)saturated_count);
    VQ_CHECK_LAUNCH("grad_scale_roll_counted");
    return VQCPC_OK;
}

int vqcpc_weight_planes_many(const float* base, const int64_t* desc, int n, int64_t total_tiles, float* amax, void* planes,
                             void* planes_t, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n == 0) return VQCPC_OK;
    VQ_REQUIRE(base && desc && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31) && amax && planes && planes_t && workspace &&
                   aligned16(base) && aligned16(planes) && aligned16(planes_t),
               "weight_planes_many: bad arguments");
    if (workspace_bytes < total_tiles * (int64_t)sizeof(float)) {
        set_error("weight_planes_many: workspace too small (one float per 32 x 32 tile)");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(weight_amax_many_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, base, desc, n, (float*)workspace);
    VQ_CHECK_LAUNCH("weight_amax_many");
    hipLaunchKernelGGL(weight_planes_many_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, base, desc, n, (const float*)workspace,
                       total_tiles, amax, (uint4*)planes, (uint4*)planes_t);
    VQ_CHECK_LAUNCH("weight_planes_many");
    return VQCPC_OK;
}

int vqcpc_grad_scale_roll_logged(float* state, int nsites, void* monitor, int log_capacity, void* stream) {
    VQ_REQUIRE(state && nsites >= 0 && nsites <= 512 && monitor && log_capacity >= 0, "grad_scale_roll_logged: bad arguments (at most 512 sites)");
    hipLaunchKernelGGL(grad_scale_roll_logged_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, state, nsites, (int*)monitor,
                       log_capacity);
    VQ_CHECK_LAUNCH("grad_scale_roll_logged");
    return VQCPC_OK;
}

}  // extern "C"
