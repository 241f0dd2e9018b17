// Definitions shared by the GEMM translation units (gemm.hip, gemm_dma.hip): epilogue descriptor, tile constants,
// the exact 3-way bf16 split.
#pragma once
#include "common.h"

namespace vq {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_S = BK + 4;   // padded row stride (floats) of the NT operand tiles
constexpr int kGemmThreads = 256;

struct EpiParams {
    const float* bias;
    int act;
    uint32_t thr;
    float inv_keep;
    uint64_t seed;
    const float* gate;
    int64_t ldgate;
    float gate_scale;
    const float* add;
    int64_t ldadd;
    const float* add2;   // second residual (only honoured together with `add`)
    int64_t ldadd2;
    int64_t row0;        // global row of the first row of this launch (a GEMM may be cut into two launches by rows): only
                         // the dropout element index needs it
    uint32_t* mask;      // E_MASKOUT: written, E_GATEBITS: read.  Bit (row, col) of the "output > 0" mask lives in word
                         // ((row >> 2) * (N / 32) + col / 32) * 4 + (row & 3), bit col % 32: the four rows a lane of the
                         // 32x32 MFMA layout holds consecutively are one 16-byte load
    int64_t split_plane; // split-K launch of the 128-tile kernel (gridDim.y > 1): floats between two partial planes
    // f16x3 kernels on PRE-SPLIT operands (round 6, csrc/gemm_grad.hip "P4"): when an operand arrives as fp16 plane pairs, the amax
    // its planes were scaled with (device scalar; the kernel derives the same power-of-two scale from it)
    const float* pl_amax_a;
    const float* pl_amax_b;
};

// epilogue feature bits (compile-time); E_RUNTIME = decide everything from EpiParams at run time (rare combinations)
enum { E_BIAS = 1, E_RELU = 2, E_DROP = 4, E_GATE = 8, E_ADD = 16, E_RUNTIME = 32, E_ADD2 = 64, E_MASKOUT = 128, E_GATEBITS = 256 };

// bijective XCD-aware remap: consecutive tiles (which share an A row panel) land on the same XCD / L2
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// ---- bf16x6 helpers -------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kX6Stride = BK + 8;                    // bf16 per LDS row: 80 B -> conflict-free ds_read_b128 across 16 rows
constexpr int kX6Plane = BM * kX6Stride * 2;         // bytes per plane (BM == BN)

// exact 3-way split of an fp32 value into bf16 pieces by truncation: x == h + m + l (as fp32 values whose low 16 bits are 0)
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r = x - __uint_as_float(h);
    m = __float_as_uint(r) & 0xFFFF0000u;
    l = __float_as_uint(r - __uint_as_float(m));     // at most 8 significant bits are left: truncation is exact
}
// pack the high halves of two fp32 bit patterns into one dword (element 0 in the low half)
__device__ __forceinline__ uint32_t pack_hi(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }
// MODE 2 ("bf16"): ONE bf16 piece per operand, round-to-nearest-even like a torch .bfloat16() cast (finite inputs)
__device__ __forceinline__ uint32_t round_bf16(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}
// two fp32 -> one dword of two bf16 (element 0 in the low half), round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950 (the
// integer form above is 4 VALU instructions per element + a byte permute per pair)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float e0, float e1) {
    const f32x2_t v = {e0, e1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint2 round4_bf16(const float4& v) {
    return make_uint2(cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w));
}
__device__ __forceinline__ void split3x4(const float4& v, uint2& h, uint2& m, uint2& l) {
    uint32_t h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    split3(v.x, h0, m0, l0);
    split3(v.y, h1, m1, l1);
    split3(v.z, h2, m2, l2);
    split3(v.w, h3, m3, l3);
    h = make_uint2(pack_hi(h0, h1), pack_hi(h2, h3));
    m = make_uint2(pack_hi(m0, m1), pack_hi(m2, m3));
    l = make_uint2(pack_hi(l0, l1), pack_hi(l2, l3));
}

// Two-plane split of the gradient arithmetic (NP = 2 kernels): h = rn_bf16(x) (cvt_pk_bf16 above: round to nearest even),
// m = rn_bf16(x - h) -- the subtraction is exact in fp32 -- so |x - h - m| <= 2^-18 |x|.  Packed pairs, low half first.
__device__ __forceinline__ void split2_pair(float a, float b, uint32_t& h, uint32_t& m) {
    h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(ra, rb);
}
__device__ __forceinline__ void split2x4(const float4& v, uint2& h, uint2& m) {
    split2_pair(v.x, v.y, h.x, m.x);
    split2_pair(v.z, v.w, h.y, m.y);
}
// row-pair interleaved form of the weight-gradient (TN) kernel: dword c = (row0[c], row1[c])
__device__ __forceinline__ void split2_pair4(const float4& r0, const float4& r1, uint4& h, uint4& m) {
    split2_pair(r0.x, r1.x, h.x, m.x);
    split2_pair(r0.y, r1.y, h.y, m.y);
    split2_pair(r0.z, r1.z, h.z, m.z);
    split2_pair(r0.w, r1.w, h.w, m.w);
}

#if VQCPC_LAB
// software-pipelined one-wave-per-SIMD variant of the 256-tile bf16x6 NT kernel (gemm_sw.hip)
bool gemm_nt_sw_ok(int64_t M, int N, int K, int flags);
int gemm_nt_sw_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                      int K, int flags, const EpiParams& ep, hipStream_t st);
// LDS-DMA variant of the 256-tile bf16x6 NT kernel (gemm_dma.hip)
bool gemm_nt_dma_ok(int64_t M, int N, int K, int flags);
int gemm_nt_dma_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                       int K, int flags, const EpiParams& ep, hipStream_t st);

#endif

}  // namespace vq
