// Shared device/host helpers for libvqcpc_hip.so (gfx950 only, wave64).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vqcpc.h"

namespace vq {

void set_error(const char* fmt, ...);

#define VQ_REQUIRE(cond, ...)                   \
    do {                                        \
        if (!(cond)) {                          \
            vq::set_error(__VA_ARGS__);         \
            return VQCPC_EINVAL;                \
        }                                       \
    } while (0)

#define VQ_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            vq::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return VQCPC_ELAUNCH;                                                    \
        }                                                                            \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;
constexpr int kNumCU = 256;   // MI355X

// ---- dropout RNG: stateless, one 32-bit integer mix per element (8 VALU ops; a 64-bit splitmix cost ~25 and doubled the
// instruction count of the K = 256 GEMM epilogues).  The element index enters modulo 2^32 (tensors of the path stay below).
// Step salt: every seed is XOR-ed with g_rng_salt before use.  It is 0 in eager execution (a fresh seed per dropout site and
// step arrives as a kernel argument); when a whole training step is replayed from a HIP graph the seed ARGUMENTS are frozen
// in the graph, so its first node (vqcpc_rng_salt_advance) draws a new salt per replay and the masks still change every
// step.  One copy per translation unit (no relocatable device code); util.hip keeps the registry of their addresses.
// __constant__: read-only inside a kernel, so the load is a hoistable scalar load; written between kernels only.
__attribute__((used)) static __constant__ uint64_t g_rng_salt = 0;
typedef uint64_t* (*SaltAddrFn)();
void register_rng_salt(SaltAddrFn fn);
namespace {
inline uint64_t* rng_salt_addr_of_this_unit() {
    uint64_t* p = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_rng_salt)) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
    }
    return p;
}
struct RngSaltRegistrar {
    RngSaltRegistrar() { register_rng_salt(&rng_salt_addr_of_this_unit); }
};
static RngSaltRegistrar g_rng_salt_registrar;
}  // namespace

// Measurement switches (environment variables of the tools under tools/) exist in LAB builds only (-DVQCPC_LAB=1,
// `VQCPC_LAB=1 python -m vqcpc_bach_amd.build` -> libvqcpc_hip_lab.so): the product library reads no tuning variable.
#ifndef VQCPC_LAB
#define VQCPC_LAB 0
#endif
#if VQCPC_LAB
inline int lab_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
constexpr int lab_env_int(const char*, int dflt) { return dflt; }
#endif

// The hash in two steps, for kernels whose element index advances by a constant stride (GEMM epilogues: rows of one
// column): x0 = idx * kRngMul + seed_lo is affine in idx, so x0(idx + d) = x0(idx) + d * kRngMul costs one add instead of
// a 32-bit multiply (quarter rate) and the 64-bit index arithmetic.
constexpr uint32_t kRngMul = 0x9E3779B1u;
__device__ __forceinline__ uint64_t rng_seed_eff(uint64_t seed) { return seed ^ g_rng_salt; }
__device__ __forceinline__ uint32_t rng_x0(uint64_t seed_eff, uint32_t idx_lo) {
    return idx_lo * kRngMul + static_cast<uint32_t>(seed_eff);
}
// Round 5: the mixer runs on the FULL-RATE 24-bit multiplier (v_mul_u32_u24 / v_mad_u32_u24) -- the 'lowbias32' finaliser of rounds
// 1-4 needs two v_mul_lo_u32, quarter-rate instructions, ~16 issue slots per element against ~10 here, and a GEMM epilogue with
// dropout evaluates it 128 times per lane and output tile (bias + relu + dropout was 17 % slower than bias + relu on the bf16
// GEMMs of configs[4]).  x ^= x >> 16 first, so that the 24 bits the multiplier sees depend on all 32 (no period of 2^24
// elements); keep rate, lag-1 / lag-2 / lag-N autocorrelation of the masks, chi^2 of the 24-bit values and the correlation
// between consecutive seeds were checked against the old mixer on 4 M indices (tools/micro/rng_quality.py): equal within noise.
__device__ __forceinline__ uint32_t rng_u24_from_x0(uint32_t x, uint32_t seed_hi) {
    x ^= seed_hi;
    x ^= x >> 16;
    uint32_t y = __umul24(x, 0x6B43A9u);
    y ^= y >> 15;
    y = __umul24(y, 0x52DCE7u) + x;
    y ^= y >> 14;
    return y >> 8;   // top 24 bits
}
__device__ __forceinline__ uint32_t rng_u24(uint64_t seed, uint64_t idx) {
    const uint64_t se = rng_seed_eff(seed);
    return rng_u24_from_x0(rng_x0(se, static_cast<uint32_t>(idx)), static_cast<uint32_t>(se >> 32));
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) { return static_cast<uint32_t>(p * 16777216.0f); }
// keep-mask value: 0 or 1/(1-p).  thr == 0 (p == 0) keeps everything with scale 1.
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
    if (thr == 0) return 1.0f;
    return rng_u24(seed, idx) >= thr ? inv_keep : 0.0f;
}

// ---- wave64 reductions ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// deterministic reduction of `nsplit` partial arrays: out[i] = (accumulate ? out[i] : 0) + sum_s ws[s*stride + i]
int launch_reduce_splits(const float* ws, int64_t stride, int nsplit, float* out, int64_t count, int accumulate,
                         hipStream_t stream);
// same, two column segments in ONE launch: (ws, stride, out, count) and (ws2, stride2, out2, count2)
int launch_reduce_splits2(const float* ws, int64_t stride, int nsplit, float* out, int64_t count, const float* ws2,
                          int64_t stride2, float* out2, int64_t count2, int accumulate, hipStream_t stream);

// Outputs of the attention kernels on the bf16 training path (BASELINE configs[4]): context / input gradients that only feed
// GEMMs are written as bf16 by the producing kernel (`out16` != 0: `base` holds bf16 elements; offsets in elements either way).
#ifdef __HIPCC__
typedef float vq_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 vq_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16x2_rn(float e0, float e1) {      // round to nearest even, element 0 in the low half
    const vq_f32x2_t v = {e0, e1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vq_bf16x2_t));
}
__device__ __forceinline__ void store4_out(float* __restrict__ base, int64_t off, float a, float b, float c, float d, int out16) {
    if (out16)
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = make_uint2(bf16x2_rn(a, b), bf16x2_rn(c, d));
    else
        *reinterpret_cast<float4*>(base + off) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store1_out(float* __restrict__ base, int64_t off, float a, int out16) {
    if (out16) reinterpret_cast<unsigned short*>(base)[off] = (unsigned short)(bf16x2_rn(a, 0.0f) & 0xFFFFu);
    else base[off] = a;
}
#endif

// L = 16 attention on the fp32 matrix cores (relattn16.hip); tokens != nullptr = block-table indirection
bool relattn16_supported(int H, int hd);
int64_t relattn16_bwd_workspace(int64_t n_blocks, int H, int hd);
int relattn16_fwd(const float* qkv, int64_t ldq, const int64_t* tokens, const float* e1, const float* e2, float* ctx,
                  int64_t ldo, float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, hipStream_t s);
int relattn16_bwd(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const int64_t* tokens, const float* probs,
                  const float* e1, const float* e2, float* d_qkv, int64_t ldg, float* ws, int64_t n_blocks, int H, int hd,
                  float drop_p, uint64_t seed, hipStream_t s, int* nsplit);
int relattn16_fwd_b16(const float* qkv, int64_t ldq, const int64_t* tokens, const float* e1, const float* e2, void* ctx_b16,
                      int64_t ldo, float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, hipStream_t s);
int relattn16_bwd_b16(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const int64_t* tokens, const float* probs,
                      const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* ws, int64_t n_blocks, int H, int hd,
                      float drop_p, uint64_t seed, hipStream_t s, int* nsplit);
int relattn16_fwd_b16io(const void* qkv_b16, int64_t ldq, const float* e1, const float* e2, void* ctx_b16, int64_t ldo,
                        float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, hipStream_t s);
int relattn16_bwd_b16io(const void* d_ctx_b16, int64_t ldo, const void* qkv_b16, int64_t ldq, const float* probs,
                        const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* ws, int64_t n_blocks, int H,
                        int hd, float drop_p, uint64_t seed, hipStream_t s, int* nsplit);

}  // namespace vq
