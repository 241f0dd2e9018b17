// General-L self-attention with the closed-form learned relative bias (any L up to kGenMaxL; the student path runs
// L = 384 in the teacher and L = 24 / 96 in the auxiliary decoder: teacher_relative.py:38-50,
// auxiliary_decoder_relative.py:57-72).  Same maths and buffers as relattn.hip:
//   S[i][j] = qs_i . k_j + qs_i . Erel[j - i + L - 1],   qs = q / sqrt(hd),
//   Erel[r] = e1[h][r] (r < L: j <= i) | e2[h][r - L + 1] (r >= L: j > i)
// (SubsampledRelativeAttention.forward, subsampled_relative_attention.py:30-122, without its pad / view / mask tensors).
//
// Mapping.  One wavefront owns a strip of 32 query rows of one (block, head) problem and walks the key tiles with
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).  The MFMA k index is a summation index, so each lane
// half g takes the contiguous columns [g*hd/2, (g+1)*hd/2) of its operand row: A / B fragments are plain float4 loads
// of a row straight from global memory (L2 resident: q, k, v of one problem are 3*L*hd*4 bytes), no LDS staging.
// The relative term is ONE extra GEMM per strip, Q_strip (32 x hd) . Erel_band^T with the band of the L + 31
// relative rows the strip can see, r in [L-32-i0, 2L-2-i0]; the skew S[ii][j] += QE[ii][j - ii + 31] is an LDS read
// with a per-register offset.  The 32 x L score strip lives in LDS (in place over the QE band), softmax is done by the
// whole wave one row at a time (coalesced probs stores), P.V reads the strip back as the A operand (ds_read_b128,
// row stride = 4 mod 32 dwords: conflict free).
//
// Backward = three kernels over the same strips (deterministic, no atomics):
//   dq : dP = (dO.V^T) * mask, dS = P (dP - rowsum(dP P)) -> dS strip in LDS and in the workspace;
//        dq = scale * (dS.K + skew(dS).Erel_band)
//   dkv: per key tile, dV = (P*mask)^T dO, dK = dS^T qs
//   de : per (head, 32 relative rows), dErel[r] = sum_n sum_i dS[i][r + i - (L-1)] qs_i, block chunks -> partials
#include "common.h"

namespace vq {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kGenMaxL = 1024;
constexpr float kNegBig = -1.0e30f;

// row of accumulator register r in the 32x32 MFMA C layout (column = lane & 31)
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

static inline int gen_kt(int L) { return (L + 31) / 32; }
static inline int gen_sw_fwd(int L) { return 32 * (gen_kt(L) + 1) + 4; }
static inline int gen_sw_bwd(int L) { return 32 * gen_kt(L) + 64 + 4; }

template <int N>
__device__ __forceinline__ void load_row(float (&dst)[N], const float* __restrict__ p, bool ok, float mul) {
#pragma unroll
    for (int v = 0; v < N / 4; ++v) {
        float4 t = ok ? *reinterpret_cast<const float4*>(p + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        dst[4 * v + 0] = t.x * mul;
        dst[4 * v + 1] = t.y * mul;
        dst[4 * v + 2] = t.z * mul;
        dst[4 * v + 3] = t.w * mul;
    }
}

__device__ __forceinline__ const float* erel_row(const float* __restrict__ e1, const float* __restrict__ e2, int h, int L,
                                                 int HD, int r) {
    r = min(max(r, 0), 2 * L - 2);
    return r < L ? e1 + ((int64_t)h * L + r) * HD : e2 + ((int64_t)h * L + (r - L + 1)) * HD;
}

// =====================================================================================================================
template <int HD>
__global__ __launch_bounds__(64) void relattn_gen_fwd_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                             const float* __restrict__ e1, const float* __restrict__ e2,
                                                             float* __restrict__ ctx, int64_t ldo,
                                                             float* __restrict__ probs, int L, int H, float scale,
                                                             uint32_t thr, float inv_keep, uint64_t seed) {
    constexpr int KH = HD / 2, CT = (HD + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int lane = threadIdx.x, g = lane >> 5, l31 = lane & 31;
    const int KT = (L + 31) / 32, SW = 32 * (KT + 1) + 4;
    const int64_t prob = blockIdx.x / KT;
    const int i0 = (int)(blockIdx.x % KT) * 32;
    const int64_t n = prob / H;
    const int h = (int)(prob % H);
    const int d = H * HD;
    const float* base = qkv + n * L * ldq + h * HD;

    float qa[KH];
    {
        const int i = i0 + l31;
        load_row<KH>(qa, base + (int64_t)min(i, L - 1) * ldq + g * KH, i < L, scale);
    }
    // ---- phase 1: QE band  strip[ii][x] = qs_ii . Erel[rlo + x]
    const int rlo = L - 32 - i0;
    for (int t = 0; t <= KT; ++t) {
        float eb[KH];
        load_row<KH>(eb, erel_row(e1, e2, h, L, HD, rlo + 32 * t + l31) + g * KH, true, 1.0f);
        floatx16 acc = {0};
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], eb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) strip[crow(r, lane) * SW + 32 * t + l31] = acc[r];
    }
    __syncthreads();
    // ---- phase 2: scores, in place (tile jt reads columns [32jt, 32jt+62], writes [32jt, 32jt+31])
    for (int jt = 0; jt < KT; ++jt) {
        const int j = 32 * jt + l31;
        float kb[KH];
        load_row<KH>(kb, base + d + (int64_t)min(j, L - 1) * ldq + g * KH, j < L, 1.0f);
        floatx16 acc = {0};
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], kb[s], acc, 0, 0, 0);
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = crow(r, lane);
            sv[r] = acc[r] + strip[ii * SW + j - ii + 31];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) strip[crow(r, lane) * SW + j] = j < L ? sv[r] : kNegBig;
    }
    __syncthreads();
    // ---- softmax, one row at a time by the whole wave; probs saved BEFORE dropout
    const int rows = min(32, L - i0);
    for (int ii = 0; ii < rows; ++ii) {
        float* row = strip + ii * SW;
        float m = kNegBig;
        for (int j = lane; j < L; j += 64) m = fmaxf(m, row[j]);
        m = wave_max(m);
        float sum = 0.0f;
        for (int j = lane; j < L; j += 64) {
            const float e = __expf(row[j] - m);
            row[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        const int64_t pbase = (prob * L + i0 + ii) * L;
        for (int j = lane; j < 32 * KT; j += 64) {
            float pd = 0.0f;
            if (j < L) {
                const float p = row[j] * inv;
                probs[pbase + j] = p;
                pd = p * drop_scale(seed, (uint64_t)(pbase + j), thr, inv_keep);
            }
            row[j] = pd;
        }
    }
    for (int ii = rows; ii < 32; ++ii)
        for (int j = lane; j < 32 * KT; j += 64) strip[ii * SW + j] = 0.0f;
    __syncthreads();
    // ---- ctx = Pd . V
    floatx16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) o[ct] = floatx16{0};
    const float* vbase = base + 2 * d;
    for (int jt = 0; jt < KT; ++jt) {
        float pa[16];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 t = *reinterpret_cast<const float4*>(strip + l31 * SW + 32 * jt + 16 * g + 4 * v);
            pa[4 * v] = t.x; pa[4 * v + 1] = t.y; pa[4 * v + 2] = t.z; pa[4 * v + 3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int jc = min(32 * jt + 16 * g + s, L - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + l31;
                const float vb = c < HD ? vbase[(int64_t)jc * ldq + c] : 0.0f;
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], vb, o[ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + crow(r, lane);
            if (i < L && c < HD) ctx[(n * L + i) * ldo + h * HD + c] = o[ct][r];
        }
    }
}

// =====================================================================================================================
// dS strip + dq.  dS is also written to dSg [n][H][L][L] for the dkv / de kernels.
template <int HD>
__global__ __launch_bounds__(64) void relattn_gen_bwd_dq_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ qkv, int64_t ldq,
    const float* __restrict__ probs, const float* __restrict__ e1, const float* __restrict__ e2,
    float* __restrict__ d_qkv, int64_t ldg, float* __restrict__ dSg, int L, int H, float scale, uint32_t thr,
    float inv_keep, uint64_t seed) {
    constexpr int KH = HD / 2, CT = (HD + 31) / 32, OFF = 32;
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int lane = threadIdx.x, g = lane >> 5, l31 = lane & 31;
    const int KT = (L + 31) / 32, SW = 32 * KT + 64 + 4;
    const int64_t prob = blockIdx.x / KT;
    const int i0 = (int)(blockIdx.x % KT) * 32;
    const int64_t n = prob / H;
    const int h = (int)(prob % H);
    const int d = H * HD;
    const float* base = qkv + n * L * ldq + h * HD;

    for (int e = lane; e < 32 * SW; e += 64) strip[e] = 0.0f;
    float doa[KH];
    {
        const int i = i0 + l31;
        load_row<KH>(doa, d_ctx + (n * L + min(i, L - 1)) * ldo + h * HD + g * KH, i < L, 1.0f);
    }
    __syncthreads();
    float rd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rd[r] = 0.0f;
    // pass 1: dP = (dO . V^T) * mask, row sums of dP * P
    for (int jt = 0; jt < KT; ++jt) {
        const int j = 32 * jt + l31;
        float vb[KH];
        load_row<KH>(vb, base + 2 * d + (int64_t)min(j, L - 1) * ldq + g * KH, j < L, 1.0f);
        floatx16 acc = {0};
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(doa[s], vb[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = crow(r, lane), i = i0 + ii;
            const bool ok = i < L && j < L;
            const int64_t idx = (prob * L + i) * L + j;
            const float p = ok ? probs[idx] : 0.0f;
            const float dp = ok ? acc[r] * drop_scale(seed, (uint64_t)idx, thr, inv_keep) : 0.0f;
            rd[r] += dp * p;
            strip[ii * SW + OFF + j] = dp;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) rd[r] += __shfl_xor(rd[r], o, 64);
    }
    // pass 2: dS = P (dP - rowsum); every lane revisits exactly the strip entries it wrote
    for (int jt = 0; jt < KT; ++jt) {
        const int j = 32 * jt + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = crow(r, lane), i = i0 + ii;
            const bool ok = i < L && j < L;
            const int64_t idx = (prob * L + i) * L + j;
            const float p = ok ? probs[idx] : 0.0f;
            const float ds = p * (strip[ii * SW + OFF + j] - rd[r]);
            strip[ii * SW + OFF + j] = ds;
            if (ok) dSg[idx] = ds;
        }
    }
    __syncthreads();
    floatx16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = floatx16{0};
    // dS . K
    for (int jt = 0; jt < KT; ++jt) {
        float pa[16];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 t = *reinterpret_cast<const float4*>(strip + l31 * SW + OFF + 32 * jt + 16 * g + 4 * v);
            pa[4 * v] = t.x; pa[4 * v + 1] = t.y; pa[4 * v + 2] = t.z; pa[4 * v + 3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int jc = min(32 * jt + 16 * g + s, L - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + l31;
                const float kb = c < HD ? base[d + (int64_t)jc * ldq + c] : 0.0f;
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], kb, acc[ct], 0, 0, 0);
            }
        }
    }
    // skew(dS) . Erel_band :  skew[ii][x] = dS[ii][x + ii - 31]  (zero padding on both sides of the strip)
    const int rlo = L - 32 - i0;
    for (int xt = 0; xt <= KT; ++xt) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int x = 32 * xt + 16 * g + s;
            const float a = strip[l31 * SW + x + l31 + 1];
            const float* er = erel_row(e1, e2, h, L, HD, rlo + x);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + l31;
                const float eb = c < HD ? er[c] : 0.0f;
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, eb, acc[ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + crow(r, lane);
            if (i < L && c < HD) d_qkv[(n * L + i) * ldg + h * HD + c] = acc[ct][r] * scale;
        }
    }
}

// =====================================================================================================================
template <int HD>
__global__ __launch_bounds__(64) void relattn_gen_bwd_dkv_kernel(
    const float* __restrict__ d_ctx, int64_t ldo, const float* __restrict__ qkv, int64_t ldq,
    const float* __restrict__ probs, const float* __restrict__ dSg, float* __restrict__ d_qkv, int64_t ldg, int L, int H,
    float scale, uint32_t thr, float inv_keep, uint64_t seed) {
    constexpr int CT = (HD + 31) / 32;
    const int lane = threadIdx.x, g = lane >> 5, l31 = lane & 31;
    const int KT = (L + 31) / 32;
    const int64_t prob = blockIdx.x / KT;
    const int j0 = (int)(blockIdx.x % KT) * 32;
    const int64_t n = prob / H;
    const int h = (int)(prob % H);
    const int d = H * HD;
    const int jA = j0 + l31;
    floatx16 dk[CT], dv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dk[ct] = dv[ct] = floatx16{0};
    for (int it = 0; it < KT; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int i = 32 * it + 16 * g + s;
            const bool ok = i < L && jA < L;
            const int64_t idx = (prob * L + i) * L + jA;
            const float p = ok ? probs[idx] * drop_scale(seed, (uint64_t)idx, thr, inv_keep) : 0.0f;
            const float ds = ok ? dSg[idx] : 0.0f;
            const int ic = min(i, L - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 32 * ct + l31;
                const float dob = c < HD ? d_ctx[(n * L + ic) * ldo + h * HD + c] : 0.0f;
                const float qb = c < HD ? qkv[(n * L + ic) * ldq + h * HD + c] * scale : 0.0f;
                dv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, dob, dv[ct], 0, 0, 0);
                dk[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, qb, dk[ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + crow(r, lane);
            if (j < L && c < HD) {
                float* gp = d_qkv + (n * L + j) * ldg + h * HD + c;
                gp[d] = dk[ct][r];
                gp[2 * d] = dv[ct][r];
            }
        }
    }
}

// =====================================================================================================================
// grid = (H * RT, chunks).  ws [chunk][H][2L-1][HD]
template <int HD>
__global__ __launch_bounds__(64) void relattn_gen_bwd_de_kernel(const float* __restrict__ qkv, int64_t ldq,
                                                                const float* __restrict__ dSg, float* __restrict__ ws,
                                                                int64_t n_blocks, int blocks_per_chunk, int L, int H,
                                                                float scale) {
    constexpr int CT = (HD + 31) / 32;
    const int lane = threadIdx.x, g = lane >> 5, l31 = lane & 31;
    const int KT = (L + 31) / 32, NE = 2 * L - 1, RT = (NE + 31) / 32;
    const int h = blockIdx.x / RT;
    const int r0 = (blockIdx.x % RT) * 32;
    const int rA = r0 + l31;
    floatx16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = floatx16{0};
    const int64_t n_begin = (int64_t)blockIdx.y * blocks_per_chunk;
    const int64_t n_end = min(n_begin + blocks_per_chunk, n_blocks);
    for (int64_t n = n_begin; n < n_end; ++n) {
        const int64_t prob = n * H + h;
        for (int it = 0; it < KT; ++it) {
            const int jmin = r0 + 32 * it - (L - 1);          // smallest key index this tile pair can touch
            if (jmin + 62 < 0 || jmin >= L) continue;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int i = 32 * it + 16 * g + s;
                const int j = rA + i - (L - 1);
                const bool ok = i < L && j >= 0 && j < L && rA < NE;
                const float a = ok ? dSg[(prob * L + i) * L + j] : 0.0f;
                const int ic = min(i, L - 1);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = 32 * ct + l31;
                    const float qb = c < HD ? qkv[(n * L + ic) * ldq + h * HD + c] * scale : 0.0f;
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qb, acc[ct], 0, 0, 0);
                }
            }
        }
    }
    float* dst = ws + ((int64_t)blockIdx.y * H + h) * NE * HD;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + crow(r, lane);
            if (rr < NE && c < HD) dst[(int64_t)rr * HD + c] = acc[ct][r];
        }
    }
}

__global__ __launch_bounds__(256) void relattn_gen_de_split(const float* __restrict__ tot, int H, int L, int HD,
                                                            float* __restrict__ d_e1, float* __restrict__ d_e2) {
    const int NE = 2 * L - 1;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= H * NE * HD) return;
    const int c = o % HD, r = (o / HD) % NE, h = o / (HD * NE);
    const float acc = tot[o];
    if (r < L) d_e1[((int64_t)h * L + r) * HD + c] = acc;
    else d_e2[((int64_t)h * L + (r - L + 1)) * HD + c] = acc;
    if (r == 0) d_e2[((int64_t)h * L) * HD + c] = 0.0f;      // e2 row 0 is never read by the closed form
}

// ---- host side ------------------------------------------------------------------------------------------------------
static int gen_chunks(int64_t n_blocks, int L, int H) {
    const int RT = (2 * L - 1 + 31) / 32;
    const int64_t want = std::max<int64_t>(1, 2048 / ((int64_t)H * RT));
    return (int)std::min<int64_t>(n_blocks, want);
}

bool relattn_gen_supported(int L, int H, int hd) {
    return L >= 1 && L <= kGenMaxL && H >= 1 && (hd == 16 || hd == 32 || hd == 64);
}

int64_t relattn_gen_bwd_workspace(int64_t n_blocks, int L, int H, int hd) {
    const int64_t ds = n_blocks * H * (int64_t)L * L;
    const int64_t part = ((int64_t)gen_chunks(n_blocks, L, H) + 1) * H * (2 * L - 1) * hd;
    return (round_up(ds, 64) + part) * (int64_t)sizeof(float);
}

template <int HD>
static int gen_fwd_t(const float* qkv, int64_t ldq, const float* e1, const float* e2, float* ctx, int64_t ldo, float* probs,
                     int64_t n_blocks, int L, int H, float drop_p, uint64_t seed, hipStream_t s) {
    const size_t lds = (size_t)32 * gen_sw_fwd(L) * sizeof(float);
    auto kern = relattn_gen_fwd_kernel<HD>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t grid = n_blocks * H * gen_kt(L);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, s, qkv, ldq, e1, e2, ctx, ldo, probs, L, H,
                       1.0f / sqrtf((float)HD), drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed);
    VQ_CHECK_LAUNCH("relattn_gen_fwd");
    return VQCPC_OK;
}

template <int HD>
static int gen_bwd_t(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const float* probs, const float* e1,
                     const float* e2, float* d_qkv, int64_t ldg, float* d_e1, float* d_e2, int64_t n_blocks, int L, int H,
                     float drop_p, uint64_t seed, float* ws, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)HD), inv_keep = 1.0f / (1.0f - drop_p);
    const uint32_t thr = drop_threshold(drop_p);
    float* dSg = ws;
    float* part = ws + round_up(n_blocks * H * (int64_t)L * L, 64);
    const int64_t grid = n_blocks * H * gen_kt(L);
    {
        const size_t lds = (size_t)32 * gen_sw_bwd(L) * sizeof(float);
        auto kern = relattn_gen_bwd_dq_kernel<HD>;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds, s, d_ctx, ldo, qkv, ldq, probs, e1, e2, d_qkv, ldg, dSg,
                           L, H, scale, thr, inv_keep, seed);
        VQ_CHECK_LAUNCH("relattn_gen_bwd_dq");
    }
    hipLaunchKernelGGL(relattn_gen_bwd_dkv_kernel<HD>, dim3((unsigned)grid), dim3(64), 0, s, d_ctx, ldo, qkv, ldq, probs,
                       dSg, d_qkv, ldg, L, H, scale, thr, inv_keep, seed);
    VQ_CHECK_LAUNCH("relattn_gen_bwd_dkv");
    const int chunks = gen_chunks(n_blocks, L, H);
    const int bpc = (int)ceil_div(n_blocks, chunks);
    const int nchunk = (int)ceil_div(n_blocks, bpc);
    const int RT = (2 * L - 1 + 31) / 32;
    hipLaunchKernelGGL(relattn_gen_bwd_de_kernel<HD>, dim3(H * RT, nchunk), dim3(64), 0, s, qkv, ldq, dSg, part, n_blocks,
                       bpc, L, H, scale);
    VQ_CHECK_LAUNCH("relattn_gen_bwd_de");
    const int total = H * (2 * L - 1) * HD;
    float* tot = part + (int64_t)chunks * total;
    int rc = launch_reduce_splits(part, total, nchunk, tot, total, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(relattn_gen_de_split, dim3(ceil_div(total, 256)), dim3(256), 0, s, tot, H, L, HD, d_e1, d_e2);
    VQ_CHECK_LAUNCH("relattn_gen_de_split");
    return VQCPC_OK;
}

int relattn_gen_fwd(const float* qkv, int64_t ldq, const float* e1, const float* e2, float* ctx, int64_t ldo, float* probs,
                    int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed, hipStream_t s) {
    if (hd == 16) return gen_fwd_t<16>(qkv, ldq, e1, e2, ctx, ldo, probs, n_blocks, L, H, drop_p, seed, s);
    if (hd == 32) return gen_fwd_t<32>(qkv, ldq, e1, e2, ctx, ldo, probs, n_blocks, L, H, drop_p, seed, s);
    return gen_fwd_t<64>(qkv, ldq, e1, e2, ctx, ldo, probs, n_blocks, L, H, drop_p, seed, s);
}

int relattn_gen_bwd(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const float* probs, const float* e1,
                    const float* e2, float* d_qkv, int64_t ldg, float* d_e1, float* d_e2, int64_t n_blocks, int L, int H,
                    int hd, float drop_p, uint64_t seed, float* ws, hipStream_t s) {
    if (hd == 16)
        return gen_bwd_t<16>(d_ctx, ldo, qkv, ldq, probs, e1, e2, d_qkv, ldg, d_e1, d_e2, n_blocks, L, H, drop_p, seed, ws, s);
    if (hd == 32)
        return gen_bwd_t<32>(d_ctx, ldo, qkv, ldq, probs, e1, e2, d_qkv, ldg, d_e1, d_e2, n_blocks, L, H, drop_p, seed, ws, s);
    return gen_bwd_t<64>(d_ctx, ldo, qkv, ldq, probs, e1, e2, d_qkv, ldg, d_e1, d_e2, n_blocks, L, H, drop_p, seed, ws, s);
}

}  // namespace vq
