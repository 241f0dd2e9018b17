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
#include "common.h"

namespace vq {

typedef float floatx16 __attribute__((ext_vector_type(16)));

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
};

// bijective XCD-aware remap: consecutive tiles (which share an A row panel) land on the same XCD / L2
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

__global__ __launch_bounds__(kGemmThreads, 2) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                                 int K, int tiles_n, EpiParams ep) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_S];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(t / tiles_n) * BM;
    const int n0 = (t % tiles_n) * BN;

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // global -> register staging: 4 float4 of A and 4 of B per thread per k-tile
    const int ld_row = tid >> 3, ld_c4 = (tid & 7) * 4;
    float4 ra[4], rb[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ld_row + 32 * i;
            const bool kok = k0 + ld_c4 < K;
            ra[i] = (kok && m0 + r < M) ? *reinterpret_cast<const float4*>(A + (m0 + r) * lda + k0 + ld_c4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (kok && n0 + r < N) ? *reinterpret_cast<const float4*>(B + (int64_t)(n0 + r) * ldb + k0 + ld_c4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ld_row + 32 * i;
            *reinterpret_cast<float4*>(As + r * LDS_S + ld_c4) = ra[i];
            *reinterpret_cast<float4*>(Bs + r * LDS_S + ld_c4) = rb[i];
        }
    };

    load_tiles(0);
    store_tiles();
    __syncthreads();
    const float* ap = As + (wm * 64 + li) * LDS_S + kh * 4;
    const float* bp = Bs + (wn * 64 + li) * LDS_S + kh * 4;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) load_tiles(k0 + BK);           // HBM/L2 latency hides under the MFMAs below
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
            __syncthreads();
            store_tiles();
            __syncthreads();
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + wn * 64 + nt * 32 + li;
            if (col >= N) continue;
            const float bv = ep.bias ? ep.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= M) continue;
                float v = acc[mt][nt][r] + bv;
                if (ep.act == 1) v = fmaxf(v, 0.0f);
                if (ep.thr) v *= drop_scale(ep.seed, (uint64_t)row * N + col, ep.thr, ep.inv_keep);
                if (ep.gate) v *= (ep.gate[row * ep.ldgate + col] > 0.0f ? ep.gate_scale : 0.0f);
                if (ep.add) v += ep.add[row * ep.ldadd + col];
                C[row * ldc + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr int TM = 32;   // rows of the contraction (M) dimension per step

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
    float4 ra[4], rb[4];
    auto load_tiles = [&](int64_t mm) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = mm + ld_row + 8 * i;
            const bool rok = r < m_end;
            ra[i] = (rok && n0 + ld_c4 < N) ? *reinterpret_cast<const float4*>(A + r * lda + n0 + ld_c4)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (rok && k0 + ld_c4 < K) ? *reinterpret_cast<const float4*>(B + r * ldb + k0 + ld_c4)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ld_row + 8 * i;
            *reinterpret_cast<float4*>(As + r * BM + ld_c4) = ra[i];
            *reinterpret_cast<float4*>(Bs + r * BN + ld_c4) = rb[i];
        }
    };

    if (m_begin < m_end) {
        load_tiles(m_begin);
        store_tiles();
    }
    __syncthreads();
    const float* ap = As + kh * BM + wm * 64 + li;
    const float* bp = Bs + kh * BN + wn * 64 + li;
    for (int64_t mm = m_begin; mm < m_end; mm += TM) {
        const bool more = mm + TM < m_end;
        if (more) load_tiles(mm + TM);
#pragma unroll
        for (int s = 0; s < TM / 2; ++s) {
            const float a0 = ap[2 * s * BM], a1 = ap[2 * s * BM + 32];
            const float b0 = bp[2 * s * BN], b1 = bp[2 * s * BN + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            bsum[0] += a0;
            bsum[1] += a1;
        }
        if (more) {
            __syncthreads();
            store_tiles();
            __syncthreads();
        }
    }

    float* out = ws + (int64_t)blockIdx.y * N * K;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = k0 + wn * 64 + nt * 32 + li;
            if (col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < N) out[(int64_t)row * K + col] = acc[mt][nt][r];
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

static int tn_splits(int64_t M, int N, int K) {
    const int64_t tiles = ceil_div(N, BM) * ceil_div(K, BN);
    int64_t s = ceil_div(1024, tiles);
    s = std::min<int64_t>(s, ceil_div(M, 8 * TM));   // at least 256 rows per split
    return (int)std::max<int64_t>(s, 1);
}

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                  const float* bias, int act, float drop_p, uint64_t seed, const float* gate, int64_t ldgate,
                  float gate_scale, const float* add, int64_t ldadd, void* stream) {
    VQ_REQUIRE(A && B && C, "gemm_nt: null pointer");
    VQ_REQUIRE(M >= 0 && N >= 1 && K >= 4 && K % 4 == 0, "gemm_nt: bad shape M=%lld N=%d K=%d (K %% 4 == 0 required)",
               (long long)M, N, K);
    VQ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N, "gemm_nt: bad leading dimensions");
    VQ_REQUIRE(aligned16(A) && aligned16(B), "gemm_nt: A and B must be 16-byte aligned");
    VQ_REQUIRE(act == 0 || act == 1, "gemm_nt: act must be 0 (none) or 1 (relu)");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "gemm_nt: bad dropout probability");
    VQ_REQUIRE((!gate || ldgate >= N) && (!add || ldadd >= N), "gemm_nt: bad gate/add strides");
    if (M == 0) return VQCPC_OK;
    const int tiles_n = (int)ceil_div(N, BN);
    const int64_t tiles = ceil_div(M, BM) * tiles_n;
    VQ_REQUIRE(tiles < (1ll << 31), "gemm_nt: too many tiles");
    EpiParams ep{bias, act, drop_threshold(drop_p), 1.0f / (1.0f - drop_p), seed, gate, ldgate, gate_scale, add, ldadd};
    hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)tiles), dim3(kGemmThreads), 0, (hipStream_t)stream, A, lda, B, ldb, C,
                       ldc, M, N, K, tiles_n, ep);
    VQ_CHECK_LAUNCH("gemm_nt");
    return VQCPC_OK;
}

int64_t vqcpc_gemm_tn_workspace(int64_t M, int N, int K) {
    const int s = tn_splits(std::max<int64_t>(M, 1), N, K);
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
    const int splits = tn_splits(M, N, K);
    const int64_t rows_per_split = round_up(ceil_div(M, splits), TM);
    const int tiles_k = (int)ceil_div(K, BN);
    const int tiles = (int)ceil_div(N, BM) * tiles_k;
    float* ws = (float*)workspace;
    float* ws_bias = db ? ws + (int64_t)splits * N * K : nullptr;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(kGemmThreads), 0, s, A, lda, B, ldb, M, N, K, tiles_k,
                       rows_per_split, ws, ws_bias);
    VQ_CHECK_LAUNCH("gemm_tn");
    int rc = launch_reduce_splits(ws, (int64_t)N * K, splits, dW, (int64_t)N * K, accumulate, s);
    if (rc) return rc;
    if (db) rc = launch_reduce_splits(ws_bias, N, splits, db, N, accumulate, s);
    return rc;
}

}  // extern "C"
