// Wavefront-primitive kernels: fused embedding-table gather + positional concat, fused residual + dropout + LayerNorm.
#include <stdlib.h>

#include "common.h"
#include "gemm_common.h"   // round4_bf16 (bf16 copies of LayerNorm outputs for the bf16 GEMM path)

namespace vq {

// =====================================================================================================================
// embedding + positional (forward): one float4 per lane, grid-stride over [n_rows][d/4]
// =====================================================================================================================
__global__ __launch_bounds__(256) void embed_pos_fwd_kernel(const int64_t* __restrict__ tokens, int64_t n_rows, int tpb,
                                                            int nv, const float* __restrict__ table, int vmax, int dlin,
                                                            const float* __restrict__ chan, const float* __restrict__ ev,
                                                            int pos, float* __restrict__ out) {
    const int d = dlin + (ev ? 2 : 1) * pos;          // ev == nullptr: no event part (teacher input)
    const int d4 = d >> 2;
    const int64_t total = n_rows * d4;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (; e < total; e += step) {
        const int64_t row = e / d4;
        const int col = (int)(e - row * d4) << 2;
        const int p = (int)(row % tpb);
        const int v = p % nv, evt = p / nv;
        float4 val;
        if (col < dlin) {
            const int64_t tok = tokens[row];
            val = *reinterpret_cast<const float4*>(table + ((int64_t)v * vmax + tok) * dlin + col);
        } else if (col < dlin + pos) {
            val = *reinterpret_cast<const float4*>(chan + v * pos + (col - dlin));
        } else {
            val = *reinterpret_cast<const float4*>(ev + evt * pos + (col - dlin - pos));
        }
        *reinterpret_cast<float4*>(out + row * d + col) = val;
    }
}

// =====================================================================================================================
// embedding backward, stage 1.  grid = (nchunks, n_voices).  Workgroup (chunk, v) walks the rows of voice v in its chunk;
// lane `c` owns column c of an LDS table [vmax][d] (+ one row of positional sums), so the read-modify-write needs no
// atomics and is deterministic.  Partials: ws[chunk][v][vmax + 1][d]  (row vmax = sums of the chan|event columns,
// event sums are further split per event in a second small array).
// =====================================================================================================================
constexpr int kEmbRowsPerChunk = 2048;   // rows of ALL voices per chunk (large inputs)
constexpr int kEmbU = 16;
// Small inputs (the student step: 8 sequences x 384 tokens) get shorter chunks: with 2048-row chunks two chunks x four
// voices = 8 workgroups each walked 512 rows one batch of 16 after the other (190 us); ~48 chunks keep the walk to a few
// batches per workgroup at the price of 48 partial tables for the reduction.
static int64_t emb_chunk_rows(int64_t n_rows, int tpb) {
    if (n_rows >= 32 * kEmbRowsPerChunk) return kEmbRowsPerChunk;
    const int64_t r = std::max<int64_t>(1, ceil_div(ceil_div(n_rows, 48), tpb)) * tpb;
    return std::min<int64_t>(r, kEmbRowsPerChunk);
}

// PARTIAL = the last wave of the column loop owns fewer than 16 columns (d % 64 in [1, 15]): see the loop
template <bool PARTIAL>
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(const int64_t* __restrict__ tokens, int64_t n_rows, int tpb,
                                                            int nv, int vmax, int dlin, int pos, int has_ev,
                                                            const float* __restrict__ g, float* __restrict__ ws,
                                                            int chunk_rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int d = dlin + (has_ev ? 2 : 1) * pos;
    const int nev = has_ev ? tpb / nv : 0;
    // LDS: tab [vmax][dlin] | chan_sum [pos] | ev_sum [nev][pos]
    const int lds_floats = vmax * dlin + pos + nev * pos;
    for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) lds[i] = 0.0f;
    __syncthreads();
    // voice is the FAST launch index: the nv workgroups of a chunk run together and their interleaved rows (row % nv == v)
    // make one contiguous stream (with the chunk index fast, concurrent workgroups were 2048 rows apart)
    const int v = blockIdx.x % nv;
    const int chunk = blockIdx.x / nv;
    const int64_t row0 = (int64_t)chunk * chunk_rows;
    const int64_t row1 = min(row0 + chunk_rows, n_rows);
    // rows of this voice: row % tpb % nv == v.  chunk_rows is a multiple of tpb (checked on the host).
    // The token ids of a batch are handed out with v_readlane from lanes 0..15, which must have executed the load.  With
    // d % 64 in [1, 15] the last wave of the plain column loop has fewer than 16 active lanes: the PARTIAL instantiation runs
    // the loop over wave-uniform bases with every lane active (lanes past the last column read column 0 and skip the LDS
    // update).  It is 4 x slower (386 vs 98 us at C1: the all-lanes form keeps 32 more loads per batch in flight and a
    // branch around every update), so the host selects it only for those widths.
    for (int colb = PARTIAL ? (int)(threadIdx.x & ~63u) : (int)threadIdx.x; colb < d; colb += blockDim.x) {
        const bool cok = !PARTIAL || colb + (int)(threadIdx.x & 63) < d;
        const int col = !PARTIAL ? colb : (cok ? colb + (int)(threadIdx.x & 63) : 0);
        const int kind = col < dlin ? 0 : (col < dlin + pos ? 1 : 2);                 // table column | channel | event
        const int cbase = kind == 0 ? col : (kind == 1 ? vmax * dlin + (col - dlin) : vmax * dlin + pos + (col - dlin - pos));
        const int cmul = kind == 0 ? dlin : (kind == 2 ? pos : 0);
        // kEmbU rows in flight per lane, branch-free (the tail re-reads the last row of the voice and adds zero): issue
        // the loads first, then the (ordered) LDS read-modify-writes
        const int64_t last = row0 + v + (row1 - 1 - row0 - v) / nv * nv;         // last row of this voice in the chunk
        // event index of a row = (row % tpb) / nv.  row0 is a multiple of tpb and the rows of a voice are nv apart, so the index
        // of consecutive rows of the voice simply counts 0, 1, .., nev - 1, 0, ..: kept as a wrapped counter (the 64-bit
        // modulo / division per row were ~20 000 scalar instructions per wave: counters in profiles/r03_pmc_kernels_c1.txt)
        const int nevw = tpb / nv;
        int ev_next = 0;
        for (int64_t base = row0 + v; base < row1; base += (int64_t)nv * kEmbU) {
            float gv[kEmbU];
            int tk[kEmbU], evi[kEmbU];
            // token ids of the batch: one load (lane u: row u) + v_readlane instead of one broadcast load per row (the same
            // change took block_table_segsum from 3.1 to 4.3 TB/s: half the vector-memory operations of a batch)
            const int tokv = (int)tokens[min(base + (int64_t)((int)(threadIdx.x & 63) % kEmbU) * nv, last)];
#pragma unroll
            for (int u = 0; u < kEmbU; ++u) {
                const int64_t want = base + (int64_t)u * nv;
                const int64_t row = min(want, last);
                const float x = g[row * d + col];
                gv[u] = (cok && want < row1) ? x : 0.0f;
                tk[u] = __builtin_amdgcn_readlane(tokv, u);
                evi[u] = ev_next;                          // rows clamped to `last` add zero: their index is irrelevant
                ev_next = ev_next + 1 == nevw ? 0 : ev_next + 1;
            }
            // tab | csum | esum are one contiguous LDS array: one branch-free cell index per lane (a per-element
            // if / else-if / else made the wave that holds the 16 positional columns run all three arms for every row:
            // 465 -> 324 us at C1; requesting the next batch before these read-modify-writes was slower again, 423 us)
#pragma unroll
            for (int u = 0; u < kEmbU; ++u)
                if (cok) lds[cbase + (kind == 2 ? evi[u] : tk[u]) * cmul] += gv[u];
        }
    }
    __syncthreads();
    float* dst = ws + ((int64_t)chunk * nv + v) * lds_floats;
    for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) dst[i] = lds[i];
}

// =====================================================================================================================
// residual + dropout + LayerNorm.  One wavefront per row, 4 rows per workgroup, lane owns columns lane*4 + 256*it.
// =====================================================================================================================
constexpr int kLnMaxIt = 4;   // d <= 1024

// streaming accesses of the LayerNorm kernels (every element is touched once): non-temporal loads / stores keep them out
// of the way of the caches -- backward 225 -> 199 us per call at C1
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
    nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
}

// four bf16 values (8 bytes) of a streamed row -> fp32 (exact)
__device__ __forceinline__ float4 nt_load4_bf16(const unsigned short* p) {
    typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
    const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p));
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xFFFF0000u));
}

// NIT = 256-column groups per row (as in the backward): gamma / beta stay in registers for the whole row loop, and the NEXT row of
// the wave is requested before the current one is reduced and written (a row is two dependent wave reductions: without the
// prefetch a wave has one 1 KB load in flight)
// XB16 (round 5, bf16 path, !HAS_R): the residual sum `x` was written in bf16 by the producing GEMM epilogue (2 bytes per element in)
template <bool HAS_R, int NIT = kLnMaxIt, bool XB16 = false>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ r, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int64_t M,
                                                         int d, float eps, uint32_t thr, float inv_keep, uint64_t seed,
                                                         unsigned short* __restrict__ y_b16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_d = 1.0f / (float)d;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gm[NIT], bt[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = lane * 4 + it * 256;
        gm[it] = col < d ? *reinterpret_cast<const float4*>(gamma + col) : zero4;
        bt[it] = col < d ? *reinterpret_cast<const float4*>(beta + col) : zero4;
    }
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    float4 xn[NIT], rn[NIT];                       // raw operands of the wave's next row
#define LNF_FETCH(ROW)                                                                   \
    _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                 \
        const int col = lane * 4 + it * 256;                                             \
        const bool ok = (ROW) < M && col < d;                                            \
        xn[it] = !ok ? zero4 : XB16 ? nt_load4_bf16(reinterpret_cast<const unsigned short*>(x) + (ROW) * ldx + col) \
                                    : nt_load4(x + (ROW) * ldx + col);                   \
        if (HAS_R) rn[it] = ok ? nt_load4(r + (ROW) * d + col) : zero4;                  \
    }
    LNF_FETCH(row)
    for (; row < M; row += step) {
        float4 s[NIT];
        float sum = 0.0f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = lane * 4 + it * 256;
            float4 xv = xn[it];
            if (HAS_R && col < d) {
                const float4 rv = rn[it];
                const uint64_t e = (uint64_t)row * d + col;
                xv.x += rv.x * drop_scale(seed, e + 0, thr, inv_keep);
                xv.y += rv.y * drop_scale(seed, e + 1, thr, inv_keep);
                xv.z += rv.z * drop_scale(seed, e + 2, thr, inv_keep);
                xv.w += rv.w * drop_scale(seed, e + 3, thr, inv_keep);
            }
            s[it] = xv;                            // columns past d hold zeros: they add nothing to the sum
            sum += (xv.x + xv.y) + (xv.z + xv.w);
        }
        const int64_t nrow = row + step;
        LNF_FETCH(nrow)
        const float mu = wave_sum(sum) * inv_d;
        float sq = 0.0f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = lane * 4 + it * 256;
            if (col < d) {
                const float a = s[it].x - mu, b = s[it].y - mu, c = s[it].z - mu, e = s[it].w - mu;
                sq += (a * a + b * b) + (c * c + e * e);
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(sq) * inv_d + eps);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = lane * 4 + it * 256;
            if (col < d) {
                float4 o;
                o.x = (s[it].x - mu) * rs * gm[it].x + bt[it].x;
                o.y = (s[it].y - mu) * rs * gm[it].y + bt[it].y;
                o.z = (s[it].z - mu) * rs * gm[it].z + bt[it].z;
                o.w = (s[it].w - mu) * rs * gm[it].w + bt[it].w;
                if (y) nt_store4(y + row * d + col, o);       // NULL: the bf16 copy is the only consumer (bf16 path, round 5)
                if (y_b16) *reinterpret_cast<uint2*>(y_b16 + row * d + col) = round4_bf16(o);   // GEMM operand copy
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
#undef LNF_FETCH
}

// backward: d_s, d_r and per-workgroup partial d_gamma / d_beta (ws[block][2][d])
// HAS_R: x and r are separate inputs (s = x + dropout(r) is rebuilt here).  MASKED (and !HAS_R): `x` IS the residual sum s (the
// producing GEMM's epilogue formed it); the dropout mask of the sub-layer output is still regenerated from (seed, index)
// for d_r = d_s . mask -- one input stream fewer.
// NIT = 256-column groups per row (d <= 256 NIT): with the generic 4 the kernel holds 152 registers (3 waves per SIMD); NIT = 1
// (d <= 256) and 2 (d <= 512) keep the per-row arrays small enough for 8 / 5 waves -- more rows in flight per CU
// DYB16 (bf16 path, with XB16): the incoming gradient dy is bf16 as well -- the input gradient of the next sub-layer's first GEMM
// left its epilogue so (vqcpc_gemm_nt_bf16 with a bf16 output only); same arithmetic on the upcast values.
template <bool HAS_R, bool MASKED = HAS_R, int NIT = kLnMaxIt, bool XB16 = false, bool DYB16 = false>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         int64_t ldx, const float* __restrict__ r,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ d_s,
                                                         float* __restrict__ d_r, float* __restrict__ ws, int64_t M, int d,
                                                         uint32_t thr, float inv_keep, uint64_t seed,
                                                         unsigned short* __restrict__ dr_b16,
                                                         unsigned short* __restrict__ ds_b16) {
    constexpr int kRedW = NIT * 256;                    // columns a wave's partials span
    __shared__ float red[4 * 2 * kRedW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int nit = NIT;
    const float inv_d = 1.0f / (float)d;
    float4 dg[NIT], db[NIT], gm[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        dg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int col = lane * 4 + it * 256;
        gm[it] = (it < nit && col < d) ? *reinterpret_cast<const float4*>(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[NIT], gy[NIT], msk[NIT];
        float s1 = 0.0f, s2 = 0.0f;   // sum(g), sum(g * xhat) with g = dy * gamma
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = lane * 4 + it * 256;
            if (it < nit && col < d) {
                float4 xv = XB16 ? nt_load4_bf16(reinterpret_cast<const unsigned short*>(x) + row * ldx + col)
                                 : nt_load4(x + row * ldx + col);
                msk[it] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (MASKED) {
                    const uint64_t e = (uint64_t)row * d + col;
                    msk[it].x = drop_scale(seed, e + 0, thr, inv_keep);
                    msk[it].y = drop_scale(seed, e + 1, thr, inv_keep);
                    msk[it].z = drop_scale(seed, e + 2, thr, inv_keep);
                    msk[it].w = drop_scale(seed, e + 3, thr, inv_keep);
                }
                if (HAS_R) {
                    const float4 rv = nt_load4(r + row * d + col);
                    xv.x += rv.x * msk[it].x;
                    xv.y += rv.y * msk[it].y;
                    xv.z += rv.z * msk[it].z;
                    xv.w += rv.w * msk[it].w;
                }
                const float4 dv = DYB16 ? nt_load4_bf16(reinterpret_cast<const unsigned short*>(dy) + row * d + col)
                                        : nt_load4(dy + row * d + col);
                xh[it] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                gy[it] = make_float4(dv.x * gm[it].x, dv.y * gm[it].y, dv.z * gm[it].z, dv.w * gm[it].w);
                s1 += (gy[it].x + gy[it].y) + (gy[it].z + gy[it].w);
                s2 += (gy[it].x * xh[it].x + gy[it].y * xh[it].y) + (gy[it].z * xh[it].z + gy[it].w * xh[it].w);
                dg[it].x += dv.x * xh[it].x;
                dg[it].y += dv.y * xh[it].y;
                dg[it].z += dv.z * xh[it].z;
                dg[it].w += dv.w * xh[it].w;
                db[it].x += dv.x;
                db[it].y += dv.y;
                db[it].z += dv.z;
                db[it].w += dv.w;
            }
        }
        const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = lane * 4 + it * 256;
            if (it < nit && col < d) {
                float4 o;
                o.x = rs * (gy[it].x - m1 - xh[it].x * m2);
                o.y = rs * (gy[it].y - m1 - xh[it].y * m2);
                o.z = rs * (gy[it].z - m1 - xh[it].z * m2);
                o.w = rs * (gy[it].w - m1 - xh[it].w * m2);
                if (d_s) nt_store4(d_s + row * d + col, o);
                // bf16 path (round 5): the gradient of the residual branch may leave in bf16 only -- its one consumer is the
                // residual operand of the next input-gradient GEMM's epilogue (vqcpc_gemm_nt_bf16, add_bf16)
                if (ds_b16) *reinterpret_cast<uint2*>(ds_b16 + row * d + col) = round4_bf16(o);
                if (MASKED && (d_r != nullptr || dr_b16 != nullptr)) {
                    o.x *= msk[it].x;
                    o.y *= msk[it].y;
                    o.z *= msk[it].z;
                    o.w *= msk[it].w;
                    if (d_r != nullptr) nt_store4(d_r + row * d + col, o);      // bf16 path: only the bf16 copy is consumed
                }
                // bf16 copy of the gradient of the sub-layer output r (= d_s when there is no dropout): GEMM operand
                if (dr_b16) *reinterpret_cast<uint2*>(dr_b16 + row * d + col) = round4_bf16(o);
            }
        }
    }
    // reduce the 4 waves' column partials through LDS, then one partial per workgroup
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = lane * 4 + it * 256;
        if (it < nit && col < d) {
            *reinterpret_cast<float4*>(&red[(wave * 2 + 0) * kRedW + col]) = dg[it];
            *reinterpret_cast<float4*>(&red[(wave * 2 + 1) * kRedW + col]) = db[it];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d; i += 256) {
        const int which = i / d, col = i % d;
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += red[(w * 2 + which) * kRedW + col];
        ws[(int64_t)blockIdx.x * 2 * d + i] = acc;
    }
}

// embedding backward, stage 3: the reduced partials tmp[voice][table | chan | event] go to the three gradient tensors; the
// event sums also run over the voices (ascending).  One launch instead of a reduction per voice and per output.
__global__ __launch_bounds__(256) void embed_pos_scatter_kernel(const float* __restrict__ tmp, int nv, int64_t tab, int pos,
                                                                int nev, float* __restrict__ d_table,
                                                                float* __restrict__ d_chan, float* __restrict__ d_event) {
    const int64_t per = tab + pos + (int64_t)nev * pos;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n1 = (int64_t)nv * (tab + pos);
    if (i < n1) {
        const int v = (int)(i / (tab + pos));
        const int64_t j = i - (int64_t)v * (tab + pos);
        const float x = tmp[(int64_t)v * per + j];
        if (j < tab) d_table[(int64_t)v * tab + j] = x;
        else d_chan[(int64_t)v * pos + (j - tab)] = x;
    } else if (i < n1 + (int64_t)nev * pos) {
        const int64_t e = i - n1;
        float acc = 0.0f;
        for (int v = 0; v < nv; ++v) acc += tmp[(int64_t)v * per + tab + pos + e];
        d_event[e] = acc;
    }
}

static int ln_blocks(int64_t M) { return (int)std::min<int64_t>(ceil_div(M, 4), 2048); }
// backward: 512 workgroups (2 per CU) stream faster than 2048 and leave a quarter of the column partials to reduce:
// 573 -> 488 us at 557 056 x 256, 522 -> 516 us at 278 528 x 512 (tools/bench_ln.py; 384 and fewer fall off again)
// round 3, with the per-width specialisations of the backward kernel (72 / 98 registers at d <= 256 / 512 instead of 152): the
// one-input form (x is the residual sum) streams best from 1024 workgroups -- 471 -> 432 us at 557 056 x 256, 423 -> 366 us at
// 278 528 x 512 -- the two-input form still from 512 (513 vs 557 us)
// (at d = 1024, the generic kernel: 303 us from 512 workgroups, 318 from 1024)
static int ln_bwd_blocks(int64_t M, bool has_r, int d) {
    static const int cap_env = lab_env_int("VQCPC_LN_BWD_BLOCKS", 0);
    const int cap = cap_env > 0 ? cap_env : ((has_r || d > 512) ? 512 : 1024);
    return (int)std::min<int64_t>(ceil_div(M, 4), cap);
}
static int ln_bwd_blocks_max(int64_t M, int d) { return std::max(ln_bwd_blocks(M, true, d), ln_bwd_blocks(M, false, d)); }

// =====================================================================================================================
// Block-table gather / segment sum.  The input of the FIRST encoder layer takes only vmax * L distinct values (token id x
// position in the block), so its QKV projection is a (vmax*L) x 3d table and the per-token product is a lookup:
//   out[r][:] = table[tokens[r] * L + r % L][:]
// and the weight gradient needs only the per-table-row sums of d out:
//   d_table[t * L + p][:] = sum over rows r with tokens[r] == t and r % L == p of g[r][:]
// (three M x 3d x d GEMMs -- forward, dgrad, wgrad -- become one gather, one segment sum and three (vmax*L)-row GEMMs).
// =====================================================================================================================
__global__ __launch_bounds__(256) void block_table_gather_kernel(const float* __restrict__ table,
                                                                 const int64_t* __restrict__ tokens,
                                                                 float* __restrict__ out, int64_t M, int L, int C4) {
    const int64_t total = M * C4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / C4;
        const int c = (int)(e - r * C4);
        const int64_t id = tokens[r] * L + (r % L);
        reinterpret_cast<float4*>(out)[e] = reinterpret_cast<const float4*>(table)[id * C4 + c];
    }
}

// grid = (chunks, L, column tiles of 256).  Workgroup (chunk, p, ct): lane owns one column of an LDS accumulator
// [vmax][256]; it walks the rows of position p of its chunk of blocks in ascending order, kSegU rows in flight -> the
// read-modify-writes of a cell are ordered: deterministic, no atomics.  Partials ws[chunk][vmax][L][C].
// Measured at C1 (557 056 x 768 floats, tools/bench_segsum.py): 0.58 ms = 3.0 TB/s with 32 rows in flight per lane; a
// probe without the LDS update streams at 3.3 TB/s, so the LDS read-modify-write is not the limiter (and ds_add_f32
// atomics are 3x slower than the plain RMW here).
constexpr int kSegU = 32;

// G16: `g` points to bf16 elements (the bf16 path's first-layer attention backward writes d q | k | v so): half the bytes in, the
// same fp32 accumulation of the upcast values.  A lane then owns TWO adjacent columns (one dword per row: 2-byte loads run at a
// quarter of the rate, measured 5.7 ms against 1.36 ms for the fp32 form at configs[4]) and a workgroup 512 columns.
template <bool G16>
__global__ __launch_bounds__(256) void block_table_segsum_kernel(const float* __restrict__ g,
                                                                 const int64_t* __restrict__ tokens,
                                                                 float* __restrict__ ws, int64_t n_blocks,
                                                                 int blocks_per_chunk, int L, int vmax, int C) {
    constexpr int W = G16 ? 2 : 1;                                       // columns per lane
    extern __shared__ __attribute__((aligned(16))) float acc[];          // [vmax][256 W]
    for (int i = threadIdx.x; i < vmax * 256 * W; i += 256) acc[i] = 0.0f;
    __syncthreads();
    // (column tile, position) are the FAST launch indices: the L * ctiles workgroups of a chunk run together and read its
    // blocks as one contiguous stream (with the chunk index fast, concurrent workgroups each walked their own 1 KB-per-row
    // strided stream)
    const int ctiles = (C + 256 * W - 1) / (256 * W);
    const int ct = blockIdx.x % ctiles, p = (blockIdx.x / ctiles) % L, chunk = blockIdx.x / (ctiles * L);
    const int col = (ct * 256 + threadIdx.x) * W;
    const bool cok = col < C;
    const int64_t b0 = (int64_t)chunk * blocks_per_chunk;
    const int64_t b1 = min(b0 + blocks_per_chunk, n_blocks);
    const float* gp = g + (cok ? col : 0);
    const unsigned short* gp16 = reinterpret_cast<const unsigned short*>(g) + (cok ? col : 0);
    float* mine = acc + threadIdx.x * W;
    static_assert(kSegU <= 64, "one lane per row of a batch holds its token");
    for (int64_t b = b0; b < b1; b += kSegU) {
        float v[kSegU][W];
        // the token ids of the batch's rows: ONE load (lane u: row u) instead of one broadcast load per row -- half the vector-
        // memory operations of a batch; v_readlane hands them out as scalars
        const int tokv = (int)tokens[min(b + (int)(threadIdx.x & 63) % kSegU, b1 - 1) * L + p];
#pragma unroll
        for (int u = 0; u < kSegU; ++u) {                                // branch-free: the tail re-reads the last row ...
            const int64_t row = min(b + u, b1 - 1) * L + p;
            if constexpr (G16) {
                const uint32_t t = *reinterpret_cast<const uint32_t*>(gp16 + row * C);
                v[u][0] = b + u < b1 ? __uint_as_float(t << 16) : 0.0f;
                v[u][W - 1] = b + u < b1 ? __uint_as_float(t & 0xFFFF0000u) : 0.0f;
            } else {
                v[u][0] = b + u < b1 ? gp[row * C] : 0.0f;               // ... and adds zero
            }
        }
#pragma unroll
        for (int u = 0; u < kSegU; ++u) {                                // one lane per cell, rows ascending
            float* cell = mine + __builtin_amdgcn_readlane(tokv, u) * (256 * W);
#pragma unroll
            for (int w = 0; w < W; ++w) cell[w] += v[u][w];
        }
    }
    __syncthreads();
    if (cok) {
        float* dst = ws + (int64_t)chunk * vmax * L * C;
        for (int t = 0; t < vmax; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w) dst[((int64_t)t * L + p) * C + col + w] = acc[(t * 256 + threadIdx.x) * W + w];
    }
}

// Gradient of a plain row gather out[m] = table[idx[m]] for tables of any size (the decoder's source-code and shifted
// target-token embeddings: decoders/decoder.py:212-215,439,474-480).  The caller passes idx sorted (stable, so equal keys
// keep ascending m) with the permutation; the workgroup at the first position of a run sums the run's rows in that
// order -- deterministic, no atomics -- and rows nobody refers to keep the zero fill.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ g, int64_t ldg,
                                                            const int64_t* __restrict__ sorted_idx,
                                                            const int64_t* __restrict__ perm, float* __restrict__ d_table,
                                                            int64_t M, int C) {
    constexpr int U = 8;                                              // rows in flight per lane; added in ascending order
    const int64_t m = blockIdx.x;
    const int64_t v = sorted_idx[m];
    if (m > 0 && sorted_idx[m - 1] == v) return;
    int64_t lo = m, hi = M;                                           // end of the run: first position whose key differs
    while (lo + 1 < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted_idx[mid] == v) lo = mid; else hi = mid;
    }
    const int64_t end = hi;
    for (int col = threadIdx.x; col < C; col += 256) {
        float acc = 0.0f;
        for (int64_t mm = m; mm < end; mm += U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = mm + u < end ? g[perm[mm + u] * ldg + col] : 0.0f;
#pragma unroll
            for (int u = 0; u < U; ++u) acc += x[u];
        }
        d_table[v * C + col] = acc;
    }
}

static int segsum_chunks(int64_t n_blocks) { return (int)std::max<int64_t>(1, std::min<int64_t>(32, n_blocks / 256)); }

}  // namespace vq

using namespace vq;

extern "C" {

int vqcpc_embed_pos_fwd(const int64_t* tokens, int64_t n_rows, int tokens_per_block, int n_voices, const float* table,
                        int vmax, int dlin, const float* chan, const float* event, int pos, float* out, void* stream) {
    if (n_rows == 0) return VQCPC_OK;
    VQ_REQUIRE(tokens && table && chan && out, "embed_pos_fwd: null pointer");
    VQ_REQUIRE(n_rows >= 0 && tokens_per_block > 0 && n_voices > 0 && tokens_per_block % n_voices == 0 && vmax > 0,
               "embed_pos_fwd: bad shape");
    VQ_REQUIRE(dlin % 4 == 0 && pos % 4 == 0 && dlin > 0, "embed_pos_fwd: dlin and pos must be multiples of 4");
    VQ_REQUIRE(n_rows % tokens_per_block == 0, "embed_pos_fwd: n_rows must be a whole number of blocks");
    const int64_t total = n_rows * ((dlin + (event ? 2 : 1) * pos) / 4);
    const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 8192);
    hipLaunchKernelGGL(embed_pos_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tokens, n_rows,
                       tokens_per_block, n_voices, table, vmax, dlin, chan, event, pos, out);
    VQ_CHECK_LAUNCH("embed_pos_fwd");
    return VQCPC_OK;
}

int64_t vqcpc_embed_pos_bwd_workspace(int64_t n_rows, int tokens_per_block, int n_voices, int vmax, int dlin, int pos) {
    const int64_t nchunks = ceil_div(std::max<int64_t>(n_rows, 1), emb_chunk_rows(n_rows, tokens_per_block));
    const int64_t per = (int64_t)vmax * dlin + pos + (int64_t)(tokens_per_block / n_voices) * pos;
    return (nchunks + 1) * n_voices * per * (int64_t)sizeof(float);          // partials + one reduced [voice][per] block
}

int vqcpc_embed_pos_bwd(const int64_t* tokens, int64_t n_rows, int tokens_per_block, int n_voices, int vmax, int dlin,
                        int pos, const float* g_out, float* d_table, float* d_chan, float* d_event, void* workspace,
                        int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(tokens && g_out && d_table && d_chan && workspace, "embed_pos_bwd: null pointer");
    VQ_REQUIRE(n_rows > 0 && tokens_per_block > 0 && n_voices > 0 && tokens_per_block % n_voices == 0 &&
                   kEmbRowsPerChunk % tokens_per_block == 0 && n_rows % tokens_per_block == 0,
               "embed_pos_bwd: bad shape");
    const int nev = d_event ? tokens_per_block / n_voices : 0;
    const size_t lds = ((size_t)vmax * dlin + pos + (size_t)nev * pos) * sizeof(float);
    VQ_REQUIRE(lds <= 160 * 1024, "embed_pos_bwd: table of %d x %d floats does not fit the LDS", vmax, dlin);
    if (workspace_bytes < vqcpc_embed_pos_bwd_workspace(n_rows, tokens_per_block, n_voices, vmax, dlin, pos)) {
        set_error("embed_pos_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const int chunk_rows = (int)emb_chunk_rows(n_rows, tokens_per_block);
    const int nchunks = (int)ceil_div(n_rows, chunk_rows);
    hipStream_t s = (hipStream_t)stream;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)embed_pos_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
        (void)hipFuncSetAttribute((const void*)embed_pos_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int d_cols = dlin + (d_event ? 2 : 1) * pos;
    const bool partial_wave = (d_cols % 64) >= 1 && (d_cols % 64) <= 15;
    hipLaunchKernelGGL(partial_wave ? embed_pos_bwd_kernel<true> : embed_pos_bwd_kernel<false>, dim3(nchunks * n_voices), dim3(256), lds, s, tokens, n_rows,
                       tokens_per_block, n_voices, vmax, dlin, pos, d_event ? 1 : 0, g_out, (float*)workspace, chunk_rows);
    VQ_CHECK_LAUNCH("embed_pos_bwd");
    // stage 2: partials ws[chunk][voice][table | chan | event] -> parallel deterministic column reductions (a single
    // thread per output walking 272 chunks was latency-bound: 140 us)
    // (one reduction over the chunks for all voices and outputs at once + one scatter: 2 launches instead of 2 nv + 1)
    float* wsf = (float*)workspace;
    const int64_t per = (int64_t)vmax * dlin + pos + (int64_t)nev * pos;
    float* tmp = wsf + (int64_t)nchunks * n_voices * per;
    int rc = launch_reduce_splits(wsf, (int64_t)n_voices * per, nchunks, tmp, (int64_t)n_voices * per, 0, s);
    if (rc) return rc;
    const int64_t total = (int64_t)n_voices * ((int64_t)vmax * dlin + pos) + (int64_t)nev * pos;
    hipLaunchKernelGGL(embed_pos_scatter_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, tmp, n_voices,
                       (int64_t)vmax * dlin, pos, nev, d_table, d_chan, d_event);
    VQ_CHECK_LAUNCH("embed_pos_scatter");
    return VQCPC_OK;
}

int vqcpc_block_table_gather(const float* table, const int64_t* tokens, float* out, int64_t M, int L, int vmax, int C,
                             void* stream) {
    if (M == 0) return VQCPC_OK;
    VQ_REQUIRE(table && tokens && out && M >= 0 && L >= 1 && vmax >= 1 && C >= 4 && C % 4 == 0 && M % L == 0,
               "block_table_gather: bad arguments");
    VQ_REQUIRE(aligned16(table) && aligned16(out), "block_table_gather: buffers must be 16-byte aligned");
    const int blocks = (int)std::min<int64_t>(ceil_div(M * (C / 4), 256), 16384);
    hipLaunchKernelGGL(block_table_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, tokens, out, M, L,
                       C / 4);
    VQ_CHECK_LAUNCH("block_table_gather");
    return VQCPC_OK;
}

int64_t vqcpc_block_table_segsum_workspace(int64_t M, int L, int vmax, int C) {
    return (int64_t)segsum_chunks(std::max<int64_t>(M, 1) / std::max(L, 1)) * vmax * L * C * (int64_t)sizeof(float);
}

static int segsum_launch(const float* g, bool g16, const int64_t* tokens, float* d_table, int64_t M, int L, int vmax, int C,
                         void* workspace, int64_t workspace_bytes, void* stream) {
    VQ_REQUIRE(g && tokens && d_table && workspace && M >= 1 && L >= 1 && vmax >= 1 && C >= 1 && M % L == 0,
               "block_table_segsum: bad arguments");
    const int W = g16 ? 2 : 1;
    const size_t lds = (size_t)vmax * 256 * W * sizeof(float);
    VQ_REQUIRE(lds <= 160 * 1024, "block_table_segsum: vocabulary of %d tokens does not fit the LDS accumulator", vmax);
    VQ_REQUIRE(!g16 || (C % 2 == 0 && (reinterpret_cast<uintptr_t>(g) & 3u) == 0), "block_table_segsum_b16: C must be even, g 4-byte aligned");
    if (workspace_bytes < vqcpc_block_table_segsum_workspace(M, L, vmax, C)) {
        set_error("block_table_segsum: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    const int64_t n_blocks = M / L;
    const int chunks = segsum_chunks(n_blocks);
    const int bpc = (int)ceil_div(n_blocks, chunks);
    const int nchunk = (int)ceil_div(n_blocks, bpc);
    hipStream_t s = (hipStream_t)stream;
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)block_table_segsum_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)block_table_segsum_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (g16)
        hipLaunchKernelGGL(block_table_segsum_kernel<true>, dim3((unsigned)(nchunk * L * ceil_div(C, 512))), dim3(256), lds, s, g, tokens,
                           (float*)workspace, n_blocks, bpc, L, vmax, C);
    else
        hipLaunchKernelGGL(block_table_segsum_kernel<false>, dim3((unsigned)(nchunk * L * ceil_div(C, 256))), dim3(256), lds, s, g, tokens,
                           (float*)workspace, n_blocks, bpc, L, vmax, C);
    VQ_CHECK_LAUNCH("block_table_segsum");
    const int64_t total = (int64_t)vmax * L * C;
    return launch_reduce_splits((const float*)workspace, total, nchunk, d_table, total, 0, s);
}

int vqcpc_block_table_segsum(const float* g, const int64_t* tokens, float* d_table, int64_t M, int L, int vmax, int C,
                             void* workspace, int64_t workspace_bytes, void* stream) {
    return segsum_launch(g, false, tokens, d_table, M, L, vmax, C, workspace, workspace_bytes, stream);
}

int vqcpc_block_table_segsum_b16(const void* g_bf16, const int64_t* tokens, float* d_table, int64_t M, int L, int vmax, int C,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    return segsum_launch((const float*)g_bf16, true, tokens, d_table, M, L, vmax, C, workspace, workspace_bytes, stream);
}

int vqcpc_embedding_bwd(const float* g, int64_t ldg, const int64_t* sorted_idx, const int64_t* perm, float* d_table, int64_t M,
                        int64_t V, int C, void* stream) {
    VQ_REQUIRE(d_table && V >= 1 && C >= 1 && M >= 0, "embedding_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_table, 0, (size_t)V * C * sizeof(float), s) != hipSuccess) {
        set_error("embedding_bwd: memset failed");
        return VQCPC_ELAUNCH;
    }
    if (M == 0) return VQCPC_OK;
    VQ_REQUIRE(g && sorted_idx && perm && ldg >= C && M < (int64_t)1 << 31, "embedding_bwd: bad arguments");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)M), dim3(256), 0, s, g, ldg, sorted_idx, perm, d_table, M, C);
    VQ_CHECK_LAUNCH("embedding_bwd");
    return VQCPC_OK;
}

int vqcpc_add_layernorm_fwd(const float* x, int64_t ldx, const float* r, const float* gamma, const float* beta, float* y,
                            float* mean, float* rstd, int64_t M, int d, float eps, float drop_p, uint64_t seed,
                            void* stream) {
    return vqcpc_add_layernorm_fwd_b16(x, ldx, r, gamma, beta, y, nullptr, mean, rstd, M, d, eps, drop_p, seed, stream);
}

static int ln_fwd_launch(const void* xv, bool xb16, int64_t ldx, const float* r, const float* gamma, const float* beta, float* y,
                         void* y_bf16, float* mean, float* rstd, int64_t M, int d, float eps, float drop_p, uint64_t seed,
                         void* stream) {
    if (M == 0) return VQCPC_OK;
    const float* x = (const float*)xv;              // XB16 kernels reinterpret it
    VQ_REQUIRE(x && gamma && beta && (y || y_bf16) && mean && rstd, "add_layernorm_fwd: null pointer");
    VQ_REQUIRE(M >= 0 && d >= 4 && d % 4 == 0 && d <= 1024 && ldx % 4 == 0 && ldx >= d, "add_layernorm_fwd: bad shape");
    VQ_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "add_layernorm_fwd: bad dropout probability");
    VQ_REQUIRE(!xb16 || (!r && (reinterpret_cast<uintptr_t>(xv) & 7u) == 0), "layernorm_fwd_xb16: one bf16 input stream, 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t thr = drop_threshold(drop_p);
    const float ik = 1.0f / (1.0f - drop_p);
#define LN_FWD(HR, NITV, XB)                                                                                              \
    hipLaunchKernelGGL((add_ln_fwd_kernel<HR, NITV, XB>), dim3(ln_blocks(M)), dim3(256), 0, s, x, ldx, r, gamma, beta, y, mean, \
                       rstd, M, d, eps, thr, ik, seed, (unsigned short*)y_bf16)
#define LN_FWD_D(HR, XB)                    \
    if (d <= 256) LN_FWD(HR, 1, XB);        \
    else if (d <= 512) LN_FWD(HR, 2, XB);   \
    else LN_FWD(HR, 4, XB)
    if (xb16) {
        LN_FWD_D(false, true);
    } else if (r) {
        LN_FWD_D(true, false);
    } else {
        LN_FWD_D(false, false);
    }
#undef LN_FWD_D
#undef LN_FWD
    VQ_CHECK_LAUNCH("add_layernorm_fwd");
    return VQCPC_OK;
}

int vqcpc_add_layernorm_fwd_b16(const float* x, int64_t ldx, const float* r, const float* gamma, const float* beta, float* y,
                                void* y_bf16, float* mean, float* rstd, int64_t M, int d, float eps, float drop_p,
                                uint64_t seed, void* stream) {
    return ln_fwd_launch(x, false, ldx, r, gamma, beta, y, y_bf16, mean, rstd, M, d, eps, drop_p, seed, stream);
}

int vqcpc_layernorm_fwd_xb16(const void* x_bf16, int64_t ldx, const float* gamma, const float* beta, float* y, void* y_bf16,
                             float* mean, float* rstd, int64_t M, int d, float eps, void* stream) {
    return ln_fwd_launch(x_bf16, true, ldx, nullptr, gamma, beta, y, y_bf16, mean, rstd, M, d, eps, 0.0f, 0, stream);
}

int64_t vqcpc_add_layernorm_bwd_workspace(int64_t M, int d) {
    return (int64_t)ln_bwd_blocks_max(std::max<int64_t>(M, 1), d) * 2 * d * (int64_t)sizeof(float);
}

int vqcpc_add_layernorm_bwd(const float* dy, const float* x, int64_t ldx, const float* r, const float* gamma,
                            const float* mean, const float* rstd, float* d_s, float* d_r, float* d_gamma, float* d_beta,
                            int64_t M, int d, float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes,
                            void* stream) {
    return vqcpc_add_layernorm_bwd_b16(dy, x, ldx, r, gamma, mean, rstd, d_s, d_r, nullptr, d_gamma, d_beta, M, d, drop_p, seed,
                                       workspace, workspace_bytes, stream);
}

static int ln_bwd_launch(const float* dy, const void* xv, bool xb16, int64_t ldx, const float* r, const float* gamma,
                         const float* mean, const float* rstd, float* d_s, float* d_r, void* d_r_bf16, float* d_gamma,
                         float* d_beta, int64_t M, int d, float drop_p, uint64_t seed, void* workspace,
                         int64_t workspace_bytes, void* stream, void* d_s_bf16 = nullptr, bool dyb16 = false) {
    // d_gamma == d_beta == NULL: the column partials stay in `workspace` ([vqcpc_add_layernorm_bwd_partials(M, d, r != NULL)][2 d]: d gamma | d beta)
    // for the caller to reduce later (vqcpc_reduce_grouped: the trainers sum the partials of every LayerNorm of a backward pass
    // in one launch)
    const float* x = (const float*)xv;              // XB16 kernels reinterpret it
    VQ_REQUIRE(dy && x && gamma && mean && rstd && (d_s || d_s_bf16) && workspace && ((d_gamma != nullptr) == (d_beta != nullptr)),
               "add_layernorm_bwd: null pointer");
    VQ_REQUIRE(M >= 1 && d >= 4 && d % 4 == 0 && d <= 1024 && ldx % 4 == 0 && ldx >= d, "add_layernorm_bwd: bad shape");
    VQ_REQUIRE(!d_s_bf16 || (reinterpret_cast<uintptr_t>(d_s_bf16) & 7u) == 0, "layernorm_bwd: bf16 d_s must be 8-byte aligned");
    VQ_REQUIRE(!xb16 || (!r && (reinterpret_cast<uintptr_t>(xv) & 7u) == 0), "layernorm_bwd_xb16: one bf16 input stream, 8-byte aligned");
    VQ_REQUIRE(!dyb16 || (xb16 && (reinterpret_cast<uintptr_t>(dy) & 7u) == 0), "layernorm_bwd_b16io: bf16 dy goes with a bf16 x, 8-byte aligned");
    if (workspace_bytes < vqcpc_add_layernorm_bwd_workspace(M, d)) {
        set_error("add_layernorm_bwd: workspace too small");
        return VQCPC_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const uint32_t thr = drop_threshold(drop_p);
    const float ik = 1.0f / (1.0f - drop_p);
    const int blocks = ln_bwd_blocks(M, r != nullptr, d);
#define LN_BWD(HR, MK, NITV, XB, ...)                                                                                     \
    hipLaunchKernelGGL((add_ln_bwd_kernel<HR, MK, NITV, XB, ##__VA_ARGS__>), dim3(blocks), dim3(256), 0, s, dy, x, ldx, r, gamma, mean, rstd, \
                       d_s, d_r, (float*)workspace, M, d, thr, ik, seed, (unsigned short*)d_r_bf16, (unsigned short*)d_s_bf16)
#define LN_BWD_D(HR, MK, XB, ...)                             \
    if (d <= 256) LN_BWD(HR, MK, 1, XB, ##__VA_ARGS__);       \
    else if (d <= 512) LN_BWD(HR, MK, 2, XB, ##__VA_ARGS__);  \
    else LN_BWD(HR, MK, 4, XB, ##__VA_ARGS__)
    const bool masked = thr && (d_r || d_r_bf16);   // x is the residual sum s = x0 + dropout(r) itself: d_r = d_s . mask(seed), no r stream
    if (dyb16) {
        if (masked) { LN_BWD_D(false, true, true, true); } else { LN_BWD_D(false, false, true, true); }
    } else if (xb16) {
        if (masked) { LN_BWD_D(false, true, true); } else { LN_BWD_D(false, false, true); }
    } else if (r) {
        LN_BWD_D(true, true, false);
    } else if (masked) {
        LN_BWD_D(false, true, false);
    } else {
        LN_BWD_D(false, false, false);
    }
#undef LN_BWD_D
#undef LN_BWD
    VQ_CHECK_LAUNCH("add_layernorm_bwd");
    if (!d_gamma) return VQCPC_OK;
    return launch_reduce_splits2((const float*)workspace, (int64_t)2 * d, blocks, d_gamma, d, (const float*)workspace + d,
                                 (int64_t)2 * d, d_beta, d, 0, s);
}

int vqcpc_add_layernorm_bwd_b16(const float* dy, const float* x, int64_t ldx, const float* r, const float* gamma,
                                const float* mean, const float* rstd, float* d_s, float* d_r, void* d_r_bf16, float* d_gamma,
                                float* d_beta, int64_t M, int d, float drop_p, uint64_t seed, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    return ln_bwd_launch(dy, x, false, ldx, r, gamma, mean, rstd, d_s, d_r, d_r_bf16, d_gamma, d_beta, M, d, drop_p, seed, workspace,
                         workspace_bytes, stream);
}

int vqcpc_layernorm_bwd_xb16(const float* dy, const void* x_bf16, int64_t ldx, const float* gamma, const float* mean,
                             const float* rstd, float* d_s, void* d_s_bf16, float* d_r, void* d_r_bf16, float* d_gamma,
                             float* d_beta, int64_t M, int d, float drop_p, uint64_t seed, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    return ln_bwd_launch(dy, x_bf16, true, ldx, nullptr, gamma, mean, rstd, d_s, d_r, d_r_bf16, d_gamma, d_beta, M, d, drop_p, seed,
                         workspace, workspace_bytes, stream, d_s_bf16);
}

int vqcpc_layernorm_bwd_b16io(const void* dy_bf16, const void* x_bf16, int64_t ldx, const float* gamma, const float* mean,
                              const float* rstd, float* d_s, void* d_s_bf16, float* d_r, void* d_r_bf16, float* d_gamma,
                              float* d_beta, int64_t M, int d, float drop_p, uint64_t seed, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    return ln_bwd_launch((const float*)dy_bf16, x_bf16, true, ldx, nullptr, gamma, mean, rstd, d_s, d_r, d_r_bf16, d_gamma, d_beta, M, d,
                         drop_p, seed, workspace, workspace_bytes, stream, d_s_bf16, true);
}

int vqcpc_add_layernorm_bwd_partials(int64_t M, int d, int has_r) { return ln_bwd_blocks(std::max<int64_t>(M, 1), has_r != 0, d); }

}  // extern "C"
