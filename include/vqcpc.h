/* vqcpc.h -- C ABI of libvqcpc_hip.so: the MI355X (gfx950) kernels of the VQ-CPC encoder training step.
 *
 * The reference (SonyCSLParis/vqcpc-bach) is pure Python/PyTorch and has NO plugin / FFI interface; its "boundary" for
 * this path is the Python class surface.  Each entry point below therefore names the reference
 * tensor expression it replaces (file:line relative to /root/reference).  A maintainer binds them with ctypes
 * (see INTEGRATION.md); vqcpc_bach_amd/hip.py is exactly that binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative VQCPC_E* code and never throws; vqcpc_last_error() gives the
 *     thread-local message of the last failure;
 *   - all pointers are DEVICE pointers to caller-owned memory (fp32 / int64 / int32 as declared), 16-byte aligned,
 *     row-major; `ld*` arguments are row strides in ELEMENTS; nothing is allocated inside: kernels that need scratch take
 *     a caller-provided workspace whose size comes from the matching *_workspace() query (a pure host function);
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default stream); no hidden
 *     synchronisation; functions are stateless and re-entrant;
 *   - activations are "block-major": row = block * L + token, i.e. the reference's time-first (L, N, E) tensors
 *     transposed to (N, L, E) and flattened.
 */
#ifndef VQCPC_H
#define VQCPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQCPC_ABI_VERSION 2

#define VQCPC_OK 0
#define VQCPC_EINVAL (-1)    /* bad argument (shape, alignment, null pointer) */
#define VQCPC_ELAUNCH (-2)   /* HIP launch / runtime error */
#define VQCPC_EWORKSPACE (-3) /* workspace too small */

/* ------------------------------------------------------------------------------------------------------------------
 * PROCESS-WIDE STATE.  Every compute entry point is a pure function of its arguments, re-entrant and stream-ordered (work is
 * enqueued on the `stream` argument, nothing is synchronised), EXCEPT for the following settings, which are plain
 * process-wide values read at launch time.  They are configuration, not data: set them before the work they apply to is
 * enqueued and do not change them from a second thread while another one launches.
 *   vqcpc_gemm_set_mode        which GEMM ARITHMETIC vqcpc_gemm_* launch: 0 fp32 MFMA / 1 bf16x6 / 8 bf16 (read by vqcpc_gemm_nt,
 *                              _tn, _tn_grouped, _nt_splitk, _nt_relu_mask, _nt_gatebits and by the *_workspace / *_supported /
 *                              *_groupable queries, whose answers belong to the mode they were asked in).  The product library
 *                              has NO kernel-selection switch (round 5: the A/B bits +2 / +4 of this call and
 *                              vqcpc_gemm_bf16_set_variant exist in lab builds only)
 *   vqcpc_gemm_set_gradient_products / vqcpc_gemm_gradient_scope   the bf16-pair gradient arithmetic of round 3 (kept for
 *                              comparison): the scope is a counter, opened and closed by ONE thread around its backward pass.
 *                              The f16x3 gradient GEMMs that training uses (vqcpc_gemm_nt_grad / _tn_grad) are explicit entry
 *                              points whose only state -- the per-call-site scale floats -- is caller-owned device memory
 *   vqcpc_relattn_force_general                         A/B switch of the attention dispatch (tests)
 *   vqcpc_rng_salt_set / vqcpc_rng_salt_advance         a DEVICE-side value (one copy per translation unit) XOR-ed into every
 *                              dropout seed: 0 outside a replayed step graph; a captured step sets it in its first node and
 *                              puts it back to 0 in its last, so work enqueued after a replay on the same stream sees 0
 *   vqcpc_last_error                                    thread-local message of the last failing call of the calling thread
 * Several trainers in one process may therefore interleave their steps at STEP granularity, on one thread or on several
 * (tested: tests/test_graphs_gpu.py::test_two_trainers_interleaved_in_one_process_equal_each_run_apart and
 * ::test_two_trainers_stepping_from_two_threads_equal_each_run_apart): the Python host layer serialises training steps with one
 * process-wide lock (vqcpc_bach_amd/graphs.py STEP_LOCK -- torch.autograd runs every backward node of a device on ONE engine
 * thread, so two concurrent backward passes would interleave there whatever the caller's threads do); callers of the C ABI that
 * step from several threads must agree on the settings above.
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_abi_version(void);
const char* vqcpc_last_error(void);
/* returns and clears the HIP runtime's last (non-sticky) error of the calling thread: every launch of this library is checked with
 * hipGetLastError(), which would otherwise attribute an earlier failure of SOMEBODY ELSE's runtime call (a rejected stream capture)
 * to the next kernel launched here */
int vqcpc_clear_runtime_error(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Dropout RNG shared by all kernels: keep(seed, i) = (u24(mix32((uint32)i * 0x9E3779B1 + lo(seed) ^ hi(seed))) >= p * 2^24)
 * with mix32(x): x ^= x >> 16; y = mul24(x, 0x6B43A9); y ^= y >> 15; y = mul24(y, 0x52DCE7) + x; y ^= y >> 14 (mul24 = the low 32 bits of
 * the product of the operands' low 24 bits: the full-rate v_mul_u32_u24; round 5 -- rounds 1-4 used the 'lowbias32' finaliser, two
 * quarter-rate 32-bit multiplies per element).
 * vqcpc_dropout_mask writes that keep-mask (1.0f / 0.0f) for i in [0, n) so that tests can reproduce every
 * in-kernel mask (element index conventions are documented per kernel).
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Token range check -- the `IndexError: index out of range in self` of nn.Embedding
 * (VQCPCB/data_processor/data_processor.py:26-32 tables, looked up at bach_cpc_data_processor.py:55-63), made asynchronous:
 *   clamped[i] = min(max(tokens[i], 0), limits[i % n_voices] - 1);  *flag |= 1 if any id was outside its table.
 * `limits` is a HOST array of n_voices (<= 16) table sizes (V_c + 1 with the mask token); every kernel of this library
 * that indexes by token id consumes the clamped copy, and the caller raises when it next reads `flag`.
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_check_tokens(const int64_t* tokens, int64_t n, int n_voices, const int32_t* limits, int64_t* clamped, int32_t* flag,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused per-voice embedding + input projection + positional concatenation.
 * Replaces BachCPCDataProcessor.embed (VQCPCB/data_processor/bach_cpc_data_processor.py:42-68) followed by
 * input_linear and the two positional concatenations of RelativeTransformerDownscaler.forward
 * (VQCPCB/downscalers/relative_transformer_downscaler.py:104-115).
 *   table  [n_voices][vmax][dlin] : host-side product  E_v . W_in^T + b_in  (so lookup == embed + linear)
 *   chan   [n_voices][pos], event [tokens_per_block / n_voices][pos]
 *   out    [n_rows][d], d = dlin + 2*pos :  out[r] = [ table[v][tokens[r]] | chan[v] | event[e] ],
 *          p = r % tokens_per_block, v = p % n_voices, e = p / n_voices.
 * bwd accumulates d_table / d_chan / d_event (overwritten, deterministic two-stage reduction).
 * event == NULL (d_event == NULL in bwd): rows are [table | chan] only, width dlin + pos -- the teacher's input
 * `cat(linear_to_input_transformer(embed(x)), channel_embeddings)` (teacher_relative.py:63-75).
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_embed_pos_fwd(const int64_t* tokens, int64_t n_rows, int tokens_per_block, int n_voices, const float* table,
                        int vmax, int dlin, const float* chan, const float* event, int pos, float* out, void* stream);
int64_t vqcpc_embed_pos_bwd_workspace(int64_t n_rows, int tokens_per_block, int n_voices, int vmax, int dlin, int pos);
int vqcpc_embed_pos_bwd(const int64_t* tokens, int64_t n_rows, int tokens_per_block, int n_voices, int vmax, int dlin,
                        int pos, const float* g_out, float* d_table, float* d_chan, float* d_event, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* Block-table lookup of the first layer's QKV projection.  The input of the first encoder layer of a block has only
 * vmax * L distinct rows (token id x position), so `in_proj(x)` (multihead_attention_custom.py:171) is computed once on
 * those rows and looked up per token:  out[r][:] = table[tokens[r] * L + r % L][:],  table [vmax * L][C].
 * segsum is its backward: d_table[t * L + p][:] = sum_{r: tokens[r] = t, r % L = p} g[r][:]  (deterministic). */
int vqcpc_block_table_gather(const float* table, const int64_t* tokens, float* out, int64_t M, int L, int vmax, int C,
                             void* stream);
int64_t vqcpc_block_table_segsum_workspace(int64_t M, int L, int vmax, int C);
int vqcpc_block_table_segsum(const float* g, const int64_t* tokens, float* d_table, int64_t M, int L, int vmax, int C,
                             void* workspace, int64_t workspace_bytes, void* stream);
/* the same with g [M][C] in bf16 (configs[4] bf16 path: the first layer's attention backward writes d q | k | v so,
 * vqcpc_relattn16_bwd_b16 with the token indirection): fp32 accumulation of the upcast values, same order -- bit-identical to
 * vqcpc_block_table_segsum on the same values in fp32; half the bytes in. */
int vqcpc_block_table_segsum_b16(const void* g_bf16, const int64_t* tokens, float* d_table, int64_t M, int L, int vmax, int C,
                                 void* workspace, int64_t workspace_bytes, void* stream);

/* Gradient of a plain row gather out[m] = table[idx[m]] (forward = vqcpc_block_table_gather with L = 1) for a table of
 * any size V: the decoder's `source_embeddings(source)` on merged codes and its shifted target-token lookup
 * (VQCPCB/decoders/decoder.py:212-215,439,474-480).  sorted_idx / perm = idx sorted ascending by a STABLE sort and the
 * permutation that sorts it; d_table [V][C] is overwritten (rows no index refers to become 0); sums run in ascending m. */
int vqcpc_embedding_bwd(const float* g, int64_t ldg, const int64_t* sorted_idx, const int64_t* perm, float* d_table, int64_t M,
                        int64_t V, int C, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32).  Replace every F.linear / nn.Linear on the path:
 * multihead_attention_custom.py:171,346 (in_proj / out_proj), transformer_custom.py:285 (linear1/linear2),
 * relative_transformer_downscaler.py:132 (output_linear), mlp_upscaler.py:21-34 and their autograd backward.
 *
 * vqcpc_gemm_nt:  C[M,N] = epilogue( A[M,K] . B[N,K]^T )      (forward: B = weight; dgrad: B = weight^T)
 *   epilogue order:  + bias[N] (nullable) -> ReLU (act==1) -> dropout(p, seed; element index m*N+n, scaled 1/(1-p))
 *                    -> * (gate[m,n] > 0 ? gate_scale : 0) (nullable) -> + add[m,n] (+ add2[m,n]) (nullable) -> store.
 *                    add / add2 may alias C element-wise (each element is read before it is written by the same lane).
 * vqcpc_gemm_tn:  dW[N,K] (+)= A[M,N]^T . B[M,K],  db[N] (+)= column sums of A (db nullable)
 *   split over M into partials in `workspace`, then reduced deterministically; accumulate!=0 adds to dW/db.
 * K % 4 == 0, lda/ldb % 4 == 0 required (16-byte vector loads).
 * ------------------------------------------------------------------------------------------------------------------ */
/* GEMM arithmetic: mode 0 = fp32 operands on v_mfma_f32_32x32x2_f32 (bit-exact fp32 fmaf chains);
 * mode 1 = "bf16x6": each fp32 operand is split exactly into 3 bf16 pieces and a product is evaluated with 6 bf16 MFMAs
 * (v_mfma_f32_32x32x16_bf16) and fp32 accumulation -- fp32-class accuracy at 2.67x the fp32 MFMA rate.
 * mode 8 = plain bf16: operands rounded to bf16 (nearest even), ONE bf16 MFMA per product, fp32 accumulation and epilogue
 * (reduced precision; BASELINE configs[4] names it; 128 x 128 tile kernels).  vqcpc_gemm_get_mode returns 0 / 1 / 2.
 * Process-wide; the initial value comes from the environment variable VQCPC_GEMM_MODE (0, 1, or 8; default 0). */
int vqcpc_gemm_set_mode(int mode);
int vqcpc_gemm_get_mode(void);
/* Gradient arithmetic of the bf16x6 mode -- OPT-IN, default 6 (= the forward's exact split everywhere).  With 3, the
 * 256 x 256-tile GEMMs launched while a gradient scope is open (the trainers open one around loss.backward(), i.e. the
 * input-gradient NT GEMMs and the weight-gradient TN GEMMs of torch.autograd's backward of F.linear) split each operand into
 * TWO bf16 planes by rounding (h = rn(x), m = rn(x - h): |x - h - m| <= 2^-18 |x|) and evaluate hh + hm + mh: three MFMAs
 * per product, ~2^-17 relative per product, fp32 accumulation.  Forward GEMMs (and hence losses and the code assignment)
 * are never affected.  vqcpc_gemm_gradient_scope(1) opens a scope, (0) closes it (nesting counted); process-wide. */
int vqcpc_gemm_set_gradient_products(int products);
int vqcpc_gemm_get_gradient_products(void);
int vqcpc_gemm_gradient_scope(int open);
/* Gradient GEMMs on THREE fp16 MFMAs per product at fp32-class accuracy (round 5; csrc/gemm_grad.hip) -- the dgrad / wgrad of
 * torch.autograd's backward of F.linear (vqcpc_encoder_trainer.py:311-313 `loss.backward()`, transformer_custom.py:279-289,
 * multihead_attention_custom.py:171,346) for the 256 x 256-tile shapes.  Each operand element x of a tensor with the
 * power-of-two scale s is carried as h = rtz_f16(x s) (saturating) and m = rn_f16(x s - h): |x s - h - m| <= 2^-21 |x s|, a
 * product is hh + hm + mh on v_mfma_f32_32x32x16_f16 with fp32 accumulation, the result is multiplied by 2^-(eA + eB).
 * Explicit entry points, no process-wide switch: the caller chooses them for its backward pass (and, with vqcpc_gemm_nt_f16x3 below,
 * for the forward products of a training step; evaluation / inference callers use vqcpc_gemm_nt).
 * `scale_state`: 4 floats per CALL SITE, caller-owned device memory: [0] / [1] = amax |A| / |B| of the PREVIOUS step (read:
 * the scale maps it into [2^12, 2^13)), [2] / [3] = amax of THIS call's operands (atomic max by the kernel; zero them before).
 * vqcpc_grad_scale_roll(state, nsites) moves [2], [3] -> [0], [1] (sites that ran) and zeroes [2], [3] for `nsites`
 * consecutive sites: once per step; vqcpc_grad_amax primes [0] / [1] of a site on its first use (atomic max of |x| into *slot).
 * vqcpc_gemm_nt_grad: C = epilogue(A . B^T) with epilogue none | + add | + add + add2 | gate-bit mask * gate_scale (the forms
 *   of the input-gradient GEMMs); M, N multiples of 256, K of 32 (persistent over the tiles: the caller cuts ragged rounds).
 *   add == C (same pointer and leading dimension, add2 == NULL): C += A . B^T in place, by fp32 atomic adds at the L2 (one add
 *   per element: the value load-add-store gives, without the epilogue's operand loads).  add2 == C (round 6): C += A . B^T + add
 *   the same way -- the form of the query-path input gradient that lands on the kept rows of d x (a strided C).
 * vqcpc_gemm_tn_grad: dW = A^T . B (+ db = column sums of A, fp32 exact), as vqcpc_gemm_tn (accumulate 0 | 1); N, K multiples
 *   of 256, M of 32. */
int vqcpc_gemm_nt_grad_supported(int64_t M, int N, int K);
int vqcpc_gemm_nt_grad(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                       const float* add, int64_t ldadd, const float* add2, int64_t ldadd2, const void* gate_mask,
                       float gate_scale, float* scale_state, void* stream);
/* vqcpc_gemm_nt_grad_splitk: the same product for FEW output tiles and a long K -- the remainder rows of a launch whose 256-tiles
 * do not fill whole rounds of the persistent kernel (139 264 x 256: two rounds + 32 tiles).  `splits` K slices run as one launch
 * (grid = tiles x splits, partial products into `workspace` planes), one pass sums the planes in ascending order and applies the
 * epilogue: + bias, dropout (row0 = global row of the first row of this call, for the element index (row0 + m) * N + c), + add,
 * + add2 (add may be C: in place).  M, N multiples of 256, K / splits a multiple of 32; deterministic. */
int64_t vqcpc_gemm_nt_grad_splitk_workspace(int64_t M, int N, int splits);
int vqcpc_gemm_nt_grad_splitk(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                              int splits, const float* bias, float drop_p, uint64_t seed, int64_t row0, const float* add,
                              int64_t ldadd, const float* add2, int64_t ldadd2, void* workspace, int64_t workspace_bytes,
                              float* scale_state, void* stream);
/* vqcpc_gemm_nt_grad_tail: the same product for the TAIL rows of a ragged launch on 64 x 128 output tiles, one per workgroup (the last
 * 8 192 rows of 139 264 x 256 occupy all 256 CUs; no partial planes, no third round of 256-tiles).  Same arithmetic, product order
 * and epilogue expression as the 256-tile kernel: without dropout a row carries the bits vqcpc_gemm_nt_grad / _f16x3 give it.
 * Epilogue: + bias, dropout (element index (row0 + m) * N + c), + add (may be C: in place), + add2.  M a multiple of 64, N of 128,
 * K of 32.  Replaces the same lines as vqcpc_gemm_nt_grad (transformer_custom.py:279-289, multihead_attention_custom.py:171,346). */
int vqcpc_gemm_nt_grad_tail_supported(int64_t M, int N, int K);
int vqcpc_gemm_nt_grad_tail(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                            const float* bias, float drop_p, uint64_t seed, int64_t row0, const float* add, int64_t ldadd,
                            const float* add2, int64_t ldadd2, float* scale_state, void* stream);
int vqcpc_gemm_tn_grad_supported(int64_t M, int N, int K);
int64_t vqcpc_gemm_tn_grad_workspace(int64_t M, int N, int K);
int vqcpc_gemm_tn_grad(const float* A, int64_t lda, const float* B, int64_t ldb, float* dW, float* db, int64_t M, int N, int K,
                       int accumulate, void* workspace, int64_t workspace_bytes, float* scale_state, void* stream);
/* The same kernel for the FORWARD products of a training step (opt-in: ops.FWD_ARITH / VQCPC_FWD_ARITH=f16x3; the library default and
 * every evaluation / inference call stay on vqcpc_gemm_nt): C = epilogue(A . B^T + bias) with the forward epilogues of vqcpc_gemm_nt
 * that the 256-tile launches use -- act 0: bias | bias + add | bias + dropout + add (F.linear + dropout + residual,
 * transformer_custom.py:279-289); act 1 with mask_out: bias + relu (+ dropout) and the bit mask of the positive outputs
 * (vqcpc_gemm_nt_relu_mask).  Same dropout element index and mask layout as vqcpc_gemm_nt; shapes and scale_state as
 * vqcpc_gemm_nt_grad. */
int vqcpc_gemm_nt_f16x3(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                        const float* bias, int act, float drop_p, uint64_t seed, const float* add, int64_t ldadd, void* mask_out,
                        float* scale_state, void* stream);
/* Round 6 -- PRE-SPLIT ("P4") operands of the f16x3 products.  A P4 image has the shape, leading dimension and BYTES of the fp32 matrix
 * it mirrors: every aligned group of four consecutive contraction elements (16 bytes) holds {h0 h1 | h2 h3 | m0 m1 | m2 m3}, fp16 pairs
 * (element 0 in the low half) of x * 2^e with h = rtz_f16, m = rn_f16(x 2^e - h), e from the tensor's amax as for scale_state.
 * vqcpc_weight_planes_many: every weight of a trainer's flat parameter buffer ONCE per step (descriptor table of
 *   vqcpc_transpose_many: 4 int64 per matrix = offset, rows, cols, first 32 x 32 tile): amax[i] = max |W_i| of this step, `planes` =
 *   P4 of W_i at the same offset (groups along the columns: B operand of the forward x W^T; needs cols % 4 == 0), `planes_t` = P4 of
 *   W_i^T at the same offset of the transposed-weight arena (groups along the rows: B operand of dy W; needs rows % 4 == 0).
 *   Replaces the split of the same 256 x K weight tile by every workgroup for every output tile (2 176 times per 557 056-row launch).
 * vqcpc_gemm_nt_g3_pl: vqcpc_gemm_nt_grad | vqcpc_gemm_nt_f16x3 with B (and optionally A) given as P4 images + the device scalar
 *   their planes were scaled with (pl_amax_b required, pl_amax_a NULL for an fp32 A): same loads, same LDS image, same products in
 *   the same order -- bit-identical to the fp32-operand entry points whenever those run under the same amax
 *   (tests/test_kernels_gpu.py::test_p4_*).  scale_state still receives the amax of an fp32 A.
 * vqcpc_gemm_nt_g3_small: the 64 x 128-tile kernel of vqcpc_gemm_nt_grad_tail as a general entry point -- tail rows of ragged
 *   launches on P4 operands AND whole products too small for 256-tiles (M % 64 == 0, N % 128 == 0, K % 32 == 0: the 3 072 - 12 288-row
 *   products of the student / decoder steps); operands fp32 or P4 (pl_amax_* NULL: fp32); epilogue in the order of vqcpc_gemm_nt:
 *   + bias, relu (act == 1), dropout, * (gate > 0 ? gate_scale : 0), + add (may be C), + add2. */
int vqcpc_weight_planes_many(const float* base, const int64_t* desc, int n, int64_t total_tiles, float* amax, void* planes,
                             void* planes_t, void* workspace, int64_t workspace_bytes, void* stream);     /* workspace: total_tiles floats */
int vqcpc_gemm_nt_g3_pl(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                        const float* bias, int act, float drop_p, uint64_t seed, const float* add, int64_t ldadd, const float* add2,
                        int64_t ldadd2, const void* gate_mask, float gate_scale, void* mask_out, float* scale_state,
                        const float* pl_amax_a, const float* pl_amax_b, void* stream);
int vqcpc_gemm_nt_g3_small(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                           const float* bias, int act, float drop_p, uint64_t seed, int64_t row0, const float* gate, int64_t ldgate,
                           float gate_scale, const float* add, int64_t ldadd, const float* add2, int64_t ldadd2, float* scale_state,
                           const float* pl_amax_a, const float* pl_amax_b, void* stream);
int vqcpc_grad_amax(const float* x, int64_t ld, int64_t rows, int cols, float* amax_slot, void* stream);
int vqcpc_grad_scale_roll(float* state, int nsites, void* stream);
/* the same, and *saturated_count (uint32, device) += the number of (site, operand) pairs whose amax of this step lay beyond the fp16
 * range under the scale the step used -- elements above 65504 / scale were clamped there: the monitor of a scale that lagged by more
 * than its 16-32 x head-room (the trainers warn at the end of an epoch when it is non-zero) */
int vqcpc_grad_scale_roll_counted(float* state, int nsites, void* saturated_count, void* stream);
/* the same for nsites <= 512 with a LOG (round 6; what the trainers call): `monitor` = int32[3 + log_capacity] on the device:
 * [0] += saturated (site, operand) pairs, [1] = rolls of this table so far (its step index, incremented by every call), [2] = steps
 * with at least one saturated pair, [3 + k] = step index of the k-th such step (the first log_capacity of them).  A step whose scale
 * lagged is thereby MARKED: the trainers' epoch() reports the count and the step indices (`f16x3_scale_saturations`), the next step
 * runs under the followed scale.  Head-room: the scale puts the previous step's amax into [2^11, 2^12), fp16 saturates at 65504 =
 * 2^16: a tensor may grow 16 x (up to 32 x) from one step to the next before its largest elements are clamped. */
int vqcpc_grad_scale_roll_logged(float* state, int nsites, void* monitor, int log_capacity, void* stream);
int vqcpc_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                  const float* bias, int act, float drop_p, uint64_t seed, const float* gate, int64_t ldgate,
                  float gate_scale, const float* add, int64_t ldadd, const float* add2, int64_t ldadd2, void* stream);
int64_t vqcpc_gemm_tn_workspace(int64_t M, int N, int K);
/* Deferred reduction of the weight-gradient partial sums: vqcpc_gemm_tn / vqcpc_gemm_tn_bf16 with accumulate == 2 leave their
 * `splits` partial sums in the workspace (dW partials at float offset s * N * K, db partials at splits * N * K + s * N) and
 * vqcpc_reduce_grouped_vec sums many such products in ONE launch (at the end of a backward pass: 17 launches less per CPC step),
 * with the per-element arithmetic of the immediate reduction.  *_deferred_splits returns the split count, or 0 when the product
 * is too small for the float4 reduction (then reduce immediately). */
int vqcpc_gemm_tn_deferred_splits(int64_t M, int N, int K);
int vqcpc_gemm_tn_bf16_deferred_splits(int64_t M, int N, int K);
int vqcpc_reduce_grouped_vec(int n, const void* const* ws, const int64_t* stride, const int* nsplit, void* const* out,
                             const int64_t* count, int accumulate, void* stream);
int vqcpc_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* dW, float* db, int64_t M, int N, int K,
                  int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* Grouped weight gradients: n independent products dW_i (+)= A_i^T B_i, db_i (+)= column sums of A_i in a few launches (32
 * problems per launch pair: products + partial-sum reduction).  For the student / decoder steps, whose ~100 weight gradients
 * per backward pass (3072 rows x 512 ... 2048) cannot fill the chip one at a time; torch.autograd issues them one by one
 * (decoders/decoder.py:431-543, student_encoder_trainer.py:256-296), the trainers of this library defer them to the end of
 * loss.backward().  Arrays are HOST arrays of length n (device pointers inside); db[i] may be NULL; a gradient buffer that
 * occurs twice is accumulated in issue order (full-tile problems first, then the ragged ones, each class in the caller's order).  vqcpc_gemm_tn_groupable: the product is one the grouped kernel serves in the
 * current GEMM mode (bf16x6 / rounded-operand modes, shapes that do not take the 256-tile kernel). */
int vqcpc_gemm_tn_groupable(int64_t M, int N, int K);
int64_t vqcpc_gemm_tn_grouped_workspace(int n, const int64_t* M, const int* N, const int* K);
int vqcpc_gemm_tn_grouped(int n, const void* const* A, const int64_t* lda, const void* const* B, const int64_t* ldb,
                          void* const* dW, void* const* db, const int64_t* M, const int* N, const int* K, int accumulate,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* out[C][R] = in[R][C]^T (weight transposes for the dgrad form of vqcpc_gemm_nt) */
int vqcpc_transpose(const float* in, float* out, int R, int C, void* stream);
/* The transposes of n row-major matrices that live in one buffer, in one launch: matrix i = base_in + desc[4 i] with
 * desc[4 i + 1] rows and desc[4 i + 2] columns, its transpose is written at the same offset of base_out; desc[4 i + 3] = number
 * of 32 x 32 tiles of the matrices before it (ascending), total_tiles = their sum over all n.  `desc` is device memory.
 * (dgrad operands W^T of every nn.Linear of a backward pass: the trainers' flat parameter buffer -> a flat arena.) */
int vqcpc_transpose_many(const float* base_in, float* base_out, const int64_t* desc, int n, int64_t total_tiles, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused self-attention with learned relative bias.
 * Replaces MultiheadAttentionCustom.forward :247-343 and SubsampledRelativeAttention.forward
 * (VQCPCB/transformer/subsampled_relative_attention.py:30-122) through its closed form
 *   bias[h,i,j] = q[h,i].e1[h, L-1-(i-j)] (j <= i) | q[h,i].e2[h, j-i] (j > i).
 *   qkv [n_blocks*L][ldq]: q | k | v at column offsets 0, d, 2d; head h = columns [h*hd, (h+1)*hd); q is UNSCALED
 *   e1, e2 [H*L][hd];  ctx [n_blocks*L][ldo] (heads merged);  probs [n_blocks][H][L][L] = softmax BEFORE dropout
 *   dropout element index = ((block*H + h)*L + i)*L + j.
 * bwd: d_qkv [n_blocks*L][ldg] (all 3d columns written), d_e1/d_e2 overwritten (deterministic partials in workspace).
 * ------------------------------------------------------------------------------------------------------------------ */
/* L = 16 and L = 4 run the one-wavefront-per-block kernels; any other L <= 1024 (teacher_relative.py L = 384,
 * auxiliary_decoder_relative.py L = 24 / 96) runs the strip kernels of relattn_x.hip (Lq = Lk, no mask) with identical semantics.
 * vqcpc_relattn_force_general(1) routes every L through the latter (parity tests between the two). */
int vqcpc_relattn_force_general(int on);
int vqcpc_relattn_fwd(const float* qkv, int64_t ldq, const float* e1, const float* e2, float* ctx, int64_t ldo,
                      float* probs, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed, void* stream);
int64_t vqcpc_relattn_bwd_workspace(int64_t n_blocks, int L, int H, int hd);
int vqcpc_relattn_bwd(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const float* probs, const float* e1,
                      const float* e2, float* d_qkv, int64_t ldg, float* d_e1, float* d_e2, int64_t n_blocks, int L, int H,
                      int hd, float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream);

/* First-layer variant: q | k | v are rows of the block table of vqcpc_block_table_gather (table [vmax * L][ldt], row of
 * token r of a block = tokens[r] * L + r); the kernels read them through the token indirection from the L2-resident
 * table instead of a gathered (n_blocks * L, 3d) copy.  d_qkv is still per token (its segment sum is the table gradient:
 * vqcpc_block_table_segsum).  L in {16, 4} only; workspace = vqcpc_relattn_bwd_workspace. */
/* bf16-output forms of the L = 16 attention for the bf16 training path (configs[4]): the attention context (forward) and the
 * gradient of the in_proj output (backward) only feed GEMMs there, which read bf16 from HBM; the kernels round to nearest even
 * on the way out instead of writing fp32 for a cast pass.  ctx_b16 [n_blocks*16][ldo], d_qkv_b16 [n_blocks*16][ldg] bf16
 * (leading dimensions in elements); tokens non-NULL: qkv is the first layer's block table.  Workspace of vqcpc_relattn_bwd. */
int vqcpc_relattn16_b16_supported(int L, int H, int hd);
int vqcpc_relattn16_fwd_b16(const float* qkv, int64_t ldq, const int64_t* tokens, const float* e1, const float* e2, void* ctx_b16,
                            int64_t ldo, float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* stream);
int vqcpc_relattn16_bwd_b16(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const int64_t* tokens,
                            const float* probs, const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* d_e1,
                            float* d_e2, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* workspace,
                            int64_t workspace_bytes, void* stream);
/* the same for any block length the LDS-tiled kernels serve (L in {16, 4}; L = 16 is forwarded to the matrix-core kernels). */
int vqcpc_relattn_b16_supported(int L, int H, int hd);
int vqcpc_relattn_fwd_b16(const float* qkv, int64_t ldq, const float* e1, const float* e2, void* ctx_b16, int64_t ldo,
                          float* probs, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed, void* stream);
int vqcpc_relattn_bwd_b16(const float* d_ctx, int64_t ldo, const float* qkv, int64_t ldq, const float* probs, const float* e1,
                          const float* e2, void* d_qkv_b16, int64_t ldg, float* d_e1, float* d_e2, int64_t n_blocks, int L,
                          int H, int hd, float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream);
/* all-bf16 forms (no token indirection): qkv_b16 [n_blocks*16][ldq] is the bf16 output of the in_proj GEMM, d_ctx_b16 the bf16
 * output of the out-proj input-gradient GEMM -- the two kernels' dominant streams at half the bytes. */
int vqcpc_relattn16_fwd_b16io(const void* qkv_b16, int64_t ldq, const float* e1, const float* e2, void* ctx_b16, int64_t ldo,
                              float* probs, int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* stream);
int vqcpc_relattn16_bwd_b16io(const void* d_ctx_b16, int64_t ldo, const void* qkv_b16, int64_t ldq, const float* probs,
                              const float* e1, const float* e2, void* d_qkv_b16, int64_t ldg, float* d_e1, float* d_e2,
                              int64_t n_blocks, int H, int hd, float drop_p, uint64_t seed, void* workspace,
                              int64_t workspace_bytes, void* stream);
int vqcpc_relattn_tab_fwd(const float* table, int64_t ldt, const int64_t* tokens, const float* e1, const float* e2, float* ctx,
                          int64_t ldo, float* probs, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed,
                          void* stream);
int vqcpc_relattn_tab_bwd(const float* d_ctx, int64_t ldo, const float* table, int64_t ldt, const int64_t* tokens,
                          const float* probs, const float* e1, const float* e2, float* d_qkv, int64_t ldg, float* d_e1,
                          float* d_e2, int64_t n_blocks, int L, int H, int hd, float drop_p, uint64_t seed, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Rectangular, masked variant for the decoder training step (VQCPCB/decoders/decoder.py:431-543,
 * transformer_custom.py:355-386): Lq = r * Lk queries over Lk keys, q / k / v from separate row-major buffers
 * (self-attention: three column blocks of one in_proj output; cross-attention, multihead_attention_custom.py:173-196:
 * q from the target rows, k | v from the memory rows).  With p = i / r,
 *   bias[h,i,j] = q[h,i].e1[h, Lk-1-(p-j)] (j <= p) | q[h,i].e2[h, j-p] (j > p)
 * (SubsampledRelativeAttention.forward with seq_len_tgt = r * seq_len_src, subsampled_relative_attention.py:30-122) and
 * mask = 0 none | 1 causal (keep j <= p) | 2 anticausal (keep j >= p)  (decoder.py:292-308; masked probabilities are 0).
 *   q [n_seq*Lq][ldq], k [n_seq*Lk][ldk], v [n_seq*Lk][ldv] (head h = columns [h*hd, (h+1)*hd), q UNSCALED),
 *   e1, e2 [H*Lk][hd], ctx [n_seq*Lq][ldo], probs [n_seq][H][Lq][Lk] = softmax BEFORE dropout,
 *   dropout element index = ((seq*H + h)*Lq + i)*Lk + j.   Lk <= 1024, hd in {16, 32, 64, 128}.
 * Key tiles a strip of queries cannot see under the mask are skipped (half of the causal self-attention).
 * bwd: same mask as fwd; d_q / d_k / d_v written in full (same layouts, own leading dimensions), d_e1 / d_e2 overwritten. */
int vqcpc_relattn_x_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                        const float* e1, const float* e2, float* ctx, int64_t ldo, float* probs, int64_t n_seq, int Lq,
                        int Lk, int H, int hd, int mask, float drop_p, uint64_t seed, void* stream);
int64_t vqcpc_relattn_x_bwd_workspace(int64_t n_seq, int Lq, int Lk, int H, int hd);
int vqcpc_relattn_x_bwd(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* k, int64_t ldk,
                        const float* v, int64_t ldv, const float* probs, const float* e1, const float* e2, float* d_q,
                        int64_t ldgq, float* d_k, int64_t ldgk, float* d_v, int64_t ldgv, float* d_e1, float* d_e2,
                        int64_t n_seq, int Lq, int Lk, int H, int hd, int mask, float drop_p, uint64_t seed, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* Query-subsampled variant for the LAST layer of a stack: `output[::F]` (relative_transformer_downscaler.py:125) keeps
 * only positions 0, F, 2F.. and everything after the attention is per-token, so only those queries are evaluated
 * (keys / values still span the block).  q [n_blocks*L/F][ldq] (projected from x[::F], unscaled), kv [n_blocks*L][ldkv]
 * (k | v at columns 0, d), ctx [n_blocks*L/F][ldo], probs [n_blocks][H][L/F][L].  F = 4.
 * Results equal the full layer followed by the row selection; dropped rows have zero gradient in the reference too. */
int vqcpc_relattn_sub_fwd(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const float* e1, const float* e2,
                          float* ctx, int64_t ldo, float* probs, int64_t n_blocks, int L, int F, int H, int hd,
                          float drop_p, uint64_t seed, void* stream);
int64_t vqcpc_relattn_sub_bwd_workspace(int64_t n_blocks, int L, int F, int H, int hd);
int vqcpc_relattn_sub_bwd(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* kv, int64_t ldkv,
                          const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, float* d_kv,
                          int64_t ldgkv, float* d_e1, float* d_e2, int64_t n_blocks, int L, int F, int H, int hd,
                          float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream);
/* bf16-output forms of the query-subsampled kernels (the bf16 training path): ctx_b16 / d_kv_b16 hold bf16 elements; d_q (1/8 of
 * the gradient bytes, operand of an fp32 GEMM with two residual inputs) stays fp32. */
int vqcpc_relattn_sub_b16_supported(int L, int F, int H, int hd);
int vqcpc_relattn_sub_fwd_b16(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const float* e1, const float* e2,
                              void* ctx_b16, int64_t ldo, float* probs, int64_t n_blocks, int L, int F, int H, int hd,
                              float drop_p, uint64_t seed, void* stream);
int vqcpc_relattn_sub_bwd_b16(const float* d_ctx, int64_t ldo, const float* q, int64_t ldq, const float* kv, int64_t ldkv,
                              const float* probs, const float* e1, const float* e2, float* d_q, int64_t ldgq, void* d_kv_b16,
                              int64_t ldgkv, float* d_e1, float* d_e2, int64_t n_blocks, int L, int F, int H, int hd,
                              float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused residual + dropout + LayerNorm:  y = LN(x + dropout(r)) * gamma + beta   (eps inside the sqrt, biased var).
 * Replaces transformer_custom.py:282-283 and :288-289.  x has row stride ldx (the [::4] subsample of
 * relative_transformer_downscaler.py:125 is a stride, not a copy); r, y contiguous [M][d].
 * dropout element index = m*d + c.   mean/rstd [M] are saved for the backward.
 * bwd: d_s [M][d] = gradient w.r.t. (x + dropout(r)) = gradient of the x path; d_r = d_s * mask / (1-p) (may alias d_s
 * when p == 0; pass NULL then); d_gamma / d_beta overwritten.
 * r == NULL: `x` is the residual sum s = x0 + dropout(r) itself (formed by the producing GEMM's epilogue: vqcpc_gemm_nt with
 * bias, drop_p, seed and add = x0 uses the same mask, element index m*N + c with N == d) -- forward: y = LN(s); backward
 * with drop_p > 0 and d_r != NULL: d_r = d_s * mask / (1-p) with the mask regenerated from (seed, index), no r stream.
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_add_layernorm_fwd(const float* x, int64_t ldx, const float* r, const float* gamma, const float* beta, float* y,
                            float* mean, float* rstd, int64_t M, int d, float eps, float drop_p, uint64_t seed,
                            void* stream);
int64_t vqcpc_add_layernorm_bwd_workspace(int64_t M, int d);
int vqcpc_add_layernorm_bwd(const float* dy, const float* x, int64_t ldx, const float* r, const float* gamma,
                            const float* mean, const float* rstd, float* d_s, float* d_r, float* d_gamma, float* d_beta,
                            int64_t M, int d, float drop_p, uint64_t seed, void* workspace, int64_t workspace_bytes,
                            void* stream);
/* the same two kernels with an extra bf16 copy of the output that feeds GEMMs in the bf16 path (configs[4]):
 * y_bf16 [M][d] = bf16(y); d_r_bf16 [M][d] = bf16(d_r) (bf16(d_s) when drop_p == 0, where d_r == d_s).  NULL = no copy.
 * forward: y may be NULL when y_bf16 is given (the bf16 copy is the only consumer: 6 instead of 10 bytes per element). */
int vqcpc_add_layernorm_fwd_b16(const float* x, int64_t ldx, const float* r, const float* gamma, const float* beta, float* y,
                                void* y_bf16, float* mean, float* rstd, int64_t M, int d, float eps, float drop_p,
                                uint64_t seed, void* stream);
int vqcpc_add_layernorm_bwd_b16(const float* dy, const float* x, int64_t ldx, const float* r, const float* gamma,
                                const float* mean, const float* rstd, float* d_s, float* d_r, void* d_r_bf16, float* d_gamma,
                                float* d_beta, int64_t M, int d, float drop_p, uint64_t seed, void* workspace,
                                int64_t workspace_bytes, void* stream);
/* the s-form (r == NULL) with the residual sum itself in bf16: x_bf16 [M][ldx] bf16, written by the producing GEMM's epilogue
 * (vqcpc_gemm_nt_bf16 with bias, drop_p, seed, a residual operand and a bf16 output only).  Same arithmetic on the upcast values
 * (bit-identical to the fp32-input kernels given the same values); 2 instead of 4 bytes per element in.  configs[4] bf16 path
 * (transformer_custom.py:279-289: norm1 / norm2 of a layer); x_bf16 8-byte aligned, ldx % 4 == 0. */
int vqcpc_layernorm_fwd_xb16(const void* x_bf16, int64_t ldx, const float* gamma, const float* beta, float* y, void* y_bf16,
                             float* mean, float* rstd, int64_t M, int d, float eps, void* stream);
int vqcpc_layernorm_bwd_xb16(const float* dy, const void* x_bf16, int64_t ldx, const float* gamma, const float* mean,
                             const float* rstd, float* d_s, void* d_s_bf16, float* d_r, void* d_r_bf16, float* d_gamma,
                             float* d_beta, int64_t M, int d, float drop_p, uint64_t seed, void* workspace,
                             int64_t workspace_bytes, void* stream);
/* d_s_bf16 [M][d] = bf16(d_s), NULL = none; d_s may be NULL when d_s_bf16 is given: the gradient of the residual branch then exists in
 * bf16 only -- its one consumer is the bf16 residual operand of the next input-gradient GEMM (vqcpc_gemm_nt_bf16, add_bf16). */
/* ... and with the incoming gradient in bf16 too: dy_bf16 [M][d] bf16 (8-byte aligned), written by the epilogue of the input-gradient
 * GEMM that produced it (vqcpc_gemm_nt_bf16 with a bf16 residual operand and a bf16 output only).  Same arithmetic on the upcast
 * values: bit-identical to vqcpc_layernorm_bwd_xb16 given the same values in fp32; 2 instead of 4 bytes per element in.  configs[4]
 * bf16 path: the gradient of the transformer stack's main stream between sub-layers (transformer_custom.py:279-289 under
 * loss.backward(), vqcpc_encoder_trainer.py:311-313). */
int vqcpc_layernorm_bwd_b16io(const void* dy_bf16, const void* x_bf16, int64_t ldx, const float* gamma, const float* mean,
                              const float* rstd, float* d_s, void* d_s_bf16, float* d_r, void* d_r_bf16, float* d_gamma,
                              float* d_beta, int64_t M, int d, float drop_p, uint64_t seed, void* workspace,
                              int64_t workspace_bytes, void* stream);
/* d_gamma == d_beta == NULL in either backward form: the [vqcpc_add_layernorm_bwd_partials(M, d, r != NULL)][2 d] column partials (d gamma | d beta)
 * stay in `workspace` for a later vqcpc_reduce_grouped.  vqcpc_reduce_grouped: n independent reductions out_i[c] (+)= sum over
 * s < nsplit_i of ws_i[s * stride_i + c], c < count_i, 32 per launch (host arrays of device pointers; a repeated output is
 * accumulated in argument order).  The trainers sum the LayerNorm weight / bias partials of a whole backward pass this way
 * (one launch instead of one reduction + one accumulation per LayerNorm). */
int vqcpc_add_layernorm_bwd_partials(int64_t M, int d, int has_r);   /* has_r: the call passes a separate r (two-input form) */
int vqcpc_reduce_grouped(int n, const void* const* ws, const int64_t* stride, const int* nsplit, void* const* out,
                         const int64_t* count, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Product vector quantiser: nearest code per sub-vector + straight-through output + commitment/codebook loss.
 * Replaces ProductVectorQuantizer.forward distances/argmin/one-hot matmul/_loss/STE
 * (VQCPCB/quantizer/vector_quantizer.py:105-148, :72-83) without the (rows, K, D) intermediate.
 *   z [R][D], codebooks [ncb][K][dsub] (D = ncb*dsub), idx [R][ncb] int64, zq_sg [R][D] = z + (q - z), loss [R].
 * Distance order is canonical: d = 0; for t ascending d = d + (z_t - e_t)^2 with separately rounded sub/mul/add;
 * k ascending, strict '<' (first index wins ties) -- bit-identical to oracle/vqcpc_oracle.py:vq_distances_canonical.
 * assign != 0: idx is computed (argmin); assign == 0: idx is an INPUT (label-corruption path, vector_quantizer.py:119-132).
 * squared != 0: loss = (1 + beta) * sum (q - z)^2 computed as q_latent + beta * e_latent;
 * squared == 0: loss = (1 + beta) * || (q - z) + 1e-5 ||_2.
 * bwd: d_z = g_zq + g_loss * d(loss)/dz ; d_codebooks [ncb][K][dsub] = segment-sum of g_loss * d(loss)/dq
 * (this is the slot the north_star calls "codebook update": the reference trains codebooks by Adam, not EMA).
 * ------------------------------------------------------------------------------------------------------------------  * zq_sg == loss == NULL (with assign = 1): index-only mode for inference consumers (decoders/decoder.py:327-336,
 * encoder.py:137-159 only read encoding_indices).
 */
int vqcpc_vq_fwd(const float* z, const float* codebooks, int64_t R, int ncb, int K, int dsub, float beta, int squared,
                 int assign, int64_t* idx, float* zq_sg, float* loss, void* stream);
int64_t vqcpc_vq_bwd_workspace(int64_t R, int ncb, int K, int dsub);
int vqcpc_vq_bwd(const float* z, const float* codebooks, const int64_t* idx, const float* g_zq, const float* g_loss,
                 int64_t R, int ncb, int K, int dsub, float beta, int squared, float* d_z, float* d_codebooks,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * MlpUpscaler middle: h' = SELU(dropout(h))  (VQCPCB/upscalers/mlp_upscaler.py:21-34); element index = flat index.
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_dropout_selu_fwd(const float* h, float* out, int64_t n, float drop_p, uint64_t seed, void* stream);
int vqcpc_dropout_selu_bwd(const float* h, const float* g_out, float* g_h, int64_t n, float drop_p, uint64_t seed,
                           void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused bilinear scores + InfoNCE + accuracy.  Replaces FksModule.forward (VQCPCB/vqcpc_helper.py:86-98) on the
 * positive and the N negative sets, the reshuffling of vqcpc_encoder_trainer.py:240-263, nce_loss
 * (vqcpc_helper.py:5-29) and the score matrix (:269).
 *   c [B][cdim], W [zdim][cdim][K], z_pos [B][K][zdim], z_neg [B][N][K][zdim]
 *   f_pos [B][K], f_neg [B][K][N] (saved for bwd), loss_b [B] = -sum_k (pos - logsumexp([neg, pos])),
 *   hits [B][K] = (pos > max_n neg) as 0/1.   loss = mean_b loss_b, accuracy[k] = mean_b hits.
 * bwd (g [B] = dLoss/d loss_b): d_c, d_W (deterministic two-stage), d_z_pos, d_z_neg.
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_nce_fwd(const float* c, const float* W, const float* z_pos, const float* z_neg, int B, int K, int N, int zdim,
                  int cdim, float* f_pos, float* f_neg, float* loss_b, float* hits, void* stream);
int64_t vqcpc_nce_bwd_workspace(int B, int K, int N, int zdim, int cdim);
int vqcpc_nce_bwd(const float* c, const float* W, const float* z_pos, const float* z_neg, const float* f_pos,
                  const float* f_neg, const float* g, int B, int K, int N, int zdim, int cdim, float* d_c, float* d_W,
                  float* d_z_pos, float* d_z_neg, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Optimiser on one flat fp32 buffer (all parameters of the trainer are views into it; so are the gradients, which is
 * also the RCCL all-reduce bucket).  Replaces nn.utils.clip_grad_norm_(., 5) + torch.optim.Adam.step
 * (VQCPCB/vqcpc_encoder_trainer.py:92,313-314).
 *   vqcpc_sumsq:   out[0] = sum g^2 (double accumulation, deterministic; workspace from vqcpc_sumsq_workspace)
 *   vqcpc_adam_step: coef = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) read ON DEVICE (no host sync); g *= coef;
 *                  m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 *   grad_scale multiplies g first (1/world_size for a summed all-reduce).
 * ------------------------------------------------------------------------------------------------------------------ */
int64_t vqcpc_sumsq_workspace(int64_t n);
int vqcpc_sumsq(const float* g, int64_t n, float grad_scale, double* out, void* workspace, int64_t workspace_bytes,
                void* stream);
int vqcpc_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step, float grad_scale, float max_norm, const double* sumsq, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * GRU cell of the CPC context network (CModule.forward, vqcpc_helper.py:54-76: nn.GRU, gate order r | z | n, h0 = 0,
 * dropout on the outputs of every layer but the last).  gi = x W_ih^T + b_ih and gh = h_prev W_hh^T + b_hh come from
 * vqcpc_gemm_nt; these entry points do the gate arithmetic.  gi, gh [B][3H]; h_prev (NULL = zeros), h_out, y_out [B][H];
 *   r = sigmoid(gi_r + gh_r), u = sigmoid(gi_z + gh_z), n = tanh(gi_n + r gh_n), h_out = (1 - u) n + u h_prev,
 *   y_out (nullable) = dropout(h_out), element index = idx_base + b*H + c.
 * bwd: dh = d_y * mask (nullable) + d_h (nullable); writes d_gi, d_gh [B][3H] and d_hprev = dh * u (the recurrent part
 * d_gh . W_hh is a vqcpc_gemm_nt with this tensor as its `add` operand).
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_gru_cell_fwd(const float* gi, const float* gh, const float* h_prev, float* h_out, float* y_out, int64_t B, int H,
                       float drop_p, uint64_t seed, uint64_t idx_base, void* stream);
int vqcpc_gru_cell_bwd(const float* gi, const float* gh, const float* h_prev, const float* d_y, const float* d_h, float* d_gi,
                       float* d_gh, float* d_hprev, int64_t B, int H, float drop_p, uint64_t seed, uint64_t idx_base,
                       void* stream);
/* One launch per time step (recurrent product + gate arithmetic fused; H % 64 == 0 -- vqcpc_gru_step_supported):
 * fwd: gh[B][3H] = h_prev W_hh^T + b_hh (h_prev NULL = zeros: gh = b_hh), then the cell of vqcpc_gru_cell_fwd on gi, gh;
 * bwd (step t -> t-1): dh = dgh_next[B][3H] . whh_t[H][3H]^T + dhp[B][H] (+ d_y * mask), then the cell backward of step t-1
 *      (gi, gh, h_prev of THAT step): d_gi, d_gh [B][3H]; dhp is overwritten with dh * u (the next launch's direct term).
 * Replaces one vqcpc_gemm_nt + one gate launch (+ a split-K reduction) per step of nn.GRU (vqcpc_helper.py:54-76). */
int vqcpc_gru_step_supported(int64_t B, int H);
int vqcpc_gru_step_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h_prev, float* gh, float* h_out,
                       float* y_out, int64_t B, int H, float drop_p, uint64_t seed, uint64_t idx_base, void* stream);
int vqcpc_gru_step_bwd(const float* dgh_next, const float* whh_t, float* dhp, const float* gi, const float* gh,
                       const float* h_prev, const float* d_y, float* d_gi, float* d_gh, int64_t B, int H, float drop_p,
                       uint64_t seed, uint64_t idx_base, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Student (distilled VQ-VAE) step, SURVEY.md section 8 row A23.
 *
 * vqcpc_softmax_ce: one row = one (batch row, masked event, channel) logit vector.
 *   target != NULL        : loss[r] = -log_softmax(logits[r])[target[r]]      utils.categorical_crossentropy (utils.py:24-49)
 *   target_logits != NULL : loss[r] = -sum_v softmax(target_logits[r])[v] log_softmax(logits[r])[v]
 *                                                                  utils.distilled_categorical_crossentropy (utils.py:131-159)
 *   grad [R][V] = d loss[r] / d logits[r] = softmax(logits[r]) - target distribution  (backward = vqcpc_scale_rows).
 * vqcpc_upscale_*: AuxiliaryDecoderRelative.upscale (auxiliary_decoder_relative.py:116-130):
 *   out[(r*f + u)][:] = x[r][:] + emb[u][:]; bwd: dx[r] = sum_u g[r*f+u], d_emb[u] = sum_r g[r*f+u] (f <= 8).
 * ------------------------------------------------------------------------------------------------------------------ */
/* 'same_sequence' negative construction on the device (BachCPCDataloaderGenerator._build_negatives_sameSeq,
 * dataloaders/bach_cpc_dataloader.py:163-181): first (B, blocks_first * tokens_per_block), second likewise, int64 tokens in
 * (tick, voice) order; out (B, blocks_first + blocks_second - 1, blocks_second, tokens_per_block):
 *   out[b][n][k] = first block n (n < blocks_first) | second block j' with j = n - blocks_first, j' = j < k ? j : j + 1.
 * negative_samples = f(x_left, x_right), negative_samples_back = f(x_right, x_left) (:131-132). */
int vqcpc_same_sequence_negatives(const int64_t* first, const int64_t* second, int64_t* out, int64_t B, int blocks_first,
                                  int blocks_second, int tokens_per_block, void* stream);
int vqcpc_softmax_ce(const float* logits, int64_t ld, const int64_t* target, const float* target_logits, int64_t ldt,
                     float* loss, float* grad, int64_t R, int V, void* stream);
int vqcpc_scale_rows(const float* in, const float* g, float* out, int64_t R, int V, void* stream);
int vqcpc_upscale_fwd(const float* x, const float* emb, float* out, int64_t rows, int f, int d, void* stream);
int64_t vqcpc_upscale_bwd_workspace(int64_t rows, int f, int d);
int vqcpc_upscale_bwd(const float* g, float* dx, float* d_emb, int64_t rows, int f, int d, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * bf16 GEMM path for BASELINE configs[4] (reduced precision; never used for the fp32 headline configuration).
 * Operands are bf16 (uint16 bit patterns) IN HBM; fp32 accumulation; one v_mfma_f32_32x32x16_bf16 per product.
 *   vqcpc_cast_bf16      out[r][c] = bf16_rne(in[r * ld_in + c])  (what torch's .bfloat16() does), dense [rows][cols]
 *   vqcpc_gemm_nt_bf16   C / Cb = epilogue(A[M,K] . B[N,K]^T): replaces F.linear on bf16-cast operands
 *                        (transformer_custom.py:282-289 FFN, multihead_attention_custom.py:171-196,338 projections).
 *                        C (fp32) and / or Cb (bf16) receive the result; epilogue as vqcpc_gemm_nt (bias, act = 1 relu,
 *                        dropout, gate: out *= gate > 0 ? gate_scale : 0 with an fp32 `gate` or a bf16 `gate_bf16` operand,
 *                        add: an fp32 `add` or -- round 5, the residual stream of the bf16 path kept in bf16: the LayerNorm's
 *                        bf16 output is then its ONLY output -- a bf16 `add_bf16` operand, bias / bias + dropout epilogues; the
 *                        residual sum may itself leave as Cb only: vqcpc_layernorm_fwd_xb16 / _bwd_xb16 read it so).
 *                        M, N multiples of 256, K of 64 (vqcpc_gemm_nt_bf16_supported); lda / ldb / ldcb / ldgate_bf16 /
 *                        ldadd_bf16 in bf16 elements.
 *   vqcpc_gemm_tn_bf16   dW[N,K] (+)= A[M,N]^T . B[M,K], db[N] (+)= column sums of A: the weight / bias gradient of F.linear
 *                        on bf16 operands (what autograd derives for the calls above); as vqcpc_gemm_tn.  M multiple of 128,
 *                        N and K of 256 (vqcpc_gemm_tn_bf16_supported).
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_cast_bf16(const float* in, int64_t ld_in, void* out, int64_t rows, int cols, void* stream);
int vqcpc_gemm_nt_bf16_supported(int64_t M, int N, int K);
/* vqcpc_gemm_nt_bf16 / vqcpc_gemm_tn_bf16 deliver their operands global -> LDS by DMA in whole 128-byte lines -- NT: K tiles of
 * 64, one barrier per K tile (K % 128 == 0, otherwise the register-staged ping-pong kernel with K tiles of 32: bit-identical
 * results); TN: 64-row slots in their row-major form, fragments by ds_read_b64_tr_b16.  (The A/B switch between the DMA and
 * the register-staged kernels, vqcpc_gemm_bf16_set_variant, is a lab-build entry point.) */
int vqcpc_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, void* Cb, int64_t ldcb,
                       int64_t M, int N, int K, const float* bias, int act, float drop_p, uint64_t seed, const float* gate,
                       int64_t ldgate, const void* gate_bf16, int64_t ldgate_bf16, float gate_scale, const float* add,
                       int64_t ldadd, const void* add_bf16, int64_t ldadd_bf16, void* stream);
int vqcpc_gemm_tn_bf16_supported(int64_t M, int N, int K);
int64_t vqcpc_gemm_tn_bf16_workspace(int64_t M, int N, int K);
int vqcpc_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* dW, float* db, int64_t M, int N, int K,
                       int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* dst[i][0 .. counts[i]) += src[i][..] for up to 8 small tensors in one launch (host arrays of device pointers): the
 * `param.grad += g` of the small per-layer parameters (LayerNorm gamma / beta, relative-position tables) that autograd's
 * AccumulateGrad would run as one kernel each (transformer_custom.py:282-289, subsampled_relative_attention.py:23-26). */
int vqcpc_accumulate8(float* const* dst, const float* const* src, const int* counts, int n_tensors, void* stream);

/* Number of distinct merged product codes (sum_c idx[c] K^c) among the rows of up to two (rows, num_codebooks) int64 index
 * tensors -> out[0] (as float): the `num_codewords` / `num_codewords_negative` metrics of VQCPCEncoderTrainer.epoch
 * (vqcpc_encoder_trainer.py:320-331, len(torch.unique(.))) without a sort and without a host sync.  One bit per possible
 * code in LDS: `_supported` is 0 when codebook_size ^ num_codebooks exceeds 2^20 (the caller counts with a sort then). */
int vqcpc_count_distinct_codes_supported(int num_codebooks, int codebook_size);
int vqcpc_count_distinct_codes(const int64_t* idx_a, int64_t rows_a, const int64_t* idx_b, int64_t rows_b, int num_codebooks,
                               int codebook_size, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * relu / dropout gate of the feed-forward block as a bit mask (transformer_custom.py:285: linear2(dropout(relu(linear1(x))))).
 *   vqcpc_gemm_nt_relu_mask   C = dropout(relu(A . B^T + bias)) as vqcpc_gemm_nt(act = 1, drop_p, seed), and one bit per
 *                             element "C > 0" into `mask` (vqcpc_gemm_gatebits_bytes(M, N) bytes; word
 *                             ((row >> 2) * N/32 + col/32) * 4 + (row & 3), bit col % 32)
 *   vqcpc_gemm_nt_gatebits    C = (A . B^T) * (bit ? gate_scale : 0): the backward of relu + dropout folded into the dgrad
 *                             GEMM as with vqcpc_gemm_nt(gate = activation), reading 1/32 of the bytes
 *   bf16x6 mode, M and N multiples of 256, K of 32 (vqcpc_gemm_gatebits_supported); same numbers as the fp32-gate calls.
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_gemm_gatebits_supported(int64_t M, int N, int K);
int64_t vqcpc_gemm_gatebits_bytes(int64_t M, int N);
int vqcpc_gemm_nt_relu_mask(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                            int K, const float* bias, float drop_p, uint64_t seed, void* mask, void* stream);
int vqcpc_gemm_nt_gatebits(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                           int K, const void* mask, float gate_scale, void* stream);

/* Split-K form of vqcpc_gemm_nt for launches that cannot fill the chip (few 128 x 128 tiles, long K: the d_model-wide
 * projections of the student step, encoder_student_trainer.py:203-262 -> transformer_custom.py:270-295 at 3072 / 768 rows).
 *   vqcpc_gemm_nt_splitk_workspace   bytes of partial-sum workspace, or 0 when (M, N, K) is not a split-K shape in the
 *                                    current GEMM mode (then call vqcpc_gemm_nt).
 *   vqcpc_gemm_nt_splitk             C = A . B^T (+ bias) (+ add); the K range is cut into partial planes in `workspace`,
 *                                    summed in a fixed order (deterministic).  Epilogues other than bias / add: vqcpc_gemm_nt. */
int64_t vqcpc_gemm_nt_splitk_workspace(int64_t M, int N, int K);
/* rows of an (M, N, K) product that vqcpc_gemm_nt hands to whole rounds of its 256-tile kernel when it cuts the launch by rows
 * (M when it does not): the caller may run rows [main, M) through vqcpc_gemm_nt_splitk and rows [0, main) through vqcpc_gemm_nt */
int64_t vqcpc_gemm_nt_main_rows(int64_t M, int N, int K);
int vqcpc_gemm_nt_splitk(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N,
                         int K, const float* bias, const float* add, int64_t ldadd, void* workspace, int64_t workspace_bytes,
                         void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Whole-step HIP-graph replay (vqcpc_bach_amd/graphs.py).  A captured step freezes its kernel ARGUMENTS, so the values
 * that must change every step live on the device:
 *   - dropout: every seed is XOR-ed with a step salt (0 outside graph replay).  vqcpc_rng_salt_advance is the first node
 *     of a captured step: counter[0] += 1, salt = splitmix64(base ^ counter[0]); vqcpc_rng_salt_set writes it directly
 *     (value 0 restores eager behaviour).  This is the ONE piece of library state: a process-wide device value.
 *   - Adam (vqcpc_adam_step_dev = vqcpc_adam_step with lr read from lr_dev[0] and the step count t -- the bias
 *     corrections 1 - beta^t -- from step_dev[0], the counter that vqcpc_rng_salt_advance increments).
 * ------------------------------------------------------------------------------------------------------------------ */
int vqcpc_rng_salt_set(uint64_t value, void* stream);
int vqcpc_rng_salt_advance(uint64_t* counter, uint64_t base, void* stream);
/* the salt of the CURRENT counter value again, without advancing it: first node of a later captured stage of the same step
 * (a step cut into several graphs around eagerly issued, bucketed all-reduces) */
int vqcpc_rng_salt_from_counter(const uint64_t* counter, uint64_t base, void* stream);
int vqcpc_adam_step_dev(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                        float eps, const uint64_t* step_dev, float grad_scale, float max_norm, const double* sumsq,
                        void* stream);

#ifdef VQCPC_LAB
/* ==================================================================================================================
 * LAB BUILDS ONLY (`VQCPC_LAB=1 python -m vqcpc_bach_amd.build` -> libvqcpc_hip_lab.so; never loaded by the training steps).
 * Rejected kernel designs kept for A/B measurements by the tools under tools/; the lab library additionally honours the
 * tools' environment switches (VQCPC_PP_ABL, VQCPC_PP_GRID, VQCPC_TN_PQ, VQCPC_S64_MAX_TILES, VQCPC_SPLITK_MIN_K / _KS,
 * VQCPC_BF16_STAGGER, VQCPC_LN_BWD_BLOCKS, VQCPC_RELATTN16_LDS, VQCPC_GEMM_ABL) and the vqcpc_gemm_set_mode bits +16
 * (gemm_dma.hip) and +32 (gemm_sw.hip).  The product library reads none of them.
 * ================================================================================================================== */
/* ------------------------------------------------------------------------------------------------------------------
 * bf16x6 GEMM on pre-split operands (csrc/gemm_planes.hip; no reference counterpart: the same F.linear products as
 * vqcpc_gemm_nt in mode 1, bit-identical results, with the exact 3-way bf16 split done once by the producer).
 *   "P3" format of an fp32 matrix X[rows][cols], cols % 16 == 0: three bf16 planes p = 0 (high), 1 (mid), 2 (low) with
 *   x == high + mid + low exactly, stored K-tile-major: element (p, r, k) at ((p * cols/16 + k/16) * rows + r) * 16 + k % 16
 *   (bf16 units); vqcpc_planes_bytes(rows, cols) = 6 * rows * cols bytes.
 *   vqcpc_split3_planes  X (row stride ld floats) -> P3;  vqcpc_join3_planes  P3 -> X (exact; tests)
 *   vqcpc_gemm_nt_planes C[M,N] = epi(A . B^T) with A = P3 of [M][K], B = P3 of [N][K]; epilogue arguments as vqcpc_gemm_nt.
 *                        M, N multiples of 256, K of 32, each operand's planes below 4 GB (vqcpc_gemm_nt_planes_supported).
 * ------------------------------------------------------------------------------------------------------------------ */
/* A/B switch of vqcpc_gemm_nt_bf16 / _tn_bf16: 1 (default) = the LDS-DMA kernels, 0 = the register-staged ones. */
int vqcpc_gemm_bf16_set_variant(int variant);
int64_t vqcpc_planes_bytes(int64_t rows, int cols);
int vqcpc_split3_planes(const float* x, int64_t ld, int64_t rows, int cols, void* planes, void* stream);
int vqcpc_join3_planes(const void* planes, int64_t rows, int cols, float* x, int64_t ld, void* stream);
int vqcpc_gemm_nt_planes_supported(int64_t M, int N, int K);
int vqcpc_gemm_nt_planes(const void* a_planes, const void* b_planes, float* C, int64_t ldc, int64_t M, int N, int K,
                         const float* bias, int act, float drop_p, uint64_t seed, const float* gate, int64_t ldgate,
                         float gate_scale, const float* add, int64_t ldadd, void* stream);

#endif /* VQCPC_LAB */

#ifdef __cplusplus
}
#endif
#endif /* VQCPC_H */
