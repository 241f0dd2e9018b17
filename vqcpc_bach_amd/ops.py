"""torch.autograd glue over the C-ABI kernels (vqcpc_bach_amd/hip.py).  PyTorch only provides device memory, the
stream and the autograd tape here; every numerical operation below is a libvqcpc_hip.so kernel.

Layout convention: activations are 2-D `(rows, features)` fp32 tensors whose rows may be strided (`stride(1) == 1`);
row = block * L + token (block-major), i.e. the reference's time-first `(L, N, E)` transposed.
"""
import bisect
import os

import torch

from . import hip


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, 'expected an fp32 device tensor'
    return t


def _rows(t):
    """2-D view with unit inner stride; returns (tensor, leading dimension)."""
    assert t.dim() == 2 and t.stride(1) == 1, f'expected row-major 2-D tensor, got strides {t.stride()}'
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def dropout_mask(n, p, seed, device):
    """Keep-mask (1/0) the kernels derive for element indices [0, n) -- test helper (vqcpc_dropout_mask)."""
    m = torch.empty(n, dtype=torch.float32, device=device)
    hip.call('vqcpc_dropout_mask', m, n, float(p), int(seed))
    return m


# ------------------------------------------------------------------------------------------------------------------
# gradient arithmetic of the backward pass (round 5): three fp16 MFMAs per product at fp32-class accuracy, csrc/gemm_grad.hip
# ------------------------------------------------------------------------------------------------------------------
# 'six'    the forward's exact six-product bf16 split everywhere (what every forward GEMM uses, always)
# 'f16x3'  dgrad / wgrad of the 256-tile shapes inside a trainer's backward on vqcpc_gemm_nt_grad / vqcpc_gemm_tn_grad: each
#          operand as two fp16 planes (11 + 11 bits) under a per-tensor power-of-two scale that follows the tensor's amax of the
#          previous step; rms error vs fp64 3-5e-7 (fp32 MFMA kernel: 3-8e-7), x 1.4 the six-product kernels
# 'bf16x3' the round-3 two-bf16-plane form (hip.set_gradient_products(3)): 18-bit operands, rms 4.4e-6 -- kept for comparison
GRAD_ARITH = os.environ.get('VQCPC_GRAD_ARITH', 'six')
TRAINING_GRAD_ARITH = 'f16x3'  # what train_model() / bench.py select in the bf16x6 mode (GraphedTraining.use_training_defaults)
_grad_arith_explicit = 'VQCPC_GRAD_ARITH' in os.environ
GRAD_ROUND_FILL = float(os.environ.get('VQCPC_GRAD_ROUND_FILL', '0.8'))    # least fill of the last round of 256-tiles (A/B switch)
GRAD_MIN_TILES = int(os.environ.get('VQCPC_GRAD_MIN_TILES', '256'))           # dgrad: fewer 256 x 256 tiles than CUs -> the six-product path (its 128-tile kernels fill the chip)
GRAD_TN_MIN_ROWS = 8192        # wgrad: as the 256-tile six-product kernel (tests set both to 0 to reach the kernels with small shapes)
LAST_GEMM_F16X3 = False       # set by gemm_nt / gemm_nt_gatebits / gemm_tn: did the last call go through the f16x3 kernels (bench.py reads it)
_GRAD_SCALES = None            # the open gradient scope's GradScales (None: outside a trainer's backward, or another arithmetic)
# arithmetic of the FORWARD products of a training step (ops.forward_arithmetic scope, the trainers' compute_losses): 'six' = the
# bf16x6 split everywhere; 'f16x3' = the whole-round 256-tile launches on three fp16 MFMAs per product with per-tensor scales, the
# kernel of the gradient GEMMs (rms error vs fp64 at the forward's shapes: 2.7e-7 against 2.4e-7 for the six-product split and
# 2.9e-7 for the exact fp32-MFMA kernel, profiles/r05_bench_grad_f16.log).  Evaluation / encode-only calls never use it.
FWD_ARITH = os.environ.get('VQCPC_FWD_ARITH', 'six')
TRAINING_FWD_ARITH = 'f16x3'   # what train_model() / bench.py select in the bf16x6 mode, with TRAINING_GRAD_ARITH
_fwd_arith_explicit = 'VQCPC_FWD_ARITH' in os.environ
_FWD_SCALES = None             # the open forward scope's GradScales


def set_forward_arithmetic(name):
    """Selects the arithmetic of the forward products of a TRAINING step (see above); returns the previous name.  Evaluation and
    inference are never affected."""
    global FWD_ARITH, _fwd_arith_explicit
    assert name in ('six', 'f16x3'), name
    prev, FWD_ARITH = FWD_ARITH, name
    _fwd_arith_explicit = True
    return prev


def set_gradient_arithmetic(name):
    """Selects the arithmetic of the gradient GEMMs launched inside a trainer's backward pass (see above); returns the previous
    name.  Forward GEMMs are chosen separately (set_forward_arithmetic)."""
    global GRAD_ARITH, _grad_arith_explicit
    assert name in ('six', 'f16x3', 'bf16x3'), name
    prev, GRAD_ARITH = GRAD_ARITH, name
    _grad_arith_explicit = True
    hip.set_gradient_products(3 if name == 'bf16x3' else 6)
    return prev


def gradient_arithmetic_state():
    """(..., the library's gradient-products setting): a caller who chose hip.set_gradient_products(3) directly (the round-3 API,
    bench.py --grad-products) gets exactly that back from restore_gradient_arithmetic_state()."""
    return (GRAD_ARITH, _grad_arith_explicit, FWD_ARITH, _fwd_arith_explicit, hip.get_gradient_products())


def restore_gradient_arithmetic_state(state):
    global GRAD_ARITH, _grad_arith_explicit, FWD_ARITH, _fwd_arith_explicit
    GRAD_ARITH, _grad_arith_explicit, FWD_ARITH, _fwd_arith_explicit = state[:4]
    hip.set_gradient_products(state[4] if len(state) > 4 else (3 if GRAD_ARITH == 'bf16x3' else 6))


def use_training_default_gradient_arithmetic():
    """What `train_model()` selects when the caller chose nothing (neither set_gradient_arithmetic() nor VQCPC_GRAD_ARITH): the
    f16x3 gradient GEMMs -- fp32-class (rms 3-5e-7 vs fp64, tools/bench_grad_f16.py; every parity suite passes in it at unchanged
    tolerances) -- i.e. the configuration bench.py measures.  The bare library default stays 'six'."""
    global GRAD_ARITH, FWD_ARITH, _grad_arith_explicit
    if not _grad_arith_explicit and hip.get_gradient_products() == 3:
        # hip.set_gradient_products(3) called directly IS a choice (the round-3 bf16x3 arithmetic): no mix of f16x3 and bf16x3
        GRAD_ARITH, _grad_arith_explicit = 'bf16x3', True
    if not _grad_arith_explicit:
        GRAD_ARITH = TRAINING_GRAD_ARITH
    # ... and the forward products of the training step on the same kernel (ops.forward_arithmetic): the same error class again
    # (2.7e-7 vs 2.4e-7 six products vs 2.9e-7 exact fp32 MFMA at the forward's shapes), the oracle parity suites green at unchanged
    # tolerances with it, code assignments as close to the exact fp32-MFMA arithmetic's as the six-product split's are
    # (tools/fwd_f16x3_flips.py); evaluation / inference stay on six products
    if not _fwd_arith_explicit:
        FWD_ARITH = TRAINING_FWD_ARITH


class GradScales:
    """Scale state of the f16x3 gradient GEMMs of ONE backward pass shape (owned by a trainer's flat parameter buffer): 4 floats
    per call site -- amax |A|, |B| of the previous step (read), of this step (written by the kernels) -- in call order.  The
    backward of a trainer issues the same GEMMs in the same order every step; a site whose (kind, M, N, K) changes is primed
    again.  `roll()` once per step: this step's amax becomes the next step's scale."""

    CAPACITY = 512
    LOG = 16

    def __init__(self, device):
        self.state = torch.zeros(4 * self.CAPACITY, dtype=torch.float32, device=device)
        # [0] (site, operand) pairs clamped by a lagging scale, ever; [1] rolls so far (this table's step index); [2] steps with a
        # clamped pair; [3 + k] step index of the k-th of them (vqcpc_grad_scale_roll_logged)
        self.monitor = torch.zeros(3 + self.LOG, dtype=torch.int32, device=device)
        self.keys = []
        self.cursor = 0
        self.reprimed = 0

    REPRIME_WARN = 64

    @property
    def saturated(self):
        return self.monitor[0:1]

    def begin(self):
        self.cursor = 0

    def site(self, key, a, lda, rows_a, cols_a, b, ldb, rows_b, cols_b):
        i = self.cursor
        self.cursor += 1
        assert i < self.CAPACITY, 'GradScales: more gradient GEMM call sites than the table holds'
        st = self.state[4 * i:4 * i + 4]
        if i == len(self.keys) or self.keys[i] != key:
            # first use of the site (or another product took its place): prime the scale with the operands' own amax
            if i == len(self.keys):
                self.keys.append(key)
            else:
                # another product took the site's place: correct (the scale is primed from the operands themselves) but it costs two
                # amax passes per re-primed site and step -- a forward whose GEMM sequence changes from step to step (a data-dependent
                # branch, alternating batch shapes) should use one scope tag per variant.  Said once per table, not silently.
                self.keys[i] = key
                st.zero_()
                self.reprimed += 1
                if self.reprimed == self.REPRIME_WARN:
                    import warnings
                    warnings.warn(f'f16x3 scale table: {self.reprimed} call sites were re-primed because the sequence of GEMM shapes '
                                  'inside the arithmetic scope changed between steps (correct, but two extra passes over the operands '
                                  'per site and step); give each variant of the step its own tag: ops.forward_arithmetic(flat, tag=...)')
            hip.call('vqcpc_grad_amax', a, lda, rows_a, cols_a, st[0:1])
            hip.call('vqcpc_grad_amax', b, ldb, rows_b, cols_b, st[1:2])
        return st

    def roll(self):
        if self.keys:
            hip.call('vqcpc_grad_scale_roll_logged', self.state, len(self.keys), self.monitor, self.LOG)


def scale_saturation_report(owner):
    """{table tag: {'pairs': n, 'steps': n, 'step_indices': [...]}} for the f16x3 scale tables on `owner` that saw a lagging scale: the
    (site, operand) pairs clamped so far, the number of steps it happened in and the indices (rolls of that table since it was
    created = training steps of that scope) of the first GradScales.LOG of them.  Reads the device (one small copy per table)."""
    out = {}
    for tag, t in (getattr(owner, '_grad_scales', None) or {}).items():
        m = t.monitor.cpu().tolist()
        if m[0]:
            out[str(tag)] = {'pairs': m[0], 'steps': m[2], 'step_indices': m[3:3 + min(m[2], t.LOG)], 'rolls': m[1]}
    return out


def scale_saturations(owner):
    """Number of (call site, operand) pairs, over every step so far, whose tensor outgrew the fp16 range under its previous-step
    scale (its largest elements were clamped for that one step) in the f16x3 scale tables that live on `owner` (a trainer's flat
    parameters).  Reads the device: call it where the host waits anyway (the trainers: end of an epoch)."""
    tabs = getattr(owner, '_grad_scales', None) or {}
    return sum(int(t.saturated.item()) for t in tabs.values())


def _grad_scales_of(owner, tag):
    tabs = getattr(owner, '_grad_scales', None)
    if tabs is None:
        tabs = owner._grad_scales = {}
    t = tabs.get(tag)
    if t is None:
        flat = getattr(owner, 'flat', owner)
        t = tabs[tag] = GradScales(flat.device)
    return t


# launches of fewer 256-tiles than CUs (one under-filled round) from this many tiles on; 1 << 30: never (A/B switch).  34 816 x 256 x K
# (136 tiles) per launch, graph-timed (tools/bench_g3_tail.py): K = 1024 124 us on six products, 79 on the f16x3 kernel; K = 256 39 / 30
GRAD_ONE_ROUND_MIN_TILES = int(os.environ.get('VQCPC_GRAD_ONE_ROUND_MIN_TILES', '128'))
_grad_cut = {}                 # (M, N) -> rows of the dgrad launch that go through the three-product kernel (0: none)


GRAD_CUT_MIN_K = 1 << 30       # ragged dgrad launches cut into whole rounds + a six-product remainder from this K on: OFF (measured slower)


def _grad_rows(M, N, K):
    """Rows of an (M, K) x (N, K)^T input-gradient product that go through vqcpc_gemm_nt_grad: all of them when its 256 x 256 tiles
    fill whole rounds of the 256 persistent workgroups to >= 80 %, none otherwise.  Ragged launches (139 264 x 256: 2.125 rounds)
    cut into whole rounds + a six-product remainder (GRAD_CUT_MIN_K) were measured twice at C1 (profiles/r05_perf_log.md) and
    were slower both times -- K = 256 alone: 0.228 -> 0.251 ms per call; every K >= 512 launch with the remainder through the
    split-K path: the step went from 24.5 to 24.8 ms -- so they stay on the six-product path, which cuts such launches itself."""
    hit = _grad_cut.get((M, N, K, GRAD_MIN_TILES, GRAD_ONE_ROUND_MIN_TILES))
    if hit is not None:
        return hit
    rows = 0
    if M >= 256 and hip.query('vqcpc_gemm_nt_grad_supported', M - M % 256, N, K):
        tn = N // 256
        tiles = (M // 256) * tn
        if GRAD_MIN_TILES == 0:
            rows = M - M % 256
        elif tiles >= GRAD_MIN_TILES and M % 256 == 0:
            if (tiles / 256.0) / -(-tiles // 256) >= GRAD_ROUND_FILL:
                rows = M
            elif K >= GRAD_CUT_MIN_K:
                rows = ((tiles // 256) * 256 // tn) * 256
        elif GRAD_ONE_ROUND_MIN_TILES <= tiles < 256 and M % 256 == 0:
            rows = M              # one under-filled round (34 816 x 256: 136 tiles)
    _grad_cut[(M, N, K, GRAD_MIN_TILES, GRAD_ONE_ROUND_MIN_TILES)] = rows
    return rows


# opt-in (VQCPC_GRAD_SPLITK=1): ragged rounds = whole rounds + a split-K remainder, both on the f16x3 kernel.  Measured at C1
# (profiles/r05_perf_log.md): 23.04 / 23.10 -> 22.99 / 22.92 ms/step -- the eight partial planes of the remainder (64 MB written and
# read back per launch) cost most of what the third round cost; off by default
GRAD_SPLITK = os.environ.get('VQCPC_GRAD_SPLITK', '0') == '1'
# ragged rounds = whole rounds of 256-tiles + the TAIL rows on 64 x 128 tiles (vqcpc_gemm_nt_grad_tail: one launch, every CU busy, no
# partial planes): at most GRAD_TAIL_MAX_FILL of a round left over (139 264 x 256: 32 tiles of a 256-tile round); VQCPC_GRAD_TAIL=0:
# such launches stay on the six-product path (A/B switch)
GRAD_TAIL = os.environ.get('VQCPC_GRAD_TAIL', '1') == '1'
GRAD_TAIL_MAX_FILL = float(os.environ.get('VQCPC_GRAD_TAIL_MAX_FILL', '0.25'))
# ... except products with an OUT-OF-PLACE residual operand and a contraction shorter than this (A/B switch): the residual loads of the
# 256-tile kernel's epilogue are its slowest form, and with K = 256 there is little K loop per tile to carry them.  Measured at C1,
# same box, 0 / 512: 22.06 / 22.08 vs 22.16 / 22.17 ms/step -- the three-product kernel wins for those launches too: stays 0
GRAD_TAIL_ADD_MIN_K = int(os.environ.get('VQCPC_GRAD_TAIL_ADD_MIN_K', '0'))
_g3_plans = {}


def _g3_plan_key(M, N, K):
    """Cache key of _g3_plan (a test may place a plan under it: a cut launch at sizes the oracle finishes in seconds)."""
    return (M, N, K, GRAD_MIN_TILES, GRAD_ONE_ROUND_MIN_TILES, GRAD_ROUND_FILL, GRAD_SPLITK, GRAD_TAIL, GRAD_TAIL_MAX_FILL)


def _g3_plan(M, N, K):
    """How an (M, K) x (N, K)^T product of a training step runs on the three-product kernel: None (not at all), (M, 0) (one launch:
    its 256-tiles fill whole rounds of the 256 persistent workgroups to >= GRAD_ROUND_FILL), (rows, splits): the whole rounds as
    one launch + the remaining rows as a split-K launch of the same kernel (vqcpc_gemm_nt_grad_splitk: `splits` K slices, so that
    the few tiles of the remainder still occupy every CU; opt-in), or (rows, -1): the whole rounds + the remaining rows on 64 x 128
    tiles (vqcpc_gemm_nt_grad_tail).  139 264 x 256 x K: 512 tiles + 8 192 tail rows = 256 small tiles."""
    key = _g3_plan_key(M, N, K)
    hit = _g3_plans.get(key)
    if hit is not None:
        return hit[0]
    plan = None
    rows = _grad_rows(M, N, K)
    if rows == M:
        plan = (M, 0)
    elif (GRAD_SPLITK and rows == 0 and GRAD_MIN_TILES > 0 and M % 256 == 0 and N % 256 == 0 and K >= 512
          and hip.query('vqcpc_gemm_nt_grad_supported', M, N, K)):
        tn = N // 256
        tiles = (M // 256) * tn
        main_rows = ((tiles // 256) * 256 // tn) * 256
        rem_tiles = (M - main_rows) // 256 * tn
        if tiles > 256 and main_rows > 0 and 0 < rem_tiles <= 128:
            for sp in (16, 8, 4, 2):
                if rem_tiles * sp <= 256 and K % sp == 0 and (K // sp) % 32 == 0 and K // sp >= 64:
                    plan = (main_rows, sp)
                    break
    if (plan is None and GRAD_TAIL and rows == 0 and GRAD_MIN_TILES > 0 and M % 256 == 0 and N % 256 == 0
            and hip.query('vqcpc_gemm_nt_grad_supported', M, N, K)):
        tn = N // 256
        tiles = (M // 256) * tn
        main_rows = ((tiles // 256) * 256 // tn) * 256
        rem_tiles = (M - main_rows) // 256 * tn
        if (tiles > 256 and main_rows > 0 and 0 < rem_tiles <= GRAD_TAIL_MAX_FILL * 256
                and hip.query('vqcpc_gemm_nt_grad_tail_supported', M - main_rows, N, K)):
            plan = (main_rows, -1)
    _g3_plans[key] = (plan,)
    return plan


# Weights as PRE-SPLIT fp16 planes, once per step (round 6, csrc/gemm_grad.hip "P4"): the B operand of a training step's NT product is
# a weight (forward: W itself, backward: its transpose from the batched-transpose arena); when the trainer's plane images are live
# (_PLANES: between the opening of the forward / gradient scope and its end) the product takes them instead of splitting the same
# 256 x K weight tile in every workgroup for every output tile.  Bit-identical products, +5-11 % per launch (tools/bench_p4.py).
# VQCPC_WEIGHT_PLANES=0: A/B switch.
WEIGHT_PLANES = os.environ.get('VQCPC_WEIGHT_PLANES', '1') != '0'
_PLANES = None                 # the _WeightTransposes whose plane images are valid right now
_PARAM_STEPS = 0               # bumped by every FlatAdam.step(): planes made before an optimiser step are never paired with a later backward


def _g3_launch(a, lda, b, ldb, out, ldc, rows, N, K, st, bias=None, act=0, drop_p=0.0, seed=0, add=None, lda_=0, add2=None, lda2_=0,
               gate_mask=None, gate_scale=1.0, mask_out=None):
    """One launch of the 256-tile three-product kernel: gradient forms (bias None) or forward forms, B from the weight planes when
    they are live."""
    pl = _PLANES.lookup_planes(b, ldb) if _PLANES is not None else None
    if pl is not None:
        hip.call('vqcpc_gemm_nt_g3_pl', a, lda, pl[0], ldb, out, ldc, rows, N, K, bias, int(act), float(drop_p), int(seed), add, lda_, add2,
                 lda2_, gate_mask, float(gate_scale), mask_out, st, None, pl[1])
    elif bias is None:
        hip.call('vqcpc_gemm_nt_grad', a, lda, b, ldb, out, ldc, rows, N, K, add, lda_, add2, lda2_, gate_mask, float(gate_scale), st)
    else:
        hip.call('vqcpc_gemm_nt_f16x3', a, lda, b, ldb, out, ldc, rows, N, K, bias, int(act), float(drop_p), int(seed), add, lda_, mask_out, st)


def _g3_nt(scales, key, a, lda, b, ldb, out, ldc, M, N, K, plan, bias=None, drop_p=0.0, seed=0, add=None, lda_=0, add2=None, lda2_=0):
    """One (M, K) x (N, K)^T product on the three-product kernel according to `plan` (_g3_plan); epilogue none | + add | + add +
    add2 (gradient forms) or + bias | + bias + add | + bias + dropout + add (forward forms)."""
    st = scales.site(key, a, lda, M, K, b, ldb, N, K)
    m_main, splits = plan

    def launch(rows0, rows):
        a_, o_ = a[rows0:], out[rows0:]
        ad_ = None if add is None else add[rows0:]
        ad2_ = None if add2 is None else add2[rows0:]
        _g3_launch(a_, lda, b, ldb, o_, ldc, rows, N, K, st, bias=bias, drop_p=drop_p, seed=seed, add=ad_, lda_=lda_, add2=ad2_, lda2_=lda2_)

    if not splits:
        launch(0, M)
        return out
    launch(0, m_main)             # the dropout element index of the main rows starts at row 0, as in the unsplit launch
    rem = M - m_main
    if splits < 0:                # the tail rows on 64 x 128 tiles
        pl = _PLANES.lookup_planes(b, ldb) if _PLANES is not None else None
        if pl is not None:
            hip.call('vqcpc_gemm_nt_g3_small', a[m_main:], lda, pl[0], ldb, out[m_main:], ldc, rem, N, K, bias, 0, float(drop_p), int(seed),
                     m_main, None, 0, 1.0, None if add is None else add[m_main:], lda_, None if add2 is None else add2[m_main:], lda2_, st,
                     None, pl[1])
            return out
        hip.call('vqcpc_gemm_nt_grad_tail', a[m_main:], lda, b, ldb, out[m_main:], ldc, rem, N, K, bias, float(drop_p), int(seed), m_main,
                 None if add is None else add[m_main:], lda_, None if add2 is None else add2[m_main:], lda2_, st)
        return out
    nbytes = hip.query('vqcpc_gemm_nt_grad_splitk_workspace', rem, N, splits)
    ws = hip.workspace(nbytes, a.device)
    hip.call('vqcpc_gemm_nt_grad_splitk', a[m_main:], lda, b, ldb, out[m_main:], ldc, rem, N, K, splits, bias, float(drop_p), int(seed),
             m_main, None if add is None else add[m_main:], lda_, None if add2 is None else add2[m_main:], lda2_, ws, nbytes, st)
    return out


def _tn_grad_ok(M, N, K):
    if not hip.query('vqcpc_gemm_tn_grad_supported', M, N, K):
        return False
    if GRAD_TN_MIN_ROWS == 0:
        return True
    tiles = (N // 256) * (K // 256)
    return M >= GRAD_TN_MIN_ROWS and tiles * min(max(1, 256 // tiles), max(1, M // 256)) >= 128


# ------------------------------------------------------------------------------------------------------------------
# raw kernel wrappers (no autograd)
# ------------------------------------------------------------------------------------------------------------------
def gemm_nt(a, b, bias=None, act=0, drop_p=0.0, seed=0, gate=None, gate_scale=1.0, add=None, add2=None, out=None):
    """out[M,N] = epilogue(a[M,K] @ b[N,K]^T); see include/vqcpc.h."""
    global LAST_GEMM_F16X3
    LAST_GEMM_F16X3 = False
    a, lda = _rows(_f32(a))
    b, ldb = _rows(_f32(b))
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    out, ldc = _rows(out)
    ldg = lda_ = lda2_ = 0
    if gate is not None:
        gate, ldg = _rows(gate)
    if add is not None:
        add, lda_ = _rows(add)
    if add2 is not None:
        add2, lda2_ = _rows(add2)
    if (_FWD_SCALES is not None and _GRAD_SCALES is None and not act and gate is None and add2 is None and hip.get_gemm_mode() == 1
            and (not drop_p or (bias is not None and add is not None)) and (bias is not None or add is None)
            and lda % 4 == 0 and ldb % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        # forward product of a training step on three fp16 MFMAs (FWD_ARITH = 'f16x3'): bias / bias + residual / bias + dropout +
        # residual epilogues, or none
        plan = _g3_plan(M, N, K)
        if plan is not None and plan[1] < 0 and add is not None and K < GRAD_TAIL_ADD_MIN_K:
            plan = None
        if plan is not None and _splitk_operands_ok(plan, out, ldc, add, lda_, None, 0, bias):
            LAST_GEMM_F16X3 = True
            return _g3_nt(_FWD_SCALES, ('fnt', M, N, K), a, lda, b, ldb, out, ldc, M, N, K, plan, bias, drop_p, seed, add, lda_)
    if (_GRAD_SCALES is not None and bias is None and not act and not drop_p and gate is None and (add2 is None or add is not None)
            and hip.get_gemm_mode() == 1):
        # inside a trainer's backward pass: the input-gradient product on three fp16 MFMAs (whole rounds of 256-tiles)
        m_g = _grad_rows(M, N, K)
        plan = _g3_plan(M, N, K) if not m_g else None
        if (plan is not None and plan[1] < 0 and add is not None and add.data_ptr() != out.data_ptr() and K < GRAD_TAIL_ADD_MIN_K):
            plan = None
        if (plan is not None and plan[1] and lda % 4 == 0 and ldb % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
                and _splitk_operands_ok(plan, out, ldc, add, lda_, add2, lda2_, None)):
            LAST_GEMM_F16X3 = True           # ragged rounds: whole rounds + split-K remainder, both on the three-product kernel
            return _g3_nt(_GRAD_SCALES, ('nt', M, N, K), a, lda, b, ldb, out, ldc, M, N, K, plan, None, 0.0, 0, add, lda_, add2, lda2_)
        if m_g and lda % 4 == 0 and ldb % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0:
            st = _GRAD_SCALES.site(('nt', M, N, K), a, lda, M, K, b, ldb, N, K)
            LAST_GEMM_F16X3 = True
            _g3_launch(a, lda, b, ldb, out, ldc, m_g, N, K, st, add=add, lda_=lda_, add2=add2, lda2_=lda2_)
            if m_g < M:            # the ragged last round: six products, through this function's own dispatch (split-K for long K)
                gemm_nt(a[m_g:], b, add=None if add is None else add[m_g:], add2=None if add2 is None else add2[m_g:], out=out[m_g:])
                LAST_GEMM_F16X3 = True
            return out
    if ((_FWD_SCALES is not None or _GRAD_SCALES is not None) and SMALL_F16X3 and hip.get_gemm_mode() == 1 and _small_f16x3_ok(M, N, K)
            and act in (0, 1) and not (act and gate is not None) and (add2 is None or add is not None)
            and lda % 4 == 0 and ldb % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        # a product of a training step too small for 256-tiles (the 3 072 - 12 288-row products of the student / decoder steps): the
        # three-product kernel on 64 x 128 tiles instead of the six-product 128-tile / split-K path, B from the weight planes
        scales = _GRAD_SCALES if _GRAD_SCALES is not None else _FWD_SCALES
        st = scales.site(('snt', M, N, K), a, lda, M, K, b, ldb, N, K)
        pl = _PLANES.lookup_planes(b, ldb) if _PLANES is not None else None
        LAST_GEMM_F16X3 = True
        hip.call('vqcpc_gemm_nt_g3_small', a, lda, b if pl is None else pl[0], ldb, out, ldc, M, N, K, bias, int(act), float(drop_p),
                 int(seed), 0, gate, ldg, float(gate_scale), add, lda_, add2, lda2_, st, None, None if pl is None else pl[1])
        return out
    if (SPLIT_K and M <= _SPLITK_MAX_ROWS and K >= 512 and not act and not drop_p and gate is None and add2 is None
            and hip.get_gemm_mode() == 1):
        # few output tiles, long K (student / decoder steps): K cut over partial planes, see include/vqcpc.h
        ws_bytes = _splitk_ws.get((M, N, K))
        if ws_bytes is None:
            ws_bytes = _splitk_ws[(M, N, K)] = hip.query('vqcpc_gemm_nt_splitk_workspace', M, N, K)
        if (ws_bytes and ldc % 4 == 0 and lda_ % 4 == 0 and out.data_ptr() % 16 == 0
                and (add is None or add.data_ptr() % 16 == 0) and (bias is None or bias.data_ptr() % 16 == 0)):
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=a.device)
            hip.call('vqcpc_gemm_nt_splitk', a, lda, b, ldb, out, ldc, M, N, K, bias, add, lda_, ws, ws_bytes)
            return out
    if (SPLIT_K and M > _SPLITK_MAX_ROWS and K >= 768 and not act and not drop_p and gate is None and add2 is None
            and hip.get_gemm_mode() == 1):
        # a launch that vqcpc_gemm_nt would cut by rows (whole rounds of 256-tiles + an under-filled 128-tile remainder:
        # 139 264 x 256 x 1024 = 2.125 rounds): the remainder rows go through the split-K path
        cut = _rowcut.get((M, N, K))
        if cut is None:
            m_main = hip.query('vqcpc_gemm_nt_main_rows', M, N, K)
            ws_bytes = hip.query('vqcpc_gemm_nt_splitk_workspace', M - m_main, N, K) if 0 < m_main < M else 0
            cut = _rowcut[(M, N, K)] = (m_main, ws_bytes)
        m_main, ws_bytes = cut
        if (ws_bytes and ldc % 4 == 0 and lda_ % 4 == 0 and out.data_ptr() % 16 == 0
                and (add is None or add.data_ptr() % 16 == 0) and (bias is None or bias.data_ptr() % 16 == 0)):
            hip.call('vqcpc_gemm_nt', a, lda, b, ldb, out, ldc, m_main, N, K, bias, 0, 0.0, 0, None, 0, 1.0, add, lda_, None, 0)
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=a.device)
            hip.call('vqcpc_gemm_nt_splitk', a[m_main:], lda, b, ldb, out[m_main:], ldc, M - m_main, N, K, bias,
                     None if add is None else add[m_main:], lda_, ws, ws_bytes)
            return out
    hip.call('vqcpc_gemm_nt', a, lda, b, ldb, out, ldc, M, N, K, bias, int(act), float(drop_p), int(seed), gate, ldg,
             float(gate_scale), add, lda_, add2, lda2_)
    return out


# sub-256-tile products of a training step on the 64 x 128-tile three-product kernel (vqcpc_gemm_nt_g3_small; VQCPC_SMALL_F16X3=0: A/B
# switch): from 128 tiles on (768 x 512 x 512 = 48 tiles: 0.95 x the six-product path, 768 x 2048 x 512 = 192 tiles: 1.19 x, 3 072-row
# shapes 1.12-1.36 x, tools/bench_small_f16x3.py), up to the row count where the split-K / 128-tile paths end
SMALL_F16X3 = os.environ.get('VQCPC_SMALL_F16X3', '1') != '0'
SMALL_F16X3_MIN_TILES = 128
SMALL_F16X3_MIN_K = 256


def _small_f16x3_ok(M, N, K):
    return (M % 64 == 0 and N % 128 == 0 and K % 32 == 0 and K >= SMALL_F16X3_MIN_K and 512 <= M <= _SPLITK_MAX_ROWS
            and (M // 64) * (N // 128) >= SMALL_F16X3_MIN_TILES)


def _splitk_operands_ok(plan, out, ldc, add, lda_, add2, lda2_, bias):
    """Alignment the float4 epilogue of the split-K remainder needs (nothing for a single launch)."""
    if plan[1] <= 0:
        return True
    ok = ldc % 4 == 0 and out.data_ptr() % 16 == 0
    for t, ld in ((add, lda_), (add2, lda2_)):
        ok = ok and (t is None or (ld % 4 == 0 and t.data_ptr() % 16 == 0))
    return ok and (bias is None or bias.data_ptr() % 16 == 0)


def gemm_nt_residual(a, b, res, res_may_alias=None):
    """res + a @ b^T, the input-gradient product that joins a residual path.  Inside a trainer's backward under the f16x3 arithmetic
    the product is ACCUMULATED INTO `res` (vqcpc_gemm_nt_grad with add == C: fp32 atomic adds at the L2, bit-identical to the
    out-of-place form and 13-15 % faster: no epilogue operand loads to wait for) and `res` itself is returned -- the caller must not
    need the old `res` afterwards; `res_may_alias`: a tensor that a DEFERRED launch may still read (a weight gradient collected
    for the grouped launch at the end of backward): if it shares `res`'s storage the out-of-place form is kept."""
    M, K = a.shape
    N = b.shape[0]
    if (_GRAD_SCALES is not None and res.dtype == torch.float32 and res.is_contiguous() and res.shape == (M, N)
            and hip.get_gemm_mode() == 1 and (_grad_rows(M, N, K) > 0 or _g3_plan(M, N, K) is not None)
            and (res_may_alias is None or res_may_alias.data_ptr() != res.data_ptr())):
        return gemm_nt(a, b, add=res, out=res)
    return gemm_nt(a, b, add=res)


_rowcut = {}                   # (M, N, K) -> (rows of the whole 256-tile rounds, split-K workspace bytes of the rest)
SPLIT_K = True                 # A/B switch (tools/bench_splitk.py)
_SPLITK_MAX_ROWS = 1 << 14     # 160 tiles of 128 x 128 at most: no shape above this many rows qualifies
_splitk_ws = {}                # (M, N, K) -> workspace bytes in bf16x6 mode (0: not a split-K shape)


def gatebits_supported(M, N, K):
    """True when the relu / dropout gate of an (M, K) -> (M, N) feed-forward projection can travel as a bit mask
    (bf16x6 mode, full 256-tiles): see include/vqcpc.h."""
    return hip.get_gemm_mode() == 1 and bool(hip.query('vqcpc_gemm_gatebits_supported', M, N, K))


GATEBITS_MIN_TILES = 160


def gatebits_worthwhile(M, N, K):
    """The bit-mask forms exist in the 256-tile ping-pong kernel only (one workgroup per CU): below ~160 tiles the
    fp32-gate forms on 128-tiles fill more of the chip and win (3072 x 2048 x 512: 47 vs 63 us, 768 rows: 33 vs 60 us; at
    192 tiles the bit forms are ahead -- tools/bench_splitk.py)."""
    return (M // 256) * (N // 256) >= GATEBITS_MIN_TILES and gatebits_supported(M, N, K)


# the relu / bit-mask forms have no tail-row launch: their ragged launches take whole rounds of 256-tiles from this fill of the last
# round on (34 816 x 1024 x 256: 544 tiles = 2.125 rounds = 0.71 of three; same-box A/B 0.8 / 0.7: 22.17 / 22.13 -> 22.05 / 22.01 ms)
GRAD_ROUND_FILL_MASKED = float(os.environ.get('VQCPC_GRAD_ROUND_FILL_MASKED', '0.7'))
_masked_ok_cache = {}


def _masked_rows_ok(M, N, K):
    """True when an (M, K) x (N, K)^T product with a relu-mask / gate-bit epilogue runs on the three-product kernel."""
    hit = _masked_ok_cache.get((M, N, K))
    if hit is None:
        hit = _grad_rows(M, N, K) == M
        if not hit and M % 256 == 0 and N % 256 == 0 and hip.query('vqcpc_gemm_nt_grad_supported', M, N, K):
            tiles = (M // 256) * (N // 256)
            hit = tiles > 256 and (tiles / 256.0) / -(-tiles // 256) >= GRAD_ROUND_FILL_MASKED
        _masked_ok_cache[(M, N, K)] = hit
    return hit


def gemm_nt_relu_mask(a, b, bias, drop_p=0.0, seed=0):
    """(dropout(relu(a @ b^T + bias)), bit mask of its positive elements): ops.gemm_nt(act=1, ...) plus the mask that
    gemm_nt_gatebits reads in the backward instead of the activation."""
    a, lda = _rows(_f32(a))
    b, ldb = _rows(_f32(b))
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    mask = torch.empty(M * (N // 32), dtype=torch.int32, device=a.device)
    global LAST_GEMM_F16X3
    LAST_GEMM_F16X3 = False
    if (_FWD_SCALES is not None and _GRAD_SCALES is None and hip.get_gemm_mode() == 1 and _masked_rows_ok(M, N, K)
            and lda % 4 == 0 and ldb % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
        st = _FWD_SCALES.site(('fntm', M, N, K), a, lda, M, K, b, ldb, N, K)
        LAST_GEMM_F16X3 = True
        _g3_launch(a, lda, b, ldb, out, N, M, N, K, st, bias=bias, act=1, drop_p=drop_p, seed=seed, mask_out=mask)
        return out, mask
    hip.call('vqcpc_gemm_nt_relu_mask', a, lda, b, ldb, out, N, M, N, K, bias, float(drop_p), int(seed), mask)
    return out, mask


def gemm_nt_gatebits(a, b, mask, gate_scale=1.0):
    """(a @ b^T) * (bit ? gate_scale : 0) == ops.gemm_nt(a, b, gate=activation, gate_scale=...)."""
    a, lda = _rows(_f32(a))
    b, ldb = _rows(_f32(b))
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    global LAST_GEMM_F16X3
    LAST_GEMM_F16X3 = False
    if _GRAD_SCALES is not None and _masked_rows_ok(M, N, K):
        st = _GRAD_SCALES.site(('ntg', M, N, K), a, lda, M, K, b, ldb, N, K)
        LAST_GEMM_F16X3 = True
        _g3_launch(a, lda, b, ldb, out, N, M, N, K, st, gate_mask=mask, gate_scale=gate_scale)
        return out
    hip.call('vqcpc_gemm_nt_gatebits', a, lda, b, ldb, out, N, M, N, K, mask, float(gate_scale))
    return out


def gemm_tn(a, b, want_bias=True, into=None):
    """dW[N,K] = a[M,N]^T @ b[M,K], db[N] = column sums of a.  `into` = (dW, db) accumulates into existing buffers."""
    a, lda = _rows(_f32(a))
    b, ldb = _rows(_f32(b))
    M, N = a.shape
    K = b.shape[1]
    assert b.shape[0] == M
    if into is not None:
        dw, db = into
        assert dw.shape == (N, K) and dw.is_contiguous() and (db is None or (db.shape == (N,) and db.is_contiguous()))
    else:
        dw = torch.empty(N, K, dtype=torch.float32, device=a.device)
        db = torch.empty(N, dtype=torch.float32, device=a.device) if want_bias else None
    global LAST_TN_DEFERRED, LAST_GEMM_F16X3
    LAST_TN_DEFERRED = False
    LAST_GEMM_F16X3 = False
    if _GRAD_SCALES is not None and hip.get_gemm_mode() == 1 and _tn_grad_ok(M, N, K):
        LAST_GEMM_F16X3 = True
        # inside a trainer's backward pass: the weight gradient on three fp16 MFMAs per product
        st = _GRAD_SCALES.site(('tn', M, N, K), a, lda, M, N, b, ldb, M, K)
        nbytes = hip.query('vqcpc_gemm_tn_grad_workspace', M, N, K)
        ws = hip.workspace(nbytes, a.device)
        hip.call('vqcpc_gemm_tn_grad', a, lda, b, ldb, dw, db, M, N, K, 0 if into is None else 1, ws, nbytes, st)
        return dw, db
    if into is not None and _DIRECT_WGRAD and GROUP_WGRADS and hip.query('vqcpc_gemm_tn_groupable', M, N, K):
        # inside a trainer's backward pass: small weight gradients are collected and issued together when the pass ends
        # (flush_wgrads, called by direct_weight_gradients.__exit__): a few grouped launches instead of two per weight
        _PENDING_WGRADS.append((a, lda, b, ldb, dw, db, M, N, K))
        LAST_TN_DEFERRED = True
        return dw, db
    nbytes = hip.query('vqcpc_gemm_tn_workspace', M, N, K)
    ws = hip.workspace(nbytes, a.device)
    if into is not None and _DIRECT_WGRAD and DEFER_TN_REDUCTIONS:
        # inside a trainer's backward pass the partial sums of the LARGE weight gradients stay in their workspaces and are
        # reduced by ONE grouped launch when the pass ends (flush_reductions): one launch instead of one per weight
        splits = hip.query('vqcpc_gemm_tn_deferred_splits', M, N, K)
        if splits:
            hip.call('vqcpc_gemm_tn', a, lda, b, ldb, dw, db, M, N, K, 2, ws, nbytes)
            _defer_tn_reduction(ws, splits, N, K, dw, db)
            return dw, db
    hip.call('vqcpc_gemm_tn', a, lda, b, ldb, dw, db, M, N, K, 0 if into is None else 1, ws, nbytes)
    return dw, db


GROUP_WGRADS = os.environ.get('VQCPC_GROUP_WGRADS', '1') != '0'      # A/B switch: grouped launches of the small weight gradients
# opt-in (VQCPC_DEFER_TN_REDUCE=1): ONE grouped reduction of the large weight gradients' partial sums at the end of backward instead
# of one per product.  Bit-identical, 16 launches less per CPC step -- and no faster (27.34 / 27.43 vs 27.36 / 27.39 ms per step at
# C1, profiles/r04_perf_log.md): the deferred pass reads 270 MB of partials back from HBM that the immediate ones find in the L2 /
# MALL, which costs what the launches saved.  Off by default.
DEFER_TN_REDUCTIONS = os.environ.get('VQCPC_DEFER_TN_REDUCE', '0') == '1'
_PENDING_VEC_REDUCTIONS = []   # (workspace tensor, byte offset, stride, nsplit, out tensor, count): float4 form (large segments)


def _defer_tn_reduction(ws, splits, N, K, dw, db):
    _PENDING_VEC_REDUCTIONS.append((ws, 0, N * K, splits, dw, N * K))
    if db is not None:
        _PENDING_VEC_REDUCTIONS.append((ws, 4 * splits * N * K, N, splits, db, N))
LAST_TN_DEFERRED = False
_PENDING_WGRADS = []           # (a, lda, b, ldb, dw, db, M, N, K): operands stay alive until the flush


_PENDING_REDUCTIONS = []       # (workspace tensor, byte offset, stride, nsplit, out tensor, count): partial sums -> gradient buffers


def defer_ln_param_grads(ws, M, d, gamma, beta, has_r):
    """LayerNorm backward inside a trainer's gradient scope: the [partials][2 d] column sums of `ws` are added into the live
    gradient buffers of gamma / beta by ONE grouped launch when the scope closes.  Returns False when the parameters own no
    gradient buffers (plain autograd use): the caller reduces as before."""
    if not (_DIRECT_WGRAD and GROUP_WGRADS):
        return False
    gg, bg = _live_grad(gamma), _live_grad(beta)
    if gg is None or bg is None or gg.numel() != d or bg.numel() != d:
        return False
    n = hip.query('vqcpc_add_layernorm_bwd_partials', M, d, 1 if has_r else 0)
    _PENDING_REDUCTIONS.append((ws, 0, 2 * d, n, gg, d))
    _PENDING_REDUCTIONS.append((ws, 4 * d, 2 * d, n, bg, d))
    return True


def flush_reductions():
    import ctypes
    for pending, entry in ((_PENDING_VEC_REDUCTIONS, 'vqcpc_reduce_grouped_vec'), (_PENDING_REDUCTIONS, 'vqcpc_reduce_grouped')):
        items = list(pending)
        pending.clear()
        if not items:
            continue
        n = len(items)
        vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
        hip.call(entry, n, vp(*[it[0].data_ptr() + it[1] for it in items]), i64(*[it[2] for it in items]),
                 i32(*[it[3] for it in items]), vp(*[it[4].data_ptr() for it in items]), i64(*[it[5] for it in items]), 1)


def pending_wgrad_flops():
    return sum(2.0 * it[6] * it[7] * it[8] for it in _PENDING_WGRADS)


def flush_wgrads():
    """Issues the weight gradients deferred by gemm_tn(into=...) since the last flush: vqcpc_gemm_tn_grouped accumulates
    each into its gradient buffer (fixed split order per shape, problems in the order they were deferred).  Called when the
    trainers' gradient scope closes, i.e. after backward() has joined the streams it ran on.
    Measured and NOT kept (profiles/r03_perf_log.md): issuing the groups on a side stream every 8 / 16 deferred products while the
    input-gradient chain continues -- the 1000-workgroup grouped launches take the CUs the chain's small kernels are waiting
    for: C3 12.4 -> 14.2-14.9 ms/step, DEC 10.1 -> 10.4."""
    items = list(_PENDING_WGRADS)
    _PENDING_WGRADS.clear()
    if not items:
        return
    import ctypes
    n = len(items)
    vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    A = vp(*[it[0].data_ptr() for it in items])
    lda = i64(*[it[1] for it in items])
    B = vp(*[it[2].data_ptr() for it in items])
    ldb = i64(*[it[3] for it in items])
    dW = vp(*[it[4].data_ptr() for it in items])
    dB = vp(*[(it[5].data_ptr() if it[5] is not None else None) for it in items])
    M, N, K = i64(*[it[6] for it in items]), i32(*[it[7] for it in items]), i32(*[it[8] for it in items])
    nbytes = hip.query('vqcpc_gemm_tn_grouped_workspace', n, M, N, K)
    ws = hip.workspace(nbytes, items[0][0].device)
    hip.call('vqcpc_gemm_tn_grouped', n, A, lda, B, ldb, dW, dB, M, N, K, 1, ws, nbytes)


# ------------------------------------------------------------------------------------------------------------------
# bf16 path (BASELINE configs[4]: hip.set_gemm_mode(8)): operands are bf16 IN HBM (vqcpc_gemm_nt_bf16), fp32 accumulate
# ------------------------------------------------------------------------------------------------------------------
def _require_lab(what):
    if not hip.is_lab():
        raise hip.VqcpcHipError(f'{what} exists in the lab build only: VQCPC_LAB=1 python -m vqcpc_bach_amd.build, then run with VQCPC_LAB=1')


def split3_planes(x):
    """fp32 (rows, cols) -> the P3 format of csrc/gemm_planes.hip: three K-tile-major bf16 planes with
    x == high + mid + low exactly.  Returns a uint8 buffer of 6 * rows * cols bytes tagged with its logical shape."""
    _require_lab('split3_planes')
    x = _f32(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 16 == 0
    rows, cols = x.shape
    out = torch.empty(6 * rows * cols, dtype=torch.uint8, device=x.device)
    hip.call('vqcpc_split3_planes', x, x.stride(0), rows, cols, out)
    out.p3_shape = (rows, cols)
    return out


def join3_planes(planes, rows, cols):
    _require_lab('join3_planes')
    x = torch.empty(rows, cols, dtype=torch.float32, device=planes.device)
    hip.call('vqcpc_join3_planes', planes, rows, cols, x, cols)
    return x


def gemm_nt_planes(a_planes, b_planes, M, N, K, bias=None, act=0, drop_p=0.0, seed=0, gate=None, gate_scale=1.0, add=None,
                   out=None):
    """C[M, N] = epi(A . B^T) on pre-split (P3) operands: the same products, in the same order, as ops.gemm_nt in mode 1."""
    _require_lab('gemm_nt_planes')
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a_planes.device)
    hip.call('vqcpc_gemm_nt_planes', a_planes, b_planes, out, out.stride(0), M, N, K, bias, int(act), float(drop_p), int(seed),
             gate, gate.stride(0) if gate is not None else 0, float(gate_scale), add, add.stride(0) if add is not None else 0)
    return out


def cast_bf16(x):
    """fp32 (rows, cols), rows possibly strided -> dense torch.bfloat16 (round to nearest even, like .bfloat16())."""
    if x.dtype == torch.bfloat16:
        return x if x.is_contiguous() else x.contiguous()
    x, ld = _rows(_f32(x))
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    hip.call('vqcpc_cast_bf16', x, ld, out, x.shape[0], x.shape[1])
    return out


def _attach_bf16_copy(y, yb):
    """The producing layer hands the bf16 copy its LayerNorm kernel wrote to the consumer ON the fp32 tensor object: no
    global slot keyed by address (a freed-and-reused address would match a stale copy), and the copy lives exactly as
    long as the tensor it mirrors."""
    y._vqcpc_bf16 = (y.data_ptr(), y._version, yb)


def _bf16_copy_of(x):
    """The bf16 copy that the producing kernel wrote next to the fp32 tensor `x` (same values, rounded), if any: `x` must
    be the very tensor object the producer returned (same storage offset, not modified in place since)."""
    ent = getattr(x, '_vqcpc_bf16', None)
    if (ent is not None and ent[0] == x.data_ptr() and ent[1] == x._version and x.is_contiguous()
            and tuple(ent[2].shape) == tuple(x.shape)):
        return ent[2]
    return None


def bf16_native(*shapes):
    """True when the bf16 mode is on and every (M, N, K) fits the 256-tile bf16 kernel."""
    return hip.get_gemm_mode() == 2 and all(hip.query('vqcpc_gemm_nt_bf16_supported', m, n, k) for m, n, k in shapes)


def gemm_nt_bf16(a, b, bias=None, act=0, drop_p=0.0, seed=0, gate=None, gate_b=None, gate_scale=1.0, add=None, out=None,
                 out_f32=True, out_bf16=False, add_b=None):
    """epilogue(a[M,K] @ b[N,K]^T) on bf16 operands (fp32 tensors are cast first).  Returns the fp32 result, the bf16
    result, or (fp32, bf16) when both are requested."""
    a, b = cast_bf16(a), cast_bf16(b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    c = cb = None
    ldc = ldcb = 0
    if out_f32:
        c = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=a.device)
        c, ldc = _rows(c)
    if out_bf16:
        cb, ldcb = torch.empty(M, N, dtype=torch.bfloat16, device=a.device), N
    ldg = ldgb = lda_ = 0
    if gate is not None:
        gate, ldg = _rows(gate)
    if gate_b is not None:
        assert gate_b.dtype == torch.bfloat16 and gate_b.is_contiguous()
        ldgb = gate_b.shape[1]
    if add is not None:
        add, lda_ = _rows(add)
    ldab = 0
    if add_b is not None:                 # residual operand in bf16 (the LayerNorm's bf16 output)
        assert add_b.dtype == torch.bfloat16 and add_b.is_contiguous() and add is None
        ldab = add_b.shape[1]
    hip.call('vqcpc_gemm_nt_bf16', a, K, b, K, c, ldc, cb, ldcb, M, N, K, bias, int(act), float(drop_p), int(seed), gate, ldg,
             gate_b, ldgb, float(gate_scale), add, lda_, add_b, ldab)
    return (c, cb) if (out_f32 and out_bf16) else (c if out_f32 else cb)


def gemm_tn_bf16(a, b, want_bias=True, into=None):
    """dW[N,K] = a[M,N]^T @ b[M,K], db[N] = column sums of a, on bf16 operands (fp32 tensors are cast first)."""
    a, b = cast_bf16(a), cast_bf16(b)
    M, N = a.shape
    K = b.shape[1]
    assert b.shape[0] == M
    if into is not None:
        dw, db = into
        assert dw.shape == (N, K) and dw.is_contiguous() and (db is None or (db.shape == (N,) and db.is_contiguous()))
    else:
        dw = torch.empty(N, K, dtype=torch.float32, device=a.device)
        db = torch.empty(N, dtype=torch.float32, device=a.device) if want_bias else None
    nbytes = hip.query('vqcpc_gemm_tn_bf16_workspace', M, N, K)
    ws = hip.workspace(nbytes, a.device)
    if into is not None and _DIRECT_WGRAD and DEFER_TN_REDUCTIONS:
        splits = hip.query('vqcpc_gemm_tn_bf16_deferred_splits', M, N, K)
        if splits:
            hip.call('vqcpc_gemm_tn_bf16', a, N, b, K, dw, db, M, N, K, 2, ws, nbytes)
            _defer_tn_reduction(ws, splits, N, K, dw, db)
            return dw, db
    hip.call('vqcpc_gemm_tn_bf16', a, N, b, K, dw, db, M, N, K, 0 if into is None else 1, ws, nbytes)
    return dw, db


_DIRECT_WGRAD = False
BATCHED_TRANSPOSES = os.environ.get('VQCPC_BATCHED_TRANSPOSES', '1') != '0'      # A/B switch


class forward_arithmetic:
    """Context manager used by the trainers around the FORWARD pass of a training step (compute_losses): with
    `FWD_ARITH == 'f16x3'` (and the bf16x6 GEMM mode) the whole-round 256-tile products inside it run on the three-product fp16
    kernel with their own scale table on `flat_parameters` (call order, as the gradient scope's), rolled when the scope closes.
    Nothing else changes: outside the scope -- evaluation, encode_indices, inference -- every product stays on six MFMAs."""

    def __init__(self, flat_parameters, tag=None):
        self.flat = flat_parameters
        self.tag = tag

    def __enter__(self):
        global _FWD_SCALES
        self.prev = _FWD_SCALES
        self.mine = None
        if (self.prev is None and self.flat is not None and FWD_ARITH == 'f16x3' and hip.get_gemm_mode() == 1
                and torch.is_grad_enabled()):
            self.mine = _FWD_SCALES = _grad_scales_of(self.flat, ('fwd', self.tag))
            self.mine.begin()
            if WEIGHT_PLANES:                  # the weights of this step are final: their fp16 planes, once, for forward and backward
                global _PLANES
                self.planes = _PLANES = WEIGHT_T.instance(self.flat)
                self.planes.refresh_planes()
        return self

    def __exit__(self, *exc):
        global _FWD_SCALES, _PLANES
        if self.mine is not None:
            _FWD_SCALES = self.prev
            if getattr(self, 'planes', None) is not None:
                _PLANES = None
                # the gradient scope that follows uses them as they are -- unless an optimiser step comes in between
                self.planes.planes_paired = _PARAM_STEPS if exc[0] is None else None
                self.planes = None
            if exc[0] is None:
                self.mine.roll()
        return False


class direct_weight_gradients:
    """Context manager used by the trainers around `loss.backward()`: inside it the weight-gradient GEMMs accumulate
    straight into the parameters' existing `.grad` buffers (the flat all-reduce bucket) and hand `None` to autograd.
    Outside it (plain `backward()`, `torch.autograd.grad`) gradients are returned to autograd as usual."""

    def __init__(self, flat_parameters=None, tag=None):
        """flat_parameters: the trainer's parallel.FlatParameters (or its flat fp32 buffer) -- the transposed dgrad operands
        of its weights are then prepared by one launch (ops.WEIGHT_T, a cache owned by that object) instead of one per
        weight, and the scale states of the f16x3 gradient arithmetic (ops.GradScales) live on it.  tag: distinguishes several
        backward passes per step over the same buffer (the student step's bucketed form: teacher | encoder + decoder)."""
        self.flat = flat_parameters
        self.tag = tag

    def __enter__(self):
        global _DIRECT_WGRAD, _GRAD_SCALES
        self.prev, _DIRECT_WGRAD = _DIRECT_WGRAD, True
        if self.prev:                      # nested scope: the outermost one owns the deferred work and the transposed weights
            return self
        hip.gradient_scope(True)           # the GEMMs launched from here on are gradient GEMMs (hip.set_gradient_products)
        if self.flat is not None and GRAD_ARITH == 'f16x3' and hip.get_gemm_mode() == 1:
            _GRAD_SCALES = _grad_scales_of(self.flat, self.tag)
            _GRAD_SCALES.begin()
        if self.flat is not None and BATCHED_TRANSPOSES:
            WEIGHT_T.begin(self.flat)
            if WEIGHT_PLANES and _GRAD_SCALES is not None:
                global _PLANES
                inst = WEIGHT_T.current
                if inst.planes_paired is None or inst.planes_paired != _PARAM_STEPS:     # no forward scope made them for THIS step
                    inst.refresh_planes()
                inst.planes_paired = None
                _PLANES = inst
        return self

    def __exit__(self, *exc):
        global _DIRECT_WGRAD, _GRAD_SCALES
        _DIRECT_WGRAD = self.prev
        if self.prev:                      # an inner scope neither flushes nor discards what the outer scope deferred
            return False
        try:
            if exc[0] is None:
                flush_wgrads()                  # the deferred small weight gradients, still inside the gradient scope
                flush_reductions()              # ... and the LayerNorm weight / bias partial sums
                if _GRAD_SCALES is not None:
                    _GRAD_SCALES.roll()         # this step's operand amax becomes the next step's scale
            else:
                _PENDING_WGRADS.clear()
                _PENDING_REDUCTIONS.clear()
                _PENDING_VEC_REDUCTIONS.clear()
        finally:
            global _PLANES
            _GRAD_SCALES = None
            _PLANES = None
            hip.gradient_scope(False)
            WEIGHT_T.end()
        return False


def _live_grad(t):
    """The gradient buffer of a leaf parameter whose `.grad` already exists (parallel.FlatParameters installs views
    into the flat all-reduce bucket), else None."""
    if not _DIRECT_WGRAD or t is None or not t.is_leaf or not t.requires_grad:
        return None
    g = t.grad
    return g if (g is not None and g.is_contiguous() and g.dtype == torch.float32) else None


def accumulate_small(params, grads):
    """Small-parameter gradients of a fused layer: where the parameter already owns a gradient buffer (trainers:
    ops.direct_weight_gradients), all of them are added into those buffers by ONE launch and None is handed to autograd;
    otherwise the gradients are returned unchanged.  Returns the list to give back to autograd."""
    import ctypes
    out, dst, src, cnt = list(grads), [], [], []
    for i, (prm, g) in enumerate(zip(params, grads)):
        live = _live_grad(prm) if g is not None else None
        if live is not None and g.is_contiguous() and g.dtype == torch.float32 and live.numel() == g.numel():
            dst.append(live.data_ptr())
            src.append(g.data_ptr())
            cnt.append(g.numel())
            out[i] = None
    for o in range(0, len(dst), 8):
        n = min(8, len(dst) - o)
        hip.call('vqcpc_accumulate8', (ctypes.c_void_p * 8)(*dst[o:o + n]), (ctypes.c_void_p * 8)(*src[o:o + n]),
                 (ctypes.c_int * 8)(*cnt[o:o + n]), n)
    return out


def _accumulate_into(dst_tensors, src_tensors):
    """dst[i] += src[i] for up to 8 contiguous fp32 tensors per launch."""
    import ctypes
    for o in range(0, len(dst_tensors), 8):
        d, sr = dst_tensors[o:o + 8], src_tensors[o:o + 8]
        hip.call('vqcpc_accumulate8', (ctypes.c_void_p * 8)(*[t.data_ptr() for t in d]),
                 (ctypes.c_void_p * 8)(*[t.data_ptr() for t in sr]), (ctypes.c_int * 8)(*[t.numel() for t in sr]), len(d))


class SplitRowsFn(torch.autograd.Function):
    """x (R, ...) -> consecutive row blocks of the given sizes (copies).  Backward is ONE concatenation of the incoming
    gradients; autograd's own slice backward would fill a zero tensor of the full size per block, copy the block in and add
    the results up (3 blocks of the encoder output: 8 kernels instead of 1)."""

    @staticmethod
    def forward(ctx, x, *sizes):
        ctx.sizes, ctx.tail = sizes, x.shape[1:]
        out, s = [], 0
        for n in sizes:
            out.append(x[s:s + n].clone())
            s += n
        assert s == x.shape[0]
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else torch.zeros((n,) + tuple(ctx.tail), dtype=ref.dtype, device=ref.device)
                 for g, n in zip(grads, ctx.sizes)]
        return (torch.cat(parts, dim=0),) + (None,) * len(ctx.sizes)


class StackTablesFn(torch.autograd.Function):
    """(nv, vmax, emb) zero-padded stack of per-voice embedding tables (data_processor.stacked_tables): a fill + ONE launch
    forward and ONE launch backward (the slices of the incoming gradient are added straight into the tables' gradient
    buffers), instead of pad x nv + stack and their autograd mirror images (about 25 small kernels per step)."""

    @staticmethod
    def forward(ctx, *tables):
        vmax, emb = max(t.shape[0] for t in tables), tables[0].shape[1]
        out = torch.zeros(len(tables), vmax, emb, dtype=torch.float32, device=tables[0].device)
        if tables[0].is_cuda and all(t.is_contiguous() and t.dtype == torch.float32 for t in tables):
            _accumulate_into([out[c, :t.shape[0]] for c, t in enumerate(tables)], [t.detach() for t in tables])
        else:
            for c, t in enumerate(tables):
                out[c, :t.shape[0]] = t
        ctx.tables = tables
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        grads = [g[c, :t.shape[0]] for c, t in enumerate(ctx.tables)]       # rows of one voice: a contiguous block
        return tuple(accumulate_small(ctx.tables, grads)) if g.is_cuda else tuple(grads)


def wgrad(g, x, weight, bias, rows=None):
    """Weight / bias gradient of y = x W^T + b.  When the parameter already owns a gradient buffer the TN GEMM's final
    reduction ACCUMULATES straight into it and (None, None) is returned to autograd -- no temporary, no extra
    `grad += dW` pass per parameter; otherwise (dW, db) are returned as usual.  `rows` = slice of output features when
    only a row block of the parameter is differentiated (q | k,v halves of in_proj)."""
    wg, bg = _live_grad(weight), _live_grad(bias)
    tn = gemm_tn
    if g.dtype == torch.bfloat16 or x.dtype == torch.bfloat16:       # bf16 path: operands already bf16 in HBM
        assert hip.query('vqcpc_gemm_tn_bf16_supported', g.shape[0], g.shape[1], x.shape[1])
        tn = gemm_tn_bf16
    if wg is not None and (bias is None or bg is not None):
        if rows is not None:
            wg, bg = wg[rows], (bg[rows] if bg is not None else None)
        tn(g, x, into=(wg, bg))
        return None, None
    return tn(g, x, want_bias=bias is not None)


def transpose(w):
    w = _f32(w).contiguous()
    hit = WEIGHT_T.lookup(w)
    if hit is not None:
        return hit
    out = torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=w.device)
    hip.call('vqcpc_transpose', w, out, w.shape[0], w.shape[1])
    return out


class _WeightTransposes:
    """W^T of every 2-D weight a backward pass asks for, refreshed by ONE launch when the pass begins
    (`direct_weight_gradients(flat_parameters)`), instead of one 5 us launch per weight in the middle of it (25 per CPC
    step, 80 per student step).  ONE instance per flat parameter buffer (a trainer), created on first use and kept for the
    life of the process: captured step graphs (graphs.py) bake in the arena and descriptor-table device pointers and the
    frozen (n, total_tiles) arguments of the transpose_many node, so neither is ever freed or overwritten -- a new
    descriptor table (weights learned after a capture) is a NEW tensor and the old ones stay alive and valid.
    Only tensors inside the flat buffer are cached (their addresses are stable and nothing else can live there); a weight
    is learned the first time `transpose` is asked for it and served from the arena from the next pass on.  The arena is
    valid between `begin` and `end` only: weights change in the optimiser step that follows."""

    def __init__(self, flat):
        self.flat = flat                 # keeps the buffer (and hence its address) alive
        self.arena = torch.empty_like(flat)
        self.entries = {}          # data_ptr -> (offset, rows, cols)
        self.uploaded = {}         # the entries the current device table covers
        self.desc = None
        self.retired = []          # earlier descriptor tables: captured graphs may still read them
        self.total_tiles = 0
        # round 6: the same matrices as fp16 plane pairs ("P4", csrc/gemm_grad.hip), made once per step: `planes` mirrors the flat
        # buffer (B operand of the forward products), `planes_t` the arena (B operand of the input-gradient products), amax[i] = the
        # amax matrix i (in descriptor order) was scaled with.  Same lifetime rules as the arena: never freed, never moved.
        self.planes = torch.empty_like(flat)
        self.planes_t = torch.empty_like(flat)
        self.amax = torch.zeros(self.MAX_MATRICES, dtype=torch.float32, device=flat.device)
        # one float per 32 x 32 tile of every registered matrix (their maxima, reduced per matrix by the plane pass: no atomics)
        self.tile_max = torch.zeros(flat.numel() // 256 + 4 * self.MAX_MATRICES + 64, dtype=torch.float32, device=flat.device)
        self.planes_paired = None       # _PARAM_STEPS at which a forward scope made the planes: the gradient scope of the same step uses them as they are
        self.planes_live = False        # the plane images describe the current `uploaded` set
        self._plane_hits = {}           # (ptr, rows, cols, ld) -> (plane tensor, amax slot) | None
        self._sorted = []               # (offset, rows, cols, index) of the uploaded set, ascending

    MAX_MATRICES = 4096

    def _upload(self):
        if len(self.entries) != len(self.uploaded) and not (self.flat.is_cuda and torch.cuda.is_current_stream_capturing()):
            rows, tiles = [], 0
            for key, (off, r, c) in sorted(self.entries.items(), key=lambda kv: kv[1][0]):
                rows.append((off, r, c, tiles))
                tiles += ((r + 31) // 32) * ((c + 31) // 32)
            if self.desc is not None:
                self.retired.append(self.desc)
            self.desc = torch.tensor(rows, dtype=torch.int64).to(self.flat.device)
            self.total_tiles, self.uploaded = tiles, dict(self.entries)
            self._sorted = [(off, r, c, i) for i, (off, r, c, _) in enumerate(rows)]
            self._plane_hits = {}
            self.planes_live = False

    def refresh_planes(self):
        """The P4 images of every uploaded matrix and of its transpose under this step's own amax: three small launches."""
        self._upload()
        n = len(self.uploaded)
        if n and n <= self.MAX_MATRICES and self.total_tiles <= self.tile_max.numel():
            hip.call('vqcpc_weight_planes_many', self.flat, self.desc, n, self.total_tiles, self.amax, self.planes, self.planes_t,
                     self.tile_max, 4 * self.tile_max.numel())
            self.planes_live = True

    def lookup_planes(self, b, ldb):
        """(P4 image at b's address, amax slot) when the 2-D operand `b` (unit inner stride, leading dimension ldb) is a registered
        weight -- or a row / column block of one -- inside the flat buffer, or of its transpose inside the arena; else None."""
        if not self.planes_live:
            return None
        key = (b.data_ptr(), b.shape[0], b.shape[1], ldb)
        hit = self._plane_hits.get(key, 0)
        if hit != 0:
            return hit
        res = None
        ptr = key[0]
        for base_t, img, transposed in ((self.flat, self.planes, False), (self.arena, self.planes_t, True)):
            base = base_t.data_ptr()
            off = (ptr - base) // 4
            if not (base <= ptr and off < base_t.numel() and (ptr - base) % 16 == 0):
                continue
            i = bisect.bisect_right(self._sorted, (off, 1 << 62, 0, 0)) - 1
            if i < 0:
                continue
            o, r, c, idx = self._sorted[i]
            ld = r if transposed else c                     # leading dimension of the (possibly transposed) registered matrix
            rows_all = c if transposed else r
            if (off < o + r * c and ldb == ld and ld % 4 == 0 and b.shape[1] % 4 == 0
                    and (off - o) % ld + b.shape[1] <= ld and (off - o) // ld + b.shape[0] <= rows_all):
                span = (b.shape[0] - 1) * ld + b.shape[1]
                res = (img[off:off + span], self.amax[idx:idx + 1])
            break
        self._plane_hits[key] = res
        return res

    def refresh(self):
        self._upload()
        if self.uploaded:
            hip.call('vqcpc_transpose_many', self.flat, self.arena, self.desc, len(self.uploaded), self.total_tiles)

    def lookup(self, w):
        ptr = w.data_ptr()
        ent = self.uploaded.get(ptr)
        if ent is not None and ent[1] == w.shape[0] and ent[2] == w.shape[1]:
            return self.arena[ent[0]:ent[0] + ent[1] * ent[2]].view(ent[2], ent[1])
        base = self.flat.data_ptr()
        off = (ptr - base) // 4
        n = w.shape[0] * w.shape[1]
        if base <= ptr and off + n <= self.flat.numel() and ptr not in self.entries:
            # no overlap with another registered matrix (a sub-block of a weight next to the whole weight would share arena space)
            if all(off + n <= o or o + r * c <= off for (o, r, c) in self.entries.values()):
                self.entries[ptr] = (off, w.shape[0], w.shape[1])
        return None


class _WeightTransposeRegistry:
    """`ops.WEIGHT_T`: hands out the per-trainer caches and tracks the one that is active (between begin and end of a
    backward pass).  The cache lives ON its owner (a parallel.FlatParameters, or the flat tensor itself), so it dies with
    the trainer -- and with the trainer's step graphs, the only other holders of its device pointers."""

    def __init__(self):
        self.current = None

    def instance(self, owner):
        flat = owner if torch.is_tensor(owner) else owner.flat
        inst = getattr(owner, '_vqcpc_weight_t', None)
        if inst is None or inst.flat.data_ptr() != flat.data_ptr() or inst.flat.numel() != flat.numel():
            inst = _WeightTransposes(flat)
            owner._vqcpc_weight_t = inst
        return inst

    def begin(self, owner):
        inst = self.instance(owner)
        inst.refresh()
        self.current = inst

    def end(self):
        self.current = None

    def lookup(self, w):
        if self.current is None or w.dim() != 2:
            return None
        return self.current.lookup(w)


WEIGHT_T = _WeightTransposeRegistry()


# ------------------------------------------------------------------------------------------------------------------
# A2 + input_linear + positional concat
# ------------------------------------------------------------------------------------------------------------------
class EmbedPosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, table, chan, event, tokens_per_block):
        # tokens (rows,) int64 ; table (nv, vmax, dlin) ; chan (nv, pos) ; event (nev, pos) or None (no event part)
        nv, vmax, dlin = table.shape
        pos = chan.shape[1]
        rows = tokens.numel()
        out = torch.empty(rows, dlin + (2 if event is not None else 1) * pos, dtype=torch.float32, device=table.device)
        hip.call('vqcpc_embed_pos_fwd', tokens, rows, tokens_per_block, nv, table.contiguous(), vmax, dlin,
                 chan.contiguous(), event.contiguous() if event is not None else None, pos, out)
        ctx.save_for_backward(tokens)
        ctx.meta = (rows, tokens_per_block, nv, vmax, dlin, pos, event.shape[0] if event is not None else 0)
        return out

    @staticmethod
    def backward(ctx, g):
        (tokens,) = ctx.saved_tensors
        rows, tpb, nv, vmax, dlin, pos, nev = ctx.meta
        g = g.contiguous()
        d_table = torch.empty(nv, vmax, dlin, dtype=torch.float32, device=g.device)
        d_chan = torch.empty(nv, pos, dtype=torch.float32, device=g.device)
        d_event = torch.empty(nev, pos, dtype=torch.float32, device=g.device) if nev else None
        nbytes = hip.query('vqcpc_embed_pos_bwd_workspace', rows, tpb, nv, vmax, dlin, pos)
        ws = hip.workspace(nbytes, g.device)
        hip.call('vqcpc_embed_pos_bwd', tokens, rows, tpb, nv, vmax, dlin, pos, g, d_table, d_chan, d_event, ws, nbytes)
        return None, d_table, d_chan, d_event, None


class BlockTableGatherFn(torch.autograd.Function):
    """out[r] = table[tokens[r] * L + r % L]  (table (vmax * L, C)); backward = deterministic segment sum."""

    @staticmethod
    def forward(ctx, table, tokens, L):
        table = _f32(table).contiguous()
        rows, C = table.shape
        vmax = rows // L
        M = tokens.numel()
        out = torch.empty(M, C, dtype=torch.float32, device=table.device)
        hip.call('vqcpc_block_table_gather', table, tokens, out, M, L, vmax, C)
        ctx.save_for_backward(tokens)
        ctx.meta = (L, vmax, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (tokens,) = ctx.saved_tensors
        L, vmax, C = ctx.meta
        g = g.contiguous()
        M = tokens.numel()
        d_table = torch.empty(vmax * L, C, dtype=torch.float32, device=g.device)
        nbytes = hip.query('vqcpc_block_table_segsum_workspace', M, L, vmax, C)
        ws = hip.workspace(nbytes, g.device)
        hip.call('vqcpc_block_table_segsum', g, tokens, d_table, M, L, vmax, C, ws, nbytes)
        return d_table, None, None


# ------------------------------------------------------------------------------------------------------------------
# plain linear (output_linear, upscaler)
# ------------------------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.bias = bias
        return gemm_nt(x, weight, bias=bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = gemm_nt(g, transpose(weight)) if ctx.needs_input_grad[0] else None
        dw, db = wgrad(g, x, weight, ctx.bias)
        return dx, dw, db


def linear(x, weight, bias=None):
    """F.linear on the last dimension through the MFMA GEMM.  The GEMMs work on multiples of 4 features; other sizes
    (the student's codebook_dim = 3, per-voice vocabularies) are zero-padded, which changes no result."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    N, K = weight.shape
    pk, pn = -K % 4, -N % 4
    if pk or pn:
        pad = torch.nn.functional.pad
        if pk:
            x2 = pad(x2, (0, pk))
        y = LinearFn.apply(x2, pad(weight, (0, pk, 0, pn)), pad(bias, (0, pn)) if bias is not None else None)
        return y[:, :N].reshape(*lead, N)
    return LinearFn.apply(x2, weight, bias).reshape(*lead, N)


# ------------------------------------------------------------------------------------------------------------------
# one post-LN relative-attention encoder layer (A6/A7/A8), forward + hand-scheduled backward
# ------------------------------------------------------------------------------------------------------------------
SFORM_MIN_TILES = 128      # 256 x 256 tiles from which the residual sum is formed in the GEMM epilogue (tests set 0)


def _residual_sum_in_epilogue(M, N, K, nat):
    """Where the residual sum s = x + dropout(a W^T + b) is formed.  True: in the producing GEMM's epilogue (LayerNorm then
    reads one stream) -- pays where the 256-tile ping-pong kernel runs the GEMM (C1 / C4: >= 128 tiles).  False: the
    projection output and the residual stay two LayerNorm inputs -- the under-filled projections of the student / decoder-sized
    steps run on the 128-tile kernel (3072 x 512 x 512 with the dropout + residual epilogue: 87 us against 38 us with the
    bias alone) or through the split-K entry point (3072 x 512 x 2048: 59 us against 112 us; its plane-sum epilogue has no
    dropout); measured on C3: 13.5 -> 16.4 ms/step with the epilogue form everywhere."""
    if nat:
        return True
    if hip.get_gemm_mode() != 1:
        return SFORM_MIN_TILES == 0
    return (M // 256) * (N // 256) >= SFORM_MIN_TILES and M % 256 == 0 and N % 256 == 0 or SFORM_MIN_TILES == 0


def _dgrad_plus_residual_bf16(g, wt, res, carrier=False):
    """g @ wt^T + res on the bf16 path; `res` fp32 or -- the gradient of a residual branch kept in bf16 -- bf16.
    carrier: the layer that produced this layer's input reads its output gradient in bf16 (EncoderLayerFn.forward): the sum leaves
    the epilogue in bf16 only, wrapped for autograd by _bf16_grad_carrier."""
    if res.dtype != torch.bfloat16:
        return gemm_nt_bf16(g, wt, add=res)
    if carrier:
        return _bf16_grad_carrier(gemm_nt_bf16(g, wt, add_b=res, out_f32=False, out_bf16=True))
    return gemm_nt_bf16(g, wt, add_b=res)


_NAN1 = {}


def _bf16_grad_carrier(gb):
    """The gradient `gb` (bf16) of an fp32 tensor, handed to autograd as an fp32-typed tensor of the right shape that owns no
    memory of that size: a stride-0 view of ONE NaN with the real gradient attached.  The consumer (the producing layer's
    backward, which announced on its output that it reads such carriers) takes the attachment; anything else that touches the
    values -- autograd summing it with a second consumer's gradient, a view op's backward -- produces NaNs, loudly."""
    nan1 = _NAN1.get(gb.device)
    if nan1 is None:
        nan1 = _NAN1[gb.device] = torch.full((1,), float('nan'), dtype=torch.float32, device=gb.device)
    c = nan1.expand(gb.shape)
    c._vqcpc_bf16_grad = gb
    return c


def _bf16_act_carrier(yb):
    """The same for an ACTIVATION that exists in bf16 only: the output of a stack's interior layer on the bf16 path, whose one consumer
    is the next EncoderLayerFn (GEMM operand and residual operand alike, both read bf16).  fp32-typed for autograd, values NaN."""
    c = _bf16_grad_carrier(yb)
    del c._vqcpc_bf16_grad
    c._vqcpc_bf16_act = yb
    return c


def _bf16_act_of(x):
    xb = getattr(x, '_vqcpc_bf16_act', None)
    return xb if (xb is not None and tuple(xb.shape) == tuple(x.shape)) else None


def _bf16_grad_of(g):
    gb = getattr(g, '_vqcpc_bf16_grad', None)
    return gb if (gb is not None and tuple(gb.shape) == tuple(g.shape)) else None


class EncoderLayerFn(torch.autograd.Function):
    """y = LN2(x1 + drop(W2 drop(relu(W1 x1 + b1)) + b2)),  x1 = LN1(x + drop(Wo attn(x) + bo)).
    Parameter order: in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, e1, e2, linear1.weight,
    linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias, norm2.weight, norm2.bias.

    qstride = f > 1 (last layer of a stack): only rows 0, f, 2f, ... of the output are produced -- the reference computes
    all rows and keeps `output[::f]` (relative_transformer_downscaler.py:125); everything after the attention is
    per-token, so queries / out-proj / LayerNorms / FFN run on M/f rows while keys and values still cover every token.
    Identical results, ~60 % fewer FLOPs in that layer."""

    @staticmethod
    def forward(ctx, x, L, H, drop_p, seed, qstride, qkv_in, qkv_tokens, out_b16_only, wqkv, bqkv, wo, bo, e1, e2, w1, b1, w2, b2, g1,
                be1, g2, be2):
        # qkv_in: the in_proj output computed elsewhere (first layer); wqkv / bqkv are then not used here and the gradient
        # of the projection is handed back through qkv_in.  Either the per-token (M, 3d) tensor, or -- with qkv_tokens
        # (M,) int64 -- the (vmax * L, 3d) block table, which the attention kernels read through the token indirection
        # out_b16_only: the caller promises that the ONE consumer of y is the next EncoderLayerFn of the stack (an interior layer): on
        # the bf16 path y then exists in bf16 only (_bf16_act_carrier)
        x_accepts_b16_grad = bool(getattr(x, '_vqcpc_accepts_bf16_grad', False))     # set by the layer that produced x
        xb_in = _bf16_act_of(x)              # x itself in bf16 only (the previous interior layer's output)
        if xb_in is not None and not (hip.get_gemm_mode() == 2 and BF16_RESIDUAL and BF16_SUMS and qkv_in is None):
            x, xb_in = xb_in.float(), None   # a consumer off the all-bf16 path: the values, upcast
            x_accepts_b16_grad = False
        if xb_in is None:
            x, ldx = _rows(_f32(x))
        else:
            ldx = x.shape[1]
        M, d = x.shape
        hd = d // H
        nblk = M // L
        dev = x.device
        p = float(drop_p)
        f = int(qstride)
        s = [int(seed) + 0x1000 * i for i in range(4)]     # attention probs, dropout1, ffn dropout, dropout2
        ffd = w1.shape[0]
        Mq_ = M // f
        # bf16 mode (configs[4]): the layer's GEMMs take bf16 operands from HBM; activations that only feed GEMMs get a bf16
        # copy from the producing epilogue (FFN hidden) or from a cast pass (x, attention output, LayerNorm output)
        nat = bf16_native((M, 3 * d if (f == 1 and qkv_in is None) else 2 * d, d), (Mq_, d, d), (Mq_, ffd, d), (Mq_, d, ffd)) and \
            hip.query('vqcpc_gemm_tn_bf16_supported', Mq_, d, d)
        lin = gemm_nt_bf16 if nat else gemm_nt
        if xb_in is not None and not nat:               # (shapes off the 256-tile bf16 kernels)
            x, xb_in = xb_in.float(), None
            x_accepts_b16_grad = False
            x, ldx = _rows(x)
        xb = None                                       # bf16 copies: GEMM operands now, weight-gradient operands later
        if nat and (qkv_in is None or f > 1):
            xb = xb_in if xb_in is not None else _bf16_copy_of(x)      # written by the previous layer's LayerNorm kernel
            if xb is None:
                xb = cast_bf16(x)
        xsb = None
        if f == 1:
            Mq, xs, ldxs = M, x, ldx
            xsb = xb if xb_in is not None else None
            probs = torch.empty(nblk, H, L, L, dtype=torch.float32, device=dev)
            # bf16 path at L = 16: the attention context only feeds the out-proj GEMM, which reads bf16 -> the kernel writes
            # bf16 directly (no fp32 tensor, no cast pass); where the projection runs here (no block table) q | k | v are
            # bf16 as well -- the in_proj epilogue writes them so and the attention kernels read half the bytes
            b16_att = nat and ATT_B16_OUT and bool(hip.query('vqcpc_relattn16_b16_supported', L, H, hd))
            b16_qkv = b16_att and ATT_B16_IN and qkv_in is None
            if qkv_in is not None:
                qkv = _f32(qkv_in).contiguous()
            elif b16_qkv:
                qkv = gemm_nt_bf16(xb, wqkv, bias=bqkv, out_f32=False, out_bf16=True)
            else:
                qkv = lin(xb if nat else x, wqkv, bias=bqkv)
            attb_direct = None
            if b16_att:
                attb_direct = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
                att = None
                if b16_qkv:
                    hip.call('vqcpc_relattn16_fwd_b16io', qkv, 3 * d, e1, e2, attb_direct, d, probs, nblk, H, hd, p, s[0])
                else:
                    hip.call('vqcpc_relattn16_fwd_b16', qkv, 3 * d, qkv_tokens, e1, e2, attb_direct, d, probs, nblk, H, hd, p, s[0])
            elif (nat and ATT_B16_OUT and qkv_tokens is None and hip.query('vqcpc_relattn_b16_supported', L, H, hd)):
                attb_direct = torch.empty(M, d, dtype=torch.bfloat16, device=dev)      # the other block lengths (L = 4)
                att = None
                hip.call('vqcpc_relattn_fwd_b16', qkv, 3 * d, e1, e2, attb_direct, d, probs, nblk, L, H, hd, p, s[0])
            else:
                att = torch.empty(M, d, dtype=torch.float32, device=dev)
                if qkv_tokens is not None:
                    hip.call('vqcpc_relattn_tab_fwd', qkv, 3 * d, qkv_tokens, e1, e2, att, d, probs, nblk, L, H, hd, p, s[0])
                else:
                    hip.call('vqcpc_relattn_fwd', qkv, 3 * d, e1, e2, att, d, probs, nblk, L, H, hd, p, s[0])
            qproj = qkv
        else:
            assert L % f == 0 and qkv_in is None
            attb_direct = None
            Mq = M // f
            xs, ldxs = _rows(x[::f]) if xb_in is None else (None, 0)       # query / residual rows: a stride, not a copy
            xsb = xb[::f].contiguous() if nat else None
            qkv = lin(xb if nat else x, wqkv[d:], bias=bqkv[d:])           # k | v for every token   (M, 2d)
            qproj = lin(xsb if nat else xs, wqkv[:d], bias=bqkv[:d])       # q for the kept rows     (Mq, d)
            probs = torch.empty(nblk, H, L // f, L, dtype=torch.float32, device=dev)
            if nat and ATT_B16_OUT and hip.query('vqcpc_relattn_sub_b16_supported', L, f, H, hd):
                attb_direct = torch.empty(Mq, d, dtype=torch.bfloat16, device=dev)
                att = None
                hip.call('vqcpc_relattn_sub_fwd_b16', qproj, d, qkv, 2 * d, e1, e2, attb_direct, d, probs, nblk, L, f, H, hd, p, s[0])
            else:
                att = torch.empty(Mq, d, dtype=torch.float32, device=dev)
                hip.call('vqcpc_relattn_sub_fwd', qproj, d, qkv, 2 * d, e1, e2, att, d, probs, nblk, L, f, H, hd, p, s[0])
        attb = (attb_direct if attb_direct is not None else cast_bf16(att)) if nat else None
        # s1 = x + dropout(att Wo^T + bo): the residual sum is formed by the out-proj epilogue (bias -> dropout -> + x), so the
        # LayerNorm kernels read ONE input stream and the backward needs neither x nor the projection output again
        sform1 = _residual_sum_in_epilogue(Mq, d, d, nat)
        # bf16 path: x1 = LN1(...) feeds the two feed-forward GEMMs (bf16 operand) and the residual of s2 = x1 + dropout(FFN): with
        # the residual read from the bf16 copy as well (round 5) the LayerNorm writes 2 instead of 6 bytes per element and the
        # FFN2 epilogue reads 2 instead of 4 -- the residual stream between LN1 and LN2 is bf16, as the GEMM operands already are
        x1 = None if (nat and BF16_RESIDUAL) else torch.empty(Mq, d, dtype=torch.float32, device=dev)
        mean1 = torch.empty(Mq, dtype=torch.float32, device=dev)
        rstd1 = torch.empty(Mq, dtype=torch.float32, device=dev)
        x1b = torch.empty(Mq, d, dtype=torch.bfloat16, device=dev) if nat else None
        # ... and the residual sums s1, s2 themselves leave their GEMM epilogues in bf16: 2 instead of 4 bytes out of the epilogue,
        # into the LayerNorm forward and into its backward (include/vqcpc.h: vqcpc_layernorm_fwd_xb16)
        s16 = nat and BF16_RESIDUAL and BF16_SUMS
        assert xb_in is None or s16
        if s16:
            s1 = (gemm_nt_bf16(attb, wo, bias=bo, drop_p=p, seed=s[1], add_b=xsb, out_f32=False, out_bf16=True) if xb_in is not None
                  else gemm_nt_bf16(attb, wo, bias=bo, drop_p=p, seed=s[1], add=xs, out_f32=False, out_bf16=True))
            hip.call('vqcpc_layernorm_fwd_xb16', s1, d, g1, be1, x1, x1b, mean1, rstd1, Mq, d, 1e-5)
        elif sform1:
            s1 = lin(attb if nat else att, wo, bias=bo, drop_p=p, seed=s[1], add=xs)
            hip.call('vqcpc_add_layernorm_fwd_b16', s1, d, None, g1, be1, x1, x1b, mean1, rstd1, Mq, d, 1e-5, 0.0, 0)
        else:           # s1 holds the projection output a; LayerNorm adds x and the dropout itself
            s1 = lin(att, wo, bias=bo)
            hip.call('vqcpc_add_layernorm_fwd_b16', xs, ldxs, s1, g1, be1, x1, x1b, mean1, rstd1, Mq, d, 1e-5, p, s[1])
        h2b = None
        if nat:     # the FFN hidden activation exists in bf16 only: FFN2, the backward gate and the weight gradient read it
            h2b = gemm_nt_bf16(x1b, w1, bias=b1, act=1, drop_p=p, seed=s[2], out_f32=False, out_bf16=True)
            s2 = (gemm_nt_bf16(h2b, w2, bias=b2, drop_p=p, seed=s[3], add_b=x1b, out_f32=not s16, out_bf16=s16) if x1 is None else
                  gemm_nt_bf16(h2b, w2, bias=b2, drop_p=p, seed=s[3], add=x1))
            sform2 = True
            h2 = att = x1b[:0]                       # placeholders in the saved list (never read on this path)
        else:
            if gatebits_worthwhile(Mq, ffd, d):       # relu / dropout gate of the backward as a bit mask (1/32 of the bytes)
                h2, ctx.gate_mask = gemm_nt_relu_mask(x1, w1, b1, drop_p=p, seed=s[2])
            else:
                h2, ctx.gate_mask = gemm_nt(x1, w1, bias=b1, act=1, drop_p=p, seed=s[2]), None
            sform2 = _residual_sum_in_epilogue(Mq, d, ffd, nat)
            if sform2:
                s2 = gemm_nt(h2, w2, bias=b2, drop_p=p, seed=s[3], add=x1)      # x1 + dropout(FFN(x1))
            else:
                s2 = gemm_nt(h2, w2, bias=b2)                                   # FFN(x1): LayerNorm adds x1 and the dropout
        y_carrier = bool(out_b16_only and s16 and BF16_ACT_STREAM)
        y = None if y_carrier else torch.empty(Mq, d, dtype=torch.float32, device=dev)
        mean2 = torch.empty(Mq, dtype=torch.float32, device=dev)
        rstd2 = torch.empty(Mq, dtype=torch.float32, device=dev)
        yb = torch.empty(Mq, d, dtype=torch.bfloat16, device=dev) if nat else None
        if s16:
            hip.call('vqcpc_layernorm_fwd_xb16', s2, d, g2, be2, y, yb, mean2, rstd2, Mq, d, 1e-5)
            if y_carrier:
                y = _bf16_act_carrier(yb)
        elif sform2:
            hip.call('vqcpc_add_layernorm_fwd_b16', s2, d, None, g2, be2, y, yb, mean2, rstd2, Mq, d, 1e-5, 0.0, 0)
        else:
            hip.call('vqcpc_add_layernorm_fwd_b16', x1, d, s2, g2, be2, y, yb, mean2, rstd2, Mq, d, 1e-5, p, s[3])
        if nat:
            if not y_carrier:
                _attach_bf16_copy(y, yb)
            # round 5: this layer's backward reads the gradient of y as bf16 (vqcpc_layernorm_bwd_b16io) when the consumer hands it
            # over so -- the next layer's input-gradient GEMM then writes 2 instead of 4 bytes per element and LN2's backward reads 2
            if s16 and BF16_GRAD_SUMS and BF16_GRAD_STREAM:
                y._vqcpc_accepts_bf16_grad = True
        ctx.dx_b16 = bool(nat and x_accepts_b16_grad and BF16_GRAD_SUMS and BF16_GRAD_STREAM)
        ctx.x_b16_only = xb_in is not None
        if xb_in is not None:
            x = x[:0]                       # the carrier's values are never read: only its shape (ctx.x_shape)
        ctx.x_shape = (M, d)
        ctx.save_for_backward(x, qkv, qproj, probs, att, s1, x1, mean1, rstd1, h2, s2, mean2, rstd2, wqkv, wo, e1, e2, w1,
                              w2, g1, g2)
        ctx.meta = (L, H, p, s, f, qkv_in is not None)
        ctx.sform = (sform1, sform2)
        ctx.bf16 = (xb, xsb, attb, x1b, h2b) if nat else None
        ctx.biases = (bqkv, bo, b1, b2)
        ctx.ln_betas = (be1, be2)
        ctx.qkv_tokens = qkv_tokens
        ctx.mark_non_differentiable(probs)
        ctx.set_materialize_grads(False)        # no zero tensor of the size of the attention maps for the unused output
        return y, probs

    @staticmethod
    def backward(ctx, dy, _dprobs):
        (x, qkv, qproj, probs, att, s1, x1, mean1, rstd1, h2, s2, mean2, rstd2, wqkv, wo, e1, e2, w1, w2, g1,
         g2) = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros(s2.shape, dtype=torch.float32, device=s2.device)
        dyb = _bf16_grad_of(dy)              # the consumer's input gradient in bf16 (see _bf16_grad_carrier)
        L, H, p, s, f, ext_qkv = ctx.meta
        bqkv, bo, b1, b2 = ctx.biases
        be1, be2 = ctx.ln_betas
        M, d = ctx.x_shape
        if ctx.x_b16_only:                   # x existed in bf16 only (ctx.bf16 holds it): no fp32 rows to address
            ldx, xs, ldxs = d, None, 0
        else:
            x, ldx = _rows(x)
            xs, ldxs = (x, ldx) if f == 1 else _rows(x[::f])
        hd, nblk, dev = d // H, M // L, x.device
        Mq = M // f
        dy = dyb if dyb is not None else dy.contiguous()

        nat = ctx.bf16 is not None

        def ln_bwd(dyv, xin, ldxin, r, gamma, beta, mean, rstd, seed, ds_bf16=False):
            # r None: xin is the residual sum itself (s-form, include/vqcpc.h); the mask of d_r is regenerated from `seed`
            # ds_bf16 (bf16 path, xin in bf16): the gradient of the residual branch leaves in bf16 only (returned in place of ds)
            ds_bf16 = ds_bf16 and xin.dtype == torch.bfloat16
            ds = None if ds_bf16 else torch.empty(Mq, d, dtype=torch.float32, device=dev)
            dsb = torch.empty(Mq, d, dtype=torch.bfloat16, device=dev) if ds_bf16 else None
            # bf16 path: the gradient of the sub-layer output only feeds GEMMs, which read its bf16 copy -> no fp32 d_r stream
            dr = torch.empty(Mq, d, dtype=torch.float32, device=dev) if (p > 0 and not nat) else None
            drb = torch.empty(Mq, d, dtype=torch.bfloat16, device=dev) if nat else None    # GEMM-operand copy of dr
            nbytes = hip.query('vqcpc_add_layernorm_bwd_workspace', Mq, d)
            ws = hip.workspace(nbytes, dev)
            if defer_ln_param_grads(ws, Mq, d, gamma, beta, r is not None):     # gamma / beta partial sums: reduced with all the others later
                dg = db = None
            else:
                dg = torch.empty(d, dtype=torch.float32, device=dev)
                db = torch.empty(d, dtype=torch.float32, device=dev)
            if xin.dtype == torch.bfloat16:              # the residual sum was written in bf16 (s-form only)
                assert r is None
                # ... and the incoming gradient too where an input-gradient GEMM of this path produced it (round 5)
                hip.call('vqcpc_layernorm_bwd_b16io' if dyv.dtype == torch.bfloat16 else 'vqcpc_layernorm_bwd_xb16', dyv, xin, ldxin,
                         gamma, mean, rstd, ds, dsb, dr, drb, dg, db, Mq, d, p, seed, ws, nbytes)
                if ds_bf16:
                    ds = dsb
            else:
                assert dyv.dtype == torch.float32
                hip.call('vqcpc_add_layernorm_bwd_b16', dyv, xin, ldxin, r, gamma, mean, rstd, ds, dr, drb, dg, db, Mq, d, p, seed,
                         ws, nbytes)
            return ds, (dr if dr is not None else (None if (nat and p > 0) else ds)), dg, db, drb

        sform1, sform2 = ctx.sform
        g16 = nat and BF16_GRAD_SUMS            # bf16 path: d s2 / d s1 only feed the residual operand of a dgrad epilogue -> bf16
        if sform2:
            ds2, df, dg2, dbe2, dfb = ln_bwd(dy, s2, d, None, g2, be2, mean2, rstd2, s[3], ds_bf16=g16)
        else:
            ds2, df, dg2, dbe2, dfb = ln_bwd(dy, x1, d, s2, g2, be2, mean2, rstd2, s[3])
        lin = gemm_nt_bf16 if nat else gemm_nt
        if nat:
            xb, xsb, attb, x1b, h2b = ctx.bf16
            # FFN: da = (df @ W2) * [h2 > 0] / (1 - p), bf16 only (it feeds two GEMMs and nothing else)
            da = gemm_nt_bf16(dfb, transpose(w2), gate_b=h2b, gate_scale=1.0 / (1.0 - p), out_f32=False, out_bf16=True)
            dw2, db2 = wgrad(dfb, h2b, w2, b2)
            dw1, db1 = wgrad(da, x1b, w1, b1)
            # d x1 -- the gradient that enters LN1's backward -- in bf16 only where that kernel reads the residual sum in bf16 too
            dx1_b16 = BF16_GRAD_STREAM and sform1 and ds2.dtype == torch.bfloat16 and s1.dtype == torch.bfloat16
            dx1 = (gemm_nt_bf16(da, transpose(w1), add_b=ds2, out_f32=not dx1_b16, out_bf16=dx1_b16)
                   if ds2.dtype == torch.bfloat16 else gemm_nt_bf16(da, transpose(w1), add=ds2))
        else:
            # FFN: da = (df @ W2) * [h2 > 0] / (1 - p)   (relu + dropout backward folded into the GEMM epilogue)
            if ctx.gate_mask is not None:
                da = gemm_nt_gatebits(df, transpose(w2), ctx.gate_mask, gate_scale=1.0 / (1.0 - p))
            else:
                da = gemm_nt(df, transpose(w2), gate=h2, gate_scale=1.0 / (1.0 - p))
            dw2, db2 = wgrad(df, h2, w2, b2)
            dw1, db1 = wgrad(da, x1, w1, b1)
            dx1 = gemm_nt_residual(da, transpose(w1), ds2, res_may_alias=df)     # ds2 is dead afterwards
        del da, df, ds2
        # d s1 is bf16 only where the dgrad of the in_proj consumes it here (the all-bf16 attention paths below)
        g16_1 = (g16 and f == 1 and not ext_qkv and ctx.qkv_tokens is None and ATT_B16_OUT and ctx.needs_input_grad[0]
                 and bool(hip.query('vqcpc_relattn16_b16_supported', L, H, hd) or hip.query('vqcpc_relattn_b16_supported', L, H, hd)))
        if sform1:
            ds1, dA, dg1, dbe1, dAb = ln_bwd(dx1, s1, d, None, g1, be1, mean1, rstd1, s[1], ds_bf16=g16_1)
        else:
            ds1, dA, dg1, dbe1, dAb = ln_bwd(dx1, xs, ldxs, s1, g1, be1, mean1, rstd1, s[1])
        b16_io = nat and f == 1 and qkv.dtype == torch.bfloat16       # the all-bf16 attention backward reads d ctx as bf16
        if nat:
            dwo, dbo = wgrad(dAb, attb, wo, bo)
            datt = gemm_nt_bf16(dAb, transpose(wo), out_f32=not b16_io, out_bf16=b16_io)
        else:
            dwo, dbo = wgrad(dA, att, wo, bo)
            datt = gemm_nt(dA, transpose(wo))
        de1 = torch.empty_like(e1)
        de2 = torch.empty_like(e2)
        need_dx = ctx.needs_input_grad[0]
        if f == 1:
            nbytes = hip.query('vqcpc_relattn_bwd_workspace', nblk, L, H, hd)
            ws = hip.workspace(nbytes, dev)
            tok = ctx.qkv_tokens
            if (nat and not ext_qkv and tok is None and ATT_B16_OUT
                    and hip.query('vqcpc_relattn16_b16_supported', L, H, hd)):
                # bf16 path: d qkv only feeds the two GEMMs below -> written as bf16 by the attention backward itself
                dqkvb = torch.empty(M, 3 * d, dtype=torch.bfloat16, device=dev)
                if b16_io:
                    hip.call('vqcpc_relattn16_bwd_b16io', datt, d, qkv, 3 * d, probs, e1, e2, dqkvb, 3 * d, de1, de2, nblk, H,
                             hd, p, s[0], ws, nbytes)
                else:
                    hip.call('vqcpc_relattn16_bwd_b16', datt, d, qkv, 3 * d, None, probs, e1, e2, dqkvb, 3 * d, de1, de2, nblk,
                             H, hd, p, s[0], ws, nbytes)
                dwqkv, dbqkv = wgrad(dqkvb, xb, wqkv, bqkv)
                dx = _dgrad_plus_residual_bf16(dqkvb, transpose(wqkv), ds1, carrier=ctx.dx_b16) if need_dx else None
                de1, de2, dg1, dbe1, dg2, dbe2 = accumulate_small((e1, e2, g1, be1, g2, be2), (de1, de2, dg1, dbe1, dg2, dbe2))
                return (dx, None, None, None, None, None, None, None, None, dwqkv, dbqkv, dwo, dbo, de1, de2, dw1, db1, dw2, db2, dg1,
                        dbe1, dg2, dbe2)
            if (nat and not ext_qkv and tok is None and ATT_B16_OUT and hip.query('vqcpc_relattn_b16_supported', L, H, hd)):
                dqkvb = torch.empty(M, 3 * d, dtype=torch.bfloat16, device=dev)        # the other block lengths (L = 4)
                hip.call('vqcpc_relattn_bwd_b16', datt, d, qkv, 3 * d, probs, e1, e2, dqkvb, 3 * d, de1, de2, nblk, L, H, hd, p,
                         s[0], ws, nbytes)
                dwqkv, dbqkv = wgrad(dqkvb, xb, wqkv, bqkv)
                dx = _dgrad_plus_residual_bf16(dqkvb, transpose(wqkv), ds1, carrier=ctx.dx_b16) if need_dx else None
                de1, de2, dg1, dbe1, dg2, dbe2 = accumulate_small((e1, e2, g1, be1, g2, be2), (de1, de2, dg1, dbe1, dg2, dbe2))
                return (dx, None, None, None, None, None, None, None, None, dwqkv, dbqkv, dwo, dbo, de1, de2, dw1, db1, dw2, db2, dg1,
                        dbe1, dg2, dbe2)
            # bf16 path, block table (first layer): d q | k | v only feeds the segment sum below -> the attention backward writes it as
            # bf16 (the rounding the in_proj's bf16 input-gradient operand would get anyway), the segment sum reads half the bytes
            tab_b16 = (nat and tok is not None and ATT_B16_OUT and BF16_TAB_GRAD and L == 16 and qkv.shape[0] // L <= 80
                       and bool(hip.query('vqcpc_relattn16_b16_supported', L, H, hd)))      # (80 tokens: the bf16 segment sum's LDS table)
            dqkv = torch.empty(M, 3 * d, dtype=torch.bfloat16 if tab_b16 else torch.float32, device=dev)
            if tok is not None:
                if tab_b16:
                    hip.call('vqcpc_relattn16_bwd_b16', datt, d, qkv, 3 * d, tok, probs, e1, e2, dqkv, 3 * d, de1, de2, nblk, H, hd, p,
                             s[0], ws, nbytes)
                else:
                    hip.call('vqcpc_relattn_tab_bwd', datt, d, qkv, 3 * d, tok, probs, e1, e2, dqkv, 3 * d, de1, de2, nblk, L, H, hd,
                             p, s[0], ws, nbytes)
                vmax = qkv.shape[0] // L                                  # table gradient = segment sum of d qkv
                d_in = torch.empty_like(qkv)
                nb2 = hip.query('vqcpc_block_table_segsum_workspace', M, L, vmax, 3 * d)
                ws2 = hip.workspace(nb2, dev)
                hip.call('vqcpc_block_table_segsum_b16' if tab_b16 else 'vqcpc_block_table_segsum', dqkv, tok, d_in, M, L, vmax,
                         3 * d, ws2, nb2)
            else:
                hip.call('vqcpc_relattn_bwd', datt, d, qkv, 3 * d, probs, e1, e2, dqkv, 3 * d, de1, de2, nblk, L, H, hd, p, s[0],
                         ws, nbytes)
                d_in = dqkv
            if ext_qkv:          # projection lives outside: its gradient leaves through qkv_in, x keeps the residual path
                de1, de2, dg1, dbe1, dg2, dbe2 = accumulate_small((e1, e2, g1, be1, g2, be2),
                                                                  (de1, de2, dg1, dbe1, dg2, dbe2))
                return (ds1 if need_dx else None, None, None, None, None, None, d_in, None, None, None, None, dwo, dbo, de1, de2, dw1,
                        db1, dw2, db2, dg1, dbe1, dg2, dbe2)
            if nat:
                dqkvb = cast_bf16(dqkv)
                dwqkv, dbqkv = wgrad(dqkvb, xb, wqkv, bqkv)
                dx = gemm_nt_bf16(dqkvb, transpose(wqkv), add=ds1) if need_dx else None
            else:
                dwqkv, dbqkv = wgrad(dqkv, x, wqkv, bqkv)
                dx = gemm_nt_residual(dqkv, transpose(wqkv), ds1, res_may_alias=dA) if need_dx else None
        else:
            nbytes = hip.query('vqcpc_relattn_sub_bwd_workspace', nblk, L, f, H, hd)
            ws = hip.workspace(nbytes, dev)
            sub_b16 = nat and ATT_B16_OUT and bool(hip.query('vqcpc_relattn_sub_b16_supported', L, f, H, hd))
            dq = torch.empty(Mq, d, dtype=torch.float32, device=dev)
            # bf16 path: d k | v (8 x the bytes of d q) only feeds GEMMs that read bf16 -> written so by the kernel
            dkv = torch.empty(M, 2 * d, dtype=torch.bfloat16 if sub_b16 else torch.float32, device=dev)
            hip.call('vqcpc_relattn_sub_bwd_b16' if sub_b16 else 'vqcpc_relattn_sub_bwd', datt, d, qproj, d, qkv, 2 * d, probs, e1,
                     e2, dq, d, dkv, 2 * d, de1, de2, nblk, L, f, H, hd, p, s[0], ws, nbytes)
            if nat:
                dkv = cast_bf16(dkv)                                   # a no-op on a bf16 tensor
                dwq, dbq = wgrad(cast_bf16(dq), xsb, wqkv, bqkv, rows=slice(0, d))
                dwkv, dbkv = wgrad(dkv, xb, wqkv, bqkv, rows=slice(d, 3 * d))
            else:
                dwq, dbq = wgrad(dq, xs, wqkv, bqkv, rows=slice(0, d))
                dwkv, dbkv = wgrad(dkv, x, wqkv, bqkv, rows=slice(d, 3 * d))
            if dwq is None:
                dwqkv = dbqkv = None                                       # accumulated in place into in_proj's gradient
            else:
                dwqkv, dbqkv = torch.cat([dwq, dwkv], dim=0), torch.cat([dbq, dbkv], dim=0)
            dx = None
            if need_dx:
                wt = transpose(wqkv)                                       # (d, 3d): columns q | k | v
                dx = lin(dkv, wt[:, d:])                                   # every row: keys / values path
                dxs = dx[::f]                                              # kept rows also get the query + residual paths
                gemm_nt(dq, wt[:, :d], add=ds1, add2=dxs, out=dxs)
        de1, de2, dg1, dbe1, dg2, dbe2 = accumulate_small((e1, e2, g1, be1, g2, be2), (de1, de2, dg1, dbe1, dg2, dbe2))
        return (dx, None, None, None, None, None, None, None, None, dwqkv, dbqkv, dwo, dbo, de1, de2, dw1, db1, dw2, db2, dg1, dbe1,
                dg2, dbe2)


# ------------------------------------------------------------------------------------------------------------------
# N4: building blocks of the decoder training step (decoders/decoder.py:431-543, transformer_custom.py:294-386)
# ------------------------------------------------------------------------------------------------------------------
MASK_NONE, MASK_CAUSAL, MASK_ANTICAUSAL = 0, 1, 2


class AttnXFn(torch.autograd.Function):
    """Masked, rectangular relative attention (vqcpc_relattn_x_fwd/bwd).  Self-attention: `qsrc` is the (n * L, 3d)
    in_proj output and `kvsrc` None; cross-attention: `qsrc` (n * Lq, d) projected queries, `kvsrc` (n * Lk, 2d) k | v of
    the memory.  Returns (ctx (n * Lq, d), probs (n, H, Lq, Lk)); the gradient comes back in the same packing."""

    @staticmethod
    def forward(ctx, qsrc, kvsrc, e1, e2, n, Lq, Lk, H, mask, drop_p, seed):
        hd = e1.shape[1]
        d = H * hd
        qsrc = _f32(qsrc).contiguous()
        if kvsrc is None:
            assert qsrc.shape == (n * Lq, 3 * d) and Lq == Lk
            q, k, v, ldq, ldk = qsrc, qsrc[:, d:], qsrc[:, 2 * d:], 3 * d, 3 * d
        else:
            kvsrc = _f32(kvsrc).contiguous()
            assert qsrc.shape == (n * Lq, d) and kvsrc.shape == (n * Lk, 2 * d)
            q, k, v, ldq, ldk = qsrc, kvsrc, kvsrc[:, d:], d, 2 * d
        att = torch.empty(n * Lq, d, dtype=torch.float32, device=qsrc.device)
        probs = torch.empty(n, H, Lq, Lk, dtype=torch.float32, device=qsrc.device)
        hip.call('vqcpc_relattn_x_fwd', q, ldq, k, ldk, v, ldk, e1, e2, att, d, probs, n, Lq, Lk, H, hd, int(mask),
                 float(drop_p), int(seed))
        ctx.save_for_backward(qsrc, kvsrc, probs, e1, e2)
        ctx.meta = (n, Lq, Lk, H, hd, int(mask), float(drop_p), int(seed))
        ctx.mark_non_differentiable(probs)
        ctx.set_materialize_grads(False)        # the attention maps are an unused output: no 150 MB zero gradient for them
        return att, probs

    @staticmethod
    def backward(ctx, datt, _dprobs):
        qsrc, kvsrc, probs, e1, e2 = ctx.saved_tensors
        n, Lq, Lk, H, hd, mask, p, seed = ctx.meta
        d = H * hd
        dev = qsrc.device
        datt = datt.contiguous() if datt is not None else torch.zeros(n * Lq, d, dtype=torch.float32, device=dev)
        de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
        if kvsrc is None:
            dqsrc, dkvsrc = torch.empty(n * Lq, 3 * d, dtype=torch.float32, device=dev), None
            q, k, v, ldq, ldk = qsrc, qsrc[:, d:], qsrc[:, 2 * d:], 3 * d, 3 * d
            dq, dk, dv = dqsrc, dqsrc[:, d:], dqsrc[:, 2 * d:]
        else:
            dqsrc = torch.empty(n * Lq, d, dtype=torch.float32, device=dev)
            dkvsrc = torch.empty(n * Lk, 2 * d, dtype=torch.float32, device=dev)
            q, k, v, ldq, ldk = qsrc, kvsrc, kvsrc[:, d:], d, 2 * d
            dq, dk, dv = dqsrc, dkvsrc, dkvsrc[:, d:]
        nbytes = hip.query('vqcpc_relattn_x_bwd_workspace', n, Lq, Lk, H, hd)
        ws = hip.workspace(nbytes, dev)
        hip.call('vqcpc_relattn_x_bwd', datt, d, q, ldq, k, ldk, v, ldk, probs, e1, e2, dq, ldq, dk, ldk, dv, ldk, de1, de2, n,
                 Lq, Lk, H, hd, mask, p, seed, ws, nbytes)
        return dqsrc, dkvsrc, de1, de2, None, None, None, None, None, None, None


class AddLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(r))  (transformer_custom.py:372-373,376-377,382-383)."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, drop_p, seed):
        x, ldx = _rows(_f32(x))
        r = _f32(r).contiguous()
        M, d = x.shape
        y = torch.empty(M, d, dtype=torch.float32, device=x.device)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        hip.call('vqcpc_add_layernorm_fwd', x, ldx, r, gamma, beta, y, mean, rstd, M, d, 1e-5, float(drop_p), int(seed))
        ctx.save_for_backward(x, r, gamma, mean, rstd)
        ctx.meta = (float(drop_p), int(seed))
        ctx.beta = beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, r, gamma, mean, rstd = ctx.saved_tensors
        p, seed = ctx.meta
        x, ldx = _rows(x)
        M, d = x.shape
        dev = x.device
        ds = torch.empty(M, d, dtype=torch.float32, device=dev)
        dr = torch.empty(M, d, dtype=torch.float32, device=dev) if p > 0 else None
        nbytes = hip.query('vqcpc_add_layernorm_bwd_workspace', M, d)
        ws = hip.workspace(nbytes, dev)
        if defer_ln_param_grads(ws, M, d, gamma, ctx.beta, r is not None):         # trainers: summed with every other LayerNorm's partials later
            dg = db = None
        else:
            dg = torch.empty(d, dtype=torch.float32, device=dev)
            db = torch.empty(d, dtype=torch.float32, device=dev)
        hip.call('vqcpc_add_layernorm_bwd', dy.contiguous(), x, ldx, r, gamma, mean, rstd, ds, dr, dg, db, M, d, p, seed, ws,
                 nbytes)
        if dg is not None:
            dg, db = accumulate_small([gamma, ctx.beta], [dg, db])   # one launch into the live gradient buffers
        return ds, (dr if dr is not None else ds), dg, db, None, None


class FFNFn(torch.autograd.Function):
    """linear2(dropout(relu(linear1(x))))  (transformer_custom.py:379): ReLU + dropout in the first GEMM's epilogue, their
    backward as the gate of the dgrad GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, drop_p, seed):
        if gatebits_worthwhile(x.shape[0], w1.shape[0], w1.shape[1]) and x.dim() == 2:
            h, ctx.gate_mask = gemm_nt_relu_mask(x, w1, b1, drop_p=float(drop_p), seed=int(seed))
        else:
            h, ctx.gate_mask = gemm_nt(x, w1, bias=b1, act=1, drop_p=float(drop_p), seed=int(seed)), None
        y = gemm_nt(h, w2, bias=b2)
        ctx.save_for_backward(x, h, w1, w2)
        ctx.biases = (b1, b2)
        ctx.p = float(drop_p)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, w1, w2 = ctx.saved_tensors
        b1, b2 = ctx.biases
        dy = dy.contiguous()
        if ctx.gate_mask is not None:
            da = gemm_nt_gatebits(dy, transpose(w2), ctx.gate_mask, gate_scale=1.0 / (1.0 - ctx.p))
        else:
            da = gemm_nt(dy, transpose(w2), gate=h, gate_scale=1.0 / (1.0 - ctx.p))
        dw2, db2 = wgrad(dy, h, w2, b2)
        dw1, db1 = wgrad(da, x, w1, b1)
        dx = gemm_nt(da, transpose(w1)) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2, None, None


class CrossProjFn(torch.autograd.Function):
    """Encoder-decoder branch of in_proj (multihead_attention_custom.py:173-196): q = tgt W[:d]^T + b[:d],
    k | v = mem W[d:]^T + b[d:].  One autograd node so that the two row blocks of in_proj's gradient are produced
    together (or accumulated in place, ops.direct_weight_gradients)."""

    @staticmethod
    def forward(ctx, tgt, mem, w, b):
        d = w.shape[1]
        q = gemm_nt(tgt, w[:d], bias=b[:d])
        kv = gemm_nt(mem, w[d:], bias=b[d:])
        ctx.save_for_backward(tgt, mem, w)
        ctx.b = b
        return q, kv

    @staticmethod
    def backward(ctx, dq, dkv):
        tgt, mem, w = ctx.saved_tensors
        b = ctx.b
        d = w.shape[1]
        dq, dkv = dq.contiguous(), dkv.contiguous()
        dwq, dbq = wgrad(dq, tgt, w, b, rows=slice(0, d))
        dwkv, dbkv = wgrad(dkv, mem, w, b, rows=slice(d, 3 * d))
        if dwq is None:
            dw = db = None
        else:
            dw, db = torch.cat([dwq, dwkv], dim=0), torch.cat([dbq, dbkv], dim=0)
        wt = transpose(w)
        dtgt = gemm_nt(dq, wt[:, :d]) if ctx.needs_input_grad[0] else None
        dmem = gemm_nt(dkv, wt[:, d:]) if ctx.needs_input_grad[1] else None
        return dtgt, dmem, dw, db


class EmbeddingFn(torch.autograd.Function):
    """out[m] = table[idx[m]] for a table of any size; backward = deterministic segment sum over a stable sort of the
    indices (integer plumbing only) in vqcpc_embedding_bwd."""

    @staticmethod
    def forward(ctx, table, idx):
        table = _f32(table).contiguous()
        V, C = table.shape
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        M = idx.numel()
        out = torch.empty(M, C, dtype=torch.float32, device=table.device)
        hip.call('vqcpc_block_table_gather', table, idx, out, M, 1, V, C)
        ctx.save_for_backward(idx)
        ctx.meta = (V, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        V, C = ctx.meta
        g, ldg = _rows(_f32(g))
        sorted_idx, perm = torch.sort(idx, stable=True)
        d_table = torch.empty(V, C, dtype=torch.float32, device=g.device)
        hip.call('vqcpc_embedding_bwd', g, ldg, sorted_idx, perm, d_table, idx.numel(), V, C)
        return d_table, None


# ------------------------------------------------------------------------------------------------------------------
# A9/A10: product vector quantiser
# ------------------------------------------------------------------------------------------------------------------
class VQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, codebooks, beta, squared, given_idx=None):
        # z (R, D) contiguous ; codebooks (ncb, K, dsub) ; given_idx (R, ncb) int64 forces the assignment
        z = _f32(z).contiguous()
        codebooks = _f32(codebooks).contiguous()
        R, D = z.shape
        ncb, K, dsub = codebooks.shape
        assert ncb * dsub == D
        idx = (torch.empty(R, ncb, dtype=torch.int64, device=z.device) if given_idx is None
               else given_idx.to(torch.int64).contiguous().clone())
        zq = torch.empty_like(z)
        loss = torch.empty(R, dtype=torch.float32, device=z.device)
        hip.call('vqcpc_vq_fwd', z, codebooks, R, ncb, K, dsub, float(beta), int(bool(squared)), int(given_idx is None), idx,
                 zq, loss)
        ctx.save_for_backward(z, codebooks, idx)
        ctx.meta = (float(beta), int(bool(squared)))
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return zq, idx, loss

    @staticmethod
    def backward(ctx, g_zq, _g_idx, g_loss):
        z, codebooks, idx = ctx.saved_tensors
        beta, squared = ctx.meta
        R, D = z.shape
        ncb, K, dsub = codebooks.shape
        g_zq = g_zq.contiguous() if g_zq is not None else torch.zeros_like(z)
        g_loss = g_loss.contiguous() if g_loss is not None else torch.zeros(R, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        dcb = torch.empty_like(codebooks)
        nbytes = hip.query('vqcpc_vq_bwd_workspace', R, ncb, K, dsub)
        ws = hip.workspace(nbytes, z.device)
        hip.call('vqcpc_vq_bwd', z, codebooks, idx, g_zq, g_loss, R, ncb, K, dsub, beta, squared, dz, dcb, ws, nbytes)
        return dz, dcb, None, None, None


def vq_assign(z, codebooks):
    """Index-only product-VQ assignment (no straight-through output, no loss): z (R, D) -> (R, ncb) int64."""
    z = _f32(z).contiguous()
    codebooks = _f32(codebooks).contiguous()
    R, D = z.shape
    ncb, K, dsub = codebooks.shape
    assert ncb * dsub == D
    idx = torch.empty(R, ncb, dtype=torch.int64, device=z.device)
    hip.call('vqcpc_vq_fwd', z, codebooks, R, ncb, K, dsub, 0.0, 1, 1, idx, None, None)
    return idx


class DropoutSeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, p, seed):
        h = _f32(h).contiguous()
        out = torch.empty_like(h)
        hip.call('vqcpc_dropout_selu_fwd', h, out, h.numel(), float(p), int(seed))
        ctx.save_for_backward(h)
        ctx.meta = (float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        p, seed = ctx.meta
        gh = torch.empty_like(h)
        hip.call('vqcpc_dropout_selu_bwd', h, g.contiguous(), gh, h.numel(), p, seed)
        return gh, None, None


# ------------------------------------------------------------------------------------------------------------------
# A15: GRU layer of the context network (time-major rows: row = t * B + b)
# ------------------------------------------------------------------------------------------------------------------
ATT_B16_IN = os.environ.get('VQCPC_ATT_B16_IN', '1') != '0'          # A/B switch: bf16 q | k | v and d ctx INTO the L = 16 attention kernels
BF16_RESIDUAL = os.environ.get('VQCPC_BF16_RESIDUAL', '1') != '0'      # A/B switch: LN1's output in bf16 only on the bf16 path
BF16_GRAD_SUMS = os.environ.get('VQCPC_BF16_GRAD_SUMS', '1') != '0'    # A/B switch: ... and the gradients of the residual branches (LayerNorm backward -> dgrad epilogue)
BF16_SUMS = os.environ.get('VQCPC_BF16_SUMS', '1') != '0'              # A/B switch: ... and the residual sums s1 / s2 (LayerNorm inputs)
BF16_GRAD_STREAM = os.environ.get('VQCPC_BF16_GRAD_STREAM', '1') != '0'  # A/B switch: ... and the main-stream gradient between sub-layers / layers (LayerNorm backward reads bf16 dy)
BF16_TAB_GRAD = os.environ.get('VQCPC_BF16_TAB_GRAD', '0') != '0'        # opt-in (measured equal, profiles/r05_perf_log.md): the first layer's d q | k | v into the block-table segment sum in bf16
BF16_ACT_STREAM = os.environ.get('VQCPC_BF16_ACT_STREAM', '1') != '0'    # A/B switch: ... and the output of a stack's interior layers (LN2 writes bf16 only, the next out-proj epilogue reads it)
ATT_B16_OUT = os.environ.get('VQCPC_ATT_B16_OUT', '1') != '0'        # A/B switch: bf16 outputs straight from the L = 16 attention
GRU_FUSED_STEPS = os.environ.get('VQCPC_GRU_FUSED', '1') != '0'      # A/B switch: one launch per step (csrc/gru.hip)


class GRULayerFn(torch.autograd.Function):
    """One nn.GRU layer with h0 = 0.  x (T*B, in) time-major -> y: all steps (T*B, H) with dropout(p) applied
    (last_only = False, feeds the next layer) or the last step only (B, H) (last_only = True, top layer).
    Input projections of all steps are ONE GEMM; every step is a (B, H) x (H, 3H) GEMM + the gate kernel.  Backward
    walks the steps in reverse (gate kernel + one GEMM with the direct d h_prev term in its `add` epilogue) and ends
    with two TN GEMMs over all steps for the weight / bias gradients."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, T, drop_p, seed, last_only):
        x = _f32(x).contiguous()
        TB, _ = x.shape
        B, H = TB // T, w_hh.shape[1]
        dev = x.device
        p = 0.0 if last_only else float(drop_p)
        gi = gemm_nt(x, w_ih, bias=b_ih)                                     # (T*B, 3H)
        gh = torch.empty(TB, 3 * H, dtype=torch.float32, device=dev)
        h = torch.empty(TB + B, H, dtype=torch.float32, device=dev)          # rows [t*B, (t+1)*B) = h_{t-1}; h_{-1} = 0
        h[:B].zero_()
        y = None if last_only else torch.empty(TB, H, dtype=torch.float32, device=dev)
        fused = GRU_FUSED_STEPS and bool(hip.query('vqcpc_gru_step_supported', B, H)) and w_hh.is_contiguous()
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            if fused:      # ONE launch per step: recurrent product + gates (h_{-1} = 0: no product at t = 0)
                hip.call('vqcpc_gru_step_fwd', gi[sl], w_hh, b_hh, h[sl] if t > 0 else None, gh[sl],
                         h[(t + 1) * B:(t + 2) * B], None if y is None else y[sl], B, H, p, int(seed), t * B * H)
                continue
            gemm_nt(h[sl], w_hh, bias=b_hh, out=gh[sl])
            hip.call('vqcpc_gru_cell_fwd', gi[sl], gh[sl], h[sl], h[(t + 1) * B:(t + 2) * B],
                     None if y is None else y[sl], B, H, p, int(seed), t * B * H)
        ctx.save_for_backward(x, w_ih, w_hh, gi, gh, h)
        ctx.biases = (b_ih, b_hh)
        ctx.meta = (T, B, H, p, int(seed), bool(last_only))
        return h[T * B:].clone() if last_only else y

    @staticmethod
    def backward(ctx, g):
        x, w_ih, w_hh, gi, gh, h = ctx.saved_tensors
        T, B, H, p, seed, last_only = ctx.meta
        dev = x.device
        g = g.contiguous()
        dgi = torch.empty_like(gi)
        dgh = torch.empty_like(gh)
        whh_t = transpose(w_hh)                                              # (H, 3H): dgrad operand
        dh = None
        dhp = torch.empty(B, H, dtype=torch.float32, device=dev)
        fused = GRU_FUSED_STEPS and bool(hip.query('vqcpc_gru_step_supported', B, H))
        for t in range(T - 1, -1, -1):
            sl = slice(t * B, (t + 1) * B)
            d_y = (g if t == T - 1 else None) if last_only else g[sl]
            if fused and t < T - 1:
                # ONE launch per step: d h_t = dgh_{t+1} W_hh + dh_{t+1} * u_{t+1} (in dhp), then the cell backward of step t
                nx = slice((t + 1) * B, (t + 2) * B)
                hip.call('vqcpc_gru_step_bwd', dgh[nx], whh_t, dhp, gi[sl], gh[sl], h[sl] if t > 0 else None, d_y, dgi[sl],
                         dgh[sl], B, H, p, seed, t * B * H)
                continue
            hip.call('vqcpc_gru_cell_bwd', gi[sl], gh[sl], h[sl], d_y, dh, dgi[sl], dgh[sl], dhp, B, H, p, seed, t * B * H)
            if t > 0 and not fused:
                dh = gemm_nt(dgh[sl], whh_t, add=dhp)                        # d h_{t-1} = dgh W_hh + dh * u
        b_ih, b_hh = ctx.biases
        dw_hh, db_hh = wgrad(dgh, h[:T * B], w_hh, b_hh)                     # step 0 multiplies the zero rows h_{-1}
        dw_ih, db_ih = wgrad(dgi, x, w_ih, b_ih)
        dx = gemm_nt(dgi, transpose(w_ih)) if ctx.needs_input_grad[0] else None
        return dx, dw_ih, dw_hh, db_ih, db_hh, None, None, None, None


# ------------------------------------------------------------------------------------------------------------------
# A16/A17: bilinear scores + InfoNCE + hits
# ------------------------------------------------------------------------------------------------------------------
class NCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, W, z_pos, z_neg):
        c, W, z_pos, z_neg = (_f32(t).contiguous() for t in (c, W, z_pos, z_neg))
        B, cdim = c.shape
        zdim, cdim2, K = W.shape
        N = z_neg.shape[1]
        assert cdim2 == cdim and z_pos.shape == (B, K, zdim) and z_neg.shape == (B, N, K, zdim)
        dev = c.device
        f_pos = torch.empty(B, K, dtype=torch.float32, device=dev)
        f_neg = torch.empty(B, K, N, dtype=torch.float32, device=dev)
        loss_b = torch.empty(B, dtype=torch.float32, device=dev)
        hits = torch.empty(B, K, dtype=torch.float32, device=dev)
        hip.call('vqcpc_nce_fwd', c, W, z_pos, z_neg, B, K, N, zdim, cdim, f_pos, f_neg, loss_b, hits)
        ctx.save_for_backward(c, W, z_pos, z_neg, f_pos, f_neg)
        ctx.mark_non_differentiable(hits, f_pos, f_neg)
        ctx.set_materialize_grads(False)
        return loss_b, hits, f_pos, f_neg

    @staticmethod
    def backward(ctx, g_loss_b, _gh, _gp, _gn):
        c, W, z_pos, z_neg, f_pos, f_neg = ctx.saved_tensors
        if g_loss_b is None:
            g_loss_b = torch.zeros(c.shape[0], dtype=torch.float32, device=c.device)
        B, cdim = c.shape
        zdim, _, K = W.shape
        N = z_neg.shape[1]
        d_c, d_W, d_zp, d_zn = (torch.empty_like(t) for t in (c, W, z_pos, z_neg))
        nbytes = hip.query('vqcpc_nce_bwd_workspace', B, K, N, zdim, cdim)
        ws = hip.workspace(nbytes, c.device)
        hip.call('vqcpc_nce_bwd', c, W, z_pos, z_neg, f_pos, f_neg, g_loss_b.contiguous(), B, K, N, zdim, cdim, d_c, d_W,
                 d_zp, d_zn, ws, nbytes)
        return d_c, d_W, d_zp, d_zn


# ------------------------------------------------------------------------------------------------------------------
# A23 (student step): cross-entropy rows and the auxiliary decoder's upscale
# ------------------------------------------------------------------------------------------------------------------
def same_sequence_negatives(first, second, ticks_per_block=4):
    """first (B, Ta, V), second (B, Tb, V) int64 device tokens -> (B, Ka + Kb - 1, Kb, ticks_per_block, V): for every
    target block k of `second`, all blocks of `first` then the blocks of `second` except k."""
    assert first.dtype == torch.int64 and second.dtype == torch.int64 and first.is_cuda and second.is_cuda
    first, second = first.contiguous(), second.contiguous()
    B, Ta, V = first.shape
    Tb = second.shape[1]
    assert Ta % ticks_per_block == 0 and Tb % ticks_per_block == 0 and second.shape[0] == B and second.shape[2] == V
    Ka, Kb = Ta // ticks_per_block, Tb // ticks_per_block
    out = torch.empty(B, Ka + Kb - 1, Kb, ticks_per_block, V, dtype=torch.int64, device=first.device)
    hip.call('vqcpc_same_sequence_negatives', first, second, out, B, Ka, Kb, ticks_per_block * V)
    return out


class SoftmaxCEFn(torch.autograd.Function):
    """Rows of softmax cross-entropy against hard targets (int64 (R,)) or soft targets given as logits ((R, V), no
    gradient flows into them: the reference detaches the teacher, student_encoder_trainer.py:197-198)."""

    @staticmethod
    def forward(ctx, logits, target, target_logits):
        logits, ld = _rows(_f32(logits))
        R, V = logits.shape
        loss = torch.empty(R, dtype=torch.float32, device=logits.device)
        grad = torch.empty(R, V, dtype=torch.float32, device=logits.device)
        ldt = 0
        if target_logits is not None:
            target_logits, ldt = _rows(_f32(target_logits))
            assert target is None and target_logits.shape == (R, V)
        else:
            target = target.to(torch.int64).contiguous()
            assert target.shape == (R,)
        hip.call('vqcpc_softmax_ce', logits, ld, target, target_logits, ldt, loss, grad, R, V)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad)
        hip.call('vqcpc_scale_rows', grad, g.contiguous(), out, grad.shape[0], grad.shape[1])
        return out, None, None


class UpscaleFn(torch.autograd.Function):
    """(rows, d) -> (rows * f, d): row r becomes f consecutive rows r*f + u = x[r] + emb[u]."""

    @staticmethod
    def forward(ctx, x, emb):
        x, emb = _f32(x).contiguous(), _f32(emb).contiguous()
        rows, d = x.shape
        f = emb.shape[0]
        out = torch.empty(rows * f, d, dtype=torch.float32, device=x.device)
        hip.call('vqcpc_upscale_fwd', x, emb, out, rows, f, d)
        ctx.meta = (rows, f, d)
        return out

    @staticmethod
    def backward(ctx, g):
        rows, f, d = ctx.meta
        g = g.contiguous()
        dx = torch.empty(rows, d, dtype=torch.float32, device=g.device)
        demb = torch.empty(f, d, dtype=torch.float32, device=g.device)
        nbytes = hip.query('vqcpc_upscale_bwd_workspace', rows, f, d)
        ws = hip.workspace(nbytes, g.device)
        hip.call('vqcpc_upscale_bwd', g, dx, demb, rows, f, d, ws, nbytes)
        return dx, demb


# ------------------------------------------------------------------------------------------------------------------
# flat-buffer optimiser
# ------------------------------------------------------------------------------------------------------------------
class FlatAdam:
    """clip_grad_norm_(., max_norm) + Adam on one flat fp32 buffer (params / grads / m / v), no host sync."""

    def __init__(self, flat_param, flat_grad, lr, betas=(0.9, 0.999), eps=1e-8, max_norm=5.0):
        self.p, self.g = flat_param, flat_grad
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.step_count = 0
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=flat_param.device)
        self._ws_bytes = hip.query('vqcpc_sumsq_workspace', flat_param.numel())
        self._ws = hip.workspace(self._ws_bytes, flat_param.device)
        self._dev = None            # (lr_dev, step_dev) while a StepGraph owns the per-step scalars (graphs.py)

    def use_device_scalars(self, lr_dev, step_dev):
        self._dev = (lr_dev, step_dev) if lr_dev is not None else None

    def step(self, lr=None, grad_scale=1.0):
        global _PARAM_STEPS
        _PARAM_STEPS += 1
        self.step_count += 1
        n = self.p.numel()
        hip.call('vqcpc_sumsq', self.g, n, float(grad_scale), self.sumsq, self._ws, self._ws_bytes)
        if self._dev is not None and torch.cuda.is_current_stream_capturing():
            # being recorded into a step graph: learning rate and step count are read from device memory at replay time
            hip.call('vqcpc_adam_step_dev', self.p, self.g, self.m, self.v, n, self._dev[0], float(self.betas[0]),
                     float(self.betas[1]), float(self.eps), self._dev[1], float(grad_scale), float(self.max_norm), self.sumsq)
            return
        hip.call('vqcpc_adam_step', self.p, self.g, self.m, self.v, n, float(self.lr if lr is None else lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_count, float(grad_scale),
                 float(self.max_norm), self.sumsq)

    def grad_norm(self):
        return float(self.sumsq.sqrt().item())
