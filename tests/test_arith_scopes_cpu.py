"""Host logic of the f16x3 arithmetic scopes (vqcpc_bach_amd/ops.py: GradScales, forward_arithmetic, the dispatch inside gemm_nt /
gemm_nt_relu_mask, the training defaults) with the kernel library replaced by a recorder: which entry point a product takes, in
which order the scale sites are used, when they are primed and rolled.  No GPU, no library."""
import pytest
import torch


def Z(*shape):
    return torch.empty(*shape, device='meta')            # shapes and strides only: nothing is computed in these tests


class _Owner:
    def __init__(self):
        self.flat = torch.zeros(4)


@pytest.fixture
def rec(monkeypatch):
    from vqcpc_bach_amd import hip, ops
    calls = []
    monkeypatch.setattr(hip, 'call', lambda name, *args: calls.append((name, args)) or 0)
    monkeypatch.setattr(hip, 'query', lambda name, *args: 1)
    monkeypatch.setattr(hip, 'get_gemm_mode', lambda: 1)
    monkeypatch.setattr(hip, 'gradient_scope', lambda on: None, raising=False)
    monkeypatch.setattr(hip, 'set_gradient_products', lambda n: None, raising=False)
    monkeypatch.setattr(hip, 'get_gradient_products', lambda: 6, raising=False)
    monkeypatch.setattr(ops, '_grad_cut', {})
    monkeypatch.setattr(ops, '_g3_plans', {})
    monkeypatch.setattr(ops, 'GRAD_SPLITK', True)       # the opt-in split-K remainder of ragged launches is part of what is tested
    monkeypatch.setattr(ops, 'GRAD_TAIL', False)        # ... and the tail-row launch has its own test below
    monkeypatch.setattr(hip, 'workspace', lambda n, dev: torch.empty(16, dtype=torch.uint8), raising=False)
    monkeypatch.setattr(ops, '_f32', lambda t: t)        # the device check of the product path: operands here are `meta` tensors
    state = ops.gradient_arithmetic_state()
    yield calls
    ops.restore_gradient_arithmetic_state(state)


def _names(calls):
    return [c[0] for c in calls]


def test_forward_scope_routes_whole_round_products_to_the_three_product_kernel(rec):
    from vqcpc_bach_amd import ops
    ops.set_forward_arithmetic('f16x3')
    owner = _Owner()
    M, N, K = 256 * 256, 256, 1024                       # 256 tiles: one whole round
    a, w, bias, res = Z(M, K), Z(N, K), Z(N), Z(M, N)
    ragged = Z(256 * 544, K)                             # 544 tiles = 2.125 rounds: two whole rounds + a split-K remainder (32 tiles x 8)
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        ops.gemm_nt(a, w, bias=bias)
        ops.gemm_nt(a, w, bias=bias, drop_p=0.1, seed=3, add=res)
        ops.gemm_nt(a, w)                                # no epilogue: the gradient entry point serves it
        ops.gemm_nt(a, w, bias=bias, drop_p=0.1, seed=3) # dropout without a residual is not a forward form of the kernel
        ops.gemm_nt(ragged, w, bias=bias)
        ops.gemm_nt(Z(256 * 544, 256), Z(N, 256), bias=bias)     # ragged with a short K: no slices to cut -> six products
        ops.gemm_nt_relu_mask(a, Z(1024, K), Z(1024), drop_p=0.1, seed=5)
    names = _names(rec)
    assert names.count('vqcpc_grad_amax') == 10          # five sites, two operands each, primed on first use
    kernels = [n for n in names if n.startswith('vqcpc_gemm') and not n.endswith('_workspace')]
    assert kernels[:4] == ['vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_grad', 'vqcpc_gemm_nt'], kernels
    assert kernels[4:6] == ['vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_grad_splitk'], kernels
    main = [c for c in rec if c[0] == 'vqcpc_gemm_nt_f16x3'][2][1]
    rem = [c for c in rec if c[0] == 'vqcpc_gemm_nt_grad_splitk'][0][1]
    assert main[6] == 512 * 256 and rem[6] == 32 * 256 and rem[9] == 8 and rem[13] == 512 * 256      # rows, rows, slices, row0
    assert kernels[-1] == 'vqcpc_gemm_nt_f16x3' and kernels.count('vqcpc_gemm_nt_f16x3') == 4
    assert all(k in ('vqcpc_gemm_nt', 'vqcpc_gemm_nt_splitk') for k in kernels[6:-1]), kernels     # the short-K ragged launch: six products
    assert names[-1] == 'vqcpc_grad_scale_roll_logged'   # rolled when the scope closes
    tab = owner._grad_scales[('fwd', None)]
    assert tab.keys == [('fnt', M, N, K), ('fnt', M, N, K), ('fnt', M, N, K), ('fnt', 256 * 544, N, K), ('fntm', M, 1024, K)]
    # the same step again: same sites in the same order, nothing primed
    del rec[:]
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        ops.gemm_nt(a, w, bias=bias)
        st_first = rec[-1][1][-1]
    assert 'vqcpc_grad_amax' not in _names(rec)
    assert st_first.data_ptr() == tab.state.data_ptr() and st_first.numel() == 4
    # another product at a site: primed again, the stale state cleared
    del rec[:]
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        ops.gemm_nt(Z(M, 512), Z(N, 512), bias=bias)
    assert _names(rec).count('vqcpc_grad_amax') == 2 and tab.keys[0] == ('fnt', M, N, 512)


def test_ragged_launches_take_whole_rounds_plus_the_tail_rows_on_small_tiles(rec, monkeypatch):
    """The default plan of a launch whose 256-tiles leave at most a quarter of a round over (139 264 x 256 x K = 2.125 rounds): two
    whole rounds on the 256-tile kernel + the last 8 192 rows on vqcpc_gemm_nt_grad_tail, in the forward scope (bias, dropout element
    index continued at row0) and in the gradient scope (residual forms, in place); a fuller last round stays where it was."""
    from vqcpc_bach_amd import ops
    monkeypatch.setattr(ops, 'GRAD_SPLITK', False)
    monkeypatch.setattr(ops, 'GRAD_TAIL', True)
    ops.set_forward_arithmetic('f16x3')
    ops.set_gradient_arithmetic('f16x3')
    owner = _Owner()
    M, N = 256 * 544, 256
    assert ops._g3_plan(M, N, 1024) == (512 * 256, -1) and ops._g3_plan(M, N, 256) == (512 * 256, -1)
    assert ops._g3_plan(256 * 384, N, 512) is None       # 1.5 rounds: half a round left, too much for the small tiles -> six products
    assert ops._g3_plan(256 * 512, N, 512) == (256 * 512, 0)
    bias, res = Z(N), Z(M, N)
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        ops.gemm_nt(Z(M, 1024), Z(N, 1024), bias=bias, drop_p=0.1, seed=3, add=res)
        ops.gemm_nt(Z(M, 256), Z(N, 256), bias=bias)
    kernels = [n for n in _names(rec) if n.startswith('vqcpc_gemm')]
    assert kernels == ['vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_grad_tail'] * 2, kernels
    main, tail = rec[[c[0] for c in rec].index('vqcpc_gemm_nt_f16x3')][1], rec[[c[0] for c in rec].index('vqcpc_gemm_nt_grad_tail')][1]
    assert main[6] == 512 * 256 and tail[6] == 32 * 256 and tail[12] == 512 * 256 and abs(tail[10] - 0.1) < 1e-7 and tail[11] == 3
    assert tail[9] is bias and tail[-1] is main[-1]      # one scale site for both launches
    del rec[:]
    with ops.direct_weight_gradients(owner):
        ops.gemm_nt(Z(M, 768), Z(N, 768), add=res)
        acc = torch.empty(M, N, device='meta')
        out = ops.gemm_nt_residual(Z(M, 1024), Z(N, 1024), acc)
    kernels = [n for n in _names(rec) if n.startswith('vqcpc_gemm')]
    assert kernels == ['vqcpc_gemm_nt_grad', 'vqcpc_gemm_nt_grad_tail'] * 2, kernels
    assert out is acc or out.data_ptr() == acc.data_ptr()


def test_under_filled_and_masked_launches_on_the_three_product_kernel(rec, monkeypatch):
    """One under-filled round (34 816 x 256: 136 tiles) runs on the 256-tile f16x3 kernel, fewer than GRAD_ONE_ROUND_MIN_TILES tiles do
    not; the relu-mask / gate-bit forms (no tail-row launch) take three rounds for 2.125 (fill 0.71 >= GRAD_ROUND_FILL_MASKED) inside
    the scopes and stay on their six-product entry points outside."""
    from vqcpc_bach_amd import ops
    monkeypatch.setattr(ops, '_masked_ok_cache', {})
    assert ops._grad_rows(34816, 256, 1024) == 34816 and ops._grad_rows(256 * 100, 256, 1024) == 0
    assert ops._masked_rows_ok(34816, 1024, 256) and ops._masked_rows_ok(34816, 256, 256) and not ops._masked_rows_ok(256 * 300, 256, 256)
    ops.set_forward_arithmetic('f16x3')
    ops.set_gradient_arithmetic('f16x3')
    owner = _Owner()
    a, w1, b1 = Z(34816, 256), Z(1024, 256), Z(1024)
    ops.gemm_nt_relu_mask(a, w1, b1, drop_p=0.1, seed=2)
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        ops.gemm_nt_relu_mask(a, w1, b1, drop_p=0.1, seed=2)
    with ops.direct_weight_gradients(owner):
        ops.gemm_nt_gatebits(a, w1, Z(34816 * 32), gate_scale=1.1)
        ops.gemm_nt(Z(34816, 1024), Z(256, 1024))
    kernels = [n for n in _names(rec) if n.startswith('vqcpc_gemm')]
    assert kernels == ['vqcpc_gemm_nt_relu_mask', 'vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_grad', 'vqcpc_gemm_nt_grad'], kernels


def test_forward_scope_is_inert_outside_training_and_by_default(rec):
    from vqcpc_bach_amd import ops
    owner = _Owner()
    a, w, bias = Z(256 * 256, 256), Z(256, 256), Z(256)
    assert ops.FWD_ARITH == 'six'                        # bare library default
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        ops.gemm_nt(a, w, bias=bias)
    ops.set_forward_arithmetic('f16x3')
    with torch.no_grad(), ops.forward_arithmetic(owner):        # evaluation / inference: never
        ops.gemm_nt(a, w, bias=bias)
    ops.gemm_nt(a, w, bias=bias)                         # outside any scope
    assert _names(rec) == ['vqcpc_gemm_nt'] * 3
    with pytest.raises(RuntimeError):                    # an exception inside the scope: closed, not rolled
        with torch.enable_grad(), ops.forward_arithmetic(owner):
            ops.gemm_nt(a, w, bias=bias)
            raise RuntimeError('step failed')
    assert ops._FWD_SCALES is None and 'vqcpc_grad_scale_roll_logged' not in _names(rec)
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        with ops.forward_arithmetic(owner):              # nested: the outer scope owns the table
            ops.gemm_nt(a, w, bias=bias)
        assert ops._FWD_SCALES is not None
    assert _names(rec).count('vqcpc_grad_scale_roll_logged') == 1


def test_training_defaults_select_both_arithmetics_unless_the_caller_chose(rec):
    from vqcpc_bach_amd import ops
    ops.restore_gradient_arithmetic_state(('six', False, 'six', False))
    ops.use_training_default_gradient_arithmetic()
    assert (ops.GRAD_ARITH, ops.FWD_ARITH) == (ops.TRAINING_GRAD_ARITH, ops.TRAINING_FWD_ARITH) == ('f16x3', 'f16x3')
    ops.restore_gradient_arithmetic_state(('six', False, 'six', False))
    ops.set_forward_arithmetic('six')                    # an explicit choice wins over train_model()'s default
    ops.use_training_default_gradient_arithmetic()
    assert (ops.GRAD_ARITH, ops.FWD_ARITH) == ('f16x3', 'six')
    ops.restore_gradient_arithmetic_state(('six', False, 'six', False))
    ops.set_gradient_arithmetic('six')
    ops.use_training_default_gradient_arithmetic()
    assert (ops.GRAD_ARITH, ops.FWD_ARITH) == ('six', 'f16x3')


def test_weight_planes_are_made_once_per_step_and_found_by_address(rec):
    """Round 6 (ops._WeightTransposes.refresh_planes / lookup_planes): the fp16 planes of the registered weights are made when the
    forward scope opens, the gradient scope of the SAME step uses them as they are, a gradient scope after an optimiser step (or
    without a forward scope) makes them again; a weight -- or a row block of it, or a column block of its transpose in the arena --
    is found by its address, anything else is not."""
    from vqcpc_bach_amd import ops
    ops.set_forward_arithmetic('f16x3')
    ops.set_gradient_arithmetic('f16x3')

    class Owner:
        pass
    owner = Owner()
    owner.flat = torch.zeros(2 * 768 * 256 + 64)
    w = owner.flat[64:64 + 768 * 256].view(768, 256)            # in_proj-like weight at a 16-byte aligned offset
    inst = ops.WEIGHT_T.instance(owner)
    inst.entries[w.data_ptr()] = (64, 768, 256)                 # what ops.transpose(w) registers during a backward pass
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        assert _names(rec).count('vqcpc_weight_planes_many') == 1
        hit = ops._PLANES.lookup_planes(w, 256)
        assert hit is not None and hit[0].data_ptr() - inst.planes.data_ptr() == 64 * 4 and hit[1].data_ptr() == inst.amax.data_ptr()
        rows = ops._PLANES.lookup_planes(w[256:], 256)          # k | v rows of in_proj
        assert rows is not None and rows[0].data_ptr() - inst.planes.data_ptr() == (64 + 256 * 256) * 4
        assert ops._PLANES.lookup_planes(torch.zeros(768, 256), 256) is None            # not a registered weight
        assert ops._PLANES.lookup_planes(owner.flat[:64 * 4].view(16, 16), 16) is None  # inside the buffer, before the weight
    assert ops._PLANES is None
    with ops.direct_weight_gradients(owner):
        assert _names(rec).count('vqcpc_weight_planes_many') == 1          # the forward scope's planes, as they are
        wt = ops.transpose(w)                                              # (256, 768) view of the arena
        assert wt.data_ptr() - inst.arena.data_ptr() == 64 * 4
        hit = ops._PLANES.lookup_planes(wt, 768)
        assert hit is not None and hit[0].data_ptr() - inst.planes_t.data_ptr() == 64 * 4
        cols = ops._PLANES.lookup_planes(wt[:, 256:], 768)                 # k | v columns of W^T (row stride 768)
        assert cols is not None and cols[0].data_ptr() - inst.planes_t.data_ptr() == (64 + 256) * 4
    with ops.direct_weight_gradients(owner):                               # no forward scope for this pass: made again
        assert _names(rec).count('vqcpc_weight_planes_many') == 2
    with torch.enable_grad(), ops.forward_arithmetic(owner):
        pass
    ops._PARAM_STEPS += 1                                                  # an optimiser step between the forward and the backward
    with ops.direct_weight_gradients(owner):
        assert _names(rec).count('vqcpc_weight_planes_many') == 4
