"""The path `bench.py` measures and `train_model()` selects -- bf16x6 GEMM mode, f16x3 forward AND gradient products with
previous-step scales, ragged rounds cut into whole 256-tile rounds + tail rows, in-place atomic residual, under-filled rounds,
step-graph replay -- under test AT FULL SIZE ("Missing #2").  The oracle cannot run B = 256 in seconds, so the checks
are the size-independent properties of tests/test_trainer_gpu.py::test_full_size_c1_step_properties, evaluated INSIDE the training
step's own arithmetic scopes (reference: vqcpc_encoder_trainer.py:201-316, vector_quantizer.py:105-116):

  * the code assignment of the training forward on ITS OWN encoder outputs (the z that reached vqcpc_vq_fwd inside the f16x3 forward
    scope) == the oracle's canonical argmin, bit for bit;
  * loss(batch) == mean(loss(halves)) and flat gradient(batch) == mean(flat gradient(halves)) within 2e-5;
  * eager steps then replayed steps leave finite parameters / gradients and a sane gradient norm;
  * the launch log shows the benchmark's launch plan: vqcpc_gemm_nt_f16x3, vqcpc_gemm_nt_grad_tail, the in-place (add == C)
    accumulate form and an under-filled round.

Plus: the f16x3 forward against the exact fp32-MFMA forward at B = 256 (tools/fwd_f16x3_flips.py as a test), and the f16x3 kernels
on NON-Gaussian operands against fp64 with the componentwise bound csrc/gemm_grad.hip's header promises."""
import contextlib

import pytest
import torch

from oracle import vqcpc_oracle as O

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def training_defaults():
    """What GraphedTraining.use_training_defaults() selects for a caller who chose nothing (the functions train_model() calls),
    whatever earlier tests of the process chose; everything is put back afterwards."""
    from vqcpc_bach_amd import hip, ops
    hip.load()
    mode_before, arith_before = hip.gemm_mode_state(), ops.gradient_arithmetic_state()
    products_before = hip.get_gradient_products()
    hip.set_gemm_mode(0)
    hip._gemm_mode_explicit = False
    ops._grad_arith_explicit = ops._fwd_arith_explicit = False
    ops.GRAD_ARITH = ops.FWD_ARITH = 'six'
    try:
        hip.use_training_default_gemm_mode()
        ops.use_training_default_gradient_arithmetic()
        assert hip.get_gemm_mode() == 1 and ops.GRAD_ARITH == 'f16x3' and ops.FWD_ARITH == 'f16x3'
        yield
    finally:
        hip.restore_gemm_mode_state(mode_before)
        ops.restore_gradient_arithmetic_state(arith_before)
        hip.set_gradient_products(products_before)


@contextlib.contextmanager
def six_product_library_defaults():
    """The bare library defaults the round-5 full-size tests ran in: exact fp32-MFMA GEMMs, 'six' arithmetic scopes."""
    from vqcpc_bach_amd import hip, ops
    hip.load()
    mode_before, arith_before = hip.gemm_mode_state(), ops.gradient_arithmetic_state()
    hip.set_gemm_mode(0)
    ops.GRAD_ARITH = ops.FWD_ARITH = 'six'
    try:
        yield
    finally:
        hip.restore_gemm_mode_state(mode_before)
        ops.restore_gradient_arithmetic_state(arith_before)


class _CallLog:
    """hip.call recorder: names, and for the NT f16x3 entry points (M, N, K) + whether the residual operand IS the output."""

    def __enter__(self):
        from vqcpc_bach_amd import hip
        self.hip, self.raw, self.names, self.nt = hip, hip.call, [], []

        def rec(name, *args):
            self.names.append(name)
            if name in ('vqcpc_gemm_nt_grad', 'vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_grad_tail', 'vqcpc_gemm_nt_g3_pl', 'vqcpc_gemm_nt_g3_small'):
                M, N, K = args[6:9]
                in_place = False
                if name == 'vqcpc_gemm_nt_grad' and args[9] is not None:
                    in_place = args[9].data_ptr() == args[4].data_ptr()
                if name == 'vqcpc_gemm_nt_g3_pl' and args[9] is None and args[13] is not None:      # gradient form on weight planes
                    in_place = args[13].data_ptr() == args[4].data_ptr()
                self.nt.append((name, int(M), int(N), int(K), in_place))
            return self.raw(name, *args)
        hip.call = rec
        return self

    def __exit__(self, *exc):
        self.hip.call = self.raw


def _product_trainer(name, seed, dropout=0.0):
    import os
    from vqcpc_bach_amd import configs, getters
    torch.manual_seed(int(os.environ.get('VQCPC_TEST_SEED', '0')) + seed)      # parameters and the codebooks' data-dependent initialisation
    config = configs.make_config(name, dropout=dropout)
    dlg = getters.get_dataloader_generator('bach', 'vqcpc', dict(config['dataloader_generator_kwargs'], device='cuda', seed=seed))
    enc = getters.get_encoder(f'/tmp/vqcpc_test_bp_{name}', dlg, config)
    tr = getters.get_encoder_trainer(f'/tmp/vqcpc_test_bp_{name}', dlg, 'vqcpc', enc, config['auxiliary_networks_kwargs'])
    tr.to('cuda')
    tr.init_optimizers(lr=1e-4, schedule_lr=False)
    B = config['batch_size']
    batch = next(dlg.dataloaders(batch_size=B)[0])
    return tr, enc, batch, B


def _training_forward(tr, batch, tag=None):
    """compute_losses exactly as _step_compute runs it (grad enabled, inside the forward arithmetic scope); with a tag the scope
    owns a separate scale table (half batches: other shapes, their own sites)."""
    from vqcpc_bach_amd import ops
    with torch.enable_grad(), ops.forward_arithmetic(tr.flat, tag=tag):
        loss, out = tr.compute_losses(batch)
    return loss.detach(), out


def _own_z_canonical_argmin(tr, enc, batch):
    """Runs the training forward twice (the second pass under scales that have followed) with vqcpc_vq_fwd's inputs recorded: the
    indices the kernel assigned on the z it was given == the CPU oracle's canonical argmin on that very z."""
    from vqcpc_bach_amd import ops
    seen, real = [], ops.VQFn.apply

    def probe(z, cb, beta, sq, given=None):
        res = real(z, cb, beta, sq, given)
        seen.append((z.detach(), cb.detach(), res[1]))
        return res
    ops.VQFn.apply = probe
    try:
        for _ in range(2):
            seen.clear()
            loss, out = _training_forward(tr, batch)
    finally:
        ops.VQFn.apply = real
    assert seen, 'the quantiser did not run'
    rows = 0
    for z, cb, idx in seen:
        ref = O.vq_assign(z.cpu(), [c.cpu() for c in cb])
        assert torch.equal(idx.cpu(), ref), 'code assignment on the product\'s own z differs from the canonical argmin'
        rows += z.shape[0]
    used = torch.cat([idx.reshape(-1, idx.shape[-1]) for _, _, idx in seen]).unique().numel()
    assert used > seen[0][1].shape[1] // 4, f'only {used} codes in use: the index check would be vacuous'
    return float(loss), rows


def _full_size_properties(name, arith, seed, grad_tol=2e-5):
    from test_trainer_gpu import gradient_additivity_error
    from vqcpc_bach_amd.utils import SEEDS
    scope = training_defaults() if arith == 'training-defaults' else six_product_library_defaults()
    with scope:
        tr, enc, batch, B = _product_trainer(name, seed)
        assert B == 256
        tr.eval()
        with torch.no_grad():
            tr.compute_losses(batch)                                   # data-dependent codebook initialisation happens here
        tr.train()
        with _CallLog() as log:
            loss, rows = _own_z_canonical_argmin(tr, enc, batch)
            assert rows == B * (15 * 8 + 16) * (1 if name == 'C1' else 2), rows
            # a window's loss involves its own blocks only
            halves = [float(_training_forward(tr, {k: v[s] for k, v in batch.items()}, tag=f'half{i}')[0])
                      for i, s in enumerate((slice(0, B // 2), slice(B // 2, B)))]
            assert abs(loss - 0.5 * (halves[0] + halves[1])) < 2e-5 * abs(loss), (loss, halves)
            # ... and so do the gradients: every backward kernel at the benchmark's launch geometry
            err = gradient_additivity_error(tr, batch, B)
            print(f'{name} full-size gradient additivity error under {arith}: {err:.3e}')
            assert err < grad_tol, err
            # steps the way epoch() takes them in the benchmark: dropout 0.1 (with it the gradient of a residual branch and the
            # sub-layer gradient are two tensors, so the input-gradient products accumulate IN PLACE into the former), eager warm-up
            # steps, then replays of the captured step
            del tr, enc
            tr, enc, batch, B = _product_trainer(name, seed, dropout=0.1)
            tr.train()
            SEEDS.manual_seed(5)
            tr.enable_step_graph(True)
            for _ in range(tr.graph_warmup_steps + 4):        # the first step initialises the codebooks from its data (eager)
                tr.train_step(batch, train=True)
            torch.cuda.synchronize()
            assert tr._graph is not None and tr._graph.replays >= 2, 'the step was not replayed from its graph'
            assert bool(torch.isfinite(tr.flat.flat_grad).all()) and bool(torch.isfinite(tr.flat.flat).all())
            assert 0.0 < tr.optimizer.grad_norm() < 1e4
            from vqcpc_bach_amd import ops
            assert ops.scale_saturations(tr.flat) == 0
        tr.enable_step_graph(False)
        if arith == 'training-defaults':
            n = log.names
            assert n.count('vqcpc_gemm_nt_f16x3') >= 8 and n.count('vqcpc_gemm_tn_grad') >= 8, (
                n.count('vqcpc_gemm_nt_f16x3'), n.count('vqcpc_gemm_tn_grad'))
            assert n.count('vqcpc_gemm_nt_grad_tail') + n.count('vqcpc_gemm_nt_g3_small') >= 4, 'no ragged launch was cut into whole rounds + tail rows'
            assert any(e[4] for e in log.nt), 'the in-place (add == C) accumulate form of vqcpc_gemm_nt_grad did not run'
            under = [e for e in log.nt if 'tail' not in e[0] and (e[1] // 256) * (e[2] // 256) < 256]
            # round 6: from the second step on the B operands are the weights' fp16 planes, made once per step
            assert n.count('vqcpc_weight_planes_many') >= 4 and n.count('vqcpc_gemm_nt_g3_pl') >= 16 and n.count('vqcpc_gemm_nt_g3_small') >= 4, (
                n.count('vqcpc_weight_planes_many'), n.count('vqcpc_gemm_nt_g3_pl'), n.count('vqcpc_gemm_nt_g3_small'))
            if name == 'C1':             # (C4's smallest 256-tile products still fill a round)
                assert under, 'no under-filled round (fewer 256-tiles than CUs) ran on the three-product kernel'
        else:
            assert not any(x.startswith(('vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_grad', 'vqcpc_gemm_tn_grad')) for x in log.names)


@pytest.mark.parametrize('arith', ['six', 'training-defaults'])
def test_full_size_c1_step_properties_by_arithmetic(arith):
    """BASELINE configs[1] (B = 256, 34 816 blocks of 16 tokens, M = 557 056 / 139 264 / 34 816 row products)."""
    _full_size_properties('C1', arith, seed=3)


@pytest.mark.parametrize('arith', ['six', 'training-defaults'])
def test_full_size_c4_step_properties_by_arithmetic(arith):
    """BASELINE configs[4] in fp32-class arithmetic (B = 256, 69 632 blocks, d_model 512, 4 + 4 layers, 4 x 1024 codes)."""
    _full_size_properties('C4', arith, seed=7)


def test_f16x3_forward_against_exact_fp32_forward_at_full_size():
    """tools/fwd_f16x3_flips.py as a test: C1 at B = 256, same batch and parameters, dropout off -- the code assignment of the f16x3
    training forward (scales that have followed) against the exact fp32-MFMA forward: at most 2 of the 69 632 assignments differ
    (near ties: the six-product split differs from the fp32-MFMA forward as often), the loss agrees within 5e-5."""
    from vqcpc_bach_amd import hip, ops
    with training_defaults():
        tr, enc, batch, B = _product_trainer('C1', seed=7)
        tr.eval()
        with torch.no_grad():
            tr.compute_losses(batch)
        tr.train()
        keys = ('idx_left', 'idx_right', 'idx_negative')
        with _CallLog() as log:
            for _ in range(2):
                loss3, out3 = _training_forward(tr, batch)
        assert log.names.count('vqcpc_gemm_nt_f16x3') + log.names.count('vqcpc_gemm_nt_g3_pl') >= 16
        idx3 = torch.cat([out3[k].reshape(-1) for k in keys]).cpu()
        hip.set_gemm_mode(0)
        ops.FWD_ARITH = 'six'
        loss32, out32 = _training_forward(tr, batch, tag='fp32')
        idx32 = torch.cat([out32[k].reshape(-1) for k in keys]).cpu()
        hip.set_gemm_mode(1)
        loss6, out6 = _training_forward(tr, batch, tag='six')
        idx6 = torch.cat([out6[k].reshape(-1) for k in keys]).cpu()
    assert idx3.numel() == 69632
    d3, d6 = int((idx3 != idx32).sum()), int((idx6 != idx32).sum())
    print(f'codes differing from the exact fp32-MFMA forward: f16x3 {d3}, six-product split {d6} of {idx3.numel()}; '
          f'losses {float(loss3):.7f} / {float(loss6):.7f} / {float(loss32):.7f}')
    assert d3 <= 2, d3
    assert abs(float(loss3) - float(loss32)) < 5e-5 * max(1.0, abs(float(loss32)))


# ----------------------------------------------------------------------------------------------------------------------
# f16x3 kernels on non-Gaussian operands (the design's floor, csrc/gemm_grad.hip header)
# ----------------------------------------------------------------------------------------------------------------------
def _state(a, b):
    from vqcpc_bach_amd import hip
    st = torch.zeros(4, device='cuda')
    hip.call('vqcpc_grad_amax', a, a.stride(0), a.shape[0], a.shape[1], st[0:1])
    hip.call('vqcpc_grad_amax', b, b.stride(0), b.shape[0], b.shape[1], st[1:2])
    return st


def _operand(kind, rows, cols, gen):
    x = torch.randn(rows, cols, device='cuda', generator=gen)
    if kind == 'outlier':            # ONE element 2^15 x the rest: everything else sits 15 binades below the scale's anchor
        x[rows // 3, cols // 5] = 2.0 ** 15 * 3.0
    elif kind == 'decades':          # log-uniform magnitudes over 8 decades inside one tensor
        x = x.sign() * torch.pow(10.0, torch.rand(rows, cols, device='cuda', generator=gen) * 8.0 - 6.0)
    elif kind == 'sparse':           # 99 % exact zeros
        x = x * (torch.rand(rows, cols, device='cuda', generator=gen) < 0.01)
    return x.contiguous()


@pytest.mark.parametrize('kind_a,kind_b', [('outlier', 'gauss'), ('decades', 'gauss'), ('sparse', 'gauss'), ('decades', 'decades'),
                                           ('gauss', 'outlier')])
@pytest.mark.parametrize('form', ['nt', 'tn'])
def test_f16x3_on_non_gaussian_operands_vs_fp64(kind_a, kind_b, form):
    """Normwise the f16x3 products are fp32-class whatever the operand distribution: rms error against fp64 within 2 x the exact
    fp32-MFMA kernel's on the same operands, or at the 22-bit operand floor (2^-22 ~ 2.4e-7 rms, <= 4e-7) where an fp32 GEMM would be
    nearly exact (a 99 % sparse operand: five terms per dot product, no accumulation noise to hide behind); COMPONENTWISE they are 22-bit operands under ONE power-of-two scale per tensor:
    an element x of a tensor with largest magnitude A is carried with |error| <= max(2^-21 |x|, 2^-36 A) (two fp16 planes; below
    2^-14 A the low plane is subnormal), and the product drops the low x low term (<= 2^-20 |a b|).  Every output element is
    checked against that bound:  |c - c64| <= (2^-19 + fp32 accumulation) sum_k |a b| + 2^-35 (A_b sum_k |a| + A_a sum_k |b|)."""
    from vqcpc_bach_amd import hip, ops
    gen = torch.Generator(device='cuda').manual_seed(sum(ord(c) for c in kind_a + '|' + kind_b + '|' + form))
    hip.load()
    mode_before = hip.gemm_mode_state()
    hip.set_gemm_mode(1)
    try:
        if form == 'nt':
            M, N, K = 8192, 256, 512
            a, b = _operand(kind_a, M, K, gen), _operand(kind_b, N, K, gen)
            st = _state(a, b)
            out = torch.empty(M, N, device='cuda')
            hip.call('vqcpc_gemm_nt_grad', a, K, b, K, out, N, M, N, K, None, 0, None, 0, None, 1.0, st)
            a64, b64 = a.double(), b.double()
            ref = a64 @ b64.t()
            S = a64.abs() @ b64.abs().t()
            floor = (b64.abs().max() * a64.abs().sum(1))[:, None] + (a64.abs().max() * b64.abs().sum(1))[None, :]
            kc = K
            hip.set_gemm_mode(0)
            exact = ops.gemm_nt(a, b)
        else:
            M, N, K = 16384, 256, 256
            a, b = _operand(kind_a, M, N, gen), _operand(kind_b, M, K, gen)
            st = _state(a, b)
            out = torch.empty(N, K, device='cuda')
            nbytes = hip.query('vqcpc_gemm_tn_grad_workspace', M, N, K)
            ws = hip.workspace(nbytes, 'cuda')
            hip.call('vqcpc_gemm_tn_grad', a, N, b, K, out, None, M, N, K, 0, ws, nbytes, st)
            a64, b64 = a.double(), b.double()
            ref = a64.t() @ b64
            S = a64.abs().t() @ b64.abs()
            floor = (b64.abs().max() * a64.abs().sum(0))[:, None] + (a64.abs().max() * b64.abs().sum(0))[None, :]
            kc = M
            hip.set_gemm_mode(0)
            exact = ops.gemm_tn(a, b, want_bias=False)[0]
    finally:
        hip.restore_gemm_mode_state(mode_before)
    assert bool(torch.isfinite(out).all())
    err = (out.double() - ref).abs()
    bound = (2.0 ** -19 + 4.0 * kc ** 0.5 * 2.0 ** -24) * S + 2.0 ** -35 * floor
    worst = float((err / bound.clamp_min(1e-300)).max())
    rms = lambda x: float((x.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    e3, e32 = rms(out), rms(exact)
    print(f'{form} {kind_a} x {kind_b}: worst error / componentwise bound {worst:.3f}; rms vs fp64 {e3:.2e} (exact fp32 MFMA {e32:.2e})')
    assert worst <= 1.0, worst
    assert e3 < max(2.0 * e32, 4e-7), (e3, e32)
