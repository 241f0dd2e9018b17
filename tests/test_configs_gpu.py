"""BASELINE.json configurations at their MODEL dimensions against the CPU oracle, at a batch the oracle finishes in
seconds (every window is independent, so the per-window arithmetic is the configuration's own):

  C1 (configs[1]): d_model 256, 8 heads, ff 1024, [2, 2] layers, product-VQ 2 x 512 codes of dim 16, 8 + 8 blocks
  C4 (configs[4]): d_model 512, 8 heads of 64, ff 2048, [4, 4] layers, product-VQ 4 x 1024 codes of dim 16, 16 + 16 blocks

fp32-class GEMM modes (exact fp32 MFMA and the bf16x6 split): codebook indices bit-exact, losses within 5e-5, every
gradient within 5e-4 of the oracle's (relative to the tensor's largest entry) -- on parameters whose relu gates are not
within rounding of zero (see _condition_relu_gates).  Both first-layer paths (plain in_proj
GEMM / block-table lookup) are covered at C1."""
import pytest
import torch

from conftest import rel_err
from oracle import vqcpc_oracle as O
from test_trainer_gpu import FWD_TOL, GRAD_TOL, build_trainer, first_layer_path, gemm_mode  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _data_placed_codebooks(cfg, sd, batch):
    """Codebooks on downscaler outputs of the batch (what `_initialize` does), so that many codes are in use."""
    zs = []
    with torch.no_grad():
        for x in (batch['negative_samples'].reshape(-1, 4, 4), batch['x_left'], batch['x_right']):
            st = {}
            O.encoder_forward(x, sd, cfg, stages=st)
            zs.append(st['z'].reshape(-1, cfg['D']))
    zp = torch.cat(zs)
    assert zp.shape[0] >= cfg['K'] + 7 * cfg['ncb']
    dsub = cfg['D'] // cfg['ncb']
    for c in range(cfg['ncb']):
        sd[f'encoder.quantizer.embeddings.{c}'] = zp[c * 7:c * 7 + cfg['K'], c * dsub:(c + 1) * dsub].clone() + 0.01


class _ReluProbe:
    """Records the argument of every torch.relu call of the oracle (the FFN pre-activations, in call order)."""

    def __enter__(self):
        self.real, self.calls = torch.relu, []
        torch.relu = lambda x: (self.calls.append(x.detach()), self.real(x))[1]
        return self

    def __exit__(self, *exc):
        torch.relu = self.real


def _condition_relu_gates(cfg, sd, batch, tau=5e-6, rounds=10):
    """relu'(0) is a genuine discontinuity of the model: at these sizes (millions of hidden units per step) a handful of
    FFN pre-activations land within fp32 rounding of zero, and whether such a unit's gate is open is decided by the last
    ulp of a 256- or 512-term dot product -- two correct fp32 implementations (this product and the oracle, or the oracle
    in fp32 and in fp64: tools/diag_grad_error.py) then differ by that unit's whole gradient contribution (measured:
    up to 6e-3 of the largest entry of linear1.weight's gradient, 1e-4..1e-3 on everything below it).  To keep the
    comparison well-posed the TEST PARAMETERS are chosen so that no pre-activation of the oracle lies within `tau` of
    zero: the bias of an offending hidden unit is nudged by 4 tau (a different, equally arbitrary, random model)."""
    names = [f'encoder.downscaler.transformers.{s}.layers.{l}.linear1.bias'
             for s, nl in enumerate(cfg['layers']) for l in range(nl)]
    for _ in range(rounds):
        with _ReluProbe() as probe, torch.no_grad():
            O.cpc_losses(batch, sd, cfg)
        assert len(probe.calls) % len(names) == 0
        dirty = 0
        for i, pre in enumerate(probe.calls):
            bad = (pre.abs() < tau).reshape(-1, pre.shape[-1]).any(0)
            if bool(bad.any()):
                sd[names[i % len(names)]][bad] += 4 * tau
                dirty += int(bad.sum())
        if not dirty:
            return
    raise AssertionError('could not move every FFN pre-activation away from zero')


_ORACLE_CACHE = {}       # the oracle step is the slow part (seconds): shared by the GEMM-mode / first-layer parametrisations


def _oracle_step(cfg, seed):
    key = (repr(sorted(cfg.items(), key=str)), seed)
    if key not in _ORACLE_CACHE:
        sd = O.init_state(cfg, seed=seed)
        batch = O.synthetic_batch(cfg, seed=seed + 1)
        _data_placed_codebooks(cfg, sd, batch)
        _condition_relu_gates(cfg, sd, batch)
        otr = O.OracleTrainer(cfg, sd, lr=1e-4)
        ref = otr.step(batch, train=True)
        _ORACLE_CACHE[key] = (sd, batch, otr, ref)
    return _ORACLE_CACHE[key]


def _step_vs_oracle(cfg, seed, trainer_backward=False, forward_scope=False):
    sd, batch, otr, ref = _oracle_step(cfg, seed)
    tr = build_trainer(cfg, sd, lr=1e-4)
    tr.train()
    if forward_scope:                          # as the trainers' step does it: the forward inside ops.forward_arithmetic
        from vqcpc_bach_amd import ops
        with ops.forward_arithmetic(tr.flat):
            loss, out = tr.compute_losses(batch)
    else:
        loss, out = tr.compute_losses(batch)
    tr.flat.zero_grad()
    if trainer_backward:                       # as the trainers' step does it: weight gradients straight into the bucket,
        from vqcpc_bach_amd import ops         # batched transposes, and the gradient scope of the GEMM library open
        with ops.direct_weight_gradients(tr.flat):
            loss.backward()
    else:
        loss.backward()
    for k in ('idx_left', 'idx_right', 'idx_negative'):
        assert torch.equal(out[k].cpu().reshape(ref[k].shape), ref[k]), f'{k}: index assignment differs from the oracle'
    used = torch.cat([ref[k].reshape(-1, cfg['ncb']) for k in ('idx_left', 'idx_right', 'idx_negative')]).unique().numel()
    assert used > cfg['K'] // 4, f'only {used} codes in use: the index check would be vacuous'
    for k in ('loss', 'loss_contrastive', 'loss_quantize'):
        assert abs(float(out[k].detach()) - float(ref[k].detach())) < FWD_TOL * max(1.0, abs(float(ref[k].detach()))), k
    worst = 0.0
    for n, p in tr.named_parameters():
        e = rel_err(p.grad.cpu(), otr.last_grads[n])
        worst = max(worst, e)
        assert e < GRAD_TOL, (n, e)
    print(f'worst relative gradient error {worst:.2e}')
    return tr, otr, batch


def test_c1_model_dimensions_vs_oracle(first_layer_path):
    cfg = O.make_cfg('C1', B=8)
    _step_vs_oracle(cfg, seed=31)


def test_c1_model_dimensions_with_the_products_own_gate_selection(gemm_mode):
    """The module fixture forces the bit-mask feed-forward gate at every size; the PRODUCT takes it only where the 256-tile
    kernel fills the chip (>= ops.GATEBITS_MIN_TILES = 160 tiles) and the fp32 gate on 128-tiles below.  Here the product's
    own threshold is restored: at B = 8 the full layers of stack 1 (17 408 rows x 1024: 272 tiles) take the bit mask and
    every other layer (4 352 / 1 088 rows: 68 / 16 tiles) the fp32 gate, so BOTH branches are oracle-checked in one step."""
    if gemm_mode != 'bf16x6':
        pytest.skip('the bit-mask gate exists in the bf16x6 mode only')
    from vqcpc_bach_amd import ops
    assert ops.GATEBITS_MIN_TILES == 0 and ops.SFORM_MIN_TILES == 0      # the fixture's overrides ...
    ops.GATEBITS_MIN_TILES = 160                # ... replaced by the library's values (the fixture restores them afterwards)
    ops.SFORM_MIN_TILES = 128                   # two-input LayerNorm below 128 tiles (68 / 17 / 4 at this batch)
    cfg = O.make_cfg('C1', B=8)
    rows = 8 * (8 + 8 + 15 * 8) * 16
    assert ops.gatebits_worthwhile(rows, 1024, 256) and not ops.gatebits_worthwhile(rows // 4, 1024, 256)
    _step_vs_oracle(cfg, seed=31)


@pytest.mark.parametrize('cfg_name,B', [('C1', 8), ('C4', 4)])
def test_model_dimensions_with_f16x3_forward_arithmetic(gemm_mode, cfg_name, B):
    """Opt-in ops.FWD_ARITH = 'f16x3': the forward products of the training step on the three-product fp16 kernel as well (every
    eligible launch forced through it), gradients likewise -- against the oracle at the UNCHANGED tolerances: indices bit-exact,
    losses within 5e-5, every gradient within 5e-4."""
    if gemm_mode != 'bf16x6':
        pytest.skip('an arithmetic of the bf16x6 mode')
    from vqcpc_bach_amd import hip, ops
    calls, raw = [], hip.call
    prev = ops.set_gradient_arithmetic('f16x3')
    saved = ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS, ops.FWD_ARITH
    ops.GRAD_MIN_TILES = ops.GRAD_TN_MIN_ROWS = 0
    ops.FWD_ARITH = 'f16x3'
    hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
    try:
        _step_vs_oracle(O.make_cfg(cfg_name, B=B), seed=31, trainer_backward=True, forward_scope=True)
    finally:
        hip.call = raw
        ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS, ops.FWD_ARITH = saved
        ops.set_gradient_arithmetic(prev)
    n_fwd = calls.count('vqcpc_gemm_nt_f16x3')
    print(f'{n_fwd} forward launches on the three-product kernel')
    assert n_fwd >= 8, n_fwd


def test_c1_step_with_launches_cut_into_whole_tiles_plus_tail_rows(gemm_mode):
    """The launch plan of the ragged full-size products (ops._g3_plan: whole rounds on the 256-tile f16x3 kernel + the last rows on
    vqcpc_gemm_nt_grad_tail) inside a whole oracle-checked training step.  At B = 32, where C1 reaches more than 256 tiles by
    itself, the oracle comparison is not well-posed (hundreds of FFN pre-activations within rounding of zero, see
    _condition_relu_gates), so the plan is PLACED for the stack-1 shapes of the B = 8 step: 17 408 rows = 16 384 on the 256-tile
    kernel + 1 024 tail rows, for every d_model-wide product of the one full-length layer that runs on rows, forward and backward.  UNCHANGED tolerances: indices
    bit-exact, losses within 5e-5, every gradient within 5e-4."""
    if gemm_mode != 'bf16x6':
        pytest.skip('an arithmetic of the bf16x6 mode')
    from vqcpc_bach_amd import hip, ops
    calls, raw = [], hip.call
    prev = ops.set_gradient_arithmetic('f16x3')
    saved_arith, saved_plans = ops.FWD_ARITH, dict(ops._g3_plans)
    ops.FWD_ARITH = 'f16x3'
    rows = 8 * (8 + 8 + 15 * 8) * 16
    assert rows == 17408
    for N, K in ((256, 256), (256, 512), (256, 768), (256, 1024), (512, 256), (768, 256)):
        assert ops._grad_rows(rows, N, K) in (0, rows)
        ops._g3_plans[ops._g3_plan_key(rows, N, K)] = ((16384, -1),)
    shapes = []
    hip.call = lambda name, *args: (calls.append(name), shapes.append((name, args[6:9])) if name.startswith('vqcpc_gemm_nt') else None,
                                    raw(name, *args))[2]
    try:
        _step_vs_oracle(O.make_cfg('C1', B=8), seed=31, trainer_backward=True, forward_scope=True)
    finally:
        hip.call = raw
        ops.FWD_ARITH = saved_arith
        ops._g3_plans.clear()
        ops._g3_plans.update(saved_plans)
        ops.set_gradient_arithmetic(prev)
    n_tail = calls.count('vqcpc_gemm_nt_grad_tail')
    print(f'{n_tail} tail-row launches in the step:', sorted(set(sh for n, sh in shapes if n == 'vqcpc_gemm_nt_grad_tail')))
    assert n_tail >= 5, n_tail            # out_proj, linear2 and the k | v projection forward, two input gradients backward


def _full_size_c4_properties(bf16):
    """BASELINE configs[4] at FULL size (B = 256, 16 + 16 blocks, 69 632 blocks of 16 tokens per step, d_model 512, 4 + 4
    layers, 4 x 1024 codes): the launch geometry that only the benchmark used to run (row-cut rounds, split-K remainders, the
    bit-mask gate, 4 codebooks of 1024 codes in LDS), checked through size-independent properties:
      * the product's code assignment on ITS OWN 61 440 x 64 encoder outputs == the oracle's canonical argmin, bit for bit;
      * a window's loss involves its own blocks only: loss(batch) == mean(loss(first half), loss(second half));
      * one training step leaves a finite gradient bucket, finite parameters and a sane gradient norm."""
    from vqcpc_bach_amd import configs, getters, hip, ops
    from vqcpc_bach_amd.utils import SEEDS
    config = configs.make_config('C4', dropout=0.0)
    dlg = getters.get_dataloader_generator('bach', 'vqcpc', dict(config['dataloader_generator_kwargs'], device='cuda', seed=7))
    enc = getters.get_encoder('/tmp/vqcpc_test_c4_full', dlg, config)
    tr = getters.get_encoder_trainer('/tmp/vqcpc_test_c4_full', dlg, 'vqcpc', enc, config['auxiliary_networks_kwargs'])
    tr.to('cuda')
    tr.init_optimizers(lr=1e-4, schedule_lr=False)
    B = config['batch_size']
    assert B == 256
    batch = next(dlg.dataloaders(batch_size=B)[0])
    if bf16:
        hip.set_gemm_mode(8)
    try:
        tr.eval()
        with torch.no_grad():
            tr.compute_losses(batch)                               # data-dependent codebook initialisation happens here
            loss, out = tr.compute_losses(batch)
            halves = [tr.compute_losses({k: v[s] for k, v in batch.items()})[0] for s in (slice(0, B // 2), slice(B // 2, B))]
            tol = 2e-4 if bf16 else 2e-5            # bf16: a half batch takes other tiles -> other (bf16-level) summation orders
            assert abs(float(loss) - 0.5 * (float(halves[0]) + float(halves[1]))) < tol * abs(float(loss))
            tokens = enc.data_processor.preprocess(batch['negative_samples'].reshape(-1, 4, 4))
            z = enc.downscaler.forward_tokens(tokens.reshape(1, -1, 16), enc.data_processor)[0]
            assert z.shape == (B * 15 * 16, 64)
            cb = torch.stack(list(enc.quantizer.embeddings))
            assert cb.shape == (4, 1024, 16)
            idx = ops.vq_assign(z, cb)
            ref = O.vq_assign(z.cpu(), [e.detach().cpu() for e in enc.quantizer.embeddings])
            assert torch.equal(idx.cpu(), ref)
            assert torch.equal(idx, out['idx_negative'].reshape(-1, 4))
            assert idx.unique().numel() > 256                      # the check is not vacuous: many codes in use
        tr.train()
        from test_trainer_gpu import gradient_additivity_error
        err = gradient_additivity_error(tr, batch, B)          # full-size gradients: g(batch) == mean of g(halves)
        print('C4 full-size gradient additivity error', err, 'bf16' if bf16 else 'f32-class')
        assert err < (5e-3 if bf16 else 2e-5), err
        SEEDS.manual_seed(5)
        tr.train_step(batch, train=True)
        assert bool(torch.isfinite(tr.flat.flat_grad).all()) and bool(torch.isfinite(tr.flat.flat).all())
        assert 0.0 < tr.optimizer.grad_norm() < 1e4
    finally:
        if bf16:
            hip.set_gemm_mode(0)


def test_full_size_c4_step_properties():
    _full_size_c4_properties(bf16=False)


def test_full_size_c4_step_properties_bf16_mode(gemm_mode):
    if gemm_mode != 'f32':
        pytest.skip('sets its own GEMM mode')
    _full_size_c4_properties(bf16=True)


def test_c1_model_dimensions_with_three_product_gradient_arithmetic(gemm_mode):
    """Opt-in gradient arithmetic (hip.set_gradient_products(3), include/vqcpc.h): the forward is untouched -- indices
    bit-exact, losses within 5e-5 -- and every gradient stays within the SAME 5e-4 of the oracle's (measured: the worst
    tensor moves from ~1e-5 to ~2e-5)."""
    if gemm_mode != 'bf16x6':
        pytest.skip('a switch of the bf16x6 mode')
    from vqcpc_bach_amd import hip
    hip.set_gradient_products(3)
    try:
        _step_vs_oracle(O.make_cfg('C1', B=8), seed=31, trainer_backward=True)
    finally:
        hip.set_gradient_products(6)


@pytest.mark.parametrize('cfg_name,B', [('C1', 8), ('C4', 4)])
def test_model_dimensions_with_f16x3_gradient_arithmetic(gemm_mode, cfg_name, B):
    """ops.set_gradient_arithmetic('f16x3') (csrc/gemm_grad.hip: two fp16 planes per operand under a per-tensor power-of-two
    scale, three MFMAs per product): the forward is untouched -- indices bit-exact, losses within 5e-5 -- and every parameter
    gradient stays within the UNCHANGED 5e-4 of the oracle's, at the model dimensions of configs[1] / configs[4] with every
    eligible dgrad / wgrad forced through the new kernels (the tile-count thresholds only say where they are faster)."""
    if gemm_mode != 'bf16x6':
        pytest.skip('an arithmetic of the bf16x6 mode')
    from vqcpc_bach_amd import hip, ops
    calls, raw = [], hip.call
    prev = ops.set_gradient_arithmetic('f16x3')
    saved = ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS
    ops.GRAD_MIN_TILES = ops.GRAD_TN_MIN_ROWS = 0
    hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
    try:
        _step_vs_oracle(O.make_cfg(cfg_name, B=B), seed=31, trainer_backward=True)
    finally:
        hip.call = raw
        ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS = saved
        ops.set_gradient_arithmetic(prev)
    assert calls.count('vqcpc_gemm_nt_grad') >= 10 and calls.count('vqcpc_gemm_tn_grad') >= 10, (
        calls.count('vqcpc_gemm_nt_grad'), calls.count('vqcpc_gemm_tn_grad'))


def test_c4_model_dimensions_vs_oracle():
    """configs[4] at B = 4: 1088 blocks of 16 tokens through 4 + 4 layers at d_model 512, 4 x 1024 codes."""
    cfg = O.make_cfg('C4', B=4)
    assert cfg['d'] == 512 and cfg['layers'] == [4, 4] and cfg['ncb'] == 4 and cfg['K'] == 1024 and cfg['D'] == 64
    _step_vs_oracle(cfg, seed=41)


def test_c4_product_config_builds_the_same_model():
    """configs.make_config('C4') (what `bench.py --config C4` runs) has the oracle's C4 parameter shapes."""
    from vqcpc_bach_amd import configs, getters
    config = configs.make_config('C4', dropout=0.0)
    dlg = getters.get_dataloader_generator('bach', 'vqcpc', dict(config['dataloader_generator_kwargs'], device='cuda'))
    enc = getters.get_encoder('/tmp/vqcpc_test_c4', dlg, config)
    tr = getters.get_encoder_trainer('/tmp/vqcpc_test_c4', dlg, 'vqcpc', enc, config['auxiliary_networks_kwargs'])
    ref = O.init_state(O.make_cfg('C4'))
    got = {n: tuple(p.shape) for n, p in tr.named_parameters()}
    assert got == {k: tuple(v.shape) for k, v in ref.items()}
    assert dlg.num_blocks_left == 16 and dlg.num_blocks_right == 16 and config['batch_size'] == 256


class _Bf16Linear(torch.autograd.Function):
    """F.linear whose three GEMMs (forward, input gradient, weight gradient) take bf16-rounded operands and accumulate in
    fp32 -- the arithmetic of the product's bf16 mode (hip.set_gemm_mode(8)), restated for the oracle."""

    @staticmethod
    def forward(ctx, x, w, b):
        xb, wb = x.bfloat16().float(), w.bfloat16().float()
        ctx.save_for_backward(xb, wb)
        ctx.has_bias = b is not None
        y = xb @ wb.t()
        return y if b is None else y + b

    @staticmethod
    def backward(ctx, g):
        xb, wb = ctx.saved_tensors
        gb = g.bfloat16().float()
        g2, x2 = gb.reshape(-1, gb.shape[-1]), xb.reshape(-1, xb.shape[-1])
        return gb @ wb, g2.t() @ x2, (g.reshape(-1, g.shape[-1]).sum(0) if ctx.has_bias else None)


def test_c4_bf16_mode_vs_bf16_cast_oracle(gemm_mode):
    """configs[4] in the precision BASELINE names for it: the product in bf16 mode (bf16 operands in HBM, bf16 FFN hidden
    activation, fp32 accumulation, everything else fp32) against the oracle with every linear layer's operands cast to
    bf16.  Same roundings, different summation orders -- and bf16 roundings of near-equal fp32 values do flip, so the
    agreement is statistical: >= 99 % of the code assignments, losses within 1 % (0.5 % with the oracle's assignment
    forced), gradients in the bf16-cast oracle's neighbourhood (see the end of the test)."""
    if gemm_mode != 'f32':
        pytest.skip('sets its own GEMM mode')
    from vqcpc_bach_amd import hip
    cfg = O.make_cfg('C4', B=4)
    sd, batch, otr32, ref32 = _oracle_step(cfg, seed=41)
    real, real_ln = O.linear, O.layer_norm
    O.linear = lambda x, w, b=None: _Bf16Linear.apply(x, w, b)
    # round 5: on the bf16 path the output of a layer's FIRST LayerNorm exists in bf16 only (GEMM operand and residual of the
    # feed-forward block alike): the oracle's cast point moves with it (rounding = identity for the gradient)
    ln_calls = [0]

    class _Bf16Grad(torch.autograd.Function):                        # ... and the gradient that comes back through a LayerNorm's input
        @staticmethod                                                # (residual branch: LayerNorm backward -> dgrad epilogue in bf16)
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            return g.bfloat16().float()

    def ln_bf16_after_norm1(x, w, b):
        ln_calls[0] += 1
        x = _Bf16Grad.apply(x)
        x = x + (x.bfloat16().float() - x).detach()              # ... and so do the residual sums the LayerNorms read (bf16 out of
        y = real_ln(x, w, b)                                     # the out-proj / FFN2 epilogues)
        # ... and the main-stream gradient that ENTERS a LayerNorm's backward (VQCPC_BF16_GRAD_STREAM): always for norm1 (the input
        # gradient of FFN1 leaves its epilogue in bf16), for norm2 where the consuming layer's in_proj input gradient does -- every
        # layer but the one in front of a stack's query-subsampled last layer and the very last one (2, 6, 7 of the 4 + 4).  At
        # B = 4 the last layer (1 088 kept rows: not a multiple of the 256-row tile) runs the fp32 path altogether
        layer = ((ln_calls[0] - 1) // 2) % 8
        if (ln_calls[0] % 2 == 1 and layer != 7) or (ln_calls[0] % 2 == 0 and layer not in (2, 6, 7)):
            y = _Bf16Grad.apply(y)
        # ... and so does the output of a stack's interior layers (VQCPC_BF16_ACT_STREAM: norm2 of every layer but a stack's last)
        return y + (y.bfloat16().float() - y).detach() if (ln_calls[0] % 2 == 1 or layer % 4 != 3) else y

    O.layer_norm = ln_bf16_after_norm1
    try:
        otr = O.OracleTrainer(cfg, sd, lr=1e-4)
        ref = otr.step(batch, train=True)
    finally:
        O.linear, O.layer_norm = real, real_ln
    from vqcpc_bach_amd import ops
    tr = build_trainer(cfg, sd, lr=1e-4)
    tr.train()
    keys = ('idx_left', 'idx_right', 'idx_negative')
    hip.set_gemm_mode(8)
    real_vq = ops.VQFn.apply
    calls, raw = [], hip.call
    hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
    try:
        loss, out = tr.compute_losses(batch)                      # free-running: the product's own code assignment
        # second pass with the ORACLE's assignment forced (rows in the order of Encoder.encode_many: negatives, left,
        # right): a code that flips changes z_q, the InfoNCE scores and every gradient behind them by O(1), which would
        # hide the bf16-level agreement of everything else
        given = torch.cat([ref[k].reshape(-1, cfg['ncb']) for k in ('idx_negative', 'idx_left', 'idx_right')]).cuda()
        ops.VQFn.apply = lambda z, cb, beta, sq, g=None: real_vq(z, cb, beta, sq, given)
        loss_f, out_f = tr.compute_losses(batch)
        tr.flat.zero_grad()
        loss_f.backward()
    finally:
        hip.call = raw
        ops.VQFn.apply = real_vq
        hip.set_gemm_mode(0)
    # the bf16 gradient stream is what ran: 7 norm1 backwards + the 5 norm2 backwards named above read a bf16 dy, the norm2
    # backwards of layers 2 and 6 an fp32 one
    if ops.BF16_GRAD_STREAM and ops.BF16_GRAD_SUMS and ops.BF16_SUMS and ops.BF16_RESIDUAL:
        assert calls.count('vqcpc_layernorm_bwd_b16io') == 12 and calls.count('vqcpc_layernorm_bwd_xb16') == 2, (
            calls.count('vqcpc_layernorm_bwd_b16io'), calls.count('vqcpc_layernorm_bwd_xb16'))
    same = sum(int((out[k].cpu().reshape(ref[k].shape) == ref[k]).sum()) for k in keys)
    total = sum(ref[k].numel() for k in keys)
    assert same / total > 0.99, same / total
    for k in ('loss', 'loss_contrastive', 'loss_quantize'):
        assert abs(float(out[k].detach()) - float(ref[k].detach())) < 1e-2 * max(1.0, abs(float(ref[k].detach()))), k
        assert abs(float(out_f[k].detach()) - float(ref[k].detach())) < 5e-3 * max(1.0, abs(float(ref[k].detach()))), k
    errs = {n: (rel_err(p.grad.cpu(), otr.last_grads[n]), rel_err(otr.last_grads[n], otr32.last_grads[n]))
            for n, p in tr.named_parameters()}
    top = sorted(errs.items(), key=lambda kv: -kv[1][0])[:6]
    print('largest gradient errors (vs bf16-cast oracle, bf16-cast oracle vs fp32 oracle):')
    for n, (e, e32) in top:
        print(f'  {n:70s} {e:.3e} {e32:.3e}')
    # bf16 noise in the FFN pre-activations flips relu gates by the hundred, so single tensors still differ by ~10 % of
    # their largest entry; what the test pins is that the product sits in the bf16-cast oracle's neighbourhood: every
    # gradient within 25 %, and closer to the bf16-cast oracle than that oracle is to the fp32 one for most tensors
    import statistics
    assert top[0][1][0] < 0.25, top[0]
    closer = sum(1 for e, e32 in errs.values() if e < e32)
    assert closer > 0.7 * len(errs), (closer, len(errs))
    assert statistics.median(e for e, _ in errs.values()) < 0.6 * statistics.median(e32 for _, e32 in errs.values())
