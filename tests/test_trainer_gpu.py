"""End-to-end parity of the product path (vqcpc_bach_amd, HIP kernels through the C ABI) against
  (1) golden vectors produced by the reference's own Encoder / VQCPCEncoderTrainer.epoch (tests/golden/epoch_*.npz),
  (2) the CPU oracle on a mid-size seeded configuration.
Index assignment is bit-exact; fp32 forward quantities within 5e-5, gradients within 5e-4 (relative to max |ref|)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import vqcpc_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(params=['f32', 'bf16x6'], autouse=True)
def gemm_mode(request):
    """Every end-to-end parity test runs with both GEMM arithmetic modes (exact fp32 MFMA, and the bf16x6 split).  In the
    bf16x6 mode the feed-forward gate travels as a bit mask whatever the size (the product prefers the fp32 gate below 160
    tiles, ops.gatebits_worthwhile: that form is what tests/test_student_gpu.py runs), and the residual sums that LayerNorm
    normalises are formed in the GEMM epilogues whatever the size (the product: from 128 tiles, ops.SFORM_MIN_TILES; the
    two-input LayerNorm form is what tests/test_student_gpu.py and the product-selection test of test_configs_gpu.py run)."""
    from vqcpc_bach_amd import hip, ops
    hip.load()
    hip.set_gemm_mode(1 if request.param == 'bf16x6' else 0)
    saved = ops.GATEBITS_MIN_TILES, ops.SFORM_MIN_TILES
    ops.GATEBITS_MIN_TILES = 0
    ops.SFORM_MIN_TILES = 0           # residual sums formed in the GEMM epilogues whatever the size (product: >= 128 tiles)
    yield request.param
    ops.GATEBITS_MIN_TILES, ops.SFORM_MIN_TILES = saved
    hip.set_gemm_mode(0)

FWD_TOL, GRAD_TOL = 5e-5, 5e-4


def build_trainer(cfg, sd, lr=1e-3, dropout=0.0, c_dropout=None):
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.data_processor.bach_cpc_data_processor import BachCPCDataProcessor
    from vqcpc_bach_amd.dataloaders.synthetic_cpc_dataloader import SyntheticCPCDataloaderGenerator
    from vqcpc_bach_amd.downscalers.relative_transformer_downscaler import RelativeTransformerDownscaler
    from vqcpc_bach_amd.encoder import Encoder
    from vqcpc_bach_amd.quantizer.vector_quantizer import ProductVectorQuantizer
    from vqcpc_bach_amd.upscalers.mlp_upscaler import MlpUpscaler
    from vqcpc_bach_amd.vqcpc_encoder_trainer import VQCPCEncoderTrainer
    hip.load()
    dlg = SyntheticCPCDataloaderGenerator(num_blocks_left=cfg['Kl'], num_blocks_right=cfg['Kr'],
                                          num_negative_samples=cfg['N'], vocab=cfg['vocab'])
    dp = BachCPCDataProcessor(embedding_size=cfg['emb'], num_events=(cfg['Kl'] + cfg['Kr']) * 4, num_channels=4,
                              num_tokens_per_channel=cfg['vocab'], num_tokens_per_block=16)
    ds = RelativeTransformerDownscaler(input_dim=cfg['emb'], output_dim=cfg['D'], num_channels=4, downscale_factors=[4, 4],
                                       d_model=cfg['d'], n_head=cfg['H'], list_of_num_layers=cfg['layers'],
                                       dim_feedforward=cfg['ff'], dropout=dropout)
    q = ProductVectorQuantizer(codebook_size=cfg['K'], codebook_dim=cfg['D'], commitment_cost=0.25,
                               num_codebooks=cfg['ncb'], use_batch_norm=False, initialize=False, squared_l2_norm=True)
    up = MlpUpscaler(input_dim=cfg['D'], output_dim=cfg['zdim'], hidden_size=cfg['up_hidden'], dropout=dropout)
    enc = Encoder('/tmp/vqcpc_test_model', dp, ds, q, up)
    tr = VQCPCEncoderTrainer('/tmp/vqcpc_test_model', dlg, enc,
                             c_net_kwargs=dict(output_dim=cfg['cdim'], hidden_size=cfg['gru_hidden'],
                                               num_layers=cfg.get('gru_layers', 2),
                                               dropout=dropout if c_dropout is None else c_dropout,
                                               bidirectional=cfg.get('bidirectional', False)),
                             quantization_weighting=cfg.get('qw', 0.5))
    for name in ('encoder', 'c_module', 'fks_module', 'c_module_back', 'fks_module_back'):
        sub = {k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + '.')}
        if sub:
            getattr(tr, name).load_state_dict(sub)
    tr.to('cuda')
    tr.init_optimizers(lr=lr, schedule_lr=False)
    assert tr.flat.check_views()
    return tr


def golden_cfg_sd(g, prefix='sd0'):
    cfg = O.make_cfg(**json.loads(str(g['cfg_json'])))
    sd = {}
    for k, v in g.items():
        if k.startswith(prefix + '/'):
            mod, rest = k[len(prefix) + 1:].split('/', 1)
            sd[mod + '.' + rest] = T(np.array(v))
    return cfg, sd


@pytest.mark.parametrize('name', ['epoch_tiny', 'epoch_tiny_bidir', 'epoch_tiny_acc'])
def test_encoder_forward_golden(name):
    g = load_golden(name)
    cfg, sd = golden_cfg_sd(g)
    tr = build_trainer(cfg, sd)
    tr.eval()
    with torch.no_grad():
        z_up, idx, ql = tr.encoder(T(g['batch/x_left']))
    assert torch.equal(idx.cpu(), T(g['fwd_idx'])), 'codebook index assignment must be bit-exact vs the reference'
    assert rel_err(z_up.cpu(), g['fwd_zup']) < FWD_TOL
    assert rel_err(ql.cpu(), g['fwd_qloss']) < 2e-4


def _near_tie_fraction(cfg, sd, batch, tol=1e-5):
    """Per prediction step k: fraction of windows whose oracle margin f_pos - max f_neg is below `tol` in magnitude
    (averaged over the two directions of the bidirectional model, like the accuracy itself)."""
    with torch.no_grad():
        out = O.cpc_losses(batch, sd, cfg)
    ties = (out['margin'].abs() < tol).float().mean(0)
    if out['margin_back'] is not None:
        ties = (ties + (out['margin_back'].abs() < tol).float().mean(0)) / 2
    return ties.numpy()


@pytest.fixture(params=['plain_qkv', 'table_qkv'])
def first_layer_path(request):
    """The first layer's in_proj either as a GEMM over every token or as the block-table lookup (forced on here: the
    fixtures are smaller than the size from which the product switches to it by itself)."""
    from vqcpc_bach_amd.downscalers.relative_transformer_downscaler import RelativeTransformerDownscaler as D
    old = D.table_lookup_min_ratio
    D.table_lookup_min_ratio = 0 if request.param == 'table_qkv' else 10 ** 9
    yield request.param
    D.table_lookup_min_ratio = old


@pytest.mark.parametrize('name', ['epoch_tiny', 'epoch_tiny_bidir', 'epoch_tiny_clip', 'epoch_tiny_acc'])
def test_epoch_golden(name, first_layer_path):
    """epoch(train=False), then one training step, against the reference's VQCPCEncoderTrainer.epoch."""
    g = load_golden(name)
    cfg, sd = golden_cfg_sd(g)
    lr = float(g['lr'])
    tr = build_trainer(cfg, sd, lr=lr)
    batch = {k.split('/', 1)[1]: T(v) for k, v in g.items() if k.startswith('batch/')}
    ev = tr.epoch(iter([batch]), train=False, num_batches=1, corrupt_labels=False)
    for k in ('loss', 'loss_quantize', 'loss_contrastive'):
        assert abs(ev[k] - float(g[f'eval/{k}'])) < FWD_TOL * max(1.0, abs(float(g[f'eval/{k}']))), k
    assert ev['num_codewords'] == float(g['eval/num_codewords'])
    assert ev['num_codewords_negative'] == float(g['eval/num_codewords_negative'])
    # accuracy is a count of strict wins `f_pos > max f_neg`; a positive and a negative block that fall on the same code
    # give scores that differ by an ulp or not at all (the straight-through value z + (q - z) depends on z), so such
    # windows may flip with any change of rounding: they are identified with the oracle and allowed either way
    slack = _near_tie_fraction(cfg, sd, batch)
    assert np.all(np.abs(np.asarray(ev['accuracy']) - g['eval/accuracy']) <= slack + 1e-6), (ev['accuracy'], slack)
    assert abs(ev['loss_monitor'] - float(g['eval/loss_monitor'])) <= float(slack.mean()) + 1e-6

    # gradients BEFORE clipping: forward + backward only
    tr.train()
    loss, out = tr.compute_losses(batch)
    tr.flat.zero_grad()
    loss.backward()
    names = {id(p): n for n, p in tr.named_parameters()}
    for p in tr.flat.params:
        ref = g.get('grad/' + names[id(p)])
        if ref is None:
            assert float(p.grad.abs().max()) == 0.0, names[id(p)]
            continue
        assert rel_err(p.grad.cpu(), ref) < GRAD_TOL, names[id(p)]
    # clip + Adam (flat kernels), then compare the updated parameters
    tr.optimizer.step(lr=lr)
    assert abs(tr.optimizer.grad_norm() - float(g['grad_total_norm'])) < 2e-4 * float(g['grad_total_norm'])
    coef = min(1.0, 5.0 / (float(g['grad_total_norm']) + 1e-6))
    after = dict(tr.named_parameters())
    for k, v in g.items():
        if not k.startswith('sd1/'):
            continue
        mod, rest = k[4:].split('/', 1)
        pname = mod + '.' + rest
        got, ref = after[pname].detach().cpu(), T(np.array(v))
        gref = g.get('grad/' + pname)
        if gref is None:
            assert torch.equal(got, ref), pname
            continue
        assert float((got - ref).abs().max()) <= 1.01 * lr + 1e-7, pname
        sig = T(np.abs(gref) * coef > 1e-4)          # where |g| ~ eps the first Adam step is rounding noise <= lr
        if sig.any():
            assert float((got - ref)[sig].abs().max()) < 2e-2 * lr + 1e-7, pname


def test_train_step_vs_oracle_midsize():
    """Mid-size model with a 2-codebook product quantiser (the reference itself cannot run ncb > 1, SURVEY.md defect 1;
    the oracle carries the codebook axis).  Indices bit-exact, losses / gradients within tolerance."""
    cfg = O.make_cfg(emb=32, vocab=[56] * 4, d=64, H=4, layers=[2, 2], ff=128, D=32, K=64, ncb=2, zdim=32, up_hidden=64,
                     cdim=32, gru_hidden=64, B=8, N=15, Kl=4, Kr=4)
    sd = O.init_state(cfg, seed=3)
    batch = O.synthetic_batch(cfg, seed=11)
    # place the codebooks on downscaler outputs so that many codes are in use
    z_probe = O.encoder_forward(batch['negative_samples'].reshape(-1, 4, 4), sd, cfg, stages=(st := {}))
    zp = st['z'].reshape(-1, cfg['D'])
    for c in range(cfg['ncb']):
        sd[f'encoder.quantizer.embeddings.{c}'] = zp[c * 7:c * 7 + cfg['K'], c * 16:(c + 1) * 16].clone() + 0.01
    otr = O.OracleTrainer(cfg, sd, lr=1e-3)
    ref = otr.step(batch, train=True)
    tr = build_trainer(cfg, sd, lr=1e-3)
    tr.train()
    loss, out = tr.compute_losses(batch)
    tr.flat.zero_grad()
    loss.backward()
    for k in ('idx_left', 'idx_right', 'idx_negative'):
        assert torch.equal(out[k].cpu().reshape(ref[k].shape), ref[k]), k
    for k in ('loss', 'loss_contrastive', 'loss_quantize'):
        assert abs(float(out[k]) - float(ref[k])) < FWD_TOL * max(1.0, abs(float(ref[k]))), k
    assert torch.allclose(out['accuracy'].cpu(), ref['accuracy'])
    worst = 0.0
    for n, p in tr.named_parameters():
        e = rel_err(p.grad.cpu(), otr.last_grads[n])
        worst = max(worst, e)
        assert e < GRAD_TOL, (n, e)
    print('worst relative gradient error', worst)


def test_dropout_training_step_runs_and_is_reproducible():
    from vqcpc_bach_amd.utils import SEEDS
    cfg = O.make_cfg(emb=32, vocab=[56] * 4, d=64, H=4, layers=[1, 1], ff=128, D=16, K=32, ncb=1, zdim=32, up_hidden=64,
                     cdim=32, gru_hidden=32, B=4, N=3, Kl=2, Kr=2)
    sd = O.init_state(cfg, seed=5)
    batch = O.synthetic_batch(cfg, seed=12)
    losses = []
    for _ in range(2):
        torch.manual_seed(0)
        SEEDS.manual_seed(77)
        # every dropout site (attention, residuals, FFN, upscaler, GRU inter-layer) draws from the counter RNG
        tr = build_trainer(cfg, sd, lr=1e-3, dropout=0.2)
        m = tr.epoch(iter([batch, batch]), train=True, num_batches=2, corrupt_labels=False)
        losses.append(m['loss'])
        assert np.isfinite(m['loss'])
    assert losses[0] == losses[1], 'same seeds -> same masks -> same loss'


def test_c0_config_vs_oracle():
    """BASELINE.json configs[0] (encoder_cpc_small: seq_len 64, batch 8, 1 codebook x 64 codes): one training step of the
    product against the CPU oracle at the configuration's full size."""
    cfg = O.make_cfg('C0')
    sd = O.init_state(cfg, seed=7)
    batch = O.synthetic_batch(cfg, seed=8)
    st = {}
    O.encoder_forward(batch['negative_samples'].reshape(-1, 4, 4), sd, cfg, stages=st)
    zp = st['z'].reshape(-1, cfg['D'])
    sd['encoder.quantizer.embeddings.0'] = zp[:cfg['K']].clone() + 0.01       # data-placed codebook (as _initialize does)
    otr = O.OracleTrainer(cfg, sd, lr=1e-4)
    ref = otr.step(batch, train=True)
    tr = build_trainer(cfg, sd, lr=1e-4)
    tr.train()
    loss, out = tr.compute_losses(batch)
    tr.flat.zero_grad()
    loss.backward()
    for k in ('idx_left', 'idx_right', 'idx_negative'):
        assert torch.equal(out[k].cpu().reshape(ref[k].shape), ref[k]), k
    for k in ('loss', 'loss_contrastive', 'loss_quantize'):
        assert abs(float(out[k].detach()) - float(ref[k].detach())) < FWD_TOL * max(1.0, abs(float(ref[k].detach()))), k
    for n, p in tr.named_parameters():
        assert rel_err(p.grad.cpu(), otr.last_grads[n]) < GRAD_TOL, n


def test_codebook_data_initialisation_and_epoch_contract():
    """initialize=True: the first (negatives) segment seeds the codebooks (vector_quantizer.py:57-70); epoch() returns the
    reference's metric dict (vqcpc_encoder_trainer.py:343-354)."""
    from vqcpc_bach_amd import configs, getters
    config = configs.make_config('C0', dropout=0.1)
    dlg = getters.get_dataloader_generator('bach', 'vqcpc', dict(config['dataloader_generator_kwargs'], device='cuda'))
    enc = getters.get_encoder('/tmp/vqcpc_test_model', dlg, config)
    tr = getters.get_encoder_trainer('/tmp/vqcpc_test_model', dlg, 'vqcpc', enc, config['auxiliary_networks_kwargs'])
    tr.to('cuda')
    tr.init_optimizers(lr=1e-4, schedule_lr=True)
    before = enc.quantizer.embeddings[0].detach().clone()
    assert enc.quantizer.initialize
    gen_train, gen_val, _ = dlg.dataloaders(batch_size=8)
    m = tr.epoch(gen_train, train=True, num_batches=3, corrupt_labels=True)
    assert not enc.quantizer.initialize and not torch.equal(before, enc.quantizer.embeddings[0].detach())
    assert tr.flat.check_views(), 'data initialisation must write INTO the flat parameter buffer'
    assert set(m) == {'loss', 'accuracy', 'loss_quantize', 'loss_contrastive', 'num_codewords', 'num_codewords_negative',
                      'loss_monitor'}
    assert isinstance(m['accuracy'], list) and len(m['accuracy']) == dlg.num_blocks_right
    assert np.isfinite(m['loss']) and 1 <= m['num_codewords'] <= 64
    assert abs(tr.current_lr() - 1e-4 * (0.1 + 9e-5 * 3)) < 1e-12      # LambdaLR stepped once per training batch
    v = tr.epoch(gen_val, train=False, num_batches=1, corrupt_labels=False)
    assert np.isfinite(v['loss']) and tr.global_step == 3


def test_three_step_trajectory_vs_oracle_with_lr_schedule():
    """Three consecutive training steps (Adam bias correction at t = 1, 2, 3, LambdaLR stepped per batch): per-step
    losses and the parameters after the third step follow the oracle's trajectory."""
    cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=64, H=4, layers=[2, 1], ff=128, D=16, K=16, ncb=1, zdim=16, up_hidden=32,
                     cdim=16, gru_hidden=32, B=6, N=5, Kl=3, Kr=3)
    sd = O.init_state(cfg, seed=13)
    batches = [O.synthetic_batch(cfg, seed=40 + i) for i in range(3)]
    st = {}
    O.encoder_forward(batches[0]['x_left'], sd, cfg, stages=st)
    sd['encoder.quantizer.embeddings.0'] = st['z'].reshape(-1, cfg['D'])[:cfg['K']].clone() + 0.01
    otr = O.OracleTrainer(cfg, sd, lr=3e-3, schedule_lr=True)
    tr = build_trainer(cfg, sd, lr=3e-3)
    tr.schedule_lr = True
    tr.train()
    for i, b in enumerate(batches):
        ref = otr.step(b, train=True)
        assert abs(tr.current_lr() - 3e-3 * O.lr_lambda(i)) < 1e-12
        out = tr.train_step(b, train=True)
        # the trajectories separate slowly (different rounding, Adam's 1/sqrt(v) amplifies tiny gradients): 2e-4 per step
        assert abs(float(out['loss']) - float(ref['loss'])) < 2e-4 * (i + 1) * max(1.0, abs(float(ref['loss']))), i
    worst = max(float((p.detach().cpu() - otr.P[n].detach()).abs().max()) for n, p in tr.named_parameters())
    assert worst < 3 * 3e-3 * 0.05, worst            # far below the 3 * lr a sign flip of every update would give


def test_training_reduces_the_loss_on_a_fixed_batch():
    """Sanity of the whole loop: 40 Adam steps on one batch lower the InfoNCE loss and raise the accuracy."""
    cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=64, H=4, layers=[2, 2], ff=128, D=16, K=32, ncb=1, zdim=16, up_hidden=32,
                     cdim=16, gru_hidden=32, B=16, N=7, Kl=4, Kr=4)
    sd = O.init_state(cfg, seed=21)
    batch = O.synthetic_batch(cfg, seed=22)
    st = {}
    O.encoder_forward(batch['negative_samples'].reshape(-1, 4, 4), sd, cfg, stages=st)
    sd['encoder.quantizer.embeddings.0'] = st['z'].reshape(-1, cfg['D'])[:cfg['K']].clone() + 0.01
    tr = build_trainer(cfg, sd, lr=2e-3)
    first = tr.epoch(iter([batch]), train=False, num_batches=1, corrupt_labels=False)
    tr.epoch(iter([batch] * 40), train=True, num_batches=40, corrupt_labels=False)
    last = tr.epoch(iter([batch]), train=False, num_batches=1, corrupt_labels=False)
    assert last['loss_contrastive'] < 0.7 * first['loss_contrastive'], (first, last)
    assert np.mean(last['accuracy']) > np.mean(first['accuracy']) + 0.2


def gradient_additivity_error(tr, batch, B):
    """Full-size gradient pin: with dropout 0 every window contributes its own term to the loss, so the flat gradient of
    the B-window step is the mean of the gradients of its two halves.  Returns |g - (g1 + g2) / 2| / |g| on the whole flat
    gradient bucket (zero_grad + forward + backward of the trainer's own step, no optimiser update)."""
    grads = []
    for sl in (slice(0, B), slice(0, B // 2), slice(B // 2, B)):
        tr._step_compute({k: v[sl] for k, v in batch.items()})
        grads.append(tr.flat.flat_grad.double().clone())
    assert float(grads[0].norm()) > 0
    return float((grads[0] - 0.5 * (grads[1] + grads[2])).norm() / grads[0].norm())


def test_full_size_c1_step_properties():
    """BASELINE configs[1] at full size (B = 256, 34 816 blocks): size-independent properties.
      * the product's code assignment on ITS OWN 34 816 x 32 encoder outputs == the oracle's canonical argmin, bit for bit;
      * every window's loss only involves its own blocks: loss(batch) == mean(loss(first half), loss(second half));
      * the same for the GRADIENTS: flat gradient of the B = 256 step == mean of the flat gradients of its two halves, within
        2e-5 of its norm (every backward kernel at the benchmark's launch geometry);
      * attention rows are probability distributions; one training step leaves a finite gradient bucket / parameters."""
    from vqcpc_bach_amd import configs, getters, ops
    from vqcpc_bach_amd.utils import SEEDS
    config = configs.make_config('C1', dropout=0.0)
    dlg = getters.get_dataloader_generator('bach', 'vqcpc', dict(config['dataloader_generator_kwargs'], device='cuda', seed=3))
    enc = getters.get_encoder('/tmp/vqcpc_test_c1', dlg, config)
    tr = getters.get_encoder_trainer('/tmp/vqcpc_test_c1', dlg, 'vqcpc', enc, config['auxiliary_networks_kwargs'])
    tr.to('cuda')
    tr.init_optimizers(lr=1e-4, schedule_lr=False)
    batch = next(dlg.dataloaders(batch_size=256)[0])
    tr.eval()
    with torch.no_grad():
        tr.compute_losses(batch)                                   # data-dependent codebook initialisation happens here
        loss, out = tr.compute_losses(batch)
        halves = [tr.compute_losses({k: v[s] for k, v in batch.items()})[0] for s in (slice(0, 128), slice(128, 256))]
        assert abs(float(loss) - 0.5 * (float(halves[0]) + float(halves[1]))) < 2e-5 * abs(float(loss))
        # index assignment at full size against the canonical CPU argmin on the SAME z
        tokens = enc.data_processor.preprocess(batch['negative_samples'].reshape(-1, 4, 4))
        z = enc.downscaler.forward_tokens(tokens.reshape(1, -1, 16), enc.data_processor)[0]
        assert z.shape == (256 * 15 * 8, 32)
        cb = torch.stack(list(enc.quantizer.embeddings))
        idx = ops.vq_assign(z, cb)
        ref = O.vq_assign(z.cpu(), [e.detach().cpu() for e in enc.quantizer.embeddings])
        assert torch.equal(idx.cpu(), ref)
        assert torch.equal(idx, out['idx_negative'].reshape(-1, 2))
        # attention probabilities of the first layer
        x = torch.randn(64 * 16, 256, device='cuda')
        layer = enc.downscaler.transformers[0].layers[0]
        _, probs = layer.forward_rows(x)
        assert float((probs.sum(-1) - 1).abs().max()) < 1e-5 and float(probs.min()) >= 0.0
    tr.train()
    # full-size GRADIENTS (34 816 blocks through every backward kernel at the benchmark's launch geometry)
    err = gradient_additivity_error(tr, batch, 256)
    print('C1 full-size gradient additivity error', err)
    assert err < 2e-5, err
    SEEDS.manual_seed(5)
    tr.train_step(batch, train=True)
    assert bool(torch.isfinite(tr.flat.flat_grad).all()) and bool(torch.isfinite(tr.flat.flat).all())
    assert 0.0 < tr.optimizer.grad_norm() < 1e4


def test_bf16_gemm_mode_step_is_close_to_the_fp32_step(gemm_mode):
    """Reduced-precision mode for BASELINE configs[4] (`bench.py --config C4 --gemm-mode bf16`): operands of every tiled GEMM
    rounded to bf16, fp32 accumulation; everything else (attention, LayerNorm, VQ distances, GRU recurrence, losses, Adam)
    stays fp32.  Against the fp32 oracle: losses within 2 %, almost every code identical, gradients within 10 % of the
    largest entry.  (This mode is never used for the fp32 headline configuration.)"""
    if gemm_mode != 'f32':
        pytest.skip('sets its own GEMM mode')
    from vqcpc_bach_amd import hip
    cfg = O.make_cfg(emb=32, vocab=[56] * 4, d=128, H=4, layers=[2, 2], ff=256, D=32, K=64, ncb=2, zdim=32, up_hidden=64,
                     cdim=32, gru_hidden=64, B=64, N=15, Kl=4, Kr=4)
    sd = O.init_state(cfg, seed=3)
    batch = O.synthetic_batch(cfg, seed=11)
    O.encoder_forward(batch['negative_samples'].reshape(-1, 4, 4), sd, cfg, stages=(st := {}))
    zp = st['z'].reshape(-1, cfg['D'])
    for c in range(cfg['ncb']):
        sd[f'encoder.quantizer.embeddings.{c}'] = zp[c * 7:c * 7 + cfg['K'], c * 16:(c + 1) * 16].clone() + 0.01
    otr = O.OracleTrainer(cfg, sd, lr=1e-3)
    ref = otr.step(batch, train=True)
    tr = build_trainer(cfg, sd, lr=1e-3)
    tr.train()
    hip.set_gemm_mode(8)
    try:
        loss, out = tr.compute_losses(batch)
        tr.flat.zero_grad()
        loss.backward()
    finally:
        hip.set_gemm_mode(0)
    same = sum(int((out[k].cpu().reshape(ref[k].shape) == ref[k]).sum()) for k in ('idx_left', 'idx_right', 'idx_negative'))
    total = sum(ref[k].numel() for k in ('idx_left', 'idx_right', 'idx_negative'))
    assert same / total > 0.97, same / total
    for k in ('loss', 'loss_contrastive', 'loss_quantize'):
        assert abs(float(out[k].detach()) - float(ref[k].detach())) < 2e-2 * max(1.0, abs(float(ref[k].detach()))), k
    worst = max(rel_err(p.grad.cpu(), otr.last_grads[n]) for n, p in tr.named_parameters())
    assert 1e-4 < worst < 0.1, worst                     # visibly bf16, still the same gradients
