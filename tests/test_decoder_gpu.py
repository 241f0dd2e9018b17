"""Decoder training step: the product `Decoder` (HIP kernels through the C ABI) against
  (1) fixtures produced by the reference's own Decoder / TransformerDecoderLayerCustom / SubsampledRelativeAttention
      (tests/golden/decoder_*.npz, relbias_cross_*.npz),
  (2) the CPU oracle (oracle/decoder_oracle.py) on seeded inputs, including the shipped configuration's shapes
      (T = 192 target tokens, S = 12 codes, d_model 512, 4 heads of 128),
  (3) size-independent properties: causality of the target stream, anticausality of the memory access, equality with
      the square unmasked kernels.
Codes bit-exact; forward within 5e-5, gradients within 5e-4 (relative to max |ref|)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, sub_state
from oracle import decoder_oracle as D
from oracle import vqcpc_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
FWD_TOL, GRAD_TOL = 5e-5, 5e-4


@pytest.fixture(params=['f32', 'bf16x6'], autouse=True)
def gemm_mode(request):
    from vqcpc_bach_amd import hip
    hip.load()
    hip.set_gemm_mode(1 if request.param == 'bf16x6' else 0)
    yield request.param
    hip.set_gemm_mode(0)


# ---------------------------------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------------------------------
def ref_attention(q, k, v, e1, e2, n, Lq, Lk, H, mask, keep=None):
    """CPU fp32 restatement on (n*L, d) rows with the oracle's bias / mask; `keep` = dropout keep-mask * 1/(1-p)."""
    hd = e1.shape[1]
    qh = (q * hd ** -0.5).reshape(n, Lq, H, hd).transpose(1, 2)
    kh = k.reshape(n, Lk, H, hd).transpose(1, 2)
    vh = v.reshape(n, Lk, H, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    m = D.additive_mask(mask, Lk, Lq)
    if m is not None:
        s = s + m
    s = s + D.relative_bias_cross(qh, e1, e2, Lk)
    probs = torch.softmax(s, dim=-1)
    pd = probs if keep is None else probs * keep
    return (pd @ vh).transpose(1, 2).reshape(n * Lq, H * hd), probs


CASES = [(2, 2, 48, 3, 16, 2), (3, 3, 12, 6, 16, 0), (1, 2, 80, 40, 32, 2), (2, 4, 192, 192, 128, 1), (2, 4, 192, 12, 128, 2),
         (2, 2, 12, 12, 64, 2), (1, 2, 70, 70, 16, 1), (1, 2, 132, 66, 32, 1), (1, 1, 300, 300, 32, 0), (2, 3, 24, 24, 32, 1)]


@pytest.mark.parametrize('n,H,Lq,Lk,hd,mask', CASES)
@pytest.mark.parametrize('drop_p', [0.0, 0.25])
def test_relattn_x_forward_backward(n, H, Lq, Lk, hd, mask, drop_p, gemm_mode):
    if gemm_mode != 'f32':
        pytest.skip('attention does not depend on the GEMM mode')
    from vqcpc_bach_amd import ops
    g = torch.Generator().manual_seed(1000 + Lq + Lk + hd)
    d = H * hd
    cross = Lq != Lk
    q = torch.randn(n * Lq, d, generator=g)
    kv = torch.randn(n * Lk, 2 * d, generator=g)
    e1 = torch.randn(H * Lk, hd, generator=g) * 0.5
    e2 = torch.randn(H * Lk, hd, generator=g) * 0.5
    go = torch.randn(n * Lq, d, generator=g)
    seed = 777
    keep = None
    if drop_p > 0:
        keep = ops.dropout_mask(n * H * Lq * Lk, drop_p, seed, 'cuda').cpu().view(n, H, Lq, Lk) / (1 - drop_p)
    # reference
    qr, kvr, e1r, e2r = (t.clone().requires_grad_(True) for t in (q, kv, e1, e2))
    ctx_ref, probs_ref = ref_attention(qr, kvr[:, :d], kvr[:, d:], e1r, e2r, n, Lq, Lk, H, mask, keep)
    (ctx_ref * go).sum().backward()
    # kernels (self-attention packing when square: one (M, 3d) buffer)
    e1d, e2d = e1.cuda().requires_grad_(True), e2.cuda().requires_grad_(True)
    if cross:
        qd, kvd = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
        ctx, probs = ops.AttnXFn.apply(qd, kvd, e1d, e2d, n, Lq, Lk, H, mask, drop_p, seed)
    else:
        qkvd = torch.cat([q, kv], dim=1).cuda().requires_grad_(True)
        ctx, probs = ops.AttnXFn.apply(qkvd, None, e1d, e2d, n, Lq, Lk, H, mask, drop_p, seed)
    (ctx * go.cuda()).sum().backward()
    assert rel_err(probs.cpu(), probs_ref) < 2e-5
    assert rel_err(ctx.cpu(), ctx_ref) < 2e-5
    if mask:   # masked probabilities are exactly 0
        m = D.additive_mask(mask, Lk, Lq)
        assert float(probs.cpu()[:, :, m == float('-inf')].abs().max()) == 0.0
    if cross:
        dq, dkv = qd.grad.cpu(), kvd.grad.cpu()
    else:
        dq, dkv = qkvd.grad.cpu()[:, :d], qkvd.grad.cpu()[:, d:]
    assert rel_err(dq, qr.grad) < 1e-4
    assert rel_err(dkv, kvr.grad) < 1e-4
    assert rel_err(e1d.grad.cpu(), e1r.grad) < 1e-4
    if float(e2r.grad.abs().max()) > 0:
        assert rel_err(e2d.grad.cpu(), e2r.grad) < 1e-4
    else:
        assert float(e2d.grad.abs().max()) == 0.0          # causal + r = 1: the anticausal half is never used


def test_square_unmasked_case_equals_the_L16_matrix_core_kernel(gemm_mode):
    """r = 1, no mask, L = 16: the strip kernels (reached directly) against the one-wave-per-problem 16x16x4 MFMA kernel
    that `vqcpc_relattn_fwd` dispatches to at L = 16 -- two independent implementations of the same closed form."""
    if gemm_mode != 'f32':
        pytest.skip('attention does not depend on the GEMM mode')
    from vqcpc_bach_amd import hip, ops
    n, H, L, hd = 5, 4, 16, 32
    d = H * hd
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(n * L, 3 * d, generator=g).cuda()
    e1, e2 = torch.randn(H * L, hd, generator=g).cuda(), torch.randn(H * L, hd, generator=g).cuda()
    ctx_x, probs_x = ops.AttnXFn.apply(qkv, None, e1, e2, n, L, L, H, 0, 0.1, 99)
    ctx = torch.empty(n * L, d, device='cuda')
    probs = torch.empty(n, H, L, L, device='cuda')
    hip.call('vqcpc_relattn_fwd', qkv, 3 * d, e1, e2, ctx, d, probs, n, L, H, hd, 0.1, 99)
    assert rel_err(probs_x, probs) < 1e-5 and rel_err(ctx_x, ctx) < 1e-5


def test_embedding_fn_large_table_and_repeats(gemm_mode):
    if gemm_mode != 'f32':
        pytest.skip('no GEMM')
    from vqcpc_bach_amd import ops
    g = torch.Generator().manual_seed(3)
    for V, C, M in ((262144, 64, 3000), (929, 512, 6144), (5, 8, 1), (7, 12, 40)):
        table = torch.randn(V, C, generator=g)
        idx = torch.randint(0, min(V, 37), (M,), generator=g) * (V // min(V, 37))
        idx[0] = V - 1
        go = torch.randn(M, C, generator=g)
        td = table.cuda().requires_grad_(True)
        out = ops.EmbeddingFn.apply(td, idx.cuda())
        assert torch.equal(out.cpu(), table[idx])
        out.backward(go.cuda())
        ref = torch.zeros(V, C, dtype=torch.float64).index_add_(0, idx, go.double())
        assert rel_err(td.grad.cpu(), ref) < 1e-6
        again = ops.EmbeddingFn.apply(td, idx.cuda())
        td.grad = None
        again.backward(go.cuda())
        first = td.grad.clone()
        td.grad = None
        ops.EmbeddingFn.apply(td, idx.cuda()).backward(go.cuda())
        assert torch.equal(first, td.grad)                   # deterministic


# ---------------------------------------------------------------------------------------------------------------------
# one decoder layer against the reference fixture
# ---------------------------------------------------------------------------------------------------------------------
def test_decoder_layer_golden():
    from vqcpc_bach_amd import ops
    from vqcpc_bach_amd.transformer.transformer_custom import TransformerDecoderLayerCustom
    g = load_golden('decoder_layer_S3_T48')
    Tn, n, d = g['tgt'].shape
    S = g['mem'].shape[0]
    H = int(g['H'])
    layer = TransformerDecoderLayerCustom(d_model=d, nhead=H, attention_bias_type_self='relative_attention',
                                          attention_bias_type_cross='relative_attention_target_source',
                                          num_channels_encoder=1, num_events_encoder=S, num_channels_decoder=4,
                                          num_events_decoder=Tn // 4, dim_feedforward=g['sd/linear1.weight'].shape[0],
                                          dropout=0.0)
    layer.load_state_dict(sub_state(g, 'sd'))
    layer.cuda().eval()
    tgt = T(g['tgt']).cuda().requires_grad_(True)
    mem = T(g['mem']).cuda().requires_grad_(True)
    y, att = layer(tgt, mem, tgt_mask='causal', memory_mask='anticausal')         # API path, time-first
    assert rel_err(y.cpu(), g['y']) < FWD_TOL
    assert rel_err(att['a_self_decoder'].cpu(), g['a_self']) < FWD_TOL
    assert rel_err(att['a_cross'].cpu(), g['a_cross']) < FWD_TOL
    (y * T(g['g']).cuda()).sum().backward()
    assert rel_err(tgt.grad.cpu(), g['d_tgt']) < GRAD_TOL
    assert rel_err(mem.grad.cpu(), g['d_mem']) < GRAD_TOL
    for k, p in layer.named_parameters():
        ref = g['grad/' + k]
        if float(np.abs(ref).max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad.cpu(), ref) < GRAD_TOL, k


# ---------------------------------------------------------------------------------------------------------------------
# the training step
# ---------------------------------------------------------------------------------------------------------------------
def build_decoder(cfg, sd, lr=1e-3, dropout=0.0):
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.data_processor.bach_cpc_data_processor import BachCPCDataProcessor
    from vqcpc_bach_amd.data_processor.bach_data_processor import BachDataProcessor
    from vqcpc_bach_amd.decoders.decoder import Decoder
    from vqcpc_bach_amd.downscalers.relative_transformer_downscaler import RelativeTransformerDownscaler
    from vqcpc_bach_amd.encoder import Encoder
    from vqcpc_bach_amd.quantizer.vector_quantizer import ProductVectorQuantizer
    from vqcpc_bach_amd.upscalers.mlp_upscaler import MlpUpscaler
    hip.load()
    nc = len(cfg['vocab'])
    edp = BachCPCDataProcessor(embedding_size=cfg['emb'], num_events=(cfg['Kl'] + cfg['Kr']) * 4, num_channels=nc,
                               num_tokens_per_channel=cfg['vocab'], num_tokens_per_block=16)
    ds = RelativeTransformerDownscaler(input_dim=cfg['emb'], output_dim=cfg['D'], num_channels=nc, downscale_factors=[4, 4],
                                       d_model=cfg['d'], n_head=cfg['H'], list_of_num_layers=cfg['layers'],
                                       dim_feedforward=cfg['ff'], dropout=0.0)
    q = ProductVectorQuantizer(codebook_size=cfg['K'], codebook_dim=cfg['D'], commitment_cost=0.25,
                               num_codebooks=cfg['ncb'], use_batch_norm=False, initialize=False, squared_l2_norm=True)
    up = MlpUpscaler(input_dim=cfg['D'], output_dim=cfg['zdim'], hidden_size=cfg['up_hidden'], dropout=0.0)
    enc = Encoder('/tmp/vqcpc_test_decoder', edp, ds, q, up)
    dp = BachDataProcessor(embedding_size=cfg['dec_emb'], num_events=cfg['events'], num_tokens_per_channel=cfg['vocab'])
    S = cfg['events'] * nc // 16
    dec = Decoder(model_dir='/tmp/vqcpc_test_decoder', dataloader_generator=None, data_processor=dp, encoder=enc,
                  transformer_type='relative', encoder_attention_type=cfg['enc_attn'],
                  cross_attention_type=cfg['cross_attn'], d_model=cfg['dec_d'], num_encoder_layers=cfg['dec_enc_layers'],
                  num_decoder_layers=cfg['dec_dec_layers'], n_head=cfg['dec_H'], dim_feedforward=cfg['dec_ff'],
                  positional_embedding_size=cfg['dec_pos'], num_channels_encoder=1, num_events_encoder=S,
                  num_channels_decoder=nc, num_events_decoder=cfg['events'], dropout=dropout)
    if sd is not None:
        assert set(dec.state_dict()) == set(sd), 'state_dict keys must be the reference\'s'
        dec.load_state_dict(sd)
    dec.cuda()
    dec.init_optimizers(lr=lr, schedule_lr=False)
    assert dec.flat.check_views()
    return dec


@pytest.mark.parametrize('name', ['decoder_tiny', 'decoder_tiny_fullcross', 'decoder_tiny_clip'])
def test_decoder_epoch_golden(name):
    g = load_golden(name)
    cfg = D.make_cfg(**json.loads(str(g['cfg_json'])))
    lr = float(g['lr'])
    sd0 = sub_state(g, 'sd0')
    dec = build_decoder(cfg, sd0, lr=lr)
    batch = {'x': T(g['batch/x'])}
    nc = len(cfg['vocab'])
    # frozen encoder -> merged codes, bit-exact
    dec.eval()
    codes = dec.encode(batch['x'])
    assert torch.equal(codes.cpu(), T(g['codes'])), 'codes must be bit-exact'
    # forward (API-compatible entry)
    with torch.no_grad():
        fp = dec.forward(codes, batch['x'])
    assert abs(fp['monitored_quantities']['loss'] - float(g['eval/loss'])) < FWD_TOL * float(g['eval/loss'])
    for c in range(nc):
        assert rel_err(fp['weights_per_category'][c].cpu(), g[f'eval_fwd/logits.{c}']) < FWD_TOL
    assert rel_err(fp['attentions_decoder'][-1]['a_cross'].cpu(), g['eval_fwd/a_cross_last']) < FWD_TOL
    assert rel_err(fp['attentions_decoder'][-1]['a_self_decoder'].cpu(), g['eval_fwd/a_self_last']) < FWD_TOL
    assert rel_err(fp['attentions_encoder'][-1]['a_self_encoder'].cpu(), g['eval_fwd/a_enc_last']) < FWD_TOL
    ev = dec.epoch(iter([batch]), train=False, num_batches=1)
    assert set(ev) == {'loss'} and abs(ev['loss'] - float(g['eval/loss'])) < FWD_TOL * float(g['eval/loss'])

    # gradients BEFORE the clip: forward + backward only (plain autograd accumulation, then the in-place wgrad mode)
    from vqcpc_bach_amd import ops
    golden_grads = {k[5:]: v for k, v in g.items() if k.startswith('grad/')}
    named = {k: p for k, p in dec.named_parameters() if not k.startswith('encoder.')}
    assert set(named) == set(golden_grads)
    for direct in (False, True):
        dec.train()
        loss, _, _, _ = dec.compute_loss(codes, dec.data_processor.preprocess(batch['x']))
        dec.flat.zero_grad()
        if direct:
            with ops.direct_weight_gradients():
                loss.backward()
        else:
            loss.backward()
        assert dec.flat.check_views()
        for k, p in named.items():
            ref = golden_grads[k]
            if float(np.abs(ref).max()) == 0.0:
                assert float(p.grad.abs().max()) == 0.0, k
            else:
                assert rel_err(p.grad.cpu(), ref) < GRAD_TOL, (k, direct)
    assert all(p.grad is None for k, p in dec.named_parameters() if k.startswith('encoder.'))

    # one training epoch: loss, clip norm, parameters after Adam
    trn = dec.epoch(iter([batch]), train=True, num_batches=1)
    assert abs(trn['loss'] - float(g['train/loss'])) < FWD_TOL * float(g['train/loss'])
    gn = float(g['grad_total_norm'])
    assert abs(dec.optimizer.grad_norm() - gn) < 2e-4 * gn
    sd1 = sub_state(g, 'sd1')
    now = dec.state_dict()
    for k, v in sd1.items():
        if k.startswith('encoder.'):
            assert torch.equal(now[k].cpu(), sd0[k]), k                 # frozen
            continue
        gr = T(golden_grads[k]).abs()
        solid = gr > 1e-4 * gr.max() if float(gr.max()) > 0 else torch.zeros_like(gr, dtype=torch.bool)
        mine, ref = now[k].cpu() - sd0[k], v - sd0[k]
        if bool(solid.any()):
            assert rel_err(mine[solid], ref[solid]) < 2e-2, k
        assert float((mine - ref).abs().max()) <= 2.0001 * lr, k       # noise-sign elements move by at most lr each way


def seeded_decoder(cfg, seed):
    """Random decoder + encoder whose codebooks sit on encoder outputs; returns (Decoder, oracle state dict)."""
    torch.manual_seed(seed)
    dec = build_decoder(cfg, None)
    enc = dec.encoder
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in dec.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g).cuda())
        probe = torch.cat([torch.randint(0, nv, (8 * cfg['K'], 4, 1), generator=g) for nv in cfg['vocab']], dim=2)
        t = enc.data_processor.preprocess(probe)
        z = enc.downscaler.forward_tokens(t.unsqueeze(0), enc.data_processor)[0].reshape(-1, cfg['D'])
        dsub = cfg['D'] // cfg['ncb']
        for c, e in enumerate(enc.quantizer.embeddings):
            e.copy_(z[c:c + 8 * cfg['K']:8, c * dsub:(c + 1) * dsub]
                    + 0.01 * torch.randn(cfg['K'], dsub, generator=g).cuda())
    sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    return dec, sd


def test_decoder_step_matches_oracle_at_shipped_shapes(gemm_mode):
    """configs/decoder_relative_AC_AC_C_random.py shapes: 48 ticks x 4 voices = 192 target tokens, 12 codes,
    d_model 512, 4 heads (head_dim 128), 3 + 3 layers, ff 1024; batch 4 keeps the CPU oracle at a few seconds."""
    cfg = D.make_cfg(vocab=[40, 44, 48, 52], emb=16, d=64, H=4, layers=[1, 1], ff=128, D=16, K=16, ncb=1, zdim=16,
                     up_hidden=32, Kl=2, Kr=2, events=48, B=4)
    dec, sd = seeded_decoder(cfg, 123)
    g = torch.Generator().manual_seed(9)
    x = torch.cat([torch.randint(0, nv, (cfg['B'], cfg['events'], 1), generator=g) for nv in cfg['vocab']], dim=2)
    oracle = D.DecoderOracleTrainer(cfg, sd, lr=1e-3)
    ref = oracle.step({'x': x}, train=True)
    dec.eval()
    codes = dec.encode(x)
    assert torch.equal(codes.cpu(), ref['codes'])
    assert len(torch.unique(codes)) > 4
    dec.train()
    xd = dec.data_processor.preprocess(x)
    loss, logits, _, _ = dec.compute_loss(codes, xd)
    assert abs(float(loss.detach()) - float(ref['loss'].detach())) < FWD_TOL * float(ref['loss'].detach())
    for c in range(4):
        assert rel_err(logits[c].cpu(), ref['logits'][c].detach()) < FWD_TOL
    dec.flat.zero_grad()
    loss.backward()
    tol = GRAD_TOL if gemm_mode == 'f32' else 2 * GRAD_TOL
    for k, p in dec.named_parameters():
        if k.startswith('encoder.'):
            continue
        r = oracle.last_grads[k]
        if float(r.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad.cpu(), r) < tol, k


def test_target_stream_is_causal_and_memory_access_anticausal(gemm_mode):
    """Size-independent properties of the masks.  (1) Logits at position t depend on target tokens < t only (shift by
    one + causal mask), for any depth.  (2) With ONE decoder layer, logits at position t depend on codes >= t // 16 only
    (anticausal source stack + anticausal cross-attention); deeper decoders legitimately leak earlier codes forward
    through the causal self-attention, so (2) is checked on a one-layer decoder."""
    if gemm_mode != 'f32':
        pytest.skip('one GEMM mode is enough')
    nc = 4
    g = torch.Generator().manual_seed(4)
    x = torch.cat([torch.randint(0, nv, (2, 24, 1), generator=g) for nv in (20, 21, 22, 23)], dim=2).cuda()
    codes = torch.randint(0, 16, (2, 6), generator=g).cuda()

    def model(dec_layers):
        cfg = D.make_cfg(vocab=[20, 21, 22, 23], emb=16, d=32, H=2, layers=[1, 1], ff=64, D=8, K=16, ncb=1, zdim=8,
                         up_hidden=16, Kl=2, Kr=2, events=24, B=2, dec_d=64, dec_H=2, dec_enc_layers=2,
                         dec_dec_layers=dec_layers, dec_ff=128)
        dec, _ = seeded_decoder(cfg, 321)
        dec.eval()

        def all_logits(codes_, x_):
            with torch.no_grad():
                _, logits, _, _ = dec.compute_loss(codes_, x_)
            # nc x (B, events, V_c) -> position-major (B, T, Vmin)
            return torch.stack([lg[..., :20] for lg in logits], dim=2).reshape(2, 24 * nc, 20)
        return all_logits

    deep = model(2)
    base = deep(codes, x)
    t0 = 37                                             # token (event 9, voice 1)
    x2 = x.clone()
    x2[:, t0 // nc, t0 % nc] = (x2[:, t0 // nc, t0 % nc] + 1) % 20
    pert = deep(codes, x2)
    assert torch.equal(pert[:, :t0 + 1], base[:, :t0 + 1])          # positions <= t0 never see token t0
    assert not torch.equal(pert[:, t0 + 1:], base[:, t0 + 1:])

    shallow = model(1)
    base = shallow(codes, x)
    c0 = 2
    codes2 = codes.clone()
    codes2[:, c0] = (codes2[:, c0] + 1) % 16
    pert = shallow(codes2, x)
    first_blind = (c0 + 1) * 16                          # queries aligned with codes > c0 never see code c0
    assert torch.equal(pert[:, first_blind:], base[:, first_blind:])
    assert not torch.equal(pert[:, :first_blind], base[:, :first_blind])


def test_decoder_learns_and_checkpoint_roundtrip(tmp_path, gemm_mode):
    if gemm_mode != 'bf16x6':
        pytest.skip('default GEMM mode only')
    cfg = D.make_cfg(vocab=[12, 12, 12, 12], emb=16, d=32, H=2, layers=[1, 1], ff=64, D=8, K=16, ncb=1, zdim=8, up_hidden=16,
                     Kl=2, Kr=2, events=16, B=8, dec_d=64, dec_H=2, dec_enc_layers=1, dec_dec_layers=2, dec_ff=128)
    torch.manual_seed(0)
    dec, _ = seeded_decoder(cfg, 11)
    dec.model_dir = str(tmp_path)
    dec.init_optimizers(lr=2e-3, schedule_lr=True)
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randint(0, 12, (8, 16, 1), generator=g) for _ in range(4)], dim=2)
    data = [{'x': x}] * 40
    first = dec.epoch(iter(data[:1]), train=False, num_batches=1)['loss']
    dec.epoch(iter(data), train=True, num_batches=40)
    last = dec.epoch(iter(data[:1]), train=False, num_batches=1)['loss']
    assert last < 0.8 * first, (first, last)             # memorises one batch
    dec.save(early_stopped=False)
    dec2, _ = seeded_decoder(cfg, 12)
    dec2.model_dir = str(tmp_path)
    dec2.load(early_stopped=False, device='cuda')
    dec2.init_optimizers(lr=2e-3, schedule_lr=True)
    assert dec2.global_step == dec.global_step == 40 and dec2.optimizer.step_count == 40
    assert abs(dec2.epoch(iter(data[:1]), train=False, num_batches=1)['loss'] - last) < 1e-6
    a = dec.epoch(iter(data[:1]), train=True, num_batches=1)['loss']
    b = dec2.epoch(iter(data[:1]), train=True, num_batches=1)['loss']
    assert abs(a - b) < 1e-6
    for (k, p), (_, p2) in zip(dec.named_parameters(), dec2.named_parameters()):
        assert torch.equal(p, p2), k                      # resumed run continues bit-identically


def test_new_entry_points_reject_bad_arguments_and_accept_empty_inputs(gemm_mode):
    if gemm_mode != 'f32':
        pytest.skip('no GEMM')
    from vqcpc_bach_amd import hip
    d = 64
    q = torch.randn(48, d, device='cuda')
    kv = torch.randn(4, 2 * d, device='cuda')
    e = torch.randn(2 * 4, 32, device='cuda')
    ctx, probs = torch.empty(48, d, device='cuda'), torch.empty(1, 2, 48, 4, device='cuda')
    hip.call('vqcpc_relattn_x_fwd', q, d, kv, 2 * d, kv[:, d:], 2 * d, e, e, ctx, d, probs, 0, 48, 4, 2, 32, 2, 0.0, 0)   # n_seq = 0
    with pytest.raises(hip.VqcpcHipError, match='unsupported Lq'):          # Lq not a multiple of Lk
        hip.call('vqcpc_relattn_x_fwd', q, d, kv, 2 * d, kv[:, d:], 2 * d, e, e, ctx, d, probs, 1, 46, 4, 2, 32, 2, 0.0, 0)
    with pytest.raises(hip.VqcpcHipError, match='mask must be'):
        hip.call('vqcpc_relattn_x_fwd', q, d, kv, 2 * d, kv[:, d:], 2 * d, e, e, ctx, d, probs, 1, 48, 4, 2, 32, 3, 0.0, 0)
    with pytest.raises(hip.VqcpcHipError, match='workspace too small'):
        hip.call('vqcpc_relattn_x_bwd', ctx, d, q, d, kv, 2 * d, kv[:, d:], 2 * d, probs, e, e, torch.empty_like(q), d,
                 torch.empty_like(kv), 2 * d, torch.empty_like(kv)[:, d:], 2 * d, torch.empty_like(e), torch.empty_like(e), 1,
                 48, 4, 2, 32, 2, 0.0, 0, torch.empty(16, device='cuda'), 16)
    table = torch.zeros(5, 8, device='cuda').fill_(7.0)
    idx = torch.empty(0, dtype=torch.int64, device='cuda')
    hip.call('vqcpc_embedding_bwd', None, 8, idx, idx, table, 0, 5, 8)      # M = 0: the table gradient is all zeros
    assert float(table.abs().max()) == 0.0
    torch.cuda.synchronize()


def _run_bench(args, env_extra):
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert sum(l.startswith('{') for l in lines) == 1, (out.stdout[-1500:], out.stderr[-1500:])   # exactly ONE JSON line ...
    return json.loads(lines[-1])                    # ... and it is the last thing on stdout (after RCCL's printf banner)


@pytest.mark.parametrize('config,batch', [('DEC', 4), ('C0', 8)])
def test_bench_contract_through_single_rank_rccl(config, batch, gemm_mode):
    """The N > 1 code path on one GPU: VQCPC_FORCE_DIST=1 makes the trainer create a one-rank RCCL group, so weights are
    broadcast and the flat gradient is all-reduced exactly as with 8 ranks; bench.py must still print its one JSON line."""
    if gemm_mode != 'bf16x6':
        pytest.skip('the subprocess uses the default GEMM mode')
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    line = _run_bench(['--config', config, '--batch', str(batch), '--steps', '3', '--warmup', '1', '--no-cpu-baseline'],
                      dict(VQCPC_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
                           LOCAL_RANK='0'))
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 3 and line['value'] > 0 and np.isfinite(line['final_loss'])
    assert line['roofline']['bound'] in ('mfma', 'hbm') and 0 < line['roofline']['frac'] < 1 and 0 < line['roofline']['frac_of_binding'] < 1


def test_decoder_full_config_through_getters_vs_oracle(gemm_mode):
    """The shipped configuration itself (configs.make_config('DEC'): 384 target tokens, 24 codes, d_model 512, 8 heads of
    64 -> the four-waves-per-strip attention kernels, 3 + 3 layers, frozen d_model-512 encoder), built through the
    getters exactly as bench.py does, at batch 4: codes bit-exact, loss / logits / gradients against the oracle."""
    from vqcpc_bach_amd import configs, getters
    config = configs.make_config('DEC', dropout=0.0)
    torch.manual_seed(5)
    dlg = getters.get_dataloader_generator(config['dataset'], config['training_method'],
                                           dict(config['dataloader_generator_kwargs'], seed=7, device='cuda'))
    enc_cfg = config['config_encoder']
    enc_cfg['downscaler_kwargs']['dropout'] = 0.0
    enc_dlg = getters.get_dataloader_generator(enc_cfg['dataset'], enc_cfg['training_method'],
                                               dict(enc_cfg['dataloader_generator_kwargs'], seed=7, device='cuda'))
    encoder = getters.get_encoder('/tmp/vqcpc_test_decoder_full', enc_dlg, enc_cfg)
    dp = getters.get_data_processor(dlg, config['data_processor_type'], config['data_processor_kwargs'])
    dec = getters.get_decoder('/tmp/vqcpc_test_decoder_full', dlg, dp, encoder, config['decoder_type'], config['decoder_kwargs'])
    dec.cuda()
    dec.init_optimizers(lr=config['lr'], schedule_lr=config['schedule_lr'])
    sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    cfg = D.make_cfg('DEC', vocab=list(dlg.vocab), B=4)
    assert set(D.init_state(cfg)) == set(sd)                       # the oracle's view of the model has the same keys
    x = next(dlg.dataloaders(batch_size=4)[0])['x']
    oracle = D.DecoderOracleTrainer(cfg, sd, lr=config['lr'])
    ref = oracle.step({'x': x.cpu()}, train=True)
    dec.eval()
    codes = dec.encode(x)
    assert torch.equal(codes.cpu(), ref['codes'])
    dec.train()
    loss, logits, _, _ = dec.compute_loss(codes, dec.data_processor.preprocess(x))
    assert abs(float(loss.detach()) - float(ref['loss'].detach())) < FWD_TOL * float(ref['loss'].detach())
    for c in range(4):
        assert rel_err(logits[c].cpu(), ref['logits'][c]) < 2 * FWD_TOL
    dec.flat.zero_grad()
    loss.backward()
    worst = 0.0
    for k, p in dec.named_parameters():
        if k.startswith('encoder.'):
            continue
        r = oracle.last_grads[k]
        if float(r.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            worst = max(worst, rel_err(p.grad.cpu(), r))
            assert rel_err(p.grad.cpu(), r) < 2 * GRAD_TOL, k
    # one full training step through the epoch API, twice from the same state -> bit-identical (deterministic kernels)
    a = dec.epoch(iter([{'x': x}]), train=False, num_batches=1)['loss']
    b = dec.epoch(iter([{'x': x}]), train=False, num_batches=1)['loss']
    assert a == b


def test_decoder_full_config_with_f16x3_small_tile_products(gemm_mode):
    """Round 6: the decoder step the way train_model() runs it -- forward inside ops.forward_arithmetic, backward inside the gradient
    scope, the 64 x 128-tile three-product kernel for its 1 536-row products (batch 4 x 384 tokens) -- against the oracle at the
    tolerances of the test above."""
    if gemm_mode != 'bf16x6':
        pytest.skip('an arithmetic of the bf16x6 mode')
    from vqcpc_bach_amd import configs, getters, hip, ops
    config = configs.make_config('DEC', dropout=0.0)
    torch.manual_seed(5)
    dlg = getters.get_dataloader_generator(config['dataset'], config['training_method'],
                                           dict(config['dataloader_generator_kwargs'], seed=7, device='cuda'))
    enc_cfg = config['config_encoder']
    enc_cfg['downscaler_kwargs']['dropout'] = 0.0
    enc_dlg = getters.get_dataloader_generator(enc_cfg['dataset'], enc_cfg['training_method'],
                                               dict(enc_cfg['dataloader_generator_kwargs'], seed=7, device='cuda'))
    encoder = getters.get_encoder('/tmp/vqcpc_test_decoder_full3', enc_dlg, enc_cfg)
    dp = getters.get_data_processor(dlg, config['data_processor_type'], config['data_processor_kwargs'])
    dec = getters.get_decoder('/tmp/vqcpc_test_decoder_full3', dlg, dp, encoder, config['decoder_type'], config['decoder_kwargs'])
    dec.cuda()
    dec.init_optimizers(lr=config['lr'], schedule_lr=config['schedule_lr'])
    sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    cfg = D.make_cfg('DEC', vocab=list(dlg.vocab), B=4)
    x = next(dlg.dataloaders(batch_size=4)[0])['x']
    oracle = D.DecoderOracleTrainer(cfg, sd, lr=config['lr'])
    ref = oracle.step({'x': x.cpu()}, train=True)
    dec.eval()
    codes = dec.encode(x)
    assert torch.equal(codes.cpu(), ref['codes'])
    dec.train()
    prev = ops.set_gradient_arithmetic('f16x3')
    saved = ops.FWD_ARITH, ops.SMALL_F16X3_MIN_TILES
    ops.FWD_ARITH, ops.SMALL_F16X3_MIN_TILES = 'f16x3', 0
    calls, raw = [], hip.call
    hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
    try:
        with ops.forward_arithmetic(dec.flat):
            loss, logits, _, _ = dec.compute_loss(codes, dec.data_processor.preprocess(x))
        dec.flat.zero_grad()
        with ops.direct_weight_gradients(dec.flat):
            loss.backward()
    finally:
        hip.call = raw
        ops.FWD_ARITH, ops.SMALL_F16X3_MIN_TILES = saved
        ops.set_gradient_arithmetic(prev)
    assert calls.count('vqcpc_gemm_nt_g3_small') >= 30, calls.count('vqcpc_gemm_nt_g3_small')
    assert abs(float(loss.detach()) - float(ref['loss'].detach())) < FWD_TOL * float(ref['loss'].detach())
    for c in range(4):
        assert rel_err(logits[c].cpu(), ref['logits'][c]) < 2 * FWD_TOL
    for k, p in dec.named_parameters():
        if k.startswith('encoder.'):
            continue
        r = oracle.last_grads[k]
        if float(r.abs().max()) == 0.0:
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel_err(p.grad.cpu(), r) < 2 * GRAD_TOL, k


@pytest.mark.parametrize('name', ['mha_self_causal_T24', 'mha_self_full_T16', 'mha_cross_anticausal_S6_T12'])
def test_multihead_attention_forward_api_against_the_reference(name, gemm_mode):
    """`MultiheadAttentionCustom.forward(query, key, value, attn_mask=...)` -- the reference's own signature
    (multihead_attention_custom.py:122), time-first tensors and the ADDITIVE mask matrices of decoders/decoder.py:294-308 --
    against fixtures produced by the reference's module: output, per-head attention weights, gradients w.r.t. the inputs and
    every parameter."""
    from vqcpc_bach_amd.transformer.multihead_attention_custom import MultiheadAttentionCustom
    g = load_golden(name)
    H, S, Tq = int(g['H']), int(g['S']), int(g['T'])
    q0 = T(g['q'])
    n, d = q0.shape[1], q0.shape[2]
    cross = S != Tq
    m = MultiheadAttentionCustom(embed_dim=d, num_heads=H,
                                 attention_bias_type='relative_attention_target_source' if cross else 'relative_attention',
                                 num_channels_k=1, num_events_k=S, num_channels_q=1, num_events_q=Tq, dropout=0.0)
    m.load_state_dict({k[3:]: T(np.array(v)) for k, v in g.items() if k.startswith('sd/')})
    m.cuda().eval()
    q = q0.cuda().requires_grad_(True)
    mem = T(g['mem']).cuda().requires_grad_(True) if cross else q
    am = T(g['attn_mask']).cuda() if 'attn_mask' in g else None
    out, w = m(q, mem, mem, attn_mask=am)
    assert out.shape == (Tq, n, d) and w.shape == (n, H, Tq, S)
    assert rel_err(out.detach().cpu(), g['out']) < FWD_TOL
    assert rel_err(w.detach().cpu(), g['weights']) < FWD_TOL
    if am is not None:
        assert float(w.detach().cpu()[:, :, torch.isinf(T(g['attn_mask']))].abs().max()) == 0.0     # masked weights exactly 0
    (out * T(g['g']).cuda()).sum().backward()
    assert rel_err(q.grad.cpu(), g['d_q']) < GRAD_TOL
    if cross:
        assert rel_err(mem.grad.cpu(), g['d_mem']) < GRAD_TOL
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu(), g[f'grad/{k}']) < GRAD_TOL, k
    assert m(q.detach(), mem.detach(), mem.detach(), attn_mask=am, need_weights=False)[1] is None
    with pytest.raises(NotImplementedError):
        m(q.detach(), mem.detach(), mem.detach(), attn_mask=torch.randn(Tq, S, device='cuda'))
