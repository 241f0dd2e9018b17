"""SURVEY.md section 8 rows A9 (at the C1 quantiser shape, against the reference's own output), A11 (data-dependent
codebook initialisation, vector_quantizer.py:57-70) and A12 (label corruption, :119-132), plus the quantizer_type=None
branch of getters.get_encoder (NoQuantization) and the asynchronous token range check.

Integer results are bit-exact; the only floating-point comparison (the loss) states its tolerance."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import vqcpc_oracle as O
from test_trainer_gpu import build_trainer

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _quantizer(K, D, ncb, initialize):
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.quantizer.vector_quantizer import ProductVectorQuantizer
    hip.load()
    return ProductVectorQuantizer(codebook_size=K, codebook_dim=D, commitment_cost=0.25, num_codebooks=ncb,
                                  use_batch_norm=False, initialize=initialize, squared_l2_norm=True).cuda()


def test_c1_quantizer_shape_with_data_init_matches_the_reference():
    """2 x 512 codes of dim 16 on 4096 rows, run by the reference with initialize=True under torch.manual_seed(1234):
    the product, seeded the same way, initialises the SAME codebooks (bit for bit: they are copies of input rows) and
    assigns the same 8192 indices; loss within 2e-5."""
    g = load_golden('vq_c1_init')
    z = T(g['z']).cuda()
    q = _quantizer(K=512, D=32, ncb=2, initialize=True)
    q.eval()
    torch.manual_seed(int(g['init_seed']))
    with torch.no_grad():
        zq, idx, loss = q(z)
    assert not q.initialize
    assert torch.equal(torch.stack([e.detach() for e in q.embeddings]).cpu(), T(g['codebooks']))
    assert torch.equal(idx.cpu(), T(g['idx'].astype(np.int64))), f'min top-2 gap {g["top2_gap"].min():.3e}'
    assert rel_err(loss.cpu(), g['loss']) < 2e-5
    # straight-through output == inputs + (codebook[idx] - inputs), the reference's fp32 expression
    cb = T(g['codebooks'])
    quant = torch.cat([cb[c][T(g['idx'].astype(np.int64))[..., c]] for c in range(2)], dim=-1)
    assert torch.equal(zq.cpu(), T(g['z']) + (quant - T(g['z'])))


def test_data_init_rows_come_from_the_first_segment_and_match_the_oracle():
    """A11 through the merged encoder pass: `init_rows` restricts the candidates to the first (negatives) segment, as the
    reference's first encoder call does (vqcpc_encoder_trainer.py:201); semantics == oracle.vq_data_init."""
    gen = torch.Generator().manual_seed(5)
    flat = torch.randn(700, 24, generator=gen)
    n_first = 300
    q = _quantizer(K=128, D=24, ncb=3, initialize=True)
    q.eval()
    torch.manual_seed(77)
    with torch.no_grad():
        q(flat.cuda(), init_rows=slice(0, n_first))
    torch.manual_seed(77)
    ref = O.vq_data_init(flat[:n_first], [torch.empty(128, 8) for _ in range(3)])
    got = [e.detach().cpu() for e in q.embeddings]
    for c in range(3):
        assert torch.equal(got[c], ref[c]), c
        # every code is the c-th sub-vector of a DISTINCT row of the first segment
        rows = [(flat[:n_first, c * 8:(c + 1) * 8] == got[c][k]).all(1).nonzero().flatten().tolist() for k in range(128)]
        assert all(len(r) == 1 for r in rows) and len({r[0] for r in rows}) == 128
    with pytest.raises(AssertionError, match='not enough elements'):
        q2 = _quantizer(K=128, D=24, ncb=3, initialize=True)
        q2(flat[:100].cuda())


def test_label_corruption_rate_scope_and_lookup():
    """A12: in training mode ~5 % of the indices of the corrupted rows are replaced by uniform random codes, rows outside
    `corrupt_rows` (the left / right positives, vqcpc_encoder_trainer.py:227-231) are untouched, the quantised output is
    the codebook row of the GIVEN index, and eval mode never corrupts."""
    from vqcpc_bach_amd import ops
    K, D, ncb, R = 64, 32, 2, 40000
    gen = torch.Generator().manual_seed(9)
    z = torch.randn(R, D, generator=gen).cuda()
    q = _quantizer(K=K, D=D, ncb=ncb, initialize=False)
    with torch.no_grad():
        for e in q.embeddings:
            e.copy_(torch.randn(K, D // ncb, generator=gen))
    cb = torch.stack([e.detach() for e in q.embeddings])
    clean = ops.vq_assign(z, cb)
    assert torch.equal(clean.cpu(), O.vq_assign(z.cpu(), [e.cpu() for e in cb]))
    n_neg = 30000
    rows = torch.arange(n_neg, device='cuda')
    q.train()
    torch.manual_seed(3)
    zq, idx, loss = q(z, corrupt_labels=True, corrupt_rows=rows)
    changed = (idx != clean)
    assert not bool(changed[n_neg:].any()), 'positives must never be corrupted'
    # P(entry differs) = 0.05 * (1 - 1/K); 60 000 Bernoulli draws: +-5 sigma
    p = 0.05 * (1 - 1 / K)
    n = n_neg * ncb
    assert abs(int(changed[:n_neg].sum()) - n * p) < 5 * (n * p * (1 - p)) ** 0.5, int(changed[:n_neg].sum())
    # replaced codes are uniform over the codebook: chi-square-ish bound on the histogram of the replacements
    rep = idx[:n_neg][changed[:n_neg]]
    hist = torch.bincount(rep, minlength=K).float()
    assert float(hist.min()) > 0 and float(hist.max()) < 3.0 * float(hist.mean())
    # the output is the codebook row of the given (possibly corrupted) index, straight-through form
    quant = torch.cat([cb[c][idx[:, c]] for c in range(ncb)], dim=1)
    assert torch.equal(zq.detach(), z + (quant - z))
    ref_loss = ((quant - z) ** 2).sum(1) * 1.25
    assert rel_err(loss.detach().cpu(), ref_loss.cpu()) < 2e-6
    # corrupt_rows=None corrupts every row (Encoder.forward(x, corrupt_labels=True) on a single tensor)
    torch.manual_seed(4)
    _, idx_all, _ = q(z, corrupt_labels=True)
    frac = float((idx_all != clean).float().mean())
    assert 0.04 < frac < 0.06, frac
    q.eval()
    _, idx_eval, _ = q(z, corrupt_labels=True)
    assert torch.equal(idx_eval, clean)


def test_corrupted_step_gradients_flow_to_the_given_codes():
    """Training step with corrupt_labels=True: the codebook gradient of the q_latent term lands on the GIVEN codes."""
    K, D, ncb, R = 16, 8, 1, 512
    gen = torch.Generator().manual_seed(1)
    z = torch.randn(R, D, generator=gen).cuda().requires_grad_(True)
    q = _quantizer(K=K, D=D, ncb=ncb, initialize=False)
    q.train()
    torch.manual_seed(8)
    zq, idx, loss = q(z, corrupt_labels=True)
    loss.sum().backward()
    cb = q.embeddings[0].detach()
    quant = cb[idx[:, 0]]
    want = torch.zeros_like(cb).index_add_(0, idx[:, 0], 2.0 * (quant - z.detach()))
    assert rel_err(q.embeddings[0].grad.cpu(), want.cpu()) < 1e-5
    assert rel_err(z.grad.cpu(), (-0.25 * 2.0 * (quant - z.detach())).cpu()) < 1e-5


def test_no_quantization_encoder_trains_and_reports_zero_codewords():
    """quantizer_type=None (the reference's encoder_*_no_quantization configs, getters.py:150-153): the quantizer returns
    encoding_indices None and epoch() leaves the codeword counters at 0 (vqcpc_encoder_trainer.py:325)."""
    from vqcpc_bach_amd import configs, getters
    config = configs.make_config('C0', dropout=0.1)
    config['quantizer_type'] = None
    dlg = getters.get_dataloader_generator('bach', 'vqcpc', dict(config['dataloader_generator_kwargs'], device='cuda'))
    enc = getters.get_encoder('/tmp/vqcpc_test_noq', dlg, config)
    tr = getters.get_encoder_trainer('/tmp/vqcpc_test_noq', dlg, 'vqcpc', enc, config['auxiliary_networks_kwargs'])
    tr.to('cuda')
    tr.init_optimizers(lr=1e-4, schedule_lr=False)
    gen_train, _, _ = dlg.dataloaders(batch_size=8)
    m = tr.epoch(gen_train, train=True, num_batches=2, corrupt_labels=False)
    assert m['num_codewords'] == 0.0 and m['num_codewords_negative'] == 0.0 and m['loss_quantize'] == 0.0
    assert np.isfinite(m['loss']) and len(m['accuracy']) == dlg.num_blocks_right
    z, idx, ql = enc(next(gen_train)['x_left'])
    assert idx is None and float(ql.abs().max()) == 0.0


def test_out_of_range_token_raises_like_nn_embedding():
    """A token id outside its voice's table must not be used as an address: the kernels consume a clamped copy and the
    epoch raises IndexError at its (single) host synchronisation."""
    cfg = O.make_cfg(emb=16, vocab=[30, 20, 30, 30], d=64, H=4, layers=[1, 1], ff=128, D=16, K=16, ncb=1, zdim=16,
                     up_hidden=32, cdim=16, gru_hidden=32, B=4, N=3, Kl=2, Kr=2)
    sd = O.init_state(cfg, seed=1)
    tr = build_trainer(cfg, sd)
    batch = O.synthetic_batch(cfg, seed=2)
    batch = {k: v.clamp(max=19) for k, v in batch.items()}
    ok = tr.epoch(iter([batch]), train=True, num_batches=1, corrupt_labels=False)
    assert np.isfinite(ok['loss'])
    bad = {k: v.clone() for k, v in batch.items()}
    bad['x_right'][1, 2, 1] = 21            # voice 1 has 20 + 1 (mask) rows: 21 is outside
    with pytest.raises(IndexError, match='out of range'):
        tr.epoch(iter([bad]), train=True, num_batches=1, corrupt_labels=False)
    assert bool(torch.isfinite(tr.flat.flat).all()), 'the bad step used clamped ids: no wild write, parameters intact'
    neg = {k: v.clone() for k, v in batch.items()}
    neg['negative_samples'][0, 0, 0, 0, 0] = -1
    with pytest.raises(IndexError):
        tr.epoch(iter([neg]), train=False, num_batches=1, corrupt_labels=False)
    again = tr.epoch(iter([batch]), train=False, num_batches=1, corrupt_labels=False)       # flag was cleared
    assert np.isfinite(again['loss'])
