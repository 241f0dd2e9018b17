"""Pins oracle/student_oracle.py (CPU restatement of StudentEncoderTrainer.epoch) against the
fixtures that tools/gen_golden_student.py produced by running the reference itself.  CPU only."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, sub_state
from oracle import student_oracle as S
from oracle import vqcpc_oracle as O

T = torch.from_numpy
FWD_TOL = 2e-5
GRAD_TOL = 2e-4


def student_state(g, tag='sd0'):
    """{'sd0/teacher/x': arr} -> {'teacher.x': tensor} for the three checkpoint groups."""
    sd = {}
    for grp in ('encoder', 'teacher', 'auxiliary_decoder'):
        for k, v in sub_state(g, f'{tag}/{grp}').items():
            sd[f'{grp}.{k}'] = v
    return sd


def test_relative_bias_closed_form_at_L24():
    g = load_golden('relbias_L24')
    q, e1, e2 = T(g['q']), T(g['e1']), T(g['e2'])
    H = int(g['H'])
    n = q.shape[0] // H
    got = O.relative_bias(q.view(n, H, q.shape[1], q.shape[2]), e1, e2).reshape(g['bias'].shape)
    assert rel_err(got, g['bias']) < 1e-6


def test_cross_entropy_helpers():
    g = load_golden('student_ce')
    nc = g['target'].shape[2]
    value = [T(g[f'value.{c}']) for c in range(nc)]
    teacher = [T(g[f'teacher.{c}']) for c in range(nc)]
    ce = S.categorical_crossentropy(value, T(g['target']), T(g['mask']))
    assert ce.shape == g['ce'].shape and rel_err(ce, g['ce']) < 1e-6
    dce = S.distilled_categorical_crossentropy(value, teacher, T(g['mask']))
    assert dce.shape == g['dce'].shape and rel_err(dce, g['dce']) < 1e-6


def test_init_state_matches_reference_keys_and_shapes():
    g = load_golden('student_tiny')
    cfg = json.loads(str(g['cfg_json']))
    ref = student_state(g)
    mine = S.init_state(S.make_cfg(**cfg))
    assert set(mine) == set(ref)
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k


def test_mask_teacher_window_is_clipped_at_both_ends():
    x = torch.arange(2 * 10 * 4).view(2, 10, 4) % 5
    vocab = [5, 6, 7, 8]
    for m, lo, hi in ((0, 0, 3), (9, 7, 10), (4, 2, 7)):
        masked, notes = S.mask_teacher(x, m, 2, vocab)
        assert notes.sum() == 2 * 4 and bool(notes[:, m].all())
        assert torch.equal(masked[:, lo:hi], torch.tensor(vocab).view(1, 1, 4).expand(2, hi - lo, 4))
        keep = torch.ones(10, dtype=torch.bool)
        keep[lo:hi] = False
        assert torch.equal(masked[:, keep], x[:, keep])


@pytest.mark.parametrize('name', ['student_tiny', 'student_tiny_clip'])
def test_student_epoch_eval_and_train(name):
    g = load_golden(name)
    cfg = S.make_cfg(**json.loads(str(g['cfg_json'])))
    batch = {'x': T(g['batch/x'])}
    tr = S.StudentOracleTrainer(cfg, student_state(g), lr=float(g['lr']))
    nc = len(cfg['vocab'])

    # eval: the masked event is the first draw of the global CPU generator, as in the reference
    torch.manual_seed(int(g['eval_seed']))
    out = tr.step(batch, train=False)
    assert out['masked_event_index'] == int(g['eval_masked_event_index'])
    assert torch.equal(out['idx'], T(g['eval_fwd/idx']))
    assert rel_err(out['z'], g['eval_fwd/z']) < FWD_TOL
    assert rel_err(out['zq'], g['eval_fwd/zq']) < FWD_TOL
    assert rel_err(out['qloss'], g['eval_fwd/qloss']) < 1e-4
    for c in range(nc):
        assert rel_err(out['teacher_logits'][c], g[f'eval_fwd/teacher_logits.{c}']) < FWD_TOL
        assert rel_err(out['student_logits'][c], g[f'eval_fwd/student_logits.{c}']) < FWD_TOL
    torch.manual_seed(int(g['eval_seed']))
    ev = tr.epoch([batch], train=False, num_batches=1)
    torch.manual_seed(int(g['train_seed']))
    trn = tr.epoch([batch], train=True, num_batches=1)
    for tag, res in (('eval', ev), ('train', trn)):
        assert set(res) == {'loss_teacher', 'loss_quantization', 'loss_reconstruction', 'loss_encdec', 'loss_monitor'}
        for k, v in res.items():
            ref = float(g[f'{tag}/{k}'])
            assert abs(v - ref) < 2e-5 * max(1.0, abs(ref)), (tag, k, v, ref)

    # gradients before the three clips
    for k, gr in tr.last_grads.items():
        ref = g.get('grad/' + k)
        if ref is None:
            assert float(gr.abs().max()) == 0.0, k
            continue
        assert rel_err(gr, ref) < GRAD_TOL, k
    norms = dict(zip(('teacher.', 'auxiliary_decoder.', 'encoder.'), g['grad_norms_teacher_decoder_encoder']))
    for grp, ref in norms.items():
        assert abs(float(tr.last_grad_norms[grp]) - ref) < 1e-4 * ref, grp
    if name == 'student_tiny_clip':
        assert all(v > 5.0 for v in norms.values())

    # parameters after clip + one Adam step of each optimiser
    lr = float(g['lr'])
    after = student_state(g, 'sd1')
    for k, ref in after.items():
        got = tr.P[k].detach()
        gref = g.get('grad/' + k)
        if gref is None:
            assert torch.equal(got, ref), k
            continue
        grp = next(p for p in norms if k.startswith(p))
        coef = min(1.0, 5.0 / (norms[grp] + 1e-6))
        sig = T(np.abs(gref) * coef > 1e-5)
        assert float((got - ref).abs().max()) <= 1.01 * lr + 1e-7, k
        if sig.any():
            assert float((got - ref)[sig].abs().max()) < 2e-3 * lr + 1e-7, k


def test_c3_configuration_matches_the_reference_config_file():
    """encoder_student_config.py:11-98 -> cfg 'C3' (BASELINE.json configs[3])."""
    cfg = S.make_cfg('C3')
    assert (cfg['ticks'], cfg['d'], cfg['H'], cfg['ff']) == (96, 512, 8, 2048)
    assert cfg['enc_layers'] == [4, 4] and cfg['dec_layers'] == [4, 4] and cfg['teacher_layers'] == 8
    assert (cfg['K'], cfg['D'], cfg['ncb'], cfg['B'], cfg['qw'], cfg['num_events_masked']) == (32, 3, 1, 8, 0.1, 4)


@pytest.mark.parametrize('name', ['negatives_same_seq', 'negatives_same_seq_uneven'])
def test_same_sequence_negatives_restatement(name):
    """oracle restatement of _build_negatives_sameSeq against the reference's own output."""
    g = load_golden(name)
    xl, xr = T(g['x_left']), T(g['x_right'])
    neg = O.same_sequence_negatives(xl, xr)
    assert neg.dtype == torch.int64 and torch.equal(neg, T(g['negative_samples']))
    Kl, Kr = xl.shape[1] // 4, xr.shape[1] // 4
    assert neg.shape == (xl.shape[0], Kl + Kr - 1, Kr, 4, 4)
    if 'negative_samples_back' in g:
        assert torch.equal(O.same_sequence_negatives(xr, xl), T(g['negative_samples_back']))
    # a window never contains its own target block among the negatives of that target
    blocks_r = xr.reshape(xr.shape[0], Kr, 4, 4)
    for k in range(Kr):
        tail = neg[:, Kl:, k]                              # the Kr - 1 blocks taken from x_right
        others = torch.cat([blocks_r[:, :k], blocks_r[:, k + 1:]], dim=1)
        assert torch.equal(tail, others)
