"""The reference-compatible entry points that the fused training path bypasses (stand-alone module `forward`s with the
reference's tensor layouts) give the same numbers as the hot path / the reference fixtures."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import student_oracle as S
from oracle import vqcpc_oracle as O
from test_student_gpu import build_student, golden_state
from test_trainer_gpu import build_trainer, golden_cfg_sd

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 5e-5


@pytest.fixture(autouse=True)
def _lib():
    from vqcpc_bach_amd import hip
    hip.load()
    hip.set_gemm_mode(0)
    yield


@pytest.mark.parametrize('name', ['relbias_L16', 'relbias_L4', 'relbias_L24'])
def test_subsampled_relative_attention_forward(name):
    """SubsampledRelativeAttention.forward(q) (subsampled_relative_attention.py:30-122) against the reference's output."""
    from vqcpc_bach_amd.transformer.subsampled_relative_attention import SubsampledRelativeAttention
    g = load_golden(name)
    H, (bh, L, hd) = int(g['H']), g['q'].shape
    m = SubsampledRelativeAttention(head_dim=hd, num_heads=H, seq_len_src=L, seq_len_tgt=L).cuda()
    with torch.no_grad():
        m.e1.copy_(T(g['e1']))
        m.e2.copy_(T(g['e2']))
        out = m(T(g['q']).cuda())
    assert out.shape == g['bias'].shape and rel_err(out.cpu(), g['bias']) < 1e-5


def test_encoder_api_path_equals_fused_path():
    """Encoder.forward_embedded(embed(preprocess(x))) and RelativeTransformerDownscaler.forward(embedded_seq) == Encoder(x);
    TransformerEncoderLayerCustom.forward on the reference's time-first (L, N, E) layout == forward_rows."""
    g = load_golden('epoch_tiny')
    cfg, sd = golden_cfg_sd(g)
    tr = build_trainer(cfg, sd)
    tr.eval()
    enc = tr.encoder
    x = T(g['batch/x_left'])
    with torch.no_grad():
        z_up, idx, ql = enc(x)
        x_embed = enc.data_processor.embed(enc.data_processor.preprocess(x))
        z_up2, idx2, ql2 = enc.forward_embedded(x_embed)
        assert torch.equal(idx, idx2) and torch.equal(idx.cpu(), T(g['fwd_idx']))
        assert rel_err(z_up2.cpu(), g['fwd_zup']) < TOL and rel_err(ql2.cpu(), g['fwd_qloss']) < 2e-4
        layer = enc.downscaler.transformers[0].layers[0]
        L, d = layer.seq_len, enc.downscaler.d_model
        rows = torch.randn(5 * L, d, device='cuda')
        y_rows, probs = layer.forward_rows(rows)
        y_tf, att = layer(rows.view(5, L, d).transpose(0, 1))                       # (L, N, E) in, (L, N, E) out
        assert torch.equal(y_tf.transpose(0, 1).reshape(5 * L, d), y_rows)
        assert att['a_self_encoder'].shape == (5, layer.nhead, L, L)
        stack_out, atts = enc.downscaler.transformers[0](rows.view(5, L, d).transpose(0, 1))
        assert stack_out.shape == (L, 5, d) and len(atts) == len(enc.downscaler.transformers[0].layers)


def test_cpc_head_api_functions_against_reference_fixture():
    """FksModule.forward, nce_loss, quantization_loss (vqcpc_helper.py:5-51, 86-98) on the reference's tensors."""
    from vqcpc_bach_amd import vqcpc_helper as VH
    g = load_golden('cpc_heads')
    fk = VH.FksModule(z_dim=8, c_dim=6, k_max=4).cuda()
    with torch.no_grad():
        fk.W.copy_(T(g['fks_module/W']))
        c = T(g['c']).cuda()
        f_pos = fk(c, T(g['z_right']).cuda())
        zn = T(g['z_neg']).cuda()                                                  # (B, N, K, z)
        B, N, K, z = zn.shape
        f_neg = fk(c.repeat_interleave(N, 0), zn.reshape(B * N, K, z)).view(B, N, K).permute(0, 2, 1)
        assert rel_err(f_pos.cpu(), g['f_pos']) < TOL and rel_err(f_neg.cpu(), g['f_neg']) < TOL
        assert abs(float(VH.nce_loss(f_pos, f_neg)) - float(g['loss'])) < TOL * abs(float(g['loss']))
        ql = VH.quantization_loss(T(g['ql']).cuda(), T(g['qn']).cuda(), T(g['qr']).cuda())
        assert abs(float(ql) - float(g['qloss'])) < 1e-5 * abs(float(g['qloss']))


def test_cross_entropy_api_helpers_against_reference_fixture():
    """utils.categorical_crossentropy / distilled_categorical_crossentropy with a two-event mask (utils.py:24-49, 131-159)."""
    from vqcpc_bach_amd import utils
    g = load_golden('student_ce')
    nc = g['target'].shape[2]
    value = [T(g[f'value.{c}']).cuda() for c in range(nc)]
    teacher = [T(g[f'teacher.{c}']).cuda() for c in range(nc)]
    ce = utils.categorical_crossentropy(value, T(g['target']).cuda(), T(g['mask']).cuda())
    dce = utils.distilled_categorical_crossentropy(value, teacher, T(g['mask']).cuda())
    assert ce.shape == g['ce'].shape and rel_err(ce.cpu(), g['ce']) < 1e-5
    assert dce.shape == g['dce'].shape and rel_err(dce.cpu(), g['dce']) < 1e-5


def test_teacher_and_decoder_full_forward_equal_event_projection_and_reference():
    """TeacherRelative.forward(x_embed) / AuxiliaryDecoderRelative.forward(codes) return every event's logits (the
    reference contract); the training path projects only the masked event: same numbers at that event, and equal to the
    reference's logits; AuxiliaryDecoderRelative.upscale keeps the reference's time-first signature."""
    g = load_golden('student_tiny')
    cfg = S.make_cfg(**json.loads(str(g['cfg_json'])))
    tr = build_student(cfg, golden_state(g))
    tr.eval()
    x = T(g['batch/x']).cuda()
    m = int(g['eval_masked_event_index'])
    nc = len(cfg['vocab'])
    with torch.no_grad():
        masked_x, notes = tr.mask_teacher(x, cfg['num_events_masked'], masked_event_index=m)
        full = tr.teacher(tr.teacher.data_processor.embed(masked_x))
        ev = tr.teacher.forward_events(masked_x, m)
        for c in range(nc):
            assert full[c].shape == g[f'eval_fwd/teacher_logits.{c}'].shape
            assert rel_err(full[c].cpu(), g[f'eval_fwd/teacher_logits.{c}']) < TOL
            assert rel_err(ev[c].cpu(), full[c][:, m].cpu()) < 1e-5
        zq = T(g['eval_fwd/zq']).cuda()
        dfull = tr.auxiliary_decoder(zq)
        dev_ = tr.auxiliary_decoder.forward_events(zq, m)
        for c in range(nc):
            assert rel_err(dfull[c].cpu(), g[f'eval_fwd/student_logits.{c}']) < TOL
            assert rel_err(dev_[c].cpu(), dfull[c][:, m].cpu()) < 1e-5
        seq = torch.randn(6, 2, cfg['d'], device='cuda')                            # time-first (L, batch, d)
        emb = tr.auxiliary_decoder.upscale_embeddings[0]
        up = tr.auxiliary_decoder.upscale(seq, len(emb), emb)
        ref = S.upscale(seq.transpose(0, 1).cpu(), len(emb), emb.detach().cpu()).transpose(0, 1)
        assert up.shape == (6 * len(emb), 2, cfg['d']) and torch.equal(up.cpu(), ref)
        # the reference's full-sequence losses on the full logits == the step's event-only losses
        from vqcpc_bach_amd import utils
        lt = utils.categorical_crossentropy(full, x, notes).mean()
        assert abs(float(lt) - float(g['eval/loss_teacher'])) < TOL * abs(float(g['eval/loss_teacher']))
        lr = utils.distilled_categorical_crossentropy(dfull, full, notes).mean()
        assert abs(float(lr) - float(g['eval/loss_reconstruction'])) < TOL * abs(float(g['eval/loss_reconstruction']))
