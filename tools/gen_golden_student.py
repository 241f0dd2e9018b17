"""Golden vectors for the student (distilled VQ-VAE) path, produced by IMPORTING the reference (container-only tool,
same rules as tools/gen_golden.py whose sys.modules stubs it reuses).  Writes tests/golden/student_*.npz and
relbias_L24.npz.  Never copies reference source text; fixtures hold tensors only.

Run:  python tools/gen_golden_student.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as base  # noqa: E402  (installs the tensorboard / music21 stubs and the /root/reference path)
from gen_golden import npy, save, sd_arrays, perturb_1d  # noqa: E402

from VQCPCB.auxiliary_decoders.auxiliary_decoder_relative import AuxiliaryDecoderRelative  # noqa: E402
from VQCPCB.data_processor.bach_data_processor import BachDataProcessor  # noqa: E402
from VQCPCB.downscalers.relative_transformer_downscaler_linear import RelativeTransformerDownscalerLinear  # noqa: E402
from VQCPCB.encoder import Encoder  # noqa: E402
from VQCPCB.quantizer.vector_quantizer import ProductVectorQuantizer  # noqa: E402
from VQCPCB.student_encoder_trainer import StudentEncoderTrainer  # noqa: E402
from VQCPCB.teachers.teacher_relative import TeacherRelative  # noqa: E402
from VQCPCB.utils import categorical_crossentropy, distilled_categorical_crossentropy, flatten  # noqa: E402


class FakeDataloaderGenerator:
    pass


def build(cfg):
    nc = len(cfg['vocab'])
    dp = BachDataProcessor(embedding_size=cfg['emb'], num_events=cfg['ticks'], num_tokens_per_channel=cfg['vocab'])
    ds = RelativeTransformerDownscalerLinear(input_dim=cfg['emb'], output_dim=cfg['D'], num_channels=nc,
                                             downscale_factors=list(cfg['factors']), d_model=cfg['d'], n_head=cfg['H'],
                                             list_of_num_layers=list(cfg['enc_layers']), dim_feedforward=cfg['ff'],
                                             dropout=0.0)
    q = ProductVectorQuantizer(codebook_size=cfg['K'], codebook_dim=cfg['D'], commitment_cost=0.25,
                               num_codebooks=cfg['ncb'], use_batch_norm=False, initialize=False, squared_l2_norm=True)
    enc = Encoder('/tmp/vqcpc_golden_student', dp, ds, q, None)
    tdp = BachDataProcessor(embedding_size=cfg['emb'], num_events=cfg['ticks'], num_tokens_per_channel=cfg['vocab'])
    teacher = TeacherRelative(data_processor=tdp, num_layers=cfg['teacher_layers'], num_tokens_per_channel=cfg['vocab'],
                              positional_embedding_size=cfg['teacher_pos'], d_model=cfg['d'], dim_feedforward=cfg['ff'],
                              n_head=cfg['H'], num_tokens=cfg['ticks'] * nc, dropout=0.0)
    dec = AuxiliaryDecoderRelative(num_tokens_per_channel=cfg['vocab'], codebook_dim=cfg['D'],
                                   upscale_factors=list(reversed(cfg['factors'])),
                                   list_of_num_layers=list(cfg['dec_layers']), n_head=cfg['H'], d_model=cfg['d'],
                                   dim_feedforward=cfg['ff'],
                                   num_tokens_bottleneck=cfg['ticks'] * nc // int(np.prod(cfg['factors'])), dropout=0.0)
    tr = StudentEncoderTrainer('/tmp/vqcpc_golden_student', FakeDataloaderGenerator(), enc,
                               num_events_masked=cfg['num_events_masked'], teacher=teacher, auxiliary_decoder=dec,
                               quantization_weighting=cfg['qw'])
    return tr


def all_sd(prefix, tr):
    out = {}
    out.update(sd_arrays(f'{prefix}/encoder', tr.encoder))
    out.update(sd_arrays(f'{prefix}/teacher', tr.teacher))
    out.update(sd_arrays(f'{prefix}/auxiliary_decoder', tr.auxiliary_decoder))
    return out


def gen_student(name, cfg, seed, lr=1e-3):
    torch.manual_seed(seed)
    tr = build(cfg)
    for m in (tr.encoder, tr.teacher, tr.auxiliary_decoder):
        perturb_1d(m)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.cat([torch.randint(0, nv, (cfg['B'], cfg['ticks'], 1), generator=g) for nv in cfg['vocab']], dim=2)
    enc = tr.encoder
    with torch.no_grad():   # codebook on (perturbed) downscaler outputs so that several codes are in use
        probe = torch.cat([torch.randint(0, nv, (16, cfg['ticks'], 1), generator=g) for nv in cfg['vocab']], dim=2)
        zp = enc.downscaler(flatten(enc.data_processor.embed(enc.data_processor.preprocess(probe)))).reshape(-1, cfg['D'])
        for e in enc.quantizer.embeddings:
            e.copy_(zp[:cfg['K']] + 0.01 * torch.randn(cfg['K'], cfg['D'], generator=g))
    arrays = all_sd('sd0', tr)
    arrays['batch/x'] = npy(x)

    captured = {}
    enc.register_forward_hook(lambda mod, inp, out: captured.update(zq=out[0], idx=out[1], qloss=out[2]))
    ds_forward = enc.downscaler.forward          # Encoder.forward calls .forward() directly: hooks do not fire

    def ds_spy(t):
        out = ds_forward(t)
        captured.update(z=out)
        return out

    enc.downscaler.forward = ds_spy
    tr.auxiliary_decoder.register_forward_hook(lambda mod, inp, out: captured.update(student_logits=out))
    tr.teacher.register_forward_hook(lambda mod, inp, out: captured.update(teacher_logits=out, masked_embed=inp[0]))

    # ---- eval epoch (train=False); the masked event index is the first draw from the global CPU generator
    tr.init_optimizers(lr=lr, schedule_lr=False)
    torch.manual_seed(seed + 2)
    ev = tr.epoch(iter([{'x': x}]), train=False, num_batches=1)
    torch.manual_seed(seed + 2)
    arrays['eval_masked_event_index'] = np.asarray(int(torch.randint(high=cfg['ticks'], size=()).item()))
    arrays['eval_seed'] = np.asarray(seed + 2)
    for k, v in ev.items():
        arrays[f'eval/{k}'] = np.asarray(v, dtype=np.float64)
    for k in ('z', 'zq', 'idx', 'qloss'):
        arrays[f'eval_fwd/{k}'] = npy(captured[k])
    for c in range(len(cfg['vocab'])):
        arrays[f'eval_fwd/teacher_logits.{c}'] = npy(captured['teacher_logits'][c])
        arrays[f'eval_fwd/student_logits.{c}'] = npy(captured['student_logits'][c])

    # ---- train epoch: gradients captured BEFORE each clip_grad_norm_ rescales them in place
    pre_clip, norms = {}, []
    orig_clip = torch.nn.utils.clip_grad_norm_
    names = {}
    for pfx, mod in (('encoder', tr.encoder), ('teacher', tr.teacher), ('auxiliary_decoder', tr.auxiliary_decoder)):
        for n, p in mod.named_parameters():
            names[id(p)] = f'{pfx}.{n}'

    def spy(parameters, max_norm, *a, **k):
        params = list(parameters)
        for p in params:
            if p.grad is not None:
                pre_clip[names[id(p)]] = p.grad.detach().clone()
        total = orig_clip(params, max_norm, *a, **k)
        norms.append(float(total))
        return total

    torch.nn.utils.clip_grad_norm_ = spy
    try:
        torch.manual_seed(seed + 3)
        trn = tr.epoch(iter([{'x': x}]), train=True, num_batches=1)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
    torch.manual_seed(seed + 3)
    arrays['train_masked_event_index'] = np.asarray(int(torch.randint(high=cfg['ticks'], size=()).item()))
    arrays['train_seed'] = np.asarray(seed + 3)
    for k, v in trn.items():
        arrays[f'train/{k}'] = np.asarray(v, dtype=np.float64)
    for k, gr in pre_clip.items():
        arrays[f'grad/{k}'] = npy(gr)
    arrays['grad_norms_teacher_decoder_encoder'] = np.asarray(norms, dtype=np.float64)   # order of the three clips
    arrays.update(all_sd('sd1', tr))
    arrays['cfg_json'] = np.array(json.dumps(cfg))
    arrays['lr'] = np.array(lr)
    save(name, **arrays)
    print('   eval :', {k: round(v, 5) for k, v in ev.items()}, 'm =', int(arrays['eval_masked_event_index']))
    print('   train:', {k: round(v, 5) for k, v in trn.items()}, 'm =', int(arrays['train_masked_event_index']))
    print('   clip norms (teacher, decoder, encoder):', norms, ' codes used:', len(np.unique(arrays['eval_fwd/idx'])))


def gen_ce(name, seed):
    """The two loss helpers on their own, with a mask that is NOT a single event (general semantics of utils.py)."""
    g = torch.Generator().manual_seed(seed)
    B, E, vocab = 4, 6, [5, 7, 6]
    value = [torch.randn(B, E, v, generator=g) for v in vocab]
    teacher = [torch.randn(B, E, v, generator=g) for v in vocab]
    target = torch.cat([torch.randint(0, v, (B, E, 1), generator=g) for v in vocab], dim=2)
    mask = torch.zeros(B, E, len(vocab), dtype=torch.long)
    mask[:, 1] = 1
    mask[:, 4] = 1
    ce = categorical_crossentropy(value, target, mask)
    dce = distilled_categorical_crossentropy(value, teacher, mask)
    arrays = dict(target=npy(target), mask=npy(mask), ce=npy(ce), dce=npy(dce))
    for c in range(len(vocab)):
        arrays[f'value.{c}'] = npy(value[c])
        arrays[f'teacher.{c}'] = npy(teacher[c])
    save(name, **arrays)


def gen_same_sequence(name, B, Kl, Kr, seed):
    """_build_negatives_sameSeq through the reference's own method (unbound, on a stand-in `self`), inputs and outputs in
    the batch-dict layout the dataloader yields (bach_cpc_dataloader.py:133-139)."""
    import types
    import VQCPCB.dataloaders.bach_cpc_dataloader as dl
    g = torch.Generator().manual_seed(seed)
    nv, tpb = dl.num_voices, 16
    p = torch.randint(0, 50, (B, nv, (Kl + Kr) * tpb // nv), generator=g)        # (batch, voices, ticks) as the dataset yields
    fake = types.SimpleNamespace(num_blocks_right=Kr, num_blocks_left=Kl, num_tokens_per_block=tpb)
    x_left, x_right = p[:, :, :Kl * tpb // nv], p[:, :, Kl * tpb // nv:]
    N = Kl + Kr - 1
    neg = dl.BachCPCDataloaderGenerator._build_negatives_sameSeq(fake, x_left, x_right, B, N)
    arrays = dict(x_left=npy(x_left.transpose(1, 2)), x_right=npy(x_right.transpose(1, 2)),
                  negative_samples=npy(neg.transpose(3, 4)))
    if Kl == Kr:        # the reference's backward direction only works for equal block counts
        back = dl.BachCPCDataloaderGenerator._build_negatives_sameSeq(fake, x_right, x_left, B, N)
        arrays['negative_samples_back'] = npy(back.transpose(3, 4))
    save(name, **arrays)


if __name__ == '__main__':
    gen_same_sequence('negatives_same_seq', B=3, Kl=3, Kr=3, seed=70)
    gen_same_sequence('negatives_same_seq_uneven', B=2, Kl=4, Kr=2, seed=71)
    base.gen_relbias('relbias_L24', n=2, H=2, L=24, hd=8, seed=12)
    gen_ce('student_ce', seed=50)
    tiny = dict(emb=8, vocab=[11, 9, 12, 10], ticks=16, d=32, H=2, ff=64, enc_layers=[1, 1], factors=[4, 4], pos=8, D=3,
                K=8, ncb=1, beta=0.25, squared=True, teacher_layers=2, teacher_pos=8, dec_layers=[1, 1], dropout=0.0,
                num_events_masked=1, qw=0.1, B=3)
    gen_student('student_tiny', tiny, seed=60)
    # 32 ticks: teacher L = 128, decoder L = 8 / 32, two layers per stack, larger loss weight so that a clip is active
    gen_student('student_tiny_clip', dict(tiny, ticks=32, enc_layers=[2, 1], dec_layers=[1, 2], num_events_masked=2, qw=40.0,
                                          B=2), seed=61)
