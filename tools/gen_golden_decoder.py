"""Golden vectors for the decoder training step, produced by IMPORTING the reference
(container-only tool; reuses the sys.modules stubs of tools/gen_golden.py).  Writes tests/golden/decoder_*.npz and
relbias_cross_*.npz.  Fixtures hold tensors only.

Reference defect worked around here (and fixed in the build): `Decoder.epoch` (decoders/decoder.py:327-344) hands the
quantizer's `encoding_indices` of shape (B, S, num_codebooks) straight to `Decoder.forward`, whose `nn.Embedding`
then returns a 4-D tensor and `TransformerCustom.forward` raises.  `Decoder.generate` (:600) merges the codebook axis
with `Encoder.merge_codes` first, and the embedding table is sized for merged codes (:212-215), so the shim below does
the same for `epoch`: the frozen encoder's forward returns merged codes.

Run:  python tools/gen_golden_decoder.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as base  # noqa: E402,F401  (stubs + /root/reference on sys.path)
from gen_golden import npy, save, sd_arrays, perturb_1d, build_encoder  # noqa: E402

import types  # noqa: E402

for _name in ('seaborn', 'matplotlib', 'matplotlib.pyplot'):   # plotting only (decoders/decoder.py:5-7); not installed
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']

from VQCPCB.data_processor.bach_data_processor import BachDataProcessor  # noqa: E402
from VQCPCB.decoders.decoder import Decoder  # noqa: E402
from VQCPCB.transformer.subsampled_relative_attention import SubsampledRelativeAttention  # noqa: E402
from VQCPCB.transformer.transformer_custom import TransformerDecoderLayerCustom  # noqa: E402
from VQCPCB.utils import flatten  # noqa: E402


class FakeDataloaderGenerator:
    pass


def gen_relbias_cross(name, n, H, S, T, hd, seed):
    torch.manual_seed(seed)
    m = SubsampledRelativeAttention(head_dim=hd, num_heads=H, seq_len_src=S, seq_len_tgt=T)
    q = torch.randn(n * H, T, hd)
    save(name, q=npy(q), e1=npy(m.e1), e2=npy(m.e2), bias=npy(m(q)), H=np.array(H), S=np.array(S), T=np.array(T))


def masks(dec, S, T):
    return dict(causal_T=npy(dec._generate_causal_mask(T)), anticausal_S=npy(dec._generate_anticausal_mask(S)),
                anticausal_ST=npy(dec._generate_anticausal_mask(S, T)))


def gen_decoder_layer(name, n, H, S, T, d, ff, seed):
    """One TransformerDecoderLayerCustom (transformer_custom.py:294-386) with the masks `Decoder.forward` builds for
    decoder_type 'transformer_relative' (causal target, anticausal cross)."""
    torch.manual_seed(seed)
    layer = TransformerDecoderLayerCustom(d_model=d, nhead=H, attention_bias_type_self='relative_attention',
                                          attention_bias_type_cross='relative_attention_target_source',
                                          num_channels_encoder=1, num_events_encoder=S, num_channels_decoder=4,
                                          num_events_decoder=T // 4, dim_feedforward=ff, dropout=0.0)
    perturb_1d(layer)
    layer.eval()
    tgt = torch.randn(T, n, d, requires_grad=True)
    mem = torch.randn(S, n, d, requires_grad=True)
    sub = torch.triu(torch.ones(T, T)).t()
    tgt_mask = torch.zeros(T, T).masked_fill(sub == 0, float('-inf'))
    subS = torch.triu(torch.ones(S, S))
    mem_mask = torch.repeat_interleave(torch.zeros(S, S).masked_fill(subS == 0, float('-inf')), T // S, dim=0)
    y, att = layer(tgt, mem, tgt_mask=tgt_mask, memory_mask=mem_mask)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    arrays = sd_arrays('sd', layer)
    arrays.update({f'grad/{k}': npy(p.grad) for k, p in layer.named_parameters()})
    save(name, tgt=npy(tgt), mem=npy(mem), y=npy(y), a_self=npy(att['a_self_decoder']), a_cross=npy(att['a_cross']),
         g=npy(g), d_tgt=npy(tgt.grad), d_mem=npy(mem.grad), tgt_mask=npy(tgt_mask), mem_mask=npy(mem_mask),
         H=np.array(H), **arrays)


def gen_mha_forward(name, n, H, S, T, d, mask, seed):
    """MultiheadAttentionCustom.forward itself (multihead_attention_custom.py:122-353) with the additive masks of
    decoders/decoder.py:294-308: self-attention (S == T; query = key = value) or encoder-decoder attention (key = value = memory
    of S rows, T = r S queries), outputs + gradients w.r.t. the inputs and every parameter."""
    from VQCPCB.transformer.multihead_attention_custom import MultiheadAttentionCustom
    torch.manual_seed(seed)
    cross = S != T
    m = MultiheadAttentionCustom(embed_dim=d, num_heads=H,
                                 attention_bias_type='relative_attention_target_source' if cross else 'relative_attention',
                                 num_channels_k=1, num_events_k=S, num_channels_q=1, num_events_q=T, dropout=0.0)
    perturb_1d(m)
    m.eval()
    q = torch.randn(T, n, d, requires_grad=True)
    mem = torch.randn(S, n, d, requires_grad=True) if cross else q
    sq = torch.triu(torch.ones(S, S)).t()                                   # _generate_square_subsequent_mask(S)
    causal = torch.zeros(S, S).masked_fill(sq == 0, float('-inf'))
    if mask == 'causal':
        am = torch.repeat_interleave(causal, T // S, dim=0)
    elif mask == 'anticausal':
        am = torch.repeat_interleave(causal.t(), T // S, dim=0)
    else:
        am = None
    out, w = m(q, mem, mem, attn_mask=am)
    g = torch.randn_like(out)
    (out * g).sum().backward()
    arrays = sd_arrays('sd', m)
    arrays.update({f'grad/{k}': npy(p.grad) for k, p in m.named_parameters()})
    extra = dict(mem=npy(mem), d_mem=npy(mem.grad)) if cross else {}
    if am is not None:
        extra['attn_mask'] = npy(am)
    save(name, q=npy(q), out=npy(out), weights=npy(w), g=npy(g), d_q=npy(q.grad), H=np.array(H), S=np.array(S),
         T=np.array(T), **extra, **arrays)


def gen_mha_all():
    gen_mha_forward('mha_self_causal_T24', n=3, H=2, S=24, T=24, d=32, mask='causal', seed=90)
    gen_mha_forward('mha_self_full_T16', n=2, H=2, S=16, T=16, d=32, mask=None, seed=91)
    gen_mha_forward('mha_cross_anticausal_S6_T12', n=3, H=2, S=6, T=12, d=32, mask='anticausal', seed=92)


def build_decoder(cfg, enc):
    dp = BachDataProcessor(embedding_size=cfg['dec_emb'], num_events=cfg['events'], num_tokens_per_channel=cfg['vocab'])
    nc = len(cfg['vocab'])
    S = cfg['events'] * nc // int(np.prod(enc.downscaler.downscale_factors))
    return Decoder(model_dir='/tmp/vqcpc_golden_decoder', dataloader_generator=FakeDataloaderGenerator(), data_processor=dp,
                   encoder=enc, transformer_type='relative', encoder_attention_type=cfg['enc_attn'],
                   cross_attention_type=cfg['cross_attn'], d_model=cfg['dec_d'], num_encoder_layers=cfg['dec_enc_layers'],
                   num_decoder_layers=cfg['dec_dec_layers'], n_head=cfg['dec_H'], dim_feedforward=cfg['dec_ff'],
                   positional_embedding_size=cfg['dec_pos'], num_channels_encoder=1, num_events_encoder=S,
                   num_channels_decoder=nc, num_events_decoder=cfg['events'], dropout=0.0)


def gen_decoder_step(name, cfg, seed, lr=1e-3, logit_gain=1.0):
    torch.manual_seed(seed)
    enc = build_encoder(cfg)
    perturb_1d(enc)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # codebooks on downscaler outputs so that several codes are in use
        probe = torch.cat([torch.randint(0, nv, (4 * cfg['K'], 4, 1), generator=g) for nv in cfg['vocab']], dim=2)
        zp = enc.downscaler(flatten(enc.data_processor.embed(enc.data_processor.preprocess(probe)))).view(-1, cfg['D'])
        dsub = cfg['D'] // cfg['ncb']
        for c, e in enumerate(enc.quantizer.embeddings):
            e.copy_(zp[c:c + 4 * cfg['K']:4, c * dsub:(c + 1) * dsub] + 0.01 * torch.randn(cfg['K'], dsub, generator=g))
    dec = build_decoder(cfg, enc)
    with torch.no_grad():
        for k, p in dec.named_parameters():
            if not k.startswith('encoder.') and p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for m in dec.pre_softmaxes:
            m.weight.mul_(logit_gain)
    x = torch.cat([torch.randint(0, nv, (cfg['B'], cfg['events'], 1), generator=g) for nv in cfg['vocab']], dim=2)

    # reference defect (see module docstring): un-merged codes crash Decoder.forward
    crashed = False
    dec.init_optimizers(lr=lr, schedule_lr=False)
    try:
        dec.epoch(iter([{'x': x}]), train=False, num_batches=1)
    except Exception as e:   # noqa: BLE001
        crashed = True
        print(f'   reference Decoder.epoch without the merge shim: {type(e).__name__}: {str(e)[:90]}')
    assert crashed
    enc_forward = enc.forward

    def merged_forward(t, corrupt_labels=False):
        zq, idx, ql = enc_forward(t, corrupt_labels=corrupt_labels)
        return zq, enc.merge_codes(idx.clone()), ql

    enc.forward = merged_forward
    enc.__class__.__call__  # noqa: B018  (nn.Module.__call__ -> self.forward: the instance attribute is picked up)

    arrays = sd_arrays('sd0', dec)
    arrays['batch/x'] = npy(x)
    S = dec.transformer.encoder.layers[0].self_attn.attn_bias.seq_len_src
    arrays.update({f'mask/{k}': v for k, v in masks(dec, S, cfg['events'] * len(cfg['vocab'])).items()})
    with torch.no_grad():
        _, idx_raw, _ = enc_forward(x)
        arrays['codes_raw'] = npy(idx_raw)
        arrays['codes'] = npy(enc.merge_codes(idx_raw.clone()))

    # ---- eval epoch + the forward's outputs
    ev = dec.epoch(iter([{'x': x}]), train=False, num_batches=1)
    arrays['eval/loss'] = np.asarray(ev['loss'], dtype=np.float64)
    dec.eval()
    with torch.no_grad():
        fp = dec.forward(torch.from_numpy(arrays['codes']), x)
    for c, w in enumerate(fp['weights_per_category']):
        arrays[f'eval_fwd/logits.{c}'] = npy(w)
    arrays['eval_fwd/a_cross_last'] = npy(fp['attentions_decoder'][-1]['a_cross'])
    arrays['eval_fwd/a_self_last'] = npy(fp['attentions_decoder'][-1]['a_self_decoder'])
    arrays['eval_fwd/a_enc_last'] = npy(fp['attentions_encoder'][-1]['a_self_encoder'])

    # ---- train epoch, gradients captured BEFORE clip_grad_norm_ rescales them in place
    pre_clip = {}
    orig_clip = torch.nn.utils.clip_grad_norm_
    names = {id(p): n for n, p in dec.named_parameters()}

    def spy(parameters, max_norm, *a, **k):
        params = list(parameters)
        for p in params:
            if p.grad is not None:
                pre_clip[names[id(p)]] = p.grad.detach().clone()
        return orig_clip(params, max_norm, *a, **k)

    torch.nn.utils.clip_grad_norm_ = spy
    try:
        trn = dec.epoch(iter([{'x': x}]), train=True, num_batches=1)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
    arrays['train/loss'] = np.asarray(trn['loss'], dtype=np.float64)
    for k, gr in pre_clip.items():
        arrays[f'grad/{k}'] = npy(gr)
    total = torch.sqrt(sum((gr.double() ** 2).sum() for gr in pre_clip.values()))
    arrays['grad_total_norm'] = np.asarray(float(total))
    arrays.update(sd_arrays('sd1', dec))
    arrays['cfg_json'] = np.array(json.dumps(cfg))
    arrays['lr'] = np.array(lr)
    save(name, **arrays)
    print(f'   eval loss {ev["loss"]:.6f}  train loss {trn["loss"]:.6f}  grad norm {float(total):.4f}  '
          f'codes in use {len(np.unique(arrays["codes"]))}  frozen encoder grads: '
          f'{sum(k.startswith("encoder.") for k in pre_clip)}')


if __name__ == '__main__':
    if '--mha-only' in sys.argv:       # only the MultiheadAttentionCustom.forward fixtures
        gen_mha_all()
        sys.exit(0)
    gen_relbias_cross('relbias_cross_S3_T48', n=2, H=2, S=3, T=48, hd=8, seed=70)
    gen_relbias_cross('relbias_cross_S6_T12', n=3, H=3, S=6, T=12, hd=4, seed=71)
    gen_relbias_cross('relbias_cross_S40_T80', n=1, H=2, S=40, T=80, hd=8, seed=72)
    gen_decoder_layer('decoder_layer_S3_T48', n=3, H=2, S=3, T=48, d=32, ff=64, seed=73)
    gen_mha_all()
    tiny = dict(emb=8, vocab=[11, 12, 13, 14], d=32, H=2, layers=[1, 1], ff=64, D=8, K=16, ncb=2, zdim=8, up_hidden=16,
                events=12, B=3, Kl=2, Kr=2, dec_emb=8, dec_d=32, dec_H=2, dec_enc_layers=2, dec_dec_layers=2, dec_ff=64, dec_pos=4,
                enc_attn='anticausal', cross_attn='anticausal')
    gen_decoder_step('decoder_tiny', tiny, seed=80)
    gen_decoder_step('decoder_tiny_fullcross', dict(tiny, cross_attn='full', B=2), seed=81)
    gen_decoder_step('decoder_tiny_clip', dict(tiny, B=4), seed=82, logit_gain=12.0)   # global grad norm > 5
