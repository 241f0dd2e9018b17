"""CPU oracle for the decoder training step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `Decoder.epoch` / `Decoder.forward` (/root/reference/VQCPCB/decoders/decoder.py:310-370, 431-543) for
`transformer_type='relative'` with causal target self-attention, anticausal (or full) source self-attention and
anticausal (or full) cross-attention, i.e. decoder_type 'transformer_relative' / 'transformer_relative_fullCross'
(getters.py:309-350), in plain CPU fp32 tensor arithmetic on top of oracle/vqcpc_oracle.py.  Same rules as that file:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  Pinned against fixtures produced
by importing the reference itself (tools/gen_golden_decoder.py -> tests/golden/decoder_*.npz, relbias_cross_*.npz) by
tests/test_decoder_oracle_golden.py.

Parameters: one flat dict with the reference's `Decoder.state_dict()` names (the frozen encoder sits under
`encoder.`, exactly the keys oracle/vqcpc_oracle.py uses).  Tensors are batch-major (B, L, d).

Reference defect fixed here and in the build: `Decoder.epoch` forwards un-merged (B, S, num_codebooks) indices to
`Decoder.forward`, which then raises; the codes are merged with `Encoder.merge_codes` as `Decoder.generate` (:600) does.
"""
import torch

from . import vqcpc_oracle as O
from .student_oracle import categorical_crossentropy, embed

NONE, CAUSAL, ANTICAUSAL = 0, 1, 2
MASKS = {'full': NONE, 'causal': CAUSAL, 'anticausal': ANTICAUSAL, None: NONE}


def make_cfg(name=None, **over):
    """Encoder keys as oracle/vqcpc_oracle.py; decoder keys dec_*.
    'DEC' = the shipped relative decoder (configs/decoder_relative_AC_AC_C_random.py: 24 beats = 96 ticks = 384 target
    tokens, d_model 512, 8 heads, 3 + 3 layers, ff 1024, batch 32) on the reference's own transformer encoder
    (configs/encoder_random_transfo_config.py: d_model 512, 8 heads, [2, 2] layers, ff 2048, 1 x 32 codes of dim 3)."""
    cfg = O.make_cfg(name if name in O.CONFIGS else None)
    cfg.update(dict(events=48, dec_emb=32, dec_d=512, dec_H=4, dec_enc_layers=3, dec_dec_layers=3, dec_ff=1024, dec_pos=8,
                    enc_attn='anticausal', cross_attn='anticausal', dec_dropout=0.0))
    if name == 'DEC':
        cfg.update(d=512, H=8, layers=[2, 2], ff=2048, D=3, K=32, ncb=1, events=96, dec_H=8, B=32)
    cfg.update(over)
    return cfg


def init_state(cfg, seed=0):
    """Random-init state dict with the reference's `Decoder.state_dict()` names / shapes and its initialisers
    (decoder.py:87-232; TransformerCustom._reset_parameters xavier-initialises every matrix of the transformer,
    e1 / e2 included, transformer_custom.py:113-118)."""
    import math
    g = torch.Generator().manual_seed(seed)
    sd = {k: v for k, v in O.init_state(cfg, seed=seed + 1).items() if k.startswith('encoder.')}
    nc = len(cfg['vocab'])
    d, H, ff, pos = cfg['dec_d'], cfg['dec_H'], cfg['dec_ff'], cfg['dec_pos']
    hd = d // H
    T = cfg['events'] * nc
    U = 1
    for f in cfg['factors']:
        U *= f
    S = T // U
    for v, nv in enumerate(cfg['vocab']):
        sd[f'data_processor.embeddings.{v}.weight'] = torch.randn(nv + 1, cfg['dec_emb'], generator=g)
    sd['target_channel_embeddings'] = torch.randn(1, nc, pos, generator=g)
    sd['target_events_positioning_embeddings'] = torch.randn(1, U // nc, pos, generator=g)

    def xav(r, c):
        return O._xavier(g, r, c)

    def attn(pre, Lk):
        sd[pre + 'in_proj_weight'], sd[pre + 'in_proj_bias'] = xav(3 * d, d), torch.zeros(3 * d)
        sd[pre + 'out_proj.weight'], sd[pre + 'out_proj.bias'] = xav(d, d), torch.zeros(d)
        sd[pre + 'attn_bias.e1'], sd[pre + 'attn_bias.e2'] = xav(H * Lk, hd), xav(H * Lk, hd)

    def ffn_norms(pre, norms):
        sd[pre + 'linear1.weight'], sd[pre + 'linear1.bias'] = xav(ff, d), (torch.rand(ff, generator=g) * 2 - 1) / math.sqrt(d)
        sd[pre + 'linear2.weight'], sd[pre + 'linear2.bias'] = xav(d, ff), (torch.rand(d, generator=g) * 2 - 1) / math.sqrt(ff)
        for n in norms:
            sd[pre + n + '.weight'], sd[pre + n + '.bias'] = torch.ones(d), torch.zeros(d)

    for l in range(cfg['dec_enc_layers']):
        pre = f'transformer.encoder.layers.{l}.'
        attn(pre + 'self_attn.', S)
        ffn_norms(pre, ('norm1', 'norm2'))
    for l in range(cfg['dec_dec_layers']):
        pre = f'transformer.decoder.layers.{l}.'
        attn(pre + 'self_attn.', T)
        attn(pre + 'multihead_attn.', S)
        ffn_norms(pre, ('norm1', 'norm2', 'norm3'))
    sd['linear_target.weight'], sd['linear_target.bias'] = O._linear_init(g, d, cfg['dec_emb'] + 2 * pos)
    sd['sos'] = torch.randn(1, 1, d, generator=g)
    sd['source_embeddings.weight'] = torch.randn(cfg['K'] ** cfg['ncb'], d, generator=g)
    for c, nv in enumerate(cfg['vocab']):
        sd[f'pre_softmaxes.{c}.weight'], sd[f'pre_softmaxes.{c}.bias'] = O._linear_init(g, nv, d)
    return sd


def synthetic_batch(cfg, seed=1234, B=None):
    B = B or cfg['B']
    g = torch.Generator().manual_seed(seed)
    return {'x': torch.cat([torch.randint(0, nv, (B, cfg['events'], 1), generator=g) for nv in cfg['vocab']], dim=2)}


def relative_bias_cross(q, e1, e2, S):
    """Closed form of SubsampledRelativeAttention.forward (subsampled_relative_attention.py:30-122) for
    seq_len_tgt = r * seq_len_src.  With p = i // r (the source position query i is aligned with):

        bias[h, i, j] = q[h, i] . e1[h, S-1-(p-j)]   if j <= p
                      = q[h, i] . e2[h, j-p]         if j >  p

    (view (T, S) -> (S, T), pad one column, view back, drop / keep the first line -- the -100 fill values never land on
    a kept entry; r = 1 is the square form of oracle/vqcpc_oracle.py).  q: (n, H, T, hd) already scaled."""
    n, H, T, hd = q.shape
    r = T // S
    a1 = torch.einsum('nhld,hmd->nhlm', q, e1.view(H, S, hd))
    a2 = torch.einsum('nhld,hmd->nhlm', q, e2.view(H, S, hd))
    p = (torch.arange(T) // r).view(T, 1)
    j = torch.arange(S).view(1, S)
    m1 = (S - 1 - p + j).clamp(0, S - 1).expand(n, H, T, S)
    m2 = (j - p).clamp(0, S - 1).expand(n, H, T, S)
    return torch.where(j <= p, a1.gather(-1, m1), a2.gather(-1, m2))


def additive_mask(kind, S, T):
    """decoder.py:292-308: causal = 0 where j <= p, anticausal = 0 where j >= p, -inf elsewhere, p = i // (T // S)
    (`_generate_anticausal_mask(sz, sz_tgt)` repeats every row of the square mask T // S times)."""
    if kind == NONE:
        return None
    p = (torch.arange(T) // (T // S)).view(T, 1)
    j = torch.arange(S).view(1, S)
    ok = (j <= p) if kind == CAUSAL else (j >= p)
    return torch.zeros(T, S).masked_fill(~ok, float('-inf'))


def attention(xq, xkv, P, pre, H, mask_kind, p_drop=0.0, training=False, gen=None):
    """MultiheadAttentionCustom.forward (multihead_attention_custom.py:150-346): self-attention when xkv is xq (one
    in_proj GEMM, :171), otherwise the encoder-decoder branch (:173-196: q from rows [0, d) of in_proj, k | v from rows
    [d, 3d) applied to the memory).  Order of the logit terms as the reference: q.k, + attn_mask (:314-316), + relative
    bias (:328-330), softmax, dropout.  Returns (out, probs)."""
    n, T, d = xq.shape
    S = xkv.shape[1]
    hd = d // H
    W, b = P[pre + 'in_proj_weight'], P[pre + 'in_proj_bias']
    q = O.linear(xq, W[:d], b[:d]) * (float(hd) ** -0.5)
    k, v = O.linear(xkv, W[d:], b[d:]).split(d, dim=-1)
    q = q.reshape(n, T, H, hd).transpose(1, 2)
    k = k.reshape(n, S, H, hd).transpose(1, 2)
    v = v.reshape(n, S, H, hd).transpose(1, 2)
    scores = q @ k.transpose(-1, -2)
    m = additive_mask(mask_kind, S, T)
    if m is not None:
        scores = scores + m
    scores = scores + relative_bias_cross(q, P[pre + 'attn_bias.e1'], P[pre + 'attn_bias.e2'], S)
    probs = O.dropout(torch.softmax(scores, dim=-1), p_drop, training, gen)
    ctx = (probs @ v).transpose(1, 2).reshape(n, T, d)
    return O.linear(ctx, P[pre + 'out_proj.weight'], P[pre + 'out_proj.bias']), probs


def _ffn(x, P, pre, p_drop, training, gen):
    h = torch.relu(O.linear(x, P[pre + 'linear1.weight'], P[pre + 'linear1.bias']))
    return O.linear(O.dropout(h, p_drop, training, gen), P[pre + 'linear2.weight'], P[pre + 'linear2.bias'])


def source_layer(x, P, pre, H, mask_kind, p_drop=0.0, training=False, gen=None):
    """TransformerEncoderLayerCustom.forward with src_mask (transformer_custom.py:268-291)."""
    a, probs = attention(x, x, P, pre + 'self_attn.', H, mask_kind, p_drop, training, gen)
    x = O.layer_norm(x + O.dropout(a, p_drop, training, gen), P[pre + 'norm1.weight'], P[pre + 'norm1.bias'])
    x = O.layer_norm(x + O.dropout(_ffn(x, P, pre, p_drop, training, gen), p_drop, training, gen),
                     P[pre + 'norm2.weight'], P[pre + 'norm2.bias'])
    return x, probs


def target_layer(tgt, mem, P, pre, H, cross_kind, p_drop=0.0, training=False, gen=None):
    """TransformerDecoderLayerCustom.forward (transformer_custom.py:355-386): causal self-attention, cross-attention
    on the memory, FFN; post-LN after each."""
    a, p_self = attention(tgt, tgt, P, pre + 'self_attn.', H, CAUSAL, p_drop, training, gen)
    tgt = O.layer_norm(tgt + O.dropout(a, p_drop, training, gen), P[pre + 'norm1.weight'], P[pre + 'norm1.bias'])
    a, p_cross = attention(tgt, mem, P, pre + 'multihead_attn.', H, cross_kind, p_drop, training, gen)
    tgt = O.layer_norm(tgt + O.dropout(a, p_drop, training, gen), P[pre + 'norm2.weight'], P[pre + 'norm2.bias'])
    tgt = O.layer_norm(tgt + O.dropout(_ffn(tgt, P, pre, p_drop, training, gen), p_drop, training, gen),
                       P[pre + 'norm3.weight'], P[pre + 'norm3.bias'])
    return tgt, p_self, p_cross


def encode_codes(x, P, cfg):
    """decoder.py:327-336 (+ merge, see module docstring): the frozen encoder in inference (the upscaler output is
    not used by the decoder), merged indices (B, S)."""
    with torch.no_grad():
        E = {k: v.detach() for k, v in P.items() if k.startswith('encoder.')}
        tokens = O.preprocess_blocks(x)
        tokens = tokens.reshape(-1, tokens.shape[-2], tokens.shape[-1])
        tables = [E[f'encoder.data_processor.embeddings.{v}.weight'] for v in range(len(cfg['vocab']))]
        z = O.downscaler_forward(O.embed_blocks(tokens, tables), E, dict(cfg, dropout=0.0))
        idx = O.vq_assign(z.reshape(-1, z.shape[-1]), [E[f'encoder.quantizer.embeddings.{c}'] for c in range(cfg['ncb'])])
    return O.merge_codes(idx.reshape(x.shape[0], -1, cfg['ncb']), cfg['K'])


def decoder_forward(codes, x, P, cfg, training=False, gen=None, stages=None):
    """Decoder.forward (decoder.py:431-543).  codes (B, S) int64 merged; x (B, events, channels) int64.
    -> dict(loss, logits [per channel (B, events, V_c)], attention maps of the last layers)."""
    B = x.shape[0]
    nc = len(cfg['vocab'])
    d, H, pd = cfg['dec_d'], cfg['dec_H'], cfg['dec_dropout']
    src = P['source_embeddings.weight'][codes]                                        # :439
    tgt = embed(x, P, 'data_processor.', nc).reshape(B, -1, cfg['dec_emb'])           # :441-443  (B, T, emb)
    T, S = tgt.shape[1], src.shape[1]
    total_up = T // S                                                                 # decoder.py:82
    tok = torch.arange(T)
    chan = P['target_channel_embeddings'].reshape(nc, -1)                             # index t % nc         (:450-452)
    ev = P['target_events_positioning_embeddings'].reshape(total_up // nc, -1)        # index (t // nc) % (total_up // nc)
    tgt = torch.cat([tgt, chan[tok % nc].expand(B, T, -1), ev[(tok // nc) % (total_up // nc)].expand(B, T, -1)], dim=2)
    tgt = O.linear(tgt, P['linear_target.weight'], P['linear_target.bias'])           # :467
    tgt = torch.cat([P['sos'].reshape(1, 1, d).expand(B, 1, d), tgt[:, :-1]], dim=1)   # shift by one (:474-480)
    mem = src
    a_enc = None
    for l in range(cfg['dec_enc_layers']):
        mem, a_enc = source_layer(mem, P, f'transformer.encoder.layers.{l}.', H, MASKS[cfg['enc_attn']], pd, training, gen)
    out = tgt
    a_self = a_cross = None
    for l in range(cfg['dec_dec_layers']):
        out, a_self, a_cross = target_layer(out, mem, P, f'transformer.decoder.layers.{l}.', H, MASKS[cfg['cross_attn']],
                                            pd, training, gen)
    out = out.reshape(B, -1, nc, d)                                                   # :519-522
    logits = [O.linear(out[:, :, c], P[f'pre_softmaxes.{c}.weight'], P[f'pre_softmaxes.{c}.bias']) for c in range(nc)]
    loss = categorical_crossentropy(logits, x, torch.ones_like(x)).mean()              # :529-535
    if stages is not None:
        stages.update(memory=mem, output=out)
    return dict(loss=loss, logits=logits, a_enc=a_enc, a_self=a_self, a_cross=a_cross)


class DecoderOracleTrainer:
    """`Decoder.epoch` (decoder.py:310-370): frozen encoder -> codes, forward, backward, clip 5 over the decoder's
    parameters (the frozen encoder has no gradients), Adam, optional LambdaLR (:235-252, same ramp as the encoder's)."""

    def __init__(self, cfg, state_dict, lr=1e-4, schedule_lr=False):
        self.cfg = cfg
        self.P = {k: v.detach().clone() for k, v in state_dict.items()}
        for k, v in self.P.items():
            if not k.startswith('encoder.') and v.is_floating_point():
                v.requires_grad_(True)
        self.lr, self.schedule_lr = lr, schedule_lr
        self.opt_state = {}
        self.sched_step = 0
        self.last_grads = None
        self.last_grad_norm = None

    def trainable(self):
        return [k for k, v in self.P.items() if v.requires_grad]

    def step(self, batch, train, gen=None):
        x = batch['x'].long()
        codes = encode_codes(x, self.P, self.cfg)
        out = decoder_forward(codes, x, self.P, self.cfg, training=train, gen=gen)
        out['codes'] = codes
        if train:
            names = self.trainable()
            grads = torch.autograd.grad(out['loss'], [self.P[k] for k in names], allow_unused=True)
            grads = {k: (g if g is not None else torch.zeros_like(self.P[k])) for k, g in zip(names, grads)}
            self.last_grads = {k: g.clone() for k, g in grads.items()}
            self.last_grad_norm = O.clip_grad_norm(list(grads.values()), 5.0)
            lr = self.lr * (O.lr_lambda(self.sched_step) if self.schedule_lr else 1.0)
            with torch.no_grad():
                O.adam_step(self.P, grads, self.opt_state, lr)
            if self.schedule_lr:
                self.sched_step += 1
        return out

    def epoch(self, data_loader, train, num_batches, gen=None):
        total, n = 0.0, -1
        for n, batch in enumerate(data_loader):
            if num_batches is not None and n >= num_batches:
                n -= 1
                break
            total += float(self.step(batch, train, gen)['loss'].detach())
        return {'loss': total / (n + 1)}
