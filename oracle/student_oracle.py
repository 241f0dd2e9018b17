"""CPU oracle for the distilled-VQ-VAE ("student") training step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `StudentEncoderTrainer.epoch` (/root/reference/VQCPCB/student_encoder_trainer.py:220-293) and what it
calls -- SURVEY.md section 8 row A23 / BASELINE configs[3] -- on top of the primitives of oracle/vqcpc_oracle.py
(encoder layer, VQ, LayerNorm, Adam, clip).  Same rules as that file: only tests/, smoke() and bench's cpu_baseline
may import it; it is pinned against fixtures produced by importing the reference (tools/gen_golden.py ->
tests/golden/student_*.npz, checked by tests/test_oracle_golden.py).

Parameter names are the reference's state_dict keys, prefixed `encoder.{data_processor,downscaler,quantizer}.`,
`teacher.` and `auxiliary_decoder.` (the checkpoint files `teacher` and `decoder`,
student_encoder_trainer.py:85-96).  Tensors are block-major (n, L, d); the reference is time-first.
"""
import math

import torch

from . import vqcpc_oracle as O

DEFAULTS = dict(
    emb=32, vocab=[56, 56, 56, 56], ticks=96,              # 24 beats x subdivision 4 (encoder_student_config.py:11-14)
    d=512, H=8, ff=2048, enc_layers=[4, 4], factors=[4, 4], pos=8,
    D=3, K=32, ncb=1, beta=0.25, squared=True,             # quantizer_kwargs (:38-47)
    teacher_layers=8, teacher_pos=8, dec_layers=[4, 4],    # (:57-83)
    dropout=0.0, num_events_masked=4, qw=0.1, B=8,
)

CONFIGS = {
    # BASELINE.json configs[3] / SURVEY.md section 8 legend C3
    'C3': dict(),
    'tiny': dict(ticks=16, d=32, H=2, ff=64, enc_layers=[1, 1], K=8, teacher_layers=2, dec_layers=[1, 1],
                 num_events_masked=1, B=3, vocab=[11, 9, 12, 10]),
}


def make_cfg(name=None, **over):
    cfg = dict(DEFAULTS)
    if name is not None:
        cfg.update(CONFIGS[name])
    cfg.update(over)
    return cfg


# -----------------------------------------------------------------------------------------------
# parameters
# -----------------------------------------------------------------------------------------------
def _layer_init(g, d, H, ff, L):
    hd = d // H
    lay = {}
    lay['self_attn.in_proj_weight'] = O._xavier(g, 3 * d, d)
    lay['self_attn.in_proj_bias'] = torch.zeros(3 * d)
    w, _ = O._linear_init(g, d, d)
    lay['self_attn.out_proj.weight'], lay['self_attn.out_proj.bias'] = w, torch.zeros(d)
    lay['self_attn.attn_bias.e1'] = torch.randn(H * L, hd, generator=g)
    lay['self_attn.attn_bias.e2'] = torch.randn(H * L, hd, generator=g)
    lay['linear1.weight'], lay['linear1.bias'] = O._linear_init(g, ff, d)
    lay['linear2.weight'], lay['linear2.bias'] = O._linear_init(g, d, ff)
    for n in ('norm1', 'norm2'):
        lay[n + '.weight'], lay[n + '.bias'] = torch.ones(d), torch.zeros(d)
    return lay


def _stack_init(sd, prefix, g, d, H, ff, L, num_layers):
    lay = _layer_init(g, d, H, ff, L)           # _get_clones: identical init inside a stack (transformer_custom.py:138)
    for l in range(num_layers):
        for k, v in lay.items():
            sd[f'{prefix}layers.{l}.{k}'] = v.clone()


def init_state(cfg, seed=0):
    """Random-init parameters with the reference's key names and shapes."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    d, H, ff, emb, pos = cfg['d'], cfg['H'], cfg['ff'], cfg['emb'], cfg['pos']
    nc = len(cfg['vocab'])
    num_tokens = cfg['ticks'] * nc
    for who in ('encoder.data_processor.', 'teacher.data_processor.'):      # the teacher owns a second set of tables
        for v, nv in enumerate(cfg['vocab']):                               # data_processor.py:26-32 (+1 mask token)
            sd[f'{who}embeddings.{v}.weight'] = torch.randn(nv + 1, emb, generator=g)
    # --- RelativeTransformerDownscalerLinear (relative_transformer_downscaler_linear.py:16-97)
    p = 'encoder.downscaler.'
    L0 = int(math.prod(cfg['factors']))
    sd[p + 'target_channel_embeddings'] = torch.randn(1, 1, nc, pos, generator=g)
    sd[p + 'events_positioning_embeddings'] = torch.randn(1, 1, L0 // nc, pos, generator=g)
    sd[p + 'input_linear.weight'], sd[p + 'input_linear.bias'] = O._linear_init(g, d - 2 * pos, emb)
    sd[p + 'output_linear.weight'], sd[p + 'output_linear.bias'] = O._linear_init(g, cfg['D'], d)
    L = L0
    for s, (nl, f) in enumerate(zip(cfg['enc_layers'], cfg['factors'])):
        _stack_init(sd, f'{p}transformers.{s}.', g, d, H, ff, L, nl)
        sd[f'{p}linear_aggs.{s}.weight'], sd[f'{p}linear_aggs.{s}.bias'] = O._linear_init(g, d, d * f)
        L //= f
    dsub = cfg['D'] // cfg['ncb']
    for c in range(cfg['ncb']):
        sd[f'encoder.quantizer.embeddings.{c}'] = torch.randn(cfg['K'], dsub, generator=g) * 4
    # --- TeacherRelative (teacher_relative.py:9-55)
    p = 'teacher.'
    tp = cfg['teacher_pos']
    sd[p + 'channel_embeddings'] = torch.randn(1, nc, tp, generator=g)
    sd[p + 'linear_to_input_transformer.weight'], sd[p + 'linear_to_input_transformer.bias'] = O._linear_init(g, d - tp, emb)
    _stack_init(sd, p + 'transformer.', g, d, H, ff, num_tokens, cfg['teacher_layers'])
    for c, nv in enumerate(cfg['vocab']):
        sd[f'{p}pre_softmaxes.{c}.weight'], sd[f'{p}pre_softmaxes.{c}.bias'] = O._linear_init(g, nv, d)
    # --- AuxiliaryDecoderRelative (auxiliary_decoder_relative.py:12-80)
    p = 'auxiliary_decoder.'
    up = list(reversed(cfg['factors']))                                     # getters.py:463-465
    sd[p + 'linear.weight'], sd[p + 'linear.bias'] = O._linear_init(g, d, cfg['D'])
    L = num_tokens // L0                                                    # num_tokens_bottleneck (getters.py:466-468)
    for s, (nl, f) in enumerate(zip(cfg['dec_layers'], up)):
        sd[f'{p}upscale_embeddings.{s}'] = torch.randn(f, d, generator=g)
        assert L % nc == 0, 'the reference builds num_events = num_tokens // num_channels (:57-66)'
        _stack_init(sd, f'{p}transformers.{s}.', g, d, H, ff, L, nl)
        L *= f
    for c, nv in enumerate(cfg['vocab']):
        sd[f'{p}pre_softmaxes.{c}.weight'], sd[f'{p}pre_softmaxes.{c}.bias'] = O._linear_init(g, nv, d)
    return sd


def synthetic_batch(cfg, seed=1234, B=None):
    """`tensor_dict['x']` of the student dataloader: (B, ticks, voices) token ids (student_encoder_trainer.py:240)."""
    B = cfg['B'] if B is None else B
    g = torch.Generator().manual_seed(seed)
    cols = [torch.randint(0, nv, (B, cfg['ticks'], 1), generator=g) for nv in cfg['vocab']]
    return {'x': torch.cat(cols, dim=2)}


# -----------------------------------------------------------------------------------------------
# forward pieces
# -----------------------------------------------------------------------------------------------
def embed(x, P, pre, nc):
    """DataProcessor.embed (data_processor.py:34-46): (B, E, C) -> (B, E, C, emb), channel c uses table c."""
    return torch.stack([P[f'{pre}embeddings.{c}.weight'][x[..., c]] for c in range(nc)], dim=-2)


def _stack(x, P, pre, num_layers, H, p_drop, training, gen):
    for l in range(num_layers):
        x, _ = O.encoder_layer(x, P, f'{pre}layers.{l}.', H, p_drop, training, gen)
    return x


def linear_downscaler_forward(x_embed, P, cfg, pre='encoder.downscaler.', training=False, gen=None):
    """RelativeTransformerDownscalerLinear.forward (relative_transformer_downscaler_linear.py:99-139).
    x_embed (B, E, C, emb) -> z (B, nb, D).  Each stack ends in Linear(ds*d -> d) over `ds` consecutive tokens
    (reshape (L/ds, ds, n, d) -> permute -> (L/ds, n, ds*d), :129-133) instead of the `[::ds]` subsample."""
    B, E, C, emb = x_embed.shape
    L0 = int(math.prod(cfg['factors']))
    flat = x_embed.reshape(B, E * C, emb)
    nb = flat.shape[1] // L0
    n = B * nb
    x = O.linear(flat.reshape(n, L0, emb), P[pre + 'input_linear.weight'], P[pre + 'input_linear.bias'])
    chan = P[pre + 'target_channel_embeddings'].reshape(-1, cfg['pos'])
    ev = P[pre + 'events_positioning_embeddings'].reshape(-1, cfg['pos'])
    tok = torch.arange(L0)
    x = torch.cat([x, chan[tok % C].expand(n, L0, -1), ev[tok // C].expand(n, L0, -1)], dim=-1)
    for s, (nl, f) in enumerate(zip(cfg['enc_layers'], cfg['factors'])):
        x = _stack(x, P, f'{pre}transformers.{s}.', nl, cfg['H'], cfg['dropout'], training, gen)
        x = x.reshape(n, x.shape[1] // f, f * x.shape[2])
        x = O.linear(x, P[f'{pre}linear_aggs.{s}.weight'], P[f'{pre}linear_aggs.{s}.bias'])
    assert x.shape[1] == 1
    return O.linear(x[:, 0].reshape(B, nb, -1), P[pre + 'output_linear.weight'], P[pre + 'output_linear.bias'])


def encoder_forward(x, P, cfg, training=False, gen=None):
    """Encoder.forward with upscaler=None (encoder.py:76-95; encoder_student_config.py:50).
    x (B, E, C) -> (z_quantized (B, nb, D), idx (B, nb, ncb), qloss (B, nb))."""
    xe = embed(x.long(), P, 'encoder.data_processor.', len(cfg['vocab']))
    z = linear_downscaler_forward(xe, P, cfg, training=training, gen=gen)
    codebooks = [P[f'encoder.quantizer.embeddings.{c}'] for c in range(cfg['ncb'])]
    return O.vq_forward(z, codebooks, cfg['beta'], cfg['squared']) + (z,)


def teacher_forward(x_embed, P, cfg, pre='teacher.', training=False, gen=None):
    """TeacherRelative.forward (teacher_relative.py:57-87).  x_embed (B, E, C, emb) -> list of C logits (B, E, V_c)."""
    B, E, C, emb = x_embed.shape
    x = O.linear(x_embed, P[pre + 'linear_to_input_transformer.weight'], P[pre + 'linear_to_input_transformer.bias'])
    x = x.reshape(B, E * C, -1)
    ch = P[pre + 'channel_embeddings'].reshape(C, -1)
    x = torch.cat([x, ch.repeat(E, 1).expand(B, E * C, -1)], dim=2)
    x = _stack(x, P, pre + 'transformer.', cfg['teacher_layers'], cfg['H'], cfg['dropout'], training, gen)
    x = x.reshape(B, E, C, -1)
    return [O.linear(x[:, :, c], P[f'{pre}pre_softmaxes.{c}.weight'], P[f'{pre}pre_softmaxes.{c}.bias']) for c in range(C)]


def upscale(x, factor, emb):
    """AuxiliaryDecoderRelative.upscale (auxiliary_decoder_relative.py:116-130): token t -> `factor` copies, copy u
    gets emb[u] added.  x (n, L, d) -> (n, L*factor, d)."""
    n, L, d = x.shape
    assert emb.shape[0] == factor
    return (x.unsqueeze(2) + emb.view(1, 1, factor, d)).reshape(n, L * factor, d)


def aux_decoder_forward(zq, P, cfg, pre='auxiliary_decoder.', training=False, gen=None):
    """AuxiliaryDecoderRelative.forward (auxiliary_decoder_relative.py:82-114).
    zq (B, nb, D) -> list of C logits (B, E, V_c)."""
    B = zq.shape[0]
    C = len(cfg['vocab'])
    x = O.linear(zq, P[pre + 'linear.weight'], P[pre + 'linear.bias'])
    for s, (nl, f) in enumerate(zip(cfg['dec_layers'], reversed(cfg['factors']))):
        x = _stack(x, P, f'{pre}transformers.{s}.', nl, cfg['H'], cfg['dropout'], training, gen)
        x = upscale(x, f, P[f'{pre}upscale_embeddings.{s}'])
    x = x.reshape(B, x.shape[1] // C, C, -1)
    return [O.linear(x[:, :, c], P[f'{pre}pre_softmaxes.{c}.weight'], P[f'{pre}pre_softmaxes.{c}.bias']) for c in range(C)]


# -----------------------------------------------------------------------------------------------
# masking and losses
# -----------------------------------------------------------------------------------------------
def mask_teacher(x, masked_event_index, num_events_masked, vocab):
    """StudentEncoderTrainer.mask_teacher (student_encoder_trainer.py:144-184) for a given event index (the reference
    draws ONE index per batch with torch.randint(high=num_events, size=()).item(), :159-160).
    x (B, E, C) -> masked_x (tokens of events [m - k, m + k] replaced by the mask token V_c), notes_to_be_predicted."""
    B, E, C = x.shape
    m, k = masked_event_index, num_events_masked
    notes = torch.zeros_like(x)
    notes[:, m] = 1
    lo, hi = max(m - k, 0), min(m + k + 1, E)
    masked = x.clone()
    masked[:, lo:hi] = torch.tensor(vocab, dtype=x.dtype).view(1, 1, C)
    return masked, notes


def _log_softmax(x):
    return x - torch.logsumexp(x, dim=-1, keepdim=True)


def categorical_crossentropy(value, target, mask):
    """utils.categorical_crossentropy (utils.py:24-49): per channel, CE on the masked positions (batch-major order),
    summed over channels.  -> (B * masked events per row,)"""
    total = 0
    for c, logits in enumerate(value):
        sel = mask[..., c].bool()
        lp = _log_softmax(logits[sel])
        total = total - lp.gather(1, target[..., c][sel].view(-1, 1)).squeeze(1)
    return total


def distilled_categorical_crossentropy(value, target, mask):
    """utils.distilled_categorical_crossentropy (utils.py:131-159): for every (channel, event) whose mask is on for
    more than half of the batch, -sum softmax(teacher) * log_softmax(student) per batch row; summed.  -> (B,)"""
    total = 0
    for c, (student, teacher) in enumerate(zip(value, target)):
        for e in range(student.shape[1]):
            if float(mask[:, e, c].float().mean()) > 0.5:
                p = torch.softmax(teacher[:, e], dim=1)
                total = total - (p * _log_softmax(student[:, e])).sum(dim=1)
    return total


def student_losses(x, masked_event_index, P, cfg, training=False, gen=None):
    """forward_teacher + forward_encdec (student_encoder_trainer.py:120-142, 186-218)."""
    C = len(cfg['vocab'])
    x = x.long()
    masked_x, notes = mask_teacher(x, masked_event_index, cfg['num_events_masked'], cfg['vocab'])
    t_logits = teacher_forward(embed(masked_x, P, 'teacher.data_processor.', C), P, cfg, training=training, gen=gen)
    loss_teacher = categorical_crossentropy(t_logits, x, notes).mean()
    zq, idx, qloss, z = encoder_forward(x, P, cfg, training, gen)
    s_logits = aux_decoder_forward(zq, P, cfg, training=training, gen=gen)
    rec = distilled_categorical_crossentropy(s_logits, [t.detach() for t in t_logits], notes)
    loss_encdec = cfg['qw'] * qloss.mean() + rec.mean()
    return dict(loss_teacher=loss_teacher, loss_encdec=loss_encdec, loss_quantization=qloss.mean(),
                loss_reconstruction=rec.mean(), idx=idx, z=z, zq=zq, qloss=qloss, teacher_logits=t_logits,
                student_logits=s_logits, masked_x=masked_x, notes_to_be_predicted=notes)


# -----------------------------------------------------------------------------------------------
# one step / epoch
# -----------------------------------------------------------------------------------------------
GROUPS = ('teacher.', 'auxiliary_decoder.', 'encoder.')


class StudentOracleTrainer:
    """Two Adam optimisers (teacher | auxiliary decoder + encoder, student_encoder_trainer.py:51-61), three clip
    groups (:252, :268-269), the reference's epoch() return contract (:275-293)."""

    def __init__(self, cfg, state_dict, lr=1e-5, schedule_lr=False):
        self.cfg = cfg
        self.P = {k: v.detach().clone().float().requires_grad_(True) for k, v in state_dict.items()}
        self.lr, self.schedule_lr = lr, schedule_lr
        self.opt_teacher, self.opt_encdec = {}, {}
        self.sched_step = 0
        self.last_grads = None
        self.last_grad_norms = None

    def draw_masked_event(self):
        """:159-160 -- consumes the global CPU generator exactly as the reference does."""
        return int(torch.randint(high=self.cfg['ticks'], size=()).item())

    def step(self, batch, train, masked_event_index=None, gen=None):
        m = self.draw_masked_event() if masked_event_index is None else masked_event_index
        out = student_losses(batch['x'], m, self.P, self.cfg, training=train, gen=gen)
        out['masked_event_index'] = m
        if train:
            names = list(self.P.keys())
            t_names = [k for k in names if k.startswith('teacher.')]
            o_names = [k for k in names if not k.startswith('teacher.')]
            gt = torch.autograd.grad(out['loss_teacher'], [self.P[k] for k in t_names], allow_unused=True)
            go = torch.autograd.grad(out['loss_encdec'], [self.P[k] for k in o_names], allow_unused=True)
            grads = {k: (g if g is not None else torch.zeros_like(self.P[k]))
                     for k, g in list(zip(t_names, gt)) + list(zip(o_names, go))}
            self.last_grads = {k: g.clone() for k, g in grads.items()}
            self.last_grad_norms = {grp: O.clip_grad_norm([grads[k] for k in names if k.startswith(grp)], 5.0)
                                    for grp in GROUPS}
            lr = self.lr * (O.lr_lambda(self.sched_step) if self.schedule_lr else 1.0)
            with torch.no_grad():
                O.adam_step(self.P, {k: grads[k] for k in t_names}, self.opt_teacher, lr)
                # Adam's parameter order is auxiliary decoder first, then encoder (:52-55); per-tensor updates
                # are independent, so the dict order is immaterial
                O.adam_step(self.P, {k: grads[k] for k in o_names}, self.opt_encdec, lr)
            if self.schedule_lr:
                self.sched_step += 1
        return out

    def epoch(self, data_loader, train, num_batches, corrupt_labels=False, gen=None):
        keys = ('loss_teacher', 'loss_quantization', 'loss_reconstruction', 'loss_encdec')
        means = {k: 0.0 for k in keys}
        means['loss_monitor'] = 0.0
        n = -1
        for n, batch in enumerate(data_loader):
            if num_batches is not None and n >= num_batches:
                n -= 1
                break
            out = self.step(batch, train, gen=gen)
            for k in keys:
                means[k] += float(out[k].detach())
            means['loss_monitor'] += float(out['loss_reconstruction'].detach())
        return {k: v / (n + 1) for k, v in means.items()}
