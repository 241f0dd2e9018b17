"""CPU oracle for the VQ-CPC encoder training step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, in plain CPU fp32 tensor arithmetic, the algorithm of the reference's hot path
(`VQCPCEncoderTrainer.epoch`, /root/reference/VQCPCB/vqcpc_encoder_trainer.py:169-354 and everything it
calls).  It exists to CHECK the HIP path:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it;
  * the package `vqcpc_bach_amd` never imports it and has no CPU fallback (it raises if the HIP
    library is missing);
  * it is pinned against golden vectors produced by importing the reference itself
    (tools/gen_golden.py -> tests/golden/*.npz, torch 2.10.0 CPU fp32) by tests/test_oracle_golden.py.
    The reference has no tests / known-answer vectors of its own.

Every function cites the reference lines it follows.  Tensors are laid out block-major
`(blocks, L, d)`; the reference's transformer is time-first `(L, blocks, d)` -- a pure transposition.

Parameters live in one flat dict whose keys are the reference's state_dict names, prefixed by the
sub-module file name used by its checkpoints (`encoder.downscaler....`, `c_module....`, `fks_module.W`).

Third-party arithmetic: the reference's numerics are PyTorch's (unpinned in requirements.txt; 2.10.0
here).  GRU, LayerNorm, softmax, logsumexp, Adam and clip_grad_norm_ are restated from their
published definitions below and checked against the reference's outputs through the fixtures.
"""
import math

import torch

# -----------------------------------------------------------------------------------------------
# configuration
# -----------------------------------------------------------------------------------------------
DEFAULTS = dict(
    emb=32,                 # data_processor_kwargs.embedding_size (configs/encoder_random_transfo_config.py:29)
    vocab=[56, 56, 56, 56],  # synthetic (reference derives it from the music21 corpus, getters.py:506)
    d=256, H=8, layers=[2, 2], ff=1024,
    D=32, K=512, ncb=2,     # quantizer: codebook_dim, codebook_size, num_codebooks
    zdim=32, up_hidden=512,  # MlpUpscaler
    cdim=32, gru_hidden=512, gru_layers=2,
    B=256, N=15, Kl=8, Kr=8,
    dropout=0.0, beta=0.25, qw=0.5, bidirectional=False, squared=True,
    pos=8,                  # positional_embedding_size (relative_transformer_downscaler.py:34)
    factors=[4, 4],         # downscale_factors
)

CONFIGS = {
    # BASELINE.json configs[0] / SURVEY.md section 8 legend C0
    'C0': dict(d=128, H=4, layers=[1, 1], ff=512, D=16, K=64, ncb=1, B=8, Kl=2, Kr=2),
    # configs[1] / C1 (the configuration the metric is quoted on)
    'C1': dict(d=256, H=8, layers=[2, 2], ff=1024, D=32, K=512, ncb=2, B=256, Kl=8, Kr=8),
    # configs[4] / C4
    'C4': dict(d=512, H=8, layers=[4, 4], ff=2048, D=64, K=1024, ncb=4, B=256, Kl=16, Kr=16),
}


def make_cfg(name=None, **over):
    cfg = dict(DEFAULTS)
    if name is not None:
        cfg.update(CONFIGS[name])
    cfg.update(over)
    return cfg


# -----------------------------------------------------------------------------------------------
# parameter initialisation (same distributions as the reference constructors)
# -----------------------------------------------------------------------------------------------
def _xavier(gen, rows, cols):
    bound = math.sqrt(6.0 / (rows + cols))
    return (torch.rand(rows, cols, generator=gen) * 2 - 1) * bound


def _linear_init(gen, out_f, in_f):
    # nn.Linear default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)) for weight and bias
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound
    return w, b


def init_state(cfg, seed=0):
    """Random-init state dict with the reference's key names / shapes."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    d, H, ff, emb, pos = cfg['d'], cfg['H'], cfg['ff'], cfg['emb'], cfg['pos']
    hd = d // H
    for v, nv in enumerate(cfg['vocab']):  # data_processor.py:26-32 (one extra mask token)
        sd[f'encoder.data_processor.embeddings.{v}.weight'] = torch.randn(nv + 1, emb, generator=g)
    p = 'encoder.downscaler.'
    sd[p + 'target_channel_embeddings'] = torch.randn(1, 1, 4, pos, generator=g)
    sd[p + 'events_positioning_embeddings'] = torch.randn(1, 1, 4, pos, generator=g)
    sd[p + 'input_linear.weight'], sd[p + 'input_linear.bias'] = _linear_init(g, d - 2 * pos, emb)
    sd[p + 'output_linear.weight'], sd[p + 'output_linear.bias'] = _linear_init(g, cfg['D'], d)
    L = 16
    for s, nl in enumerate(cfg['layers']):
        # _get_clones deep-copies one layer (transformer_custom.py:138): identical init inside a stack
        lay = {}
        lay['self_attn.in_proj_weight'] = _xavier(g, 3 * d, d)
        lay['self_attn.in_proj_bias'] = torch.zeros(3 * d)
        w, _ = _linear_init(g, d, d)
        lay['self_attn.out_proj.weight'], lay['self_attn.out_proj.bias'] = w, torch.zeros(d)
        lay['self_attn.attn_bias.e1'] = torch.randn(H * L, hd, generator=g)
        lay['self_attn.attn_bias.e2'] = torch.randn(H * L, hd, generator=g)
        lay['linear1.weight'], lay['linear1.bias'] = _linear_init(g, ff, d)
        lay['linear2.weight'], lay['linear2.bias'] = _linear_init(g, d, ff)
        for n in ('norm1', 'norm2'):
            lay[n + '.weight'], lay[n + '.bias'] = torch.ones(d), torch.zeros(d)
        for l in range(nl):
            for k, v in lay.items():
                sd[f'{p}transformers.{s}.layers.{l}.{k}'] = v.clone()
        L //= cfg['factors'][s]
    dsub = cfg['D'] // cfg['ncb']
    for c in range(cfg['ncb']):  # vector_quantizer.py:44-48
        sd[f'encoder.quantizer.embeddings.{c}'] = torch.randn(cfg['K'], dsub, generator=g) * 4
    sd['encoder.upscaler.mlp.0.weight'], sd['encoder.upscaler.mlp.0.bias'] = _linear_init(g, cfg['up_hidden'], cfg['D'])
    sd['encoder.upscaler.mlp.3.weight'], sd['encoder.upscaler.mlp.3.bias'] = _linear_init(g, cfg['zdim'], cfg['up_hidden'])
    names = ['c_module', 'fks_module'] + (['c_module_back', 'fks_module_back'] if cfg['bidirectional'] else [])
    for name in names:
        if name.startswith('c_module'):
            hid = cfg['gru_hidden']
            bound = 1.0 / math.sqrt(hid)
            for l in range(cfg['gru_layers']):
                inp = cfg['zdim'] if l == 0 else hid
                sd[f'{name}.g_ar_fwd.weight_ih_l{l}'] = (torch.rand(3 * hid, inp, generator=g) * 2 - 1) * bound
                sd[f'{name}.g_ar_fwd.weight_hh_l{l}'] = (torch.rand(3 * hid, hid, generator=g) * 2 - 1) * bound
                sd[f'{name}.g_ar_fwd.bias_ih_l{l}'] = (torch.rand(3 * hid, generator=g) * 2 - 1) * bound
                sd[f'{name}.g_ar_fwd.bias_hh_l{l}'] = (torch.rand(3 * hid, generator=g) * 2 - 1) * bound
            sd[f'{name}.output_linear.weight'], sd[f'{name}.output_linear.bias'] = _linear_init(g, cfg['cdim'], hid)
        else:
            sd[f'{name}.W'] = torch.randn(cfg['zdim'], cfg['cdim'], cfg['Kr'], generator=g)  # vqcpc_helper.py:84
    return sd


def synthetic_batch(cfg, seed=1234, B=None):
    """Batch-dict contract of BachCPCDataloaderGenerator (dataloaders/bach_cpc_dataloader.py:183-259):
    x_left (B, Kl*4, 4), x_right (B, Kr*4, 4), negative_samples[_back] (B, N, Kr, 4, 4); last dim = voice."""
    B = cfg['B'] if B is None else B
    g = torch.Generator().manual_seed(seed)
    V, N, Kl, Kr = cfg['vocab'][0], cfg['N'], cfg['Kl'], cfg['Kr']
    return {
        'x_left': torch.randint(0, V, (B, Kl * 4, 4), generator=g),
        'x_right': torch.randint(0, V, (B, Kr * 4, 4), generator=g),
        'negative_samples': torch.randint(0, V, (B, N, Kr, 4, 4), generator=g),
        'negative_samples_back': torch.randint(0, V, (B, N, Kr, 4, 4), generator=g),
    }


def same_sequence_negatives(first, second, ticks_per_block=4):
    """BachCPCDataloaderGenerator._build_negatives_sameSeq (dataloaders/bach_cpc_dataloader.py:163-181) in the batch-dict
    layout (ticks, voices): for every target block k of `second`, every block of `first` followed by the blocks of
    `second` except k.  first (B, Ta, V), second (B, Tb, V) -> (B, Ka + Kb - 1, Kb, ticks_per_block, V).
    negative_samples = f(x_left, x_right); negative_samples_back = f(x_right, x_left) (:131-132)."""
    B, Ta, V = first.shape
    a = first.reshape(B, Ta // ticks_per_block, ticks_per_block, V)
    b = second.reshape(B, second.shape[1] // ticks_per_block, ticks_per_block, V)
    Kb = b.shape[1]
    cols = [torch.cat([a, b[:, :k], b[:, k + 1:]], dim=1) for k in range(Kb)]      # each (B, N, tpb, V)
    return torch.stack(cols, dim=2)


# -----------------------------------------------------------------------------------------------
# A1/A2: blocks and embedding
# -----------------------------------------------------------------------------------------------
def preprocess_blocks(x, tokens_per_block=16):
    """bach_cpc_data_processor.py:17-40: (..., ticks, voices) -> (..., nb, 16) int64, token p = tick*4 + voice."""
    lead = tuple(x.shape[:-2])
    flat = x.reshape(*lead, x.shape[-2] * x.shape[-1])
    assert flat.shape[-1] % tokens_per_block == 0
    return flat.reshape(*lead, flat.shape[-1] // tokens_per_block, tokens_per_block).long()


def embed_blocks(tokens, tables):
    """bach_cpc_data_processor.py:42-68: voice v = p % 4 uses table v.  (..., 16) -> (..., 16, emb)."""
    nv = len(tables)
    out = torch.stack([tables[p % nv][tokens[..., p]] for p in range(tokens.shape[-1])], dim=-2)
    return out


# -----------------------------------------------------------------------------------------------
# A4-A8: relative-attention transformer downscaler
# -----------------------------------------------------------------------------------------------
def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim: biased variance, eps inside the sqrt."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def dropout(x, p, training, gen=None):
    if not training or p == 0.0:
        return x
    keep = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


def relative_bias(q, e1, e2):
    """Closed form of SubsampledRelativeAttention.forward (subsampled_relative_attention.py:30-122) for
    seq_len_src == seq_len_tgt == L (always true on the encoder path, transformer_custom.py:245-253):

        bias[h, i, j] = q[h, i] . e1[h, L-1-(i-j)]   if j <= i      (causal half, "skewed" by pad/view)
                      = q[h, i] . e2[h, j-i]         if j >  i      (anticausal half)

    q: (n, H, L, hd) (already multiplied by hd**-0.5, multihead_attention_custom.py:247,331-333)
    e1, e2: (H*L, hd) viewed as (H, L, hd) (subsampled_relative_attention.py:44-45)."""
    n, H, L, hd = q.shape
    a1 = torch.einsum('nhld,hmd->nhlm', q, e1.view(H, L, hd))
    a2 = torch.einsum('nhld,hmd->nhlm', q, e2.view(H, L, hd))
    i = torch.arange(L).view(L, 1)
    j = torch.arange(L).view(1, L)
    m1 = (L - 1 - i + j).clamp(0, L - 1).expand(n, H, L, L)
    m2 = (j - i).clamp(0, L - 1).expand(n, H, L, L)
    return torch.where(j <= i, a1.gather(-1, m1), a2.gather(-1, m2))


def self_attention(x, P, pre, H, p_drop=0.0, training=False, gen=None):
    """MultiheadAttentionCustom.forward, self-attention branch (multihead_attention_custom.py:171,247,
    289-346).  x: (n, L, d).  Returns (out, probs) with probs (n, H, L, L) after dropout as the reference."""
    n, L, d = x.shape
    hd = d // H
    qkv = linear(x, P[pre + 'in_proj_weight'], P[pre + 'in_proj_bias'])
    q, k, v = qkv.split(d, dim=-1)
    q = q * (float(hd) ** -0.5)
    q, k, v = (t.reshape(n, L, H, hd).transpose(1, 2) for t in (q, k, v))       # (n, H, L, hd)
    scores = q @ k.transpose(-1, -2) + relative_bias(q, P[pre + 'attn_bias.e1'], P[pre + 'attn_bias.e2'])
    probs = torch.softmax(scores, dim=-1)
    probs = dropout(probs, p_drop, training, gen)
    ctx = (probs @ v).transpose(1, 2).reshape(n, L, d)
    return linear(ctx, P[pre + 'out_proj.weight'], P[pre + 'out_proj.bias']), probs


def encoder_layer(x, P, pre, H, p_drop=0.0, training=False, gen=None):
    """TransformerEncoderLayerCustom.forward (transformer_custom.py:268-291), post-LN."""
    a, probs = self_attention(x, P, pre + 'self_attn.', H, p_drop, training, gen)
    x = layer_norm(x + dropout(a, p_drop, training, gen), P[pre + 'norm1.weight'], P[pre + 'norm1.bias'])
    h = torch.relu(linear(x, P[pre + 'linear1.weight'], P[pre + 'linear1.bias']))
    h = linear(dropout(h, p_drop, training, gen), P[pre + 'linear2.weight'], P[pre + 'linear2.bias'])
    x = layer_norm(x + dropout(h, p_drop, training, gen), P[pre + 'norm2.weight'], P[pre + 'norm2.bias'])
    return x, probs


def downscaler_forward(x_embed, P, cfg, pre='encoder.downscaler.', training=False, gen=None):
    """RelativeTransformerDownscaler.forward (relative_transformer_downscaler.py:93-133).
    x_embed: (rows, nb, 16, emb) -> z (rows, nb, D)."""
    rows, nb, L, emb = x_embed.shape
    n = rows * nb
    x = linear(x_embed.reshape(n, L, emb), P[pre + 'input_linear.weight'], P[pre + 'input_linear.bias'])
    chan = P[pre + 'target_channel_embeddings'].reshape(-1, cfg['pos'])        # (4, 8)   index p % 4
    ev = P[pre + 'events_positioning_embeddings'].reshape(-1, cfg['pos'])      # (4, 8)   index p // 4
    nc = chan.shape[0]
    tok = torch.arange(L)
    x = torch.cat([x, chan[tok % nc].expand(n, L, -1), ev[tok // nc].expand(n, L, -1)], dim=-1)
    for s, (nl, f) in enumerate(zip(cfg['layers'], cfg['factors'])):
        for l in range(nl):
            x, _ = encoder_layer(x, P, f'{pre}transformers.{s}.layers.{l}.', cfg['H'], cfg['dropout'], training, gen)
        x = x[:, ::f]                                                          # output[::downscaling], :125
    assert x.shape[1] == 1
    return linear(x[:, 0].reshape(rows, nb, -1), P[pre + 'output_linear.weight'], P[pre + 'output_linear.bias'])


# -----------------------------------------------------------------------------------------------
# A9-A12: product vector quantiser
# -----------------------------------------------------------------------------------------------
def vq_distances_canonical(x, e):
    """Squared distances (rows, K) in the CANONICAL order shared with the HIP kernel:
    d = 0; for t ascending: d = fl(d + fl(fl(x_t - e_t)^2)), no fused multiply-add.
    (vector_quantizer.py:105-112 computes sum((x-e)**2, dim=2); torch's vectorised sum order depends on
    the host ISA and on dsub, so the build fixes this order -- argmin agrees with the reference on every
    fixture, see tests/test_oracle_golden.py.)"""
    d = torch.zeros(x.shape[0], e.shape[0], dtype=torch.float32)
    for t in range(x.shape[1]):
        diff = x[:, t:t + 1] - e[:, t].unsqueeze(0)
        d = d + diff * diff
    return d


def vq_assign(z_flat, codebooks):
    """argmin per codebook, first index on ties (vector_quantizer.py:115-116). -> (rows, ncb) int64"""
    ncb = len(codebooks)
    with torch.no_grad():
        cols = []
        for xc, e in zip(z_flat.detach().chunk(ncb, dim=1), codebooks):
            cols.append(torch.argmin(vq_distances_canonical(xc.contiguous(), e.detach()), dim=1))
    return torch.stack(cols, dim=1)


def vq_forward(z, codebooks, beta=0.25, squared=True, idx=None):
    """ProductVectorQuantizer.forward without batch-norm / init / label corruption
    (vector_quantizer.py:85-159).  z (..., D) -> quantized_sg (..., D), idx (..., ncb), loss (...)."""
    shape = z.shape
    flat = z.reshape(-1, shape[-1])
    if idx is None:
        idx = vq_assign(flat, codebooks)
    quantized = torch.cat([e[idx[:, c]] for c, e in enumerate(codebooks)], dim=1).reshape(shape)
    if squared:                                                                 # _loss, :72-83
        e_lat = ((quantized.detach() - z) ** 2).sum(-1)
        q_lat = ((quantized - z.detach()) ** 2).sum(-1)
    else:
        eps = 1e-5
        e_lat = torch.norm((quantized.detach() - z) + eps, dim=-1)
        q_lat = torch.norm((quantized - z.detach()) + eps, dim=-1)
    loss = q_lat + beta * e_lat
    quantized_sg = z + (quantized - z).detach()                                 # straight-through, :148
    return quantized_sg, idx.reshape(*shape[:-1], len(codebooks)), loss


def vq_data_init(flat, codebooks, gen=None):
    """ProductVectorQuantizer._initialize (vector_quantizer.py:57-70): each codebook takes the k-th slice of a
    fresh random permutation of the batch rows."""
    assert flat.shape[0] >= codebooks[0].shape[0], 'not enough elements in a batch to initialise the clusters'
    out = []
    for k, e in enumerate(codebooks):
        perm = torch.randperm(flat.shape[0], generator=gen)
        out.append(flat[perm][:e.shape[0], k * e.shape[1]:(k + 1) * e.shape[1]].detach().clone())
    return out


def merge_codes(idx, K):
    """Encoder.merge_codes (encoder.py:97-110) without its aliasing bug: sum_c idx[..., c] * K**c."""
    out = idx[..., 0].clone()
    for c in range(1, idx.shape[-1]):
        out = out + idx[..., c] * (K ** c)
    return out


# -----------------------------------------------------------------------------------------------
# A13-A18: upscaler, context GRU, bilinear scores, InfoNCE
# -----------------------------------------------------------------------------------------------
def selu(x):
    alpha, scale = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
    return scale * torch.where(x > 0, x, alpha * (torch.exp(x) - 1))


def upscaler_forward(zq, P, pre='encoder.upscaler.', p_drop=0.0, training=False, gen=None):
    """MlpUpscaler.forward (mlp_upscaler.py:21-34): Linear -> Dropout -> SELU -> Linear."""
    h = linear(zq, P[pre + 'mlp.0.weight'], P[pre + 'mlp.0.bias'])
    h = selu(dropout(h, p_drop, training, gen))
    return linear(h, P[pre + 'mlp.3.weight'], P[pre + 'mlp.3.bias'])


def gru_context(zs, P, pre, num_layers, p_drop=0.0, training=False, gen=None):
    """CModule.forward (vqcpc_helper.py:54-76): multi-layer GRU (gate order r, z, n; h0 = 0; dropout on the
    outputs of every layer but the last), last time step, Linear."""
    B, T, _ = zs.shape
    x = zs
    for l in range(num_layers):
        w_ih, w_hh = P[f'{pre}g_ar_fwd.weight_ih_l{l}'], P[f'{pre}g_ar_fwd.weight_hh_l{l}']
        b_ih, b_hh = P[f'{pre}g_ar_fwd.bias_ih_l{l}'], P[f'{pre}g_ar_fwd.bias_hh_l{l}']
        hid = w_hh.shape[1]
        h = torch.zeros(B, hid, dtype=zs.dtype)
        gi_all = linear(x, w_ih, b_ih)                                          # (B, T, 3*hid)
        outs = []
        for t in range(T):
            gi = gi_all[:, t]
            gh = linear(h, w_hh, b_hh)
            r = torch.sigmoid(gi[:, :hid] + gh[:, :hid])
            u = torch.sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
            n = torch.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
            h = (1 - u) * n + u * h
            outs.append(h)
        x = torch.stack(outs, dim=1)
        if l < num_layers - 1:
            x = dropout(x, p_drop, training, gen)
    return linear(x[:, -1], P[pre + 'output_linear.weight'], P[pre + 'output_linear.bias'])


def fks_scores(c, W, z_pos, z_neg):
    """FksModule.forward for the positive and the N negative sets (vqcpc_helper.py:86-98 and the reshuffling of
    vqcpc_encoder_trainer.py:240-263):  Wc[b,k,:] = sum_c W[:, c, k] c[b, c];  f = <Wc[b,k], z[b,k]>.
    c (B, cdim), W (zdim, cdim, K), z_pos (B, K, zdim), z_neg (B, N, K, zdim) -> (B, K), (B, K, N)."""
    wc = torch.einsum('bc,zck->bkz', c, W)
    f_pos = (wc * z_pos).sum(-1)
    f_neg = torch.einsum('bkz,bnkz->bkn', wc, z_neg)
    return f_pos, f_neg


def nce_loss(f_pos, f_neg):
    """vqcpc_helper.py:5-29: -mean_b sum_k (pos - logsumexp([negatives, pos]))."""
    allf = torch.cat([f_neg, f_pos.unsqueeze(2)], dim=2)
    m = allf.max(dim=2, keepdim=True)[0]
    lse = m.squeeze(2) + torch.log(torch.exp(allf - m).sum(2))
    return -(f_pos - lse).sum(1).mean(0)


def quantization_loss(ql_left, ql_neg, ql_right, ql_neg_back=None):
    """vqcpc_helper.py:32-51: mean over the 3B (4B) per-window sums."""
    parts = [ql_left.sum(1), ql_right.sum(1), ql_neg.flatten(1).sum(1)]
    if ql_neg_back is not None:
        parts.append(ql_neg_back.flatten(1).sum(1))
    return torch.cat(parts, dim=0).mean()


# -----------------------------------------------------------------------------------------------
# A14: Encoder.forward
# -----------------------------------------------------------------------------------------------
def encoder_forward(x, P, cfg, training=False, gen=None, stages=None):
    """Encoder.forward (encoder.py:76-95).  x (..., ticks, 4) int -> (z_up (rows, nb, zdim), idx (rows, nb, ncb),
    qloss (rows, nb)).  `stages` (dict) receives the intermediate tensors when given."""
    tokens = preprocess_blocks(x)
    tokens = tokens.reshape(-1, tokens.shape[-2], tokens.shape[-1])
    tables = [P[f'encoder.data_processor.embeddings.{v}.weight'] for v in range(len(cfg['vocab']))]
    x_embed = embed_blocks(tokens, tables)
    z = downscaler_forward(x_embed, P, cfg, training=training, gen=gen)
    codebooks = [P[f'encoder.quantizer.embeddings.{c}'] for c in range(cfg['ncb'])]
    zq, idx, qloss = vq_forward(z, codebooks, cfg['beta'], cfg['squared'])
    z_up = upscaler_forward(zq, P, p_drop=cfg['dropout'], training=training, gen=gen)
    if stages is not None:
        stages.update(tokens=tokens, embed=x_embed, z=z, idx=idx, zq=zq, qloss=qloss, zup=z_up)
    return z_up, idx, qloss


# -----------------------------------------------------------------------------------------------
# A20/A21: one training / evaluation step and the epoch wrapper
# -----------------------------------------------------------------------------------------------
def cpc_losses(batch, P, cfg, training=False, gen=None):
    """Forward half of VQCPCEncoderTrainer.epoch (vqcpc_encoder_trainer.py:195-307)."""
    neg = batch['negative_samples']
    B, N, Kr, ev, nch = neg.shape
    zq_n, idx_n, ql_n = encoder_forward(neg.reshape(B * N * Kr, ev, nch), P, cfg, training, gen)
    zq_n = zq_n.reshape(B, N, Kr, -1, zq_n.shape[-1])
    ql_n = ql_n.reshape(B, N, Kr, -1)
    if cfg['bidirectional']:
        nb_ = batch['negative_samples_back']
        zq_nb, idx_nb, ql_nb = encoder_forward(nb_.reshape(B * N * Kr, ev, nch), P, cfg, training, gen)
        zq_nb = zq_nb.reshape(B, N, Kr, -1, zq_nb.shape[-1])
        ql_nb = ql_nb.reshape(B, N, Kr, -1)
    else:
        ql_nb = None
    z_l, idx_l, ql_l = encoder_forward(batch['x_left'], P, cfg, training, gen)
    z_r, idx_r, ql_r = encoder_forward(batch['x_right'], P, cfg, training, gen)

    c = gru_context(z_l, P, 'c_module.', cfg['gru_layers'], cfg['dropout'], training, gen)
    f_pos, f_neg = fks_scores(c, P['fks_module.W'], z_r, zq_n[:, :, :, 0, :])
    margin = (f_pos - f_neg.max(2)[0]).detach()          # sign = hit; |margin| ~ ulp marks a rounding-level tie
    score = f_pos > f_neg.max(2)[0]
    contrastive = nce_loss(f_pos, f_neg)
    score_b = margin_b = None
    if cfg['bidirectional']:                                                    # :277-296
        c_b = gru_context(z_r.flip(dims=[1]), P, 'c_module_back.', cfg['gru_layers'], cfg['dropout'], training, gen)
        # Reference quirk, kept for parity: unlike the forward direction (:245-255) the backward negatives are NOT
        # permuted to negative-major before FksModule's raw `.view` (:286-292 + vqcpc_helper.py:94-95), so row
        # r = n*B + b of the (N*B, K, z) view reads z_neg_back.reshape(B*N, K, z)[r]: window b is scored against
        # negatives drawn from other windows' sets.
        zb = zq_nb[:, :, :, 0, :]
        zb = zb.reshape(B * N, Kr, zb.shape[-1]).reshape(N, B, Kr, zb.shape[-1]).permute(1, 0, 2, 3)
        f_pos_b, f_neg_b = fks_scores(c_b, P['fks_module_back.W'], z_l, zb)
        margin_b = (f_pos_b - f_neg_b.max(2)[0]).detach()
        score_b = f_pos_b > f_neg_b.max(2)[0]
        contrastive = contrastive + nce_loss(f_pos_b, f_neg_b)
    q_loss = quantization_loss(ql_l, ql_n, ql_r, ql_nb)
    loss = contrastive + cfg['qw'] * q_loss
    acc = score.sum(0).float() / B
    if score_b is not None:
        acc = (acc + score_b.sum(0).float() / B) / 2
    K = cfg['K']
    ncw = len(torch.unique(torch.cat((merge_codes(idx_l, K), merge_codes(idx_r, K)), dim=0)))
    ncw_neg = len(torch.unique(merge_codes(idx_n, K)))
    return dict(loss=loss, loss_contrastive=contrastive, loss_quantize=q_loss, accuracy=acc,
                num_codewords=ncw, num_codewords_negative=ncw_neg,
                idx_left=idx_l, idx_right=idx_r, idx_negative=idx_n, margin=margin, margin_back=margin_b)


def clip_grad_norm(grads, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_ (2-norm): coef = min(1, max_norm / (total + 1e-6)).  In place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adam_step(P, grads, state, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (vqcpc_encoder_trainer.py:92): no weight decay, no amsgrad."""
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    for k, g in grads.items():
        m = state.setdefault('m/' + k, torch.zeros_like(g))
        v = state.setdefault('v/' + k, torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        P[k].data.addcdiv_(m, denom, value=-lr / bc1)


def lr_lambda(step):
    """LambdaLR factor of init_optimizers (vqcpc_encoder_trainer.py:96-107)."""
    warm, lo, hi = 10000, 0.1, 1.0
    s1 = (hi - lo) / warm
    return max(min(lo + s1 * step, hi + (step - warm) * (-s1 * 0.1)), lo)


class OracleTrainer:
    """State + `epoch()` with the reference's return contract (vqcpc_encoder_trainer.py:343-354)."""

    def __init__(self, cfg, state_dict, lr=1e-4, schedule_lr=False):
        self.cfg = cfg
        self.P = {k: v.detach().clone().float().requires_grad_(True) for k, v in state_dict.items()}
        self.lr, self.schedule_lr = lr, schedule_lr
        self.opt_state = {}
        self.sched_step = 0
        self.last_grads = None
        self.last_grad_norm = None

    def step(self, batch, train, gen=None):
        out = cpc_losses(batch, self.P, self.cfg, training=train, gen=gen)
        if train:
            names = list(self.P.keys())
            grads = torch.autograd.grad(out['loss'], [self.P[k] for k in names], allow_unused=True)
            grads = {k: (g if g is not None else torch.zeros_like(self.P[k])) for k, g in zip(names, grads)}
            self.last_grads = {k: g.clone() for k, g in grads.items()}
            self.last_grad_norm = clip_grad_norm(list(grads.values()), 5.0)
            lr = self.lr * (lr_lambda(self.sched_step) if self.schedule_lr else 1.0)
            with torch.no_grad():
                adam_step(self.P, grads, self.opt_state, lr)
            if self.schedule_lr:
                self.sched_step += 1
        return out

    def epoch(self, data_loader, train, num_batches, corrupt_labels=False, gen=None):
        assert not corrupt_labels, 'label corruption is RNG-defined; the oracle keeps it off (SURVEY.md A12)'
        means = dict(loss=0.0, accuracy=0.0, loss_quantize=0.0, loss_contrastive=0.0, num_codewords=0.0,
                     num_codewords_negative=0.0)
        n = 0
        for n, batch in enumerate(data_loader):
            if num_batches is not None and n >= num_batches:
                n -= 1
                break
            out = self.step(batch, train, gen)
            for k in ('loss', 'loss_quantize', 'loss_contrastive'):
                means[k] += float(out[k].detach())
            means['num_codewords'] += out['num_codewords']
            means['num_codewords_negative'] += out['num_codewords_negative']
            means['accuracy'] = means['accuracy'] + out['accuracy'].detach().numpy()
        means = {k: v / (n + 1) for k, v in means.items()}
        means['accuracy'] = list(means['accuracy'])
        means['loss_monitor'] = -sum(means['accuracy']) / len(means['accuracy'])
        return means
