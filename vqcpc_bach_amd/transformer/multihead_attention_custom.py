"""The reference's MultiheadAttentionCustom (VQCPCB/transformer/multihead_attention_custom.py:8-353): in_proj_weight/bias,
out_proj, attn_bias.{e1,e2} with the same names, shapes and initialisation, and `forward(query, key, value, ...,
attn_mask=...)` with the same signature and return contract.  On the training paths the arithmetic lives in
ops.EncoderLayerFn (encoder layers, fused) and in `forward_rows` below (decoder's masked / cross attentions, ops.AttnXFn);
`forward` is the API-compatible entry for callers that reach below the trainers and runs the same kernels."""
import torch
from torch import nn

from .subsampled_relative_attention import SubsampledRelativeAttention


def classify_additive_mask(attn_mask):
    """The additive (T, S) masks the reference builds (decoders/decoder.py:294-308: 0 = keep, -inf = masked; T = r S, every
    source position repeated r times along the target axis) -> the index rule the attention kernels evaluate:
    ops.MASK_CAUSAL keeps j <= i // r, ops.MASK_ANTICAUSAL keeps j >= i // r, an all-zero mask is ops.MASK_NONE.
    Entries <= -1e4 count as masked (the finite "large negative" masks of user code: exp(-1e4 - max) == 0 in fp32).
    Any other pattern raises NotImplementedError (the reference never builds one on the path).
    One small host read per mask OBJECT and content version: the class is remembered on the tensor object itself (it dies
    with the tensor; an address-keyed cache would hand a freed mask's class to whatever the allocator puts there next) next
    to the in-place version counter it was computed at, so a mask rebuilt per forward -- as the reference does -- is read
    again and a mask kept by the caller is read once."""
    from .. import ops
    assert attn_mask.dim() == 2, 'attn_mask: (target length, source length)'
    hit = getattr(attn_mask, '_vqcpc_mask_class', None)
    if hit is not None and hit[0] == attn_mask._version:
        return hit[1]
    T, S = attn_mask.shape
    m = attn_mask.detach().to('cpu', torch.float32)
    keep = m == 0
    if not bool((keep | (m <= -1e4)).all()):
        raise NotImplementedError('attn_mask: only additive masks of 0 (keep) and -inf / <= -1e4 (masked) are supported')
    cls = None
    if bool(keep.all()):
        cls = ops.MASK_NONE
    elif T % S == 0:
        p = (torch.arange(T) // (T // S)).unsqueeze(1)
        j = torch.arange(S).unsqueeze(0)
        if torch.equal(keep, j <= p):
            cls = ops.MASK_CAUSAL
        elif torch.equal(keep, j >= p):
            cls = ops.MASK_ANTICAUSAL
    if cls is None:
        raise NotImplementedError('attn_mask: not the causal / anticausal pattern of decoders/decoder.py:294-308')
    attn_mask._vqcpc_mask_class = (attn_mask._version, cls)
    return cls


class MultiheadAttentionCustom(nn.Module):
    def __init__(self, embed_dim, num_heads, attention_bias_type, num_channels_k, num_events_k, num_channels_q,
                 num_events_q, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False, kdim=None, vdim=None):
        super().__init__()
        if not bias or add_bias_kv or add_zero_attn or kdim not in (None, embed_dim) or vdim not in (None, embed_dim):
            raise NotImplementedError('only the configuration used by the encoder path is implemented')
        if attention_bias_type not in ('relative_attention', 'relative_attention_target_source'):
            raise NotImplementedError('Not a valid type of attention bias')
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, 'embed_dim must be divisible by num_heads'
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        seq_len_src, seq_len_tgt = num_channels_k * num_events_k, num_channels_q * num_events_q
        assert seq_len_tgt % seq_len_src == 0
        self.attn_bias = SubsampledRelativeAttention(head_dim=self.head_dim, num_heads=num_heads,
                                                     seq_len_src=seq_len_src, seq_len_tgt=seq_len_tgt)
        self.seq_len = seq_len_tgt
        self.seq_len_src = seq_len_src
        nn.init.xavier_uniform_(self.in_proj_weight)          # _reset_parameters, :106-120
        nn.init.constant_(self.in_proj_bias, 0.)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward_rows(self, x, n, mask, memory=None, drop_p=0.0, seed=0):
        """Masked attention on batch-major rows: x (n * T, d) queries; self-attention when `memory` is None, otherwise
        encoder-decoder attention on memory (n * S, d) (:171-196).  mask: ops.MASK_NONE / MASK_CAUSAL / MASK_ANTICAUSAL
        (the additive masks of decoders/decoder.py:292-308 as index rules).  -> (out (n * T, d), probs (n, H, T, S))."""
        from .. import ops
        T, S = self.seq_len, self.seq_len_src
        e1, e2 = self.attn_bias.e1, self.attn_bias.e2
        if memory is None:
            assert T == S
            qkv = ops.LinearFn.apply(x, self.in_proj_weight, self.in_proj_bias)
            att, probs = ops.AttnXFn.apply(qkv, None, e1, e2, n, T, S, self.num_heads, mask, drop_p, seed)
        else:
            q, kv = ops.CrossProjFn.apply(x, memory, self.in_proj_weight, self.in_proj_bias)
            att, probs = ops.AttnXFn.apply(q, kv, e1, e2, n, T, S, self.num_heads, mask, drop_p, seed)
        return ops.LinearFn.apply(att, self.out_proj.weight, self.out_proj.bias), probs

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None, static_k=None,
                static_v=None):
        """multihead_attention_custom.py:122-353.  query (L, N, E), key / value (S, N, E) time-first; self-attention when
        the three are the same tensor, encoder-decoder attention when key is value (:154-196); attn_mask: additive (L, S)
        matrix (0 / -inf) as built by decoders/decoder.py:294-308, or None.  -> (attn_output (L, N, E), attention weights
        (N, H, L, S) -- per head, as the reference returns them -- or None)."""
        from .. import ops
        from ..utils import SEEDS
        if key_padding_mask is not None or static_k is not None or static_v is not None:
            raise NotImplementedError('key_padding_mask / static_k / static_v are not used on the path')
        L, N, E = query.shape
        assert E == self.embed_dim and key.shape == value.shape and key.shape[1:] == (N, E)
        S = key.shape[0]
        assert L == self.seq_len and S == self.seq_len_src, 'the relative-attention tables are tied to (seq_len_tgt, seq_len_src)'
        qkv_same = query is key and key is value
        if not qkv_same:
            kv_same = key is value or torch.equal(key, value)
            qkv_same = kv_same and query.shape == key.shape and torch.equal(query, key)      # :154-155, host syncs
            if not kv_same:
                raise NotImplementedError('key and value must be the same tensor (self- or encoder-decoder attention)')
        mask = ops.MASK_NONE if attn_mask is None else classify_additive_mask(attn_mask)
        p = self.dropout if self.training else 0.0
        rows = query.transpose(0, 1).reshape(N * L, E)
        memory = None if qkv_same else key.transpose(0, 1).reshape(N * S, E)
        out, probs = self.forward_rows(rows, N, mask, memory=memory, drop_p=p, seed=SEEDS.next() if p > 0 else 0)
        return out.view(N, L, E).transpose(0, 1), (probs if need_weights else None)
