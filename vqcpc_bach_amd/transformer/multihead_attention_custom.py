"""Parameter container of the reference's MultiheadAttentionCustom
(VQCPCB/transformer/multihead_attention_custom.py:8-120): in_proj_weight/bias, out_proj, attn_bias.{e1,e2}, same
names, shapes and initialisation.  The arithmetic of its forward (:122-353) lives in ops.EncoderLayerFn (encoder
path) and, for the decoder's masked / cross attentions, in `forward_rows` below (ops.AttnXFn)."""
import torch
from torch import nn

from .subsampled_relative_attention import SubsampledRelativeAttention


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
