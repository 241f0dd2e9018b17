"""Encoder half of the reference's custom transformer (VQCPCB/transformer/transformer_custom.py:121-163,220-291).
Each layer is ONE fused autograd node (ops.EncoderLayerFn): QKV GEMM -> relative attention -> out-proj GEMM ->
add+LN -> FFN GEMMs (ReLU + dropout in the epilogue) -> add+LN, with a hand-scheduled backward."""
import copy

import torch
from torch import nn

from .. import ops
from ..utils import SEEDS
from .multihead_attention_custom import MultiheadAttentionCustom


class TransformerEncoderLayerCustom(nn.Module):
    def __init__(self, d_model, nhead, attention_bias_type, num_channels, num_events, dim_feedforward=2048, dropout=0.1,
                 activation='relu'):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError('the encoder path uses relu')
        self.self_attn = MultiheadAttentionCustom(embed_dim=d_model, num_heads=nhead,
                                                  attention_bias_type=attention_bias_type, num_channels_k=num_channels,
                                                  num_events_k=num_events, num_channels_q=num_channels,
                                                  num_events_q=num_events, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.p = dropout
        self.nhead = nhead
        self.seq_len = num_channels * num_events

    def _params(self):
        a = self.self_attn
        return (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, a.attn_bias.e1, a.attn_bias.e2,
                self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm1.weight,
                self.norm1.bias, self.norm2.weight, self.norm2.bias)

    def forward_rows(self, x, qstride=1, qkv=None):
        """x: (blocks * L, d) block-major rows (may be row-strided) -> (y (blocks * L / qstride, d), probs).
        qstride = f > 1 evaluates only the output rows 0, f, 2f, ... (what `output[::f]` would keep).
        qkv: the in_proj output when the caller already has it (first layer): a (rows, 3d) tensor, or a pair
        (block table (vmax * L, 3d), tokens (rows,) int64) that the attention reads through the token indirection."""
        p = self.p if self.training else 0.0
        table, tokens = qkv if isinstance(qkv, tuple) else (qkv, None)
        return ops.EncoderLayerFn.apply(x, self.seq_len, self.nhead, p, SEEDS.next() if p > 0 else 0, qstride, table, tokens,
                                        *self._params())

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        """API-compatible entry: src (L, N, E) time-first -> (out (L, N, E), {'a_self_encoder': (N, H, L, L)})."""
        assert src_mask is None and src_key_padding_mask is None, 'masks are not used on the encoder path'
        L, N, E = src.shape
        y, probs = self.forward_rows(src.transpose(0, 1).reshape(N * L, E))
        return y.view(N, L, E).transpose(0, 1), dict(a_self_encoder=probs)


class TransformerEncoderCustom(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        # the reference deep-copies one layer (:138), so all layers of a stack start from identical weights
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward_rows(self, x, out_stride=1, first_qkv=None):
        """Runs the stack on block-major rows; `out_stride` = f returns only rows 0, f, 2f, ... of the stack output
        (the downscaler's `output[::f]`), which lets the last layer skip the dropped rows.  `first_qkv`: in_proj output
        of the first layer when the caller computed it by table lookup."""
        attentions = []
        for li, layer in enumerate(self.layers):
            last = li == len(self.layers) - 1
            x, probs = layer.forward_rows(x, qstride=out_stride if last else 1, qkv=first_qkv if li == 0 else None)
            attentions.append(dict(a_self_encoder=probs))
        return x, attentions

    def forward(self, src, mask=None, src_key_padding_mask=None):
        output, attentions = src, []
        for layer in self.layers:
            output, att = layer(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
            attentions.append(att)
        if self.norm:
            output = self.norm(output)
        return output, attentions
