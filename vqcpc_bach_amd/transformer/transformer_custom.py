"""The reference's custom transformer (VQCPCB/transformer/transformer_custom.py:17-386).
Encoder path: each layer is ONE fused autograd node (ops.EncoderLayerFn): QKV GEMM -> relative attention -> out-proj
GEMM -> add+LN -> FFN GEMMs (ReLU + dropout in the epilogue) -> add+LN, with a hand-scheduled backward.
Decoder training step (masks, cross-attention): layers are composed from ops.AttnXFn / AddLayerNormFn / FFNFn."""
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

    def forward_rows(self, x, qstride=1, qkv=None, interior=False):
        """x: (blocks * L, d) block-major rows (may be row-strided) -> (y (blocks * L / qstride, d), probs).
        qstride = f > 1 evaluates only the output rows 0, f, 2f, ... (what `output[::f]` would keep).
        qkv: the in_proj output when the caller already has it (first layer): a (rows, 3d) tensor, or a pair
        (block table (vmax * L, 3d), tokens (rows,) int64) that the attention reads through the token indirection."""
        p = self.p if self.training else 0.0
        table, tokens = qkv if isinstance(qkv, tuple) else (qkv, None)
        # interior: the caller hands y to the next layer's forward_rows and to nothing else (the bf16 path may then keep it in bf16 only)
        return ops.EncoderLayerFn.apply(x, self.seq_len, self.nhead, p, SEEDS.next() if p > 0 else 0, qstride, table, tokens,
                                        bool(interior), *self._params())

    def forward_rows_masked(self, x, n, mask):
        """Source encoder of the decoder (decoders/decoder.py:497-503): x (n * L, d) batch-major rows, self-attention
        with an index-rule mask (ops.MASK_*)."""
        p = self.p if self.training else 0.0
        s = [SEEDS.next() if p > 0 else 0 for _ in range(4)]
        a, probs = self.self_attn.forward_rows(x, n, mask, drop_p=p, seed=s[0])
        x = ops.AddLayerNormFn.apply(x, a, self.norm1.weight, self.norm1.bias, p, s[1])
        f = ops.FFNFn.apply(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, p, s[2])
        return ops.AddLayerNormFn.apply(x, f, self.norm2.weight, self.norm2.bias, p, s[3]), probs

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        """API-compatible entry: src (L, N, E) time-first -> (out (L, N, E), {'a_self_encoder': (N, H, L, L)}).
        `src_mask`: None, 'causal' / 'anticausal', or the additive (L, L) matrix the reference passes
        (decoders/decoder.py:292-308) -- recognised as one of the two index rules the attention kernels evaluate."""
        assert src_key_padding_mask is None, 'key padding masks are not used on the path'
        L, N, E = src.shape
        rows = src.transpose(0, 1).reshape(N * L, E)
        if src_mask is None:
            y, probs = self.forward_rows(rows)
        else:
            y, probs = self.forward_rows_masked(rows, N, mask_code(src_mask))
        return y.view(N, L, E).transpose(0, 1), dict(a_self_encoder=probs)


def mask_code(mask):
    """'causal' / 'anticausal' / 'full' / None / ops.MASK_* / an additive (T, S) mask tensor as the reference builds them
    (decoders/decoder.py:294-308: 0 where attention is allowed, -inf elsewhere) -> ops.MASK_*."""
    if isinstance(mask, int):
        return mask
    if torch.is_tensor(mask):
        from .multihead_attention_custom import classify_additive_mask
        return classify_additive_mask(mask)
    table = {None: ops.MASK_NONE, 'full': ops.MASK_NONE, 'causal': ops.MASK_CAUSAL, 'anticausal': ops.MASK_ANTICAUSAL}
    if mask not in table:
        raise NotImplementedError(f'attention mask {mask!r}: pass "causal", "anticausal", "full" or None (additive mask '
                                  'matrices are replaced by index rules inside the attention kernel)')
    return table[mask]


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
            x, probs = layer.forward_rows(x, qstride=out_stride if last else 1, qkv=first_qkv if li == 0 else None,
                                          interior=not last)
            attentions.append(dict(a_self_encoder=probs))
        return x, attentions

    def forward_rows_masked(self, x, n, mask):
        attentions = []
        for layer in self.layers:
            x, probs = layer.forward_rows_masked(x, n, mask)
            attentions.append(dict(a_self_encoder=probs))
        assert self.norm is None
        return x, attentions

    def forward(self, src, mask=None, src_key_padding_mask=None):
        output, attentions = src, []
        for layer in self.layers:
            output, att = layer(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask)
            attentions.append(att)
        if self.norm:
            output = self.norm(output)
        return output, attentions


class TransformerDecoderLayerCustom(nn.Module):
    """transformer_custom.py:294-386: causal (or otherwise masked) relative self-attention, relative cross-attention on
    the memory with seq_len_tgt = r * seq_len_src, FFN; post-LN after each."""

    def __init__(self, d_model, nhead, attention_bias_type_self, attention_bias_type_cross, num_channels_encoder,
                 num_events_encoder, num_channels_decoder, num_events_decoder, dim_feedforward=2048, dropout=0.1,
                 activation='relu'):
        super().__init__()
        if activation != 'relu':
            raise NotImplementedError('the decoder path uses relu')
        self.self_attn = MultiheadAttentionCustom(embed_dim=d_model, num_heads=nhead,
                                                  attention_bias_type=attention_bias_type_self,
                                                  num_channels_k=num_channels_decoder, num_events_k=num_events_decoder,
                                                  num_channels_q=num_channels_decoder, num_events_q=num_events_decoder,
                                                  dropout=dropout)
        self.multihead_attn = MultiheadAttentionCustom(embed_dim=d_model, num_heads=nhead,
                                                       attention_bias_type=attention_bias_type_cross,
                                                       num_channels_k=num_channels_encoder,
                                                       num_events_k=num_events_encoder,
                                                       num_channels_q=num_channels_decoder,
                                                       num_events_q=num_events_decoder, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.p = dropout

    def forward_rows(self, tgt, memory, n, tgt_mask, memory_mask):
        """tgt (n * T, d), memory (n * S, d) batch-major rows -> (tgt, {'a_self_decoder', 'a_cross'})."""
        p = self.p if self.training else 0.0
        s = [SEEDS.next() if p > 0 else 0 for _ in range(6)]
        a, p_self = self.self_attn.forward_rows(tgt, n, tgt_mask, drop_p=p, seed=s[0])
        tgt = ops.AddLayerNormFn.apply(tgt, a, self.norm1.weight, self.norm1.bias, p, s[1])
        a, p_cross = self.multihead_attn.forward_rows(tgt, n, memory_mask, memory=memory, drop_p=p, seed=s[2])
        tgt = ops.AddLayerNormFn.apply(tgt, a, self.norm2.weight, self.norm2.bias, p, s[3])
        f = ops.FFNFn.apply(tgt, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, p, s[4])
        tgt = ops.AddLayerNormFn.apply(tgt, f, self.norm3.weight, self.norm3.bias, p, s[5])
        return tgt, dict(a_self_decoder=p_self, a_cross=p_cross)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        """API-compatible entry, time-first (T, N, E) / (S, N, E); masks as in TransformerEncoderLayerCustom.forward."""
        assert tgt_key_padding_mask is None and memory_key_padding_mask is None
        T, N, E = tgt.shape
        S = memory.shape[0]
        y, att = self.forward_rows(tgt.transpose(0, 1).reshape(N * T, E), memory.transpose(0, 1).reshape(N * S, E), N,
                                   mask_code(tgt_mask), mask_code(memory_mask))
        return y.view(N, T, E).transpose(0, 1), att


class TransformerDecoderCustom(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])   # :183-186
        self.num_layers = num_layers
        self.norm = norm

    def forward_rows(self, tgt, memory, n, tgt_mask, memory_mask):
        attentions = []
        for layer in self.layers:
            tgt, att = layer.forward_rows(tgt, memory, n, tgt_mask, memory_mask)
            attentions.append(att)
        assert self.norm is None
        return tgt, attentions

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None):
        output, attentions = tgt, []
        for layer in self.layers:
            output, att = layer(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask)
            attentions.append(att)
        if self.norm:
            output = self.norm(output)
        return output, attentions


class TransformerCustom(nn.Module):
    """transformer_custom.py:17-118: encoder stack on the source, decoder stack on (target, memory)."""

    def __init__(self, d_model=512, nhead=8, custom_encoder=None, custom_decoder=None):
        super().__init__()
        assert custom_encoder is not None and custom_decoder is not None
        self.encoder, self.decoder = custom_encoder, custom_decoder
        self.d_model, self.nhead = d_model, nhead
        for p in self.parameters():                       # _reset_parameters, :113-118
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward_rows(self, src, tgt, n, src_mask, tgt_mask, memory_mask):
        memory, att_enc = self.encoder.forward_rows_masked(src, n, src_mask)
        out, att_dec = self.decoder.forward_rows(tgt, memory, n, tgt_mask, memory_mask)
        return out, att_dec, att_enc

    def forward(self, src, tgt, src_mask=None, tgt_mask=None, memory_mask=None, src_key_padding_mask=None,
                tgt_key_padding_mask=None, memory_key_padding_mask=None):
        if src.size(1) != tgt.size(1):
            raise RuntimeError('the batch number of src and tgt must be equal')
        if src.size(2) != self.d_model or tgt.size(2) != self.d_model:
            raise RuntimeError('the feature number of src and tgt must be equal to d_model')
        memory, att_enc = self.encoder(src, mask=src_mask)
        out, att_dec = self.decoder(tgt, memory, tgt_mask=tgt_mask, memory_mask=memory_mask)
        return out, att_dec, att_enc
