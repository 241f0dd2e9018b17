"""RelativeTransformerDownscaler (reference: VQCPCB/downscalers/relative_transformer_downscaler.py:9-133).

Hot path = forward_tokens(): int64 token blocks -> fused [embedding . input_linear | channel | event] rows
(vqcpc_embed_pos_fwd) -> stacks of fused encoder layers; the `[::4]` subsample between stacks (:125) is a row stride
(lda = 4 d) into the next GEMM / LayerNorm, never a copy -> output_linear."""
import numpy as np
import torch
from torch import nn

from .. import ops
from ..transformer.transformer_custom import TransformerEncoderCustom, TransformerEncoderLayerCustom


def _sub_supported(L, factor, hd):
    return factor == 4 and L in (16, 4) and hd in (16, 32, 64)


class Downscaler(nn.Module):
    def __init__(self, downscale_factors):
        super().__init__()
        self.downscale_factors = downscale_factors


class RelativeTransformerDownscaler(Downscaler):
    def __init__(self, input_dim, output_dim, num_channels, downscale_factors, d_model, n_head, list_of_num_layers,
                 dim_feedforward, dropout):
        super().__init__(downscale_factors)
        assert len(downscale_factors) == len(list_of_num_layers), \
            'number of transfo must match number of downscaling factors'
        self.sequence_length = int(np.prod(downscale_factors))
        positional_embedding_size = 8
        self.num_channels = num_channels
        self.num_events = self.sequence_length // self.num_channels
        self.d_model = d_model
        self.input_linear = nn.Linear(input_dim, d_model - 2 * positional_embedding_size)
        self.target_channel_embeddings = nn.Parameter(torch.randn(1, 1, self.num_channels, positional_embedding_size))
        self.events_positioning_embeddings = nn.Parameter(torch.randn(1, 1, self.num_events, positional_embedding_size))
        self.output_dim = output_dim
        self.output_linear = nn.Linear(d_model, output_dim)
        transformers = []
        num_events, nch = self.num_events, self.num_channels
        for factor, num_layers in zip(downscale_factors, list_of_num_layers):
            layer = TransformerEncoderLayerCustom(d_model=d_model, nhead=n_head, attention_bias_type='relative_attention',
                                                  num_channels=nch, num_events=num_events,
                                                  dim_feedforward=dim_feedforward, dropout=dropout)
            transformers.append(TransformerEncoderCustom(encoder_layer=layer, num_layers=num_layers))
            num_events = (num_events * nch) // factor
            if nch > 1:
                assert nch <= factor, f'First stack of downscaler transfo has to be larger than input num channels = {nch}'
                nch = 1
        self.transformers = nn.ModuleList(transformers)

    # ---- hot path ---------------------------------------------------------------------------------------------
    table_lookup_min_ratio = 4       # use the first-layer QKV table when tokens >= ratio * (vmax * L) table rows

    def _stacks(self, x, first_qkv=None):
        """x (blocks * L0, d) -> (blocks, d): run the stacks, subsampling by stride between them."""
        L = self.sequence_length
        d = x.shape[1]
        for si, (transfo, factor) in enumerate(zip(self.transformers, self.downscale_factors)):
            fq = first_qkv if si == 0 else None
            if _sub_supported(L, factor, self.d_model // transfo.layers[0].nhead):
                x, _ = transfo.forward_rows(x, out_stride=factor, first_qkv=fq)   # last layer only evaluates the kept rows
            else:
                x, _ = transfo.forward_rows(x, first_qkv=fq)
                x = x.view(-1, d)[::factor]      # keep positions 0, f, 2f, ... of every block: a row stride
            L //= factor
        assert L == 1
        return x

    def forward_tokens(self, tokens, data_processor):
        """tokens (..., num_blocks, sequence_length) int64 on the device -> (..., num_blocks, output_dim)."""
        lead = tokens.shape[:-1]
        assert tokens.shape[-1] == self.sequence_length
        tables = data_processor.stacked_tables()                                    # (nv, vmax, emb)
        # lookup(E_v)[tok] @ W_in^T + b  ==  lookup(E_v @ W_in^T + b)[tok]
        table = ops.linear(tables, self.input_linear.weight, self.input_linear.bias)      # (nv, vmax, dlin), own GEMM
        flat_tokens = tokens.reshape(-1).contiguous()
        chan = self.target_channel_embeddings.view(self.num_channels, -1)
        event = self.events_positioning_embeddings.view(self.num_events, -1)
        x = ops.EmbedPosFn.apply(flat_tokens, table, chan, event, self.sequence_length)
        x = self._stacks(x, first_qkv=self._first_layer_qkv(flat_tokens, table, chan, event))
        return ops.linear(x, self.output_linear.weight, self.output_linear.bias).view(*lead, self.output_dim)

    def _first_layer_qkv(self, flat_tokens, table, chan, event):
        """in_proj of the FIRST layer as a block-table lookup: its input row depends only on (token id, position in the
        block), i.e. vmax * L distinct rows, so the projection runs on those rows (a 912 x 768 x 256 GEMM at C1 instead of
        557 056 x 768 x 256) and every token looks its row up (inside the attention kernels: the 2.8 MB table stays in L2);
        the per-table-row sums of d qkv go back through the same small GEMM.  Not used when the first layer is also the query-subsampled last layer of its stack."""
        stack = self.transformers[0]
        if len(stack.layers) < 2 and _sub_supported(self.sequence_length, self.downscale_factors[0],
                                                    self.d_model // stack.layers[0].nhead):
            return None
        L, vmax = self.sequence_length, table.shape[1]
        if flat_tokens.numel() < self.table_lookup_min_ratio * vmax * L:    # tiny inputs: the plain projection is cheaper
            return None
        syn = torch.arange(vmax, device=flat_tokens.device).repeat_interleave(L)       # row t * L + p holds token t
        x_table = ops.EmbedPosFn.apply(syn, table, chan, event, L)                     # (vmax * L, d)
        attn = stack.layers[0].self_attn
        qkv_table = ops.linear(x_table, attn.in_proj_weight, attn.in_proj_bias)        # (vmax * L, 3d)
        if L in (16, 4):
            return qkv_table, flat_tokens                 # the L = 16 / 4 attention kernels read the table directly
        return ops.BlockTableGatherFn.apply(qkv_table, flat_tokens, L)

    # ---- API-compatible path ----------------------------------------------------------------------------------
    def forward(self, embedded_seq):
        """(batch, seq_len, input_dim) embeddings -> (batch, seq_len // prod(downscale_factors), output_dim)."""
        batch_size, seq_len, dim = embedded_seq.shape
        assert seq_len % self.sequence_length == 0
        nb = seq_len // self.sequence_length
        x = ops.linear(embedded_seq.reshape(batch_size * seq_len, dim), self.input_linear.weight, self.input_linear.bias)
        tok = torch.arange(self.sequence_length, device=x.device)
        chan = self.target_channel_embeddings.view(self.num_channels, -1)[tok % self.num_channels]
        ev = self.events_positioning_embeddings.view(self.num_events, -1)[tok // self.num_channels]
        posi = torch.cat([chan, ev], dim=1).repeat(batch_size * nb, 1)
        x = self._stacks(torch.cat([x, posi], dim=1))
        return ops.linear(x, self.output_linear.weight, self.output_linear.bias).view(batch_size, nb, self.output_dim)
