"""RelativeTransformerDownscalerLinear (reference: VQCPCB/downscalers/relative_transformer_downscaler_linear.py:9-139),
the downscaler of the student configuration: same input embedding and relative-attention stacks as
RelativeTransformerDownscaler, but every stack ends in Linear(f * d -> d) over f consecutive tokens (:129-133) instead of
keeping one token in f.  In block-major rows the reshape/permute of :129-131 is a free view (rows, d) -> (rows / f, f d)."""
import numpy as np
import torch
from torch import nn

from .. import ops
from ..transformer.transformer_custom import TransformerEncoderCustom, TransformerEncoderLayerCustom
from .relative_transformer_downscaler import Downscaler


class RelativeTransformerDownscalerLinear(Downscaler):
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
        transformers, linear_aggs = [], []
        num_events, nch = self.num_events, self.num_channels
        for factor, num_layers in zip(downscale_factors, list_of_num_layers):
            layer = TransformerEncoderLayerCustom(d_model=d_model, nhead=n_head, attention_bias_type='relative_attention',
                                                  num_channels=nch, num_events=num_events,
                                                  dim_feedforward=dim_feedforward, dropout=dropout)
            transformers.append(TransformerEncoderCustom(encoder_layer=layer, num_layers=num_layers))
            linear_aggs.append(nn.Linear(d_model * factor, d_model))
            num_events = (num_events * nch) // factor
            if nch > 1:
                assert nch <= factor, f'First stack of downscaler transfo has to be larger than input num channels = {nch}'
                nch = 1
        self.transformers = nn.ModuleList(transformers)
        self.linear_aggs = nn.ModuleList(linear_aggs)

    def _stacks(self, x):
        """x (blocks * L0, d) -> (blocks, d)."""
        d = self.d_model
        for transfo, factor, agg in zip(self.transformers, self.downscale_factors, self.linear_aggs):
            x, _ = transfo.forward_rows(x)
            x = ops.linear(x.view(-1, factor * d), agg.weight, agg.bias)
        return x

    def forward_tokens(self, tokens, data_processor):
        """tokens (..., num_blocks, sequence_length) int64 on the device -> (..., num_blocks, output_dim)."""
        lead = tokens.shape[:-1]
        assert tokens.shape[-1] == self.sequence_length
        tables = data_processor.stacked_tables()
        table = ops.linear(tables, self.input_linear.weight, self.input_linear.bias)      # (nv, vmax, dlin), own GEMM
        x = ops.EmbedPosFn.apply(tokens.reshape(-1).contiguous(), table,
                                 self.target_channel_embeddings.view(self.num_channels, -1),
                                 self.events_positioning_embeddings.view(self.num_events, -1), self.sequence_length)
        x = self._stacks(x)
        return ops.linear(x, self.output_linear.weight, self.output_linear.bias).view(*lead, self.output_dim)

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
