"""AuxiliaryDecoderRelative (reference: VQCPCB/auxiliary_decoders/auxiliary_decoder_relative.py:7-130): bidirectional
relative-attention transformer that upsamples the quantised codes back to token resolution
(L = 24 -> 96 -> 384 at BASELINE configs[3]).  Like the teacher, the hot path projects only the rows of the masked event."""
import numpy as np
import torch
from torch import nn

from .. import ops
from ..transformer.transformer_custom import TransformerEncoderCustom, TransformerEncoderLayerCustom


class AuxiliaryDecoderRelative(nn.Module):
    def __init__(self, num_tokens_per_channel, codebook_dim, upscale_factors, list_of_num_layers, n_head, d_model,
                 dim_feedforward, num_tokens_bottleneck, dropout):
        super().__init__()
        assert len(list_of_num_layers) == len(upscale_factors)
        self.num_tokens_per_channel = num_tokens_per_channel
        self.num_channels = len(num_tokens_per_channel)
        self.d_model = d_model
        self.codebook_dim = codebook_dim
        self.upscale_factors = upscale_factors
        self.num_tokens_bottleneck = num_tokens_bottleneck
        self.linear = nn.Linear(codebook_dim, d_model)
        self.upscale_embeddings = nn.ParameterList([nn.Parameter(torch.randn(u, d_model)) for u in upscale_factors])
        self.num_tokens_per_transformer_block = [num_tokens_bottleneck * int(np.prod(upscale_factors[:i]))
                                                 for i in range(len(upscale_factors))]
        transformers = []
        for num_layers, num_tokens in zip(list_of_num_layers, self.num_tokens_per_transformer_block):
            # the reference sizes the relative attention with num_events = num_tokens // num_channels (:57-66)
            assert num_tokens % self.num_channels == 0, 'tokens per transformer block must be a multiple of num_channels'
            layer = TransformerEncoderLayerCustom(d_model=d_model, nhead=n_head, attention_bias_type='relative_attention',
                                                  dim_feedforward=dim_feedforward, dropout=dropout,
                                                  num_events=num_tokens // self.num_channels,
                                                  num_channels=self.num_channels)
            transformers.append(TransformerEncoderCustom(encoder_layer=layer, num_layers=num_layers))
        self.transformers = nn.ModuleList(transformers)
        self.pre_softmaxes = nn.ModuleList([nn.Linear(d_model, n) for n in num_tokens_per_channel])

    @staticmethod
    def upscale(input, upscale_factor, upscale_embeddings):
        """Reference signature (time-first input (L, batch, d)) -> (L * factor, batch, d)."""
        L, B, d = input.shape
        assert len(upscale_embeddings) == upscale_factor
        rows = ops.UpscaleFn.apply(input.transpose(0, 1).reshape(B * L, d), upscale_embeddings)
        return rows.view(B, L * upscale_factor, d).transpose(0, 1)

    def forward_hidden(self, input):
        """(batch, num_tokens_bottleneck, codebook_dim) -> hidden rows (batch * num_tokens, d_model)."""
        B, T, _ = input.shape
        assert T == self.num_tokens_bottleneck
        x = ops.linear(input.reshape(B * T, -1), self.linear.weight, self.linear.bias)
        for transformer, emb in zip(self.transformers, self.upscale_embeddings):
            x, _ = transformer.forward_rows(x)
            x = ops.UpscaleFn.apply(x, emb)
        return x

    def project_event(self, hidden, batch_size, event):
        C, d = self.num_channels, self.d_model
        rows = hidden.view(batch_size, -1, C * d)[:, event]
        return [ops.linear(rows[:, c * d:(c + 1) * d], p.weight, p.bias) for c, p in enumerate(self.pre_softmaxes)]

    def forward_events(self, input, event):
        return self.project_event(self.forward_hidden(input), input.shape[0], event)

    def forward(self, input):
        """(batch, num_tokens_bottleneck, codebook_dim) -> list of num_channels logits (batch, num_events, V_c)."""
        B = input.shape[0]
        out = self.forward_hidden(input).view(B, -1, self.num_channels, self.d_model)
        return [ops.linear(out[:, :, c], p.weight, p.bias) for c, p in enumerate(self.pre_softmaxes)]
