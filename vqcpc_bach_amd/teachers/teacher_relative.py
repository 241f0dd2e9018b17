"""TeacherRelative (reference: VQCPCB/teachers/teacher_relative.py:8-87): the masked-LM teacher of the student step, a
relative-attention transformer over the whole sequence (L = num_events * num_channels = 384 at BASELINE configs[3]).

Hot path = forward_events(): tokens -> fused [embedding . linear_to_input_transformer | channel embedding] rows
(vqcpc_embed_pos_fwd without an event part) -> fused encoder layers over block-major rows (one block = one sequence,
general-L attention kernels) -> the per-voice output projections on the rows of ONE event only: both losses of the step
read the logits of the masked event alone (utils.py:41-46, :152-158), so the other rows are never projected."""
import torch
from torch import nn

from .. import ops
from ..transformer.transformer_custom import TransformerEncoderCustom, TransformerEncoderLayerCustom
from ..utils import flatten


class TeacherRelative(nn.Module):
    def __init__(self, data_processor, num_layers, num_tokens_per_channel, positional_embedding_size, d_model,
                 dim_feedforward, n_head, num_tokens, dropout):
        super().__init__()
        self.num_channels = len(num_tokens_per_channel)
        self.data_processor = data_processor
        input_dim = data_processor.embedding_size
        assert num_tokens % self.num_channels == 0
        self.channel_embeddings = nn.Parameter(torch.randn(1, self.num_channels, positional_embedding_size))
        self.num_layers = num_layers
        self.linear_to_input_transformer = nn.Linear(input_dim, d_model - positional_embedding_size)
        layer = TransformerEncoderLayerCustom(d_model=d_model, nhead=n_head, attention_bias_type='relative_attention',
                                              dim_feedforward=dim_feedforward, dropout=dropout,
                                              num_events=num_tokens // self.num_channels, num_channels=self.num_channels)
        self.transformer = TransformerEncoderCustom(encoder_layer=layer, num_layers=num_layers)
        self.num_tokens_per_channel = num_tokens_per_channel
        self.num_tokens = num_tokens
        self.d_model = d_model
        self.pre_softmaxes = nn.ModuleList([nn.Linear(d_model, n) for n in num_tokens_per_channel])

    # ---- hot path ---------------------------------------------------------------------------------------------
    def forward_hidden(self, tokens):
        """tokens (batch, num_events, num_channels) int64 on the device -> hidden rows (batch * num_tokens, d_model)."""
        B, E, C = tokens.shape
        assert C == self.num_channels and E * C == self.num_tokens, 'the relative attention is built for num_tokens'
        tables = self.data_processor.stacked_tables()
        w = self.linear_to_input_transformer
        table = ops.linear(tables, w.weight, w.bias)                        # lookup(E_c) W^T + b == lookup(E_c W^T + b)
        x = ops.EmbedPosFn.apply(tokens.reshape(-1).contiguous(), table, self.channel_embeddings.view(C, -1), None, C)
        x, _ = self.transformer.forward_rows(x)
        return x

    def project_event(self, hidden, batch_size, event):
        """hidden (batch * num_tokens, d) -> list of num_channels logits (batch, V_c) of one event."""
        C = self.num_channels
        rows = hidden.view(batch_size, self.num_tokens // C, C * self.d_model)[:, event]        # (batch, C d): a stride
        return [ops.linear(rows[:, c * self.d_model:(c + 1) * self.d_model], p.weight, p.bias)
                for c, p in enumerate(self.pre_softmaxes)]

    def forward_events(self, tokens, event):
        return self.project_event(self.forward_hidden(tokens), tokens.shape[0], event)

    # ---- API-compatible path ----------------------------------------------------------------------------------
    def forward(self, x):
        """x (batch, num_events, num_channels, input_dim) embeddings -> list of num_channels logits
        (batch, num_events, V_c)."""
        w = self.linear_to_input_transformer
        seq = flatten(ops.linear(x, w.weight, w.bias))
        B, T, _ = seq.shape
        E = T // self.num_channels
        seq = torch.cat([seq, self.channel_embeddings.repeat(B, E, 1)], dim=2)
        out, _ = self.transformer.forward_rows(seq.reshape(B * T, -1))
        out = out.view(B, E, self.num_channels, -1)
        return [ops.linear(out[:, :, c], p.weight, p.bias) for c, p in enumerate(self.pre_softmaxes)]
