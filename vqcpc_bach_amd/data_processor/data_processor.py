"""Token pre-processing and embedding tables (reference: VQCPCB/data_processor/data_processor.py:7-104)."""
import torch
from torch import nn

from ..utils import cuda_variable


class DataProcessor(nn.Module):
    """Holds one `nn.Embedding(V_c + 1, embedding_size)` per voice (the +1 is the mask token, :26-32).
    On the training hot path the tables are consumed by the fused embedding kernel
    (RelativeTransformerDownscaler.forward_tokens); `embed` is the API-compatible stand-alone lookup."""

    def __init__(self, embedding_size, num_events, num_tokens_per_channel, add_mask_token=True):
        super().__init__()
        self.embedding_size = embedding_size
        self.num_events = num_events
        self.num_tokens_per_channel = num_tokens_per_channel
        self.num_tokens = self.num_events * len(self.num_tokens_per_channel)
        self.num_channels = len(self.num_tokens_per_channel)
        extra = 1 if add_mask_token else 0
        self.embeddings = nn.ModuleList([nn.Embedding(n + extra, embedding_size) for n in num_tokens_per_channel])

    def preprocess(self, x):
        return cuda_variable(x.long())

    def embed(self, x):
        """(..., num_channels) -> (..., num_channels, embedding_size)"""
        return torch.stack([emb.weight[x[..., c]] for c, emb in enumerate(self.embeddings)], dim=-2)

    def stacked_tables(self):
        """(num_channels, vmax, embedding_size) zero-padded stack of the per-voice tables (autograd-visible)."""
        vmax = max(e.weight.shape[0] for e in self.embeddings)
        return torch.stack([torch.nn.functional.pad(e.weight, (0, 0, 0, vmax - e.weight.shape[0]))
                            for e in self.embeddings], dim=0)
