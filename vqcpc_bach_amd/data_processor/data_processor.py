"""Token pre-processing and embedding tables (reference: VQCPCB/data_processor/data_processor.py:7-104)."""
import ctypes

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

    # ---- token range check (nn.Embedding's IndexError, made asynchronous) ---------------------------------------
    def checked(self, tokens):
        """int64 device tokens in (..., voice-fastest) order -> a copy clamped into every voice's table range; an id
        outside its table raises at the next `raise_if_bad_tokens()` (end of epoch()) instead of being used as an
        address by the kernels (vqcpc_check_tokens)."""
        from .. import hip
        tokens = tokens.contiguous()
        if getattr(self, '_token_flag', None) is None or self._token_flag.device != tokens.device:
            self._token_flag = torch.zeros(1, dtype=torch.int32, device=tokens.device)
            self._token_limits = (ctypes.c_int32 * self.num_channels)(*[e.weight.shape[0] for e in self.embeddings])
        out = torch.empty_like(tokens)
        hip.call('vqcpc_check_tokens', tokens, tokens.numel(), self.num_channels, self._token_limits, out,
                 self._token_flag)
        return out

    def bad_token_flag(self):
        """Device int32[1] (or None if nothing was checked yet): non-zero once any token id was out of range."""
        return getattr(self, '_token_flag', None)

    def raise_if_bad_tokens(self, flag_value=None, dp=None):
        """dp: the trainer's DataParallelContext -- in a multi-rank run the flag is summed over the ranks first, so that
        every rank raises together (one rank raising alone leaves the others blocked in their next collective)."""
        flag = self.bad_token_flag()
        if flag is None:
            return
        if flag_value is None:
            if dp is not None and dp.distributed:
                flag_value = int(dp.all_reduce_sum_(flag.clone()).item())
            else:
                flag_value = int(flag.item())
        if flag_value:
            flag.zero_()
            raise IndexError(f'token id out of range: voice c takes ids in [0, {[e.weight.shape[0] for e in self.embeddings]}[c]) '
                             '(nn.Embedding would have raised "index out of range in self")')

    def embed(self, x):
        """(..., num_channels) -> (..., num_channels, embedding_size)"""
        return torch.stack([emb.weight[x[..., c]] for c, emb in enumerate(self.embeddings)], dim=-2)

    def stacked_tables(self):
        """(num_channels, vmax, embedding_size) zero-padded stack of the per-voice tables (autograd-visible)."""
        from .. import ops
        return ops.StackTablesFn.apply(*[e.weight for e in self.embeddings])
