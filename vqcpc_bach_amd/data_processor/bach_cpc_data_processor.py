"""CPC block pre-processing (reference: VQCPCB/data_processor/bach_cpc_data_processor.py:8-68)."""
import torch

from ..utils import cuda_variable
from .data_processor import DataProcessor


class BachCPCDataProcessor(DataProcessor):
    def __init__(self, embedding_size, num_events, num_channels, num_tokens_per_channel, num_tokens_per_block):
        super().__init__(embedding_size=embedding_size, num_events=num_events,
                         num_tokens_per_channel=num_tokens_per_channel)
        assert num_channels == self.num_channels
        self.num_tokens_per_block = num_tokens_per_block

    def preprocess(self, x):
        """(..., num_ticks, num_voices) -> (..., num_blocks, num_tokens_per_block) int64 on the device;
        token p of a block is (tick p // num_voices, voice p % num_voices)."""
        lead = tuple(x.shape[:-2])
        flat = x.reshape(*lead, x.shape[-2] * x.shape[-1])
        assert flat.shape[-1] % self.num_tokens_per_block == 0
        blocks = flat.reshape(*lead, flat.shape[-1] // self.num_tokens_per_block, self.num_tokens_per_block)
        return cuda_variable(blocks.long())

    def embed(self, block):
        """(..., num_tokens_per_block) -> (..., num_tokens_per_block, embedding_size); voice = position % num_voices."""
        nv = self.num_channels
        cols = [self.embeddings[p % nv].weight[block[..., p]] for p in range(block.shape[-1])]
        return torch.stack(cols, dim=-2)
