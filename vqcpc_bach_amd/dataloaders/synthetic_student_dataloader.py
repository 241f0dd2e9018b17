"""Synthetic stand-in for the student configuration's BachDataloaderGenerator (reference: getters.py:37-44,
dataloaders/bach_dataloader.py): batches are {'x': (B, sequences_size * subdivision, 4)} int64 token tensors
(student_encoder_trainer.py:240)."""
import torch

from .synthetic_cpc_dataloader import _Dataset


class SyntheticStudentDataloaderGenerator:
    def __init__(self, sequences_size=24, subdivision=4, vocab=(56, 56, 56, 56), seed=1234, device=None, rank=0, **_):
        self.vocab = list(vocab)
        self.num_channels = len(vocab)
        self.seed, self.rank, self.device = seed, rank, device
        self.dataset = _Dataset(vocab, sequences_size, subdivision)        # read by getters.get_data_processor
        self.num_events = sequences_size * subdivision

    def batch(self, batch_size, gen):
        x = torch.cat([torch.randint(0, v, (batch_size, self.num_events, 1), generator=gen) for v in self.vocab], dim=2)
        return {'x': x.to(self.device) if self.device is not None else x}

    def _stream(self, batch_size, salt):
        gen = torch.Generator().manual_seed(self.seed + self.rank + 7919 * salt)
        while True:
            yield self.batch(batch_size, gen)

    def dataloaders(self, batch_size, num_workers=0, **_):
        return self._stream(batch_size, 0), self._stream(batch_size, 1), self._stream(batch_size, 2)
