"""Synthetic stand-in for BachCPCDataloaderGenerator with the same batch-dict contract
(reference: VQCPCB/dataloaders/bach_cpc_dataloader.py:183-259): the music21 corpus is not available, BASELINE.json asks
for "synthetic 4-voice chorale token tensors".

    x_left (B, num_blocks_left * 4, 4)   x_right (B, num_blocks_right * 4, 4)
    negative_samples / negative_samples_back (B, N, num_blocks_right, 4, 4)        last dim = voice, int64 tokens
"""
import torch


class _Dataset:
    def __init__(self, vocab, sequences_size, subdivision):
        self.index2note_dicts = [{i: i for i in range(v)} for v in vocab]
        self.sequences_size = sequences_size
        self.subdivision = subdivision


class SyntheticCPCDataloaderGenerator:
    def __init__(self, num_tokens_per_block=16, num_blocks_left=8, num_blocks_right=8, negative_sampling_method='random',
                 num_negative_samples=15, vocab=(56, 56, 56, 56), seed=1234, device=None, rank=0, **_):
        assert num_tokens_per_block == 16, 'one beat of 4 ticks x 4 voices per block'
        self.num_tokens_per_block = num_tokens_per_block
        self.num_blocks_left, self.num_blocks_right = num_blocks_left, num_blocks_right
        self.negative_sampling_method = negative_sampling_method
        if negative_sampling_method == 'same_sequence':
            num_negative_samples = num_blocks_left + num_blocks_right - 1
        self.num_negative_samples = num_negative_samples
        self.num_channels = len(vocab)
        self.vocab = list(vocab)
        self.seed, self.rank, self.device = seed, rank, device
        sequences_size = (num_blocks_left + num_blocks_right)
        self.dataset_positive = _Dataset(vocab, sequences_size, 4)     # read by getters.get_data_processor
        self.dataset = self.dataset_positive

    def _batch_same_sequence(self, batch_size, gen):
        """negative_sampling_method == 'same_sequence' (bach_cpc_dataloader.py:110-181): the negatives of a window are
        the other blocks of the SAME window; built on the device by vqcpc_same_sequence_negatives (a gather), so only the
        (B, (Kl+Kr)*4, 4) positives cross PCIe.  As in the reference, num_negative_samples is ignored: N = Kl + Kr - 1."""
        from .. import ops
        assert self.device is not None and torch.device(self.device).type == 'cuda', \
            'same_sequence negatives are constructed on the device'
        V = self.vocab[0]
        Kl, Kr = self.num_blocks_left, self.num_blocks_right
        p = torch.randint(0, V, (batch_size, (Kl + Kr) * 4, 4), generator=gen).to(self.device)
        x_left, x_right = p[:, :Kl * 4].contiguous(), p[:, Kl * 4:].contiguous()
        out = {'x_left': x_left, 'x_right': x_right,
               'negative_samples': ops.same_sequence_negatives(x_left, x_right)}
        if Kl == Kr:       # the reference's backward direction needs equal block counts (:132, loop over num_blocks_right)
            out['negative_samples_back'] = ops.same_sequence_negatives(x_right, x_left)
        return out

    def batch(self, batch_size, gen):
        if self.negative_sampling_method == 'same_sequence':
            return self._batch_same_sequence(batch_size, gen)
        assert self.negative_sampling_method == 'random', self.negative_sampling_method
        V, N = self.vocab[0], self.num_negative_samples
        Kl, Kr = self.num_blocks_left, self.num_blocks_right
        out = {
            'x_left': torch.randint(0, V, (batch_size, Kl * 4, 4), generator=gen),
            'x_right': torch.randint(0, V, (batch_size, Kr * 4, 4), generator=gen),
            'negative_samples': torch.randint(0, V, (batch_size, N, Kr, 4, 4), generator=gen),
            'negative_samples_back': torch.randint(0, V, (batch_size, N, Kr, 4, 4), generator=gen),
        }
        if self.device is not None:
            out = {k: v.to(self.device) for k, v in out.items()}
        return out

    def _stream(self, batch_size, salt):
        gen = torch.Generator().manual_seed(self.seed + self.rank + 7919 * salt)
        while True:
            yield self.batch(batch_size, gen)

    def dataloaders(self, batch_size, num_workers=0, **_):
        """(train, val, test) infinite generators; each rank draws its own shard (seed = base + rank)."""
        return self._stream(batch_size, 0), self._stream(batch_size, 1), self._stream(batch_size, 2)
