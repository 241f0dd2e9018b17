"""Learned relative attention bias (reference: VQCPCB/transformer/subsampled_relative_attention.py:8-122).

The reference materialises the bias by two einsums followed by pad / view "skewing" and two triangular masks.  On the
encoder path seq_len_src == seq_len_tgt == L, and the result has the closed form (verified against the reference in
tests/golden/relbias_*.npz):

    bias[h, i, j] = q[h, i] . e1[h, L - 1 - (i - j)]   if j <= i
                  = q[h, i] . e2[h, j - i]             if j >  i

which the fused attention kernel (csrc/relattn.hip) evaluates in registers; this module only owns e1 / e2."""
import torch
from torch import nn


class SubsampledRelativeAttention(nn.Module):
    def __init__(self, head_dim, num_heads, seq_len_src, seq_len_tgt):
        super().__init__()
        assert seq_len_src <= seq_len_tgt and seq_len_tgt % seq_len_src == 0
        if seq_len_src != seq_len_tgt:
            raise NotImplementedError('subsampled (src != tgt) relative attention is decoder-only: out of scope '
                                      '(SURVEY.md section 8(f) N4)')
        self.head_dim, self.num_heads = head_dim, num_heads
        self.seq_len_src, self.seq_len_tgt = seq_len_src, seq_len_tgt
        self.subsampling_ratio = 1
        self.e1 = nn.Parameter(torch.randn(num_heads * seq_len_src, head_dim))
        self.e2 = nn.Parameter(torch.randn(num_heads * seq_len_src, head_dim))

    def forward(self, q):
        raise RuntimeError('the relative bias is fused into vqcpc_relattn_fwd; call MultiheadAttentionCustom instead')
