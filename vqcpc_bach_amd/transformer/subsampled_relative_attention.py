"""Learned relative attention bias (reference: VQCPCB/transformer/subsampled_relative_attention.py:8-122).

The reference materialises the bias by two einsums followed by pad / view "skewing" and two triangular masks.  On the
encoder path seq_len_src == seq_len_tgt == L, and the result has the closed form (verified against the reference in
tests/golden/relbias_*.npz):

    bias[h, i, j] = q[h, i] . e1[h, L - 1 - (i - j)]   if j <= i
                  = q[h, i] . e2[h, j - i]             if j >  i

which the fused attention kernels (csrc/relattn*.hip) evaluate in registers; this module owns e1 / e2 and offers the
stand-alone `forward(q)` of the reference for API compatibility."""
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
        """API-compatible stand-alone bias (the training path never calls it: the bias is fused into the attention
        kernels).  q (batch * num_heads, L, head_dim), already scaled -> rel_attn (batch * num_heads, L, L).
        Two GEMMs against e1 / e2 and the closed-form index selection that replaces the reference's pad / view skewing."""
        from .. import ops
        bh, L, hd = q.shape
        assert bh % self.num_heads == 0 and L == self.seq_len_tgt and hd == self.head_dim
        H = self.num_heads
        qh = q.reshape(bh // H, H, L, hd)
        a1 = torch.stack([ops.linear(qh[:, h], self.e1[h * L:(h + 1) * L]) for h in range(H)], dim=1)     # q . e1[h, m]
        a2 = torch.stack([ops.linear(qh[:, h], self.e2[h * L:(h + 1) * L]) for h in range(H)], dim=1)
        i = torch.arange(L, device=q.device).view(L, 1)
        j = torch.arange(L, device=q.device).view(1, L)
        m1 = (L - 1 - i + j).clamp(0, L - 1).expand(bh // H, H, L, L)
        m2 = (j - i).clamp(0, L - 1).expand(bh // H, H, L, L)
        return torch.where(j <= i, a1.gather(-1, m1), a2.gather(-1, m2)).reshape(bh, L, L)
