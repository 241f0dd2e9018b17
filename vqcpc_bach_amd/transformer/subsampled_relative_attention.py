"""Learned relative attention bias (reference: VQCPCB/transformer/subsampled_relative_attention.py:8-122).

The reference materialises the bias by two einsums followed by pad / view "skewing" and two triangular masks.  With
S = seq_len_src, T = seq_len_tgt = r * S and p = i // r the result has the closed form (verified against the reference
in tests/golden/relbias_*.npz, relbias_cross_*.npz)

    bias[h, i, j] = q[h, i] . e1[h, S - 1 - (p - j)]   if j <= p
                  = q[h, i] . e2[h, j - p]             if j >  p

(r = 1, p = i on the encoder path; r > 1 in the decoder's cross-attention), which the fused attention kernels
(csrc/relattn*.hip) evaluate in registers; this module owns e1 / e2 and offers the stand-alone `forward(q)` of the
reference for API compatibility."""
import torch
from torch import nn


class SubsampledRelativeAttention(nn.Module):
    def __init__(self, head_dim, num_heads, seq_len_src, seq_len_tgt):
        super().__init__()
        assert seq_len_src <= seq_len_tgt and seq_len_tgt % seq_len_src == 0
        self.head_dim, self.num_heads = head_dim, num_heads
        self.seq_len_src, self.seq_len_tgt = seq_len_src, seq_len_tgt
        self.subsampling_ratio = seq_len_tgt // seq_len_src
        self.e1 = nn.Parameter(torch.randn(num_heads * seq_len_src, head_dim))
        self.e2 = nn.Parameter(torch.randn(num_heads * seq_len_src, head_dim))

    def forward(self, q):
        """API-compatible stand-alone bias (the training path never calls it: the bias is fused into the attention
        kernels).  q (batch * num_heads, T, head_dim), already scaled -> rel_attn (batch * num_heads, T, S).
        Two GEMMs against e1 / e2 and the closed-form index selection that replaces the reference's pad / view skewing."""
        from .. import ops
        bh, T, hd = q.shape
        S, r = self.seq_len_src, self.subsampling_ratio
        assert bh % self.num_heads == 0 and T == self.seq_len_tgt and hd == self.head_dim
        H = self.num_heads
        qh = q.reshape(bh // H, H, T, hd)
        a1 = torch.stack([ops.linear(qh[:, h], self.e1[h * S:(h + 1) * S]) for h in range(H)], dim=1)     # q . e1[h, m]
        a2 = torch.stack([ops.linear(qh[:, h], self.e2[h * S:(h + 1) * S]) for h in range(H)], dim=1)
        p = (torch.arange(T, device=q.device) // r).view(T, 1)
        j = torch.arange(S, device=q.device).view(1, S)
        m1 = (S - 1 - p + j).clamp(0, S - 1).expand(bh // H, H, T, S)
        m2 = (j - p).clamp(0, S - 1).expand(bh // H, H, T, S)
        return torch.where(j <= p, a1.gather(-1, m1), a2.gather(-1, m2)).reshape(bh, T, S)
