"""CPC heads (reference: VQCPCB/vqcpc_helper.py:5-98)."""
import torch
import torch.nn as nn

from . import ops
from .utils import SEEDS


def nce_loss(positive, negatives):
    """-mean_b sum_k (pos - logsumexp([negatives, pos]))  (:5-29).  API-compatible stand-alone form on precomputed
    scores; the training step uses the fused kernel through `cpc_scores_and_loss`."""
    allf = torch.cat([negatives, positive.unsqueeze(2)], dim=2)
    return -(positive - torch.logsumexp(allf, dim=2)).sum(1).mean(0)


def quantization_loss(loss_quantization_left, loss_quantization_negative, loss_quantization_right,
                      loss_quantization_negative_back=None):
    """mean over the 3B (4B) per-window sums (:32-51)."""
    parts = [loss_quantization_left.sum(1), loss_quantization_right.sum(1), loss_quantization_negative.flatten(1).sum(1)]
    if loss_quantization_negative_back is not None:
        parts.append(loss_quantization_negative_back.flatten(1).sum(1))
    return torch.cat(parts, dim=0).mean()


def cpc_scores_and_loss(fks_module, c, z_pos, z_neg):
    """Fused FksModule(positives) + FksModule(negatives) + nce_loss + score matrix (vqcpc_nce_fwd / _bwd).
    c (B, c_dim), z_pos (B, K, z), z_neg (B, N, K, z) -> (loss scalar, hits (B, K) 0/1)."""
    loss_b, hits, _, _ = ops.NCEFn.apply(c, fks_module.W, z_pos, z_neg)
    return loss_b.mean(), hits


class CModule(nn.Module):
    """Autoregressive context network: multi-layer GRU, last step, Linear (:54-76).  `g_ar_fwd` is kept as the
    parameter container (state_dict keys `g_ar_fwd.weight_ih_l0`, ... of the reference's checkpoints); the recurrence
    itself runs on this library: one GEMM for the input projections of all steps, one GEMM + one gate kernel per step
    (ops.GRULayerFn), dropout between layers from the shared counter RNG (reproducible, unlike MIOpen's stateful one)."""

    def __init__(self, input_dim, hidden_size, output_dim, num_layers, dropout):
        super().__init__()
        self.g_ar_fwd = torch.nn.GRU(input_size=input_dim, hidden_size=hidden_size, num_layers=num_layers, bias=True,
                                     batch_first=True, dropout=dropout, bidirectional=False)
        self.output_linear = nn.Linear(hidden_size, output_dim)
        self.num_layers = num_layers
        self.p = dropout

    def forward(self, zs, h):
        """zs (B, T, input_dim), h must be None (the path always starts from h0 = 0, vqcpc_encoder_trainer.py:251)."""
        assert h is None, 'the CPC step starts the context network from h0 = 0'
        B, T, _ = zs.shape
        x = zs.transpose(0, 1).reshape(T * B, -1)                            # time-major rows
        g = self.g_ar_fwd
        p = self.p if self.training else 0.0
        for l in range(self.num_layers):
            last = l == self.num_layers - 1
            x = ops.GRULayerFn.apply(x, getattr(g, f'weight_ih_l{l}'), getattr(g, f'weight_hh_l{l}'),
                                     getattr(g, f'bias_ih_l{l}'), getattr(g, f'bias_hh_l{l}'), T, p,
                                     SEEDS.next() if (p > 0 and not last) else 0, last)
        return ops.linear(x, self.output_linear.weight, self.output_linear.bias)


class FksModule(nn.Module):
    def __init__(self, z_dim, c_dim, k_max):
        super().__init__()
        self.k_max = k_max
        self.W = nn.Parameter(torch.randn(z_dim, c_dim, k_max))

    def forward(self, c_t, zs):
        """log f_k(c_t, z_{t+k}) = z^T W_k c  for every k  (:86-98): c_t (B, c), zs (B, K, z) -> (B, K)."""
        z_dim, c_dim, K = self.W.shape
        w = self.W.permute(2, 0, 1).reshape(K * z_dim, c_dim)
        wc = ops.linear(c_t, w).view(c_t.shape[0], K, z_dim)
        return (wc * zs).sum(-1)
