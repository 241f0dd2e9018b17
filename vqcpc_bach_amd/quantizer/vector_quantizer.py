"""ProductVectorQuantizer (reference: VQCPCB/quantizer/vector_quantizer.py:27-159).

Codebooks stay `nn.Parameter`s trained by Adam through the q_latent term (:44-48, :72-83) -- the reference has NO EMA
update.  Nearest-code search, straight-through output and loss are one kernel (vqcpc_vq_fwd) that never materialises
the (rows, K, D) difference tensor; the codebook gradient is a deterministic segment-sum (vqcpc_vq_bwd)."""
import torch
from torch import nn

from .. import ops


class VectorQuantizer(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, inputs, **kwargs):
        raise NotImplementedError


class NoQuantization(VectorQuantizer):
    def __init__(self, codebook_dim):
        super().__init__()
        self.codebook_dim = codebook_dim

    def forward(self, inputs, **kwargs):
        return inputs, None, torch.zeros(inputs.shape[:-1], dtype=inputs.dtype, device=inputs.device)


class ProductVectorQuantizer(VectorQuantizer):
    def __init__(self, codebook_size, codebook_dim, commitment_cost, num_codebooks, use_batch_norm, initialize,
                 squared_l2_norm):
        super().__init__()
        if use_batch_norm:
            raise NotImplementedError('use_batch_norm=True is not used by any encoder config (SURVEY.md section 5): '
                                      'out of scope')
        self.num_codebooks = num_codebooks
        self.codebook_dim = codebook_dim
        self.codebook_size = codebook_size
        self._commitment_cost = commitment_cost
        assert self.codebook_dim % self.num_codebooks == 0
        self.embeddings = nn.ParameterList([
            nn.Parameter(torch.randn(self.codebook_size, self.codebook_dim // num_codebooks) * 4)
            for _ in range(num_codebooks)])
        self.initialize = initialize
        self.squared_l2_norm = squared_l2_norm
        self.use_batch_norm = use_batch_norm
        self.init_broadcast = None       # set by the DP helper: rank 0's data-initialised codebooks go to every rank

    def _initialize(self, flat_input):
        """First-call codebook init from a random permutation of the batch rows (:57-70)."""
        assert flat_input.size(-1) == self.codebook_dim
        assert flat_input.size(0) >= self.codebook_size, \
            'not enough elements in a batch to initialise the clusters. You need to increase the batch dimension.'
        with torch.no_grad():
            for k, embedding in enumerate(self.embeddings):
                # drawn from the global CPU generator like the reference's `torch.randperm(flat_input.size(0))` (:65),
                # so `torch.manual_seed(s)` before the first batch selects the same rows as it does there
                perm = torch.randperm(flat_input.size(0))[:embedding.size(0)].to(flat_input.device)
                dsub = embedding.size(1)
                embedding.copy_(flat_input[perm, k * dsub:(k + 1) * dsub])    # in place: parameters may be flat views
            if self.init_broadcast is not None:
                self.init_broadcast(list(self.embeddings))
        self.initialize = False

    def forward(self, inputs, corrupt_labels=False, init_rows=None, corrupt_rows=None, **kwargs):
        """inputs (..., codebook_dim) -> quantized_sg (..., D), encoding_indices (..., num_codebooks) int64,
        quantization_loss (...).  `init_rows` / `corrupt_rows` (slices over the flattened rows) restrict the data
        initialisation / label corruption to a sub-range when several encoder calls are merged into one."""
        shape = inputs.shape
        flat = inputs.reshape(-1, self.codebook_dim)
        if self.initialize:
            self._initialize(flat.detach()[init_rows] if init_rows is not None else flat.detach())
        codebooks = torch.stack(list(self.embeddings), dim=0)
        given = None
        if self.training and corrupt_labels:                                      # :119-132
            with torch.no_grad():
                idx = ops.vq_assign(flat.detach(), codebooks.detach())     # index-only search; the pass below is a lookup
                rnd = torch.randint_like(idx, low=0, high=self.codebook_size)
                keep = torch.rand(idx.shape, device=idx.device) > 0.05
                if corrupt_rows is not None:
                    only = torch.zeros(idx.shape[0], 1, dtype=torch.bool, device=idx.device)
                    only[corrupt_rows] = True
                    keep = keep | ~only
                given = torch.where(keep, idx, rnd)
        zq, idx, loss = ops.VQFn.apply(flat, codebooks, self._commitment_cost, self.squared_l2_norm, given)
        return zq.view(shape), idx.view(*shape[:-1], self.num_codebooks), loss.view(shape[:-1])
