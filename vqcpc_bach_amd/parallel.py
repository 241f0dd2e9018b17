"""One process per GPU over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm), plus the flat fp32
parameter / gradient buffers that double as the all-reduce bucket.

The reference has no distributed code at all.  Design for 8 x MI355X, fully connected xGMI:
  * pure data parallelism over windows: every 16-token block is independent, the model (5.7 M parameters at C1) is
    replicated; no TP / PP / SP is useful at L <= 16, d <= 512;
  * ONE all-reduce per step on the flat gradient buffer (22.8 MB fp32 at C1: ~0.1-0.5 ms over xGMI against a step of
    tens of ms, so bucketing/overlap buys < 1 % and is not used), issued BEFORE the global-norm clip so that the
    clipped norm is that of the global batch (vqcpc_encoder_trainer.py:313 semantics);
  * the sum is turned into a mean inside the optimiser kernel (grad_scale = 1 / world_size): no extra pass;
  * rank 0 broadcasts the initial weights and the data-initialised codebooks (the reference's _initialize uses a
    local randperm, vector_quantizer.py:57-70, which would make ranks diverge).
"""
import os

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this platform (RCCL needs it)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class DataParallelContext:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get('RANK', 0))
        self.world_size = int(os.environ.get('WORLD_SIZE', 1))
        self.local_rank = int(os.environ.get('LOCAL_RANK', 0))
        # tests on a 1-GPU box: VQCPC_DP_SHARE_GPU=1 puts every rank on device 0 (with VQCPC_DP_BACKEND=gloo, which moves
        # device tensors through the host): the whole multi-rank trainer path runs on the GPU kernels, only the transport
        # differs from RCCL
        if os.environ.get('VQCPC_DP_SHARE_GPU', '0') == '1':
            self.local_rank = 0
        if device is None:
            if torch.cuda.is_available():
                # a launcher that masks device visibility per rank (one visible GPU each) numbers it 0 whatever LOCAL_RANK says
                if torch.cuda.device_count() == 1 and self.local_rank > 0:
                    self.local_rank = 0
                torch.cuda.set_device(self.local_rank)     # mandatory: the reference moves to the DEFAULT device
                device = torch.device('cuda', self.local_rank)
            else:
                device = torch.device('cpu')
        self.device = torch.device(device)
        self.owns_group = False
        self.backend = None
        force = os.environ.get('VQCPC_FORCE_DIST', '0') == '1'      # exercise the RCCL path with a single rank (tests)
        self.force = force
        if (self.world_size > 1 or force) and not dist.is_initialized():
            backend = backend or os.environ.get('VQCPC_DP_BACKEND') or ('nccl' if self.device.type == 'cuda' else 'gloo')
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            kw = dict(device_id=self.device) if backend == 'nccl' else {}
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size, **kw)
            self.owns_group = True
        if dist.is_available() and dist.is_initialized():
            self.backend = dist.get_backend()

    @property
    def distributed(self):
        return self.world_size > 1 or getattr(self, 'force', False)

    def all_reduce_sum_(self, tensor):
        if self.distributed:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        return tensor

    def broadcast_(self, tensor, src=0):
        if self.distributed:
            dist.broadcast(tensor, src=src)
        return tensor

    def barrier(self):
        if self.distributed:
            dist.barrier()

    def max_over_ranks(self, value):
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        if self.distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, value):
        """One float per rank, in rank order, on every rank."""
        if not self.distributed:
            return [float(value)]
        t = torch.zeros(self.world_size, dtype=torch.float64, device=self.device)
        t[self.rank] = float(value)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def distinct_devices(self):
        """Number of DIFFERENT physical GPUs behind the ranks of the group (by PCI domain / bus / device id), or None when it
        cannot be told.  bench.py refuses a multi-rank measurement whose ranks share a GPU."""
        if not self.distributed or self.device.type != 'cuda':
            return 1
        try:
            pr = torch.cuda.get_device_properties(self.device)
            mine = [int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id)]
            t = torch.tensor(mine, dtype=torch.int64, device=self.device)
            got = [torch.empty_like(t) for _ in range(self.world_size)]
            dist.all_gather(got, t)
            return len({tuple(g.tolist()) for g in got})
        except Exception:
            return None

    def shutdown(self):
        if self.owns_group and dist.is_initialized():
            dist.destroy_process_group()


class FlatParameters:
    """Re-homes every parameter of the given modules into ONE contiguous fp32 buffer (16-byte aligned slices) and gives
    every parameter a `.grad` that is a view into a second flat buffer.  autograd accumulates into these views in
    place, so after backward the flat gradient is ready for a single all-reduce + flat optimiser kernels."""

    ALIGN = 4   # elements (16 bytes): GEMM operands and LayerNorm vectors are read as float4

    def __init__(self, modules):
        seen, self.params = set(), []
        for m in modules:
            for p in (m.parameters() if hasattr(m, 'parameters') else [m]):     # a module or a bare Parameter
                if id(p) not in seen:
                    seen.add(id(p))
                    self.params.append(p)
        assert self.params, 'no parameters'
        dev, dt = self.params[0].device, self.params[0].dtype
        assert dt == torch.float32 and all(p.device == dev and p.dtype == dt for p in self.params)
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = total
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def range_of(self, module):
        """[start, end) of the flat buffers covered by `module`'s parameters (they must be contiguous, i.e. the module
        was passed as one entry of `modules`): lets one clip + Adam kernel pair run per parameter group."""
        ids = {id(p) for p in module.parameters()}
        hits = [i for i, p in enumerate(self.params) if id(p) in ids]
        assert hits and hits == list(range(hits[0], hits[-1] + 1)), 'parameters of the module are not contiguous'
        last = hits[-1]
        end = self.offsets[last + 1] if last + 1 < len(self.params) else self.numel
        return self.offsets[hits[0]], end

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, off in zip(self.params, self.offsets):        # autograd may have replaced a view: re-attach
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def check_views(self):
        return all(p.data_ptr() == self.flat.data_ptr() + 4 * off and p.grad is not None
                   and p.grad.data_ptr() == self.flat_grad.data_ptr() + 4 * off
                   for p, off in zip(self.params, self.offsets))
