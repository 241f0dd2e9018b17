"""One rank of tests/test_multigpu_gpu.py (launched by torch.distributed.run, one process per GPU, RCCL).

Every rank builds the same model from DIFFERENT seeds (so only the rank-0 broadcast can make them equal), trains 3 steps
on its shard of a global batch and writes what it observed to <out_dir>/r<rank>.pt; the launching test asserts on it."""
import hashlib
import os
import sys

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import vqcpc_oracle as O  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def main():
    out_dir = sys.argv[1]
    from test_trainer_gpu import build_trainer
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.parallel import DataParallelContext
    hip.load()
    hip.set_gemm_mode(1)
    dp = DataParallelContext()
    rank, world = dp.rank, dp.world_size
    per_rank = 4
    cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=64, H=4, layers=[2, 1], ff=128, D=16, K=32, ncb=2, zdim=16, up_hidden=32,
                     cdim=16, gru_hidden=32, B=per_rank * world, N=5, Kl=3, Kr=3)
    f16x3_calls = [0]
    if os.environ.get('VQCPC_TEST_TRAINING_DEFAULTS', '0') == '1':
        # what train_model() selects (f16x3 forward + gradient products), at a width the 256-tile kernels take (d_model = ff = 256),
        # every eligible product forced through them: each rank owns its scale tables (its shard's amax differs from the other
        # rank's), the all-reduced gradient and hence the replicas must still be bit-identical
        from vqcpc_bach_amd import ops
        ops.set_gradient_arithmetic('f16x3')
        ops.set_forward_arithmetic('f16x3')
        ops.GRAD_MIN_TILES = ops.GRAD_TN_MIN_ROWS = 0
        cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=256, H=4, layers=[2, 1], ff=256, D=16, K=32, ncb=2, zdim=16, up_hidden=32,
                         cdim=16, gru_hidden=32, B=per_rank * world, N=5, Kl=3, Kr=3)
        raw = hip.call

        def counting(name, *args):
            f16x3_calls[0] += name in ('vqcpc_gemm_nt_f16x3', 'vqcpc_gemm_nt_grad', 'vqcpc_gemm_tn_grad', 'vqcpc_gemm_nt_g3_pl')
            return raw(name, *args)
        hip.call = counting
    sd0 = O.init_state(cfg, seed=100)                       # the model every rank must end up with (rank 0's)
    sd_mine = O.init_state(cfg, seed=100 + rank)            # what this rank starts from before the broadcast
    full = O.synthetic_batch(cfg, seed=7)
    shard = {k: v[rank * per_rank:(rank + 1) * per_rank] for k, v in full.items()}

    torch.manual_seed(1234 + rank)                          # the data-init permutation differs per rank on purpose
    tr = build_trainer(cfg, sd_mine, lr=1e-3)               # init_optimizers() broadcasts rank 0's flat parameter buffer
    assert tr.dp.world_size == world and tr.dp.distributed == dp.distributed
    names = [n for n, _ in tr.named_parameters()]
    init_equal = all(torch.equal(p.detach().cpu(), sd0[n]) for n, p in tr.named_parameters())

    # data-dependent codebook initialisation on the first forward: rank 0's codebooks everywhere
    q = tr.encoder.quantizer
    q.initialize = True
    tr.eval()
    with torch.no_grad():
        tr.compute_losses(shard)
    cb = torch.stack([e.detach() for e in q.embeddings]).contiguous()
    gathered = [torch.empty_like(cb) for _ in range(world)]
    if tr.dp.distributed:
        torch.distributed.all_gather(gathered, cb)
    else:
        gathered = [cb]
    codebook_equal = all(torch.equal(g, gathered[0]) for g in gathered) and not q.initialize

    # gradient of the GLOBAL batch from the oracle (same codebooks), vs all-reduced mean of the shard gradients
    sd_ref = {n: p.detach().cpu().clone() for n, p in tr.named_parameters()}
    otr = O.OracleTrainer(cfg, sd_ref, lr=1e-3)
    ref = otr.step(full, train=True)
    tr.train()
    if f16x3_calls[0] or os.environ.get('VQCPC_TEST_TRAINING_DEFAULTS', '0') == '1':
        from vqcpc_bach_amd import ops
        with ops.forward_arithmetic(tr.flat):
            loss, out = tr.compute_losses(shard)
        tr.flat.zero_grad()
        with ops.direct_weight_gradients(tr.flat):
            loss.backward()
    else:
        loss, out = tr.compute_losses(shard)
        tr.flat.zero_grad()
        loss.backward()
    tr.dp.all_reduce_sum_(tr.flat.flat_grad)
    tr.flat.flat_grad.mul_(1.0 / world)
    grad_worst = 0.0
    for n, p in tr.named_parameters():
        r = otr.last_grads[n]
        grad_worst = max(grad_worst, float((p.grad.cpu() - r).abs().max() / (r.abs().max() + 1e-30)))
    lo, hi = rank * per_rank, (rank + 1) * per_rank
    idx_equal = (torch.equal(out['idx_left'].cpu(), ref['idx_left'][lo:hi]) and
                 torch.equal(out['idx_right'].cpu(), ref['idx_right'][lo:hi]) and
                 torch.equal(out['idx_negative'].cpu().reshape(per_rank, -1), ref['idx_negative'].reshape(cfg['B'], -1)[lo:hi]))

    # three optimiser steps through the product's own train_step (all-reduce + grad_scale inside)
    batches = [O.synthetic_batch(cfg, seed=20 + i) for i in range(3)]
    for b in batches:
        tr.train_step({k: v[lo:hi] for k, v in b.items()}, train=True)
    m = tr.epoch(iter([{k: v[lo:hi] for k, v in full.items()}]), train=False, num_batches=1, corrupt_labels=False)
    eager_digest = digest(tr.flat.flat)

    # the data-parallel step as GRAPH REPLAYS (graphs.py: two graphs around the eagerly issued all-reduce; one graph with the
    # collective captured under VQCPC_DP_GRAPH=capture) against an eager twin that starts from the same parameters and
    # Adam state: 2 eager warm-up steps + 4 replays
    del loss, out                 # graphs.py pitfall: no autograd graph of an earlier step may be alive when a capture ends
    sd_now = {n: p.detach().cpu().clone() for n, p in tr.named_parameters()}
    twin = build_trainer(cfg, sd_now, lr=1e-3)
    twin.optimizer.m.copy_(tr.optimizer.m)
    twin.optimizer.v.copy_(tr.optimizer.v)
    twin.optimizer.step_count = tr.optimizer.step_count
    twin.train()
    tr.train()
    tr.enable_step_graph(True)
    more = [O.synthetic_batch(cfg, seed=40 + i) for i in range(6)]
    for b in more:
        tr.train_step({k: v[lo:hi].cuda() for k, v in b.items()}, train=True)
        twin.train_step({k: v[lo:hi].cuda() for k, v in b.items()}, train=True)
    g = tr._graph
    graph_replays = g.replays if g is not None else 0
    graph_two = bool(g is not None and g.finish_fn is not None)
    graph_vs_eager = float((tr.flat.flat - twin.flat.flat).abs().max() / twin.flat.flat.abs().max())
    torch.cuda.synchronize()
    torch.save(dict(rank=rank, world=world, init_equal=init_equal, codebook_equal=codebook_equal, grad_worst=grad_worst,
                    idx_equal=idx_equal, param_digest=eager_digest, loss_global=m['loss'], names=len(names),
                    graph_replays=graph_replays, graph_two=graph_two, graph_vs_eager=graph_vs_eager,
                    graph_digest=digest(tr.flat.flat), f16x3_calls=f16x3_calls[0]),
               os.path.join(out_dir, f'r{rank}.pt'))
    dp.barrier()
    dp.shutdown()


if __name__ == '__main__':
    main()
