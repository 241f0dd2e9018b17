"""N > 1 path on CPU: world_size-2 `gloo` runs of the data-parallel plumbing (vqcpc_bach_amd/parallel.py).
The compute of a rank is played by the CPU oracle (the product has no CPU path); what is checked is the DP contract:
mean of shard gradients == gradient of the global batch, flat-buffer all-reduce, rank-0 broadcasts, per-rank shards."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401
from oracle import vqcpc_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _tiny_cfg():
    return O.make_cfg(emb=8, vocab=[11] * 4, d=32, H=2, layers=[1, 1], ff=32, D=4, K=8, ncb=1, zdim=8, up_hidden=16, cdim=8,
                      gru_hidden=8, B=4, N=3, Kl=2, Kr=2)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from vqcpc_bach_amd.dataloaders.synthetic_cpc_dataloader import SyntheticCPCDataloaderGenerator
    from vqcpc_bach_amd.parallel import DataParallelContext, FlatParameters
    dp = DataParallelContext(device='cpu')
    assert dp.distributed and dp.world_size == world and dp.rank == rank
    cfg = _tiny_cfg()
    sd = O.init_state(cfg, seed=0)
    full = O.synthetic_batch(cfg, seed=5)
    B = cfg['B']
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = {k: v[lo:hi] for k, v in full.items()}

    # parameters live in one flat buffer; gradients are views into the flat all-reduce bucket
    holder = torch.nn.ParameterDict({k.replace('.', '/'): torch.nn.Parameter(v.clone()) for k, v in sd.items()})
    if rank == 1:                        # rank 1 starts from garbage: the rank-0 broadcast must fix it
        with torch.no_grad():
            for p in holder.values():
                p.add_(1.0)
    flat = FlatParameters([holder])
    assert flat.check_views()
    dp.broadcast_(flat.flat, src=0)
    P = {k.replace('/', '.'): p for k, p in holder.items()}
    for k in sd:
        assert torch.equal(P[k].detach(), sd[k]), k

    flat.zero_grad()
    loss = O.cpc_losses(shard, P, cfg)['loss']
    loss.backward()                                   # autograd accumulates IN PLACE into the flat views
    assert flat.check_views()
    dp.all_reduce_sum_(flat.flat_grad)
    flat.flat_grad.mul_(1.0 / world)                  # the product does this inside the optimiser kernel (grad_scale)

    ref_tr = O.OracleTrainer(cfg, sd)
    ref_tr.step(full, train=True)
    worst = 0.0
    for k, p in P.items():
        r = ref_tr.last_grads[k]
        worst = max(worst, float((p.grad - r).abs().max() / (r.abs().max() + 1e-12)))
    # per-rank synthetic shards differ, same-rank streams are reproducible
    a = next(SyntheticCPCDataloaderGenerator(num_blocks_left=2, num_blocks_right=2, num_negative_samples=3, rank=rank)
             .dataloaders(batch_size=2)[0])['x_left']
    t = a.float().sum().reshape(1).clone()
    gathered = [torch.zeros(1) for _ in range(world)]
    torch.distributed.all_gather(gathered, t)
    mx = dp.max_over_ranks(float(rank + 1))
    torch.save(dict(worst=worst, sums=[float(g) for g in gathered], mx=mx), os.path.join(out_dir, f'r{rank}.pt'))
    dp.barrier()
    dp.shutdown()


def test_dp_gradient_mean_equals_global_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f'r{r}.pt') for r in range(world)]
    for r in res:
        assert r['worst'] < 2e-5, r['worst']          # mean of shard grads == global-batch grads
        assert r['mx'] == float(world)
        assert r['sums'][0] != r['sums'][1]           # ranks draw different synthetic shards
    assert res[0]['sums'] == res[1]['sums']


def test_flat_parameters_views_and_alignment():
    from vqcpc_bach_amd.parallel import FlatParameters
    m = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 7))
    ref = [p.detach().clone() for p in m.parameters()]
    flat = FlatParameters([m])
    assert flat.check_views() and all(o % 4 == 0 for o in flat.offsets)
    for p, r in zip(m.parameters(), ref):
        assert torch.equal(p.detach(), r)
    x = torch.randn(4, 3)
    flat.zero_grad()
    m(x).sum().backward()
    m(x).sum().backward()                              # accumulation happens in the flat buffer
    assert flat.check_views()
    g = torch.autograd.grad(m(x).sum(), list(m.parameters()))
    for p, gi in zip(m.parameters(), g):
        assert torch.allclose(p.grad, 2 * gi, atol=1e-6)
    assert float(flat.flat_grad.abs().sum()) > 0


# ----------------------------------------------------------------------------------------------------------------------
# student step: three clip groups in ONE flat bucket, one all-reduce per step
# ----------------------------------------------------------------------------------------------------------------------
def _student_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle import student_oracle as S
    from vqcpc_bach_amd.parallel import DataParallelContext, FlatParameters
    dp = DataParallelContext(device='cpu')
    cfg = S.make_cfg('tiny', B=4)
    sd = S.init_state(cfg, seed=0)
    full = S.synthetic_batch(cfg, seed=5)
    B = cfg['B']
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = {'x': full['x'][lo:hi]}
    groups = ('teacher', 'auxiliary_decoder', 'encoder')
    holders = [torch.nn.ParameterDict({k.replace('.', '/'): torch.nn.Parameter(v.clone() + float(rank))
                                       for k, v in sd.items() if k.startswith(g + '.')}) for g in groups]
    flat = FlatParameters(holders)
    dp.broadcast_(flat.flat, src=0)                    # rank 1 started from shifted weights
    P = {k.replace('/', '.'): p for h in holders for k, p in h.items()}
    for k in sd:
        assert torch.equal(P[k].detach(), sd[k]), k
    ranges = [flat.range_of(h) for h in holders]
    assert ranges[0][0] == 0 and ranges[-1][1] == flat.numel
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))       # contiguous, non-overlapping groups

    flat.zero_grad()
    out = S.student_losses(shard['x'], 7, P, cfg)      # same masked event on both ranks -> comparable to the global batch
    (out['loss_teacher'] + out['loss_encdec']).backward()
    dp.all_reduce_sum_(flat.flat_grad)
    flat.flat_grad.mul_(1.0 / world)
    ref = S.StudentOracleTrainer(cfg, sd)
    ref.step(full, train=True, masked_event_index=7)
    # (tensors whose true gradient is zero hold 1e-10 rounding noise: absolute floor in the denominator)
    worst = max(float((p.grad - ref.last_grads[k]).abs().max() / (ref.last_grads[k].abs().max() + 1e-5))
                for k, p in P.items())
    # per-group norms of the reduced bucket == the three clip norms of the global batch
    norms = [float(flat.flat_grad[a:b].double().pow(2).sum().sqrt()) for a, b in ranges]
    refn = [float(ref.last_grad_norms[g + '.']) for g in groups]
    torch.save(dict(worst=worst, norms=norms, refn=refn), os.path.join(out_dir, f's{rank}.pt'))
    dp.barrier()
    dp.shutdown()


def test_student_dp_gradient_mean_and_group_ranges(tmp_path):
    world = 2
    mp.spawn(_student_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(tmp_path / f's{r}.pt')
        assert res['worst'] < 2e-4, res['worst']       # fp32 summation order differs between 2 shards and 1 batch
        for a, b in zip(res['norms'], res['refn']):
            assert abs(a - b) < 1e-4 * b


def test_flat_parameters_range_of_rejects_interleaved_modules():
    from vqcpc_bach_amd.parallel import FlatParameters
    a, b = torch.nn.Linear(3, 5), torch.nn.Linear(5, 2)
    flat = FlatParameters([a, b])
    ra, rb = flat.range_of(a), flat.range_of(b)
    assert ra == (0, 24) and rb == (24, 24 + 12 + 4)     # 15 + 5 -> padded 16 + 8; 10 + 2 -> 12 + 4
    both = torch.nn.ModuleList([a, b])
    assert flat.range_of(both) == (0, flat.numel)


# ----------------------------------------------------------------------------------------------------------------------
# decoder step: frozen encoder outside the bucket, bare Parameters inside it
# ----------------------------------------------------------------------------------------------------------------------
def _decoder_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import json
    from conftest import load_golden, sub_state
    from oracle import decoder_oracle as D
    from vqcpc_bach_amd.parallel import DataParallelContext, FlatParameters
    dp = DataParallelContext(device='cpu')
    g = load_golden('decoder_tiny')
    cfg = D.make_cfg(**json.loads(str(g['cfg_json'])))
    sd = sub_state(g, 'sd0')
    gen = torch.Generator().manual_seed(17)
    full = torch.cat([torch.randint(0, nv, (4, cfg['events'], 1), generator=gen) for nv in cfg['vocab']], dim=2)
    lo, hi = rank * 4 // world, (rank + 1) * 4 // world
    frozen = {k: v for k, v in sd.items() if k.startswith('encoder.')}
    # the decoder's trainable set mixes modules and bare nn.Parameters (sos, positional embeddings)
    bare = [k for k in sd if not k.startswith('encoder.') and '.' not in k]
    holder = torch.nn.ParameterDict({k.replace('.', '/'): torch.nn.Parameter(v.clone() + float(rank))
                                     for k, v in sd.items() if not k.startswith('encoder.') and k not in bare})
    bare_params = {k: torch.nn.Parameter(sd[k].clone() + float(rank)) for k in bare}
    flat = FlatParameters([holder] + list(bare_params.values()))
    assert flat.check_views() and len(bare) == 3
    dp.broadcast_(flat.flat, src=0)
    P = dict(frozen)
    P.update({k.replace('/', '.'): p for k, p in holder.items()})
    P.update(bare_params)
    for k in sd:
        assert torch.equal(P[k].detach(), sd[k]), k
    flat.zero_grad()
    x = full[lo:hi]
    loss = D.decoder_forward(D.encode_codes(x, P, cfg), x, P, cfg)['loss']
    loss.backward()
    assert flat.check_views() and all(v.grad is None for v in frozen.values())
    dp.all_reduce_sum_(flat.flat_grad)
    flat.flat_grad.mul_(1.0 / world)
    ref = D.DecoderOracleTrainer(cfg, sd)
    ref.step({'x': full}, train=True)
    worst = max(float((P[k].grad - r).abs().max() / (r.abs().max() + 1e-5)) for k, r in ref.last_grads.items())
    norm = float(flat.flat_grad.double().pow(2).sum().sqrt())
    torch.save(dict(worst=worst, norm=norm, refn=float(ref.last_grad_norm)), os.path.join(out_dir, f'd{rank}.pt'))
    dp.barrier()
    dp.shutdown()


def test_decoder_dp_gradient_mean_with_frozen_encoder(tmp_path):
    world = 2
    mp.spawn(_decoder_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(tmp_path / f'd{r}.pt')
        assert res['worst'] < 2e-4, res['worst']
        assert abs(res['norm'] - res['refn']) < 1e-4 * res['refn']


def test_bench_gpus_2_without_gpus_exits_nonzero_with_a_message():
    """`python bench.py --gpus 2` starts its own ranks; on a node without two GPUs it must refuse (non-zero, a message, no
    JSON line) instead of benchmarking one rank under a multi-GPU label."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'VQCPC_DP_SHARE_GPU')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert 'needs 2 visible GPUs' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_bench_rejects_a_launcher_whose_world_differs_from_gpus():
    import subprocess
    import sys
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr
