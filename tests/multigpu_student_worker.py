"""One rank of the student-step data-parallel tests (tests/test_multigpu_gpu.py): the bucketed step of
StudentEncoderTrainer -- all-reduce of the teacher's gradient range issued asynchronously under the encoder / decoder
half, all-reduce of the rest, join -- against the single all-reduce of the whole flat bucket, eagerly and as three graph
replays around the two collectives.  Writes <out_dir>/s<rank>.pt."""
import hashlib
import os
import sys

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import student_oracle as S  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def main():
    out_dir = sys.argv[1]
    from test_student_gpu import build_student
    from vqcpc_bach_amd import hip
    hip.load()
    hip.set_gemm_mode(1)
    cfg = S.make_cfg(ticks=16, d=64, H=4, ff=128, enc_layers=[1, 1], K=8, teacher_layers=2, dec_layers=[1, 1],
                     num_events_masked=2, B=4, vocab=[23, 19, 30, 14], emb=16)
    sd = S.init_state(cfg, seed=5)

    def make(buckets):
        os.environ['VQCPC_DP_BUCKETS'] = buckets
        tr = build_student(cfg, sd, lr=1e-3)          # init_optimizers(): process group, rank-0 broadcast
        tr.train()
        return tr

    tr_b, tr_s = make('2'), None
    dp = tr_b.dp
    rank, world = dp.rank, dp.world_size
    batches = [S.synthetic_batch(cfg, seed=60 + 10 * rank + i) for i in range(7)]      # every rank its own sequences
    events = [3, 9, 3, 3, 9, 12, 3]
    assert tr_b._dp_bucketed()
    # (1) gradients of one step: bucketed == single call, bit for bit (dropout 0: the same sums of the same two rank terms)
    b0 = {k: v.cuda() for k, v in batches[0].items()}
    st = tr_b._step_compute_teacher(b0, events[0])
    tr_b._all_reduce_teacher_async()
    out = tr_b._step_compute_encdec(st)
    tr_b._all_reduce_encdec_and_join()
    g_bucketed = tr_b.flat.flat_grad.clone()
    os.environ['VQCPC_DP_BUCKETS'] = '1'
    assert not tr_b._dp_bucketed()
    tr_b._step_compute(b0, events[0])
    tr_b._all_reduce_gradients()
    g_single = tr_b.flat.flat_grad.clone()
    grads_equal = bool(torch.equal(g_bucketed, g_single))
    grad_rel = float((g_bucketed - g_single).abs().max() / g_single.abs().max())
    del st, out
    # (2) training: bucketed eager steps vs bucketed graph replays (three graphs per step) vs the single-call form
    results = {}
    for name, buckets, graph in (('bucketed_eager', '2', False), ('bucketed_graph', '2', True), ('single_eager', '1', False)):
        tr = make(buckets)
        tr.enable_step_graph(graph)
        losses = []
        for b, m in zip(batches, events):
            o = tr.train_step({k: v.cuda() for k, v in b.items()}, train=True, masked_event_index=m)
            losses.append(float(o['loss_encdec']) + float(o['loss_teacher']))
        g = tr._graph
        results[name] = dict(params=tr.flat.flat.detach().clone(), losses=losses, replays=g.replays if g is not None else 0,
                             stages=len(g.stages) if g is not None else 0)
        tr.enable_step_graph(False)
    ref = results['bucketed_eager']['params']
    torch.cuda.synchronize()
    torch.save(dict(rank=rank, world=world, grads_equal=grads_equal, grad_rel=grad_rel,
                    digest_bucketed=digest(ref), digest_single=digest(results['single_eager']['params']),
                    graph_vs_eager=float((results['bucketed_graph']['params'] - ref).abs().max() / ref.abs().max()),
                    single_vs_bucketed=float((results['single_eager']['params'] - ref).abs().max() / ref.abs().max()),
                    digest_graph=digest(results['bucketed_graph']['params']), replays=results['bucketed_graph']['replays'],
                    stages=results['bucketed_graph']['stages'], losses=results['bucketed_eager']['losses'],
                    losses_graph=results['bucketed_graph']['losses']),
               os.path.join(out_dir, f's{rank}.pt'))
    dp.barrier()
    dp.shutdown()


if __name__ == '__main__':
    main()
