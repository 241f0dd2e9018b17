"""Whole-step HIP-graph replay (vqcpc_bach_amd/graphs.py): a replayed step must be THE SAME step as the eager one --
same losses, same parameters after several Adam steps with the LambdaLR schedule -- and dropout masks must still change
from replay to replay although the seed arguments are frozen in the graph (device-side step salt)."""
import numpy as np
import pytest
import torch

from oracle import vqcpc_oracle as O
from test_trainer_gpu import build_trainer

pytestmark = pytest.mark.gpu


def _cpc_setup(dropout):
    cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=64, H=4, layers=[2, 2], ff=128, D=16, K=32, ncb=2, zdim=16, up_hidden=32,
                     cdim=16, gru_hidden=32, B=8, N=7, Kl=4, Kr=4)
    sd = O.init_state(cfg, seed=21)
    batches = [O.synthetic_batch(cfg, seed=50 + i) for i in range(6)]
    st = {}
    O.encoder_forward(batches[0]['negative_samples'].reshape(-1, 4, 4), sd, cfg, stages=st)
    zp = st['z'].reshape(-1, cfg['D'])
    for c in range(cfg['ncb']):
        sd[f'encoder.quantizer.embeddings.{c}'] = zp[c * 5:c * 5 + cfg['K'], c * 8:(c + 1) * 8].clone() + 0.01
    return cfg, sd, batches


def _run(cfg, sd, batches, graph, dropout=0.0, schedule=True):
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.utils import SEEDS
    hip.load()
    hip.set_gemm_mode(1)
    SEEDS.manual_seed(7)
    tr = build_trainer(cfg, sd, lr=2e-3, dropout=dropout)
    tr.schedule_lr = schedule
    tr.train()
    tr.enable_step_graph(graph)
    losses = []
    try:
        for b in batches:
            out = tr.train_step({k: v.cuda() for k, v in b.items()}, train=True)
            losses.append(float(out['loss']))
        params = tr.flat.flat.detach().cpu().clone()
        replays = tr._graph.replays if tr._graph is not None else 0
    finally:
        tr.enable_step_graph(False)
        hip.set_gemm_mode(0)
    return losses, params, replays, tr


def test_graph_replay_is_the_eager_step():
    cfg, sd, batches = _cpc_setup(0.0)
    l_e, p_e, r_e, _ = _run(cfg, sd, batches, graph=False)
    l_g, p_g, r_g, tr = _run(cfg, sd, batches, graph=True)
    assert r_e == 0 and r_g == len(batches) - tr.graph_warmup_steps, (r_e, r_g)
    assert np.allclose(l_e, l_g, rtol=1e-6, atol=0), (l_e, l_g)
    # Adam's bias corrections come from pow() on the device instead of the host: at most an ulp of difference per step
    assert float((p_e - p_g).abs().max()) < 1e-6 * float(p_e.abs().max())
    assert tr.global_step == len(batches) and tr.optimizer.step_count == len(batches)
    assert abs(tr.current_lr() - 2e-3 * O.lr_lambda(len(batches))) < 1e-12


def test_graph_replay_is_the_eager_step_under_the_f16x3_arithmetic():
    """The training defaults of round 5 inside a captured step: the f16x3 kernels for the forward and the gradient products, their
    scale tables (primed in the eager warm-up steps, rolled by graph nodes) -- replayed steps == eager steps at the model
    dimensions of configs[1] (every eligible product forced through the kernels)."""
    from vqcpc_bach_amd import ops
    cfg = O.make_cfg('C1', B=4)
    sd = O.init_state(cfg, seed=23)
    batches = [O.synthetic_batch(cfg, seed=70 + i) for i in range(5)]
    prev = ops.set_gradient_arithmetic('f16x3')
    saved = ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS, ops.FWD_ARITH
    ops.GRAD_MIN_TILES = ops.GRAD_TN_MIN_ROWS = 0
    ops.FWD_ARITH = 'f16x3'
    from vqcpc_bach_amd import hip
    calls, raw = [], hip.call
    hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
    try:
        l_e, p_e, r_e, tr_e = _run(cfg, sd, batches, graph=False)
        n_eager = calls.count('vqcpc_gemm_nt_f16x3') + sum(1 for c in calls if c in ('vqcpc_gemm_nt_g3_pl', 'vqcpc_gemm_nt_g3_small'))
        assert 'vqcpc_weight_planes_many' in calls     # round 6: from the second step on B comes from the weights' fp16 planes
        l_g, p_g, r_g, tr = _run(cfg, sd, batches, graph=True)
    finally:
        hip.call = raw
        ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS, ops.FWD_ARITH = saved
        ops.set_gradient_arithmetic(prev)
    assert n_eager >= 2 * len(batches), n_eager                   # the forward did run on the three-product kernel
    assert r_e == 0 and r_g == len(batches) - tr.graph_warmup_steps, (r_e, r_g)
    assert np.allclose(l_e, l_g, rtol=1e-6, atol=0), (l_e, l_g)
    assert float((p_e - p_g).abs().max()) < 1e-6 * float(p_e.abs().max())
    assert ops.scale_saturations(tr.flat) == 0 and ops.scale_saturations(tr_e.flat) == 0


def test_graph_replays_draw_fresh_dropout_masks_and_training_still_works():
    cfg, sd, batches = _cpc_setup(0.2)
    same = [batches[0]] * 8
    l_g, p_g, r_g, tr = _run(cfg, sd, same, graph=True, dropout=0.2, schedule=False)
    assert r_g == 6
    # identical batch, identical frozen seed arguments: consecutive replays still see different masks (the loss moves by
    # far more than one Adam step at lr 2e-3 would explain if the masks were frozen -- and it is not periodic)
    assert len({round(x, 6) for x in l_g}) == len(l_g)
    assert np.isfinite(l_g).all()
    # reproducible: the salt is a function of the step number only
    l_g2, p_g2, _, _ = _run(cfg, sd, same, graph=True, dropout=0.2, schedule=False)
    assert l_g == l_g2 and torch.equal(p_g, p_g2)
    # and eager execution after release() is back to salt 0 (bit-identical to a run that never used graphs)
    l_e1, p_e1, _, _ = _run(cfg, sd, same[:3], graph=False, dropout=0.2, schedule=False)
    from vqcpc_bach_amd import hip
    hip.call('vqcpc_rng_salt_set', 12345)
    hip.call('vqcpc_rng_salt_set', 0)
    l_e2, p_e2, _, _ = _run(cfg, sd, same[:3], graph=False, dropout=0.2, schedule=False)
    assert l_e1 == l_e2 and torch.equal(p_e1, p_e2)


def test_salt_changes_the_masks_and_zero_restores_them():
    from vqcpc_bach_amd import hip, ops
    hip.load()
    m0 = ops.dropout_mask(4096, 0.3, 99, 'cuda').clone()
    hip.call('vqcpc_rng_salt_set', 0xABCDEF0123)
    try:
        m1 = ops.dropout_mask(4096, 0.3, 99, 'cuda').clone()
        a = torch.randn(256, 64, device='cuda')
        b = torch.randn(128, 64, device='cuda')
        out = ops.gemm_nt(a, b, act=1, drop_p=0.3, seed=5)              # another translation unit sees the same salt
        mask = ops.dropout_mask(256 * 128, 0.3, 5, 'cuda').reshape(256, 128)
        ref = torch.relu(a.double() @ b.double().t()) * mask.double() / 0.7
        assert float((out.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    finally:
        hip.call('vqcpc_rng_salt_set', 0)
    assert not torch.equal(m0, m1) and 0.25 < 1 - float(m1.mean()) < 0.35
    assert torch.equal(ops.dropout_mask(4096, 0.3, 99, 'cuda'), m0)


def test_student_step_graph_per_masked_event():
    """The student step (two HIP streams, three optimizers, a host-drawn masked event): one captured step per event
    index; same trajectory as eager execution."""
    from oracle import student_oracle as S
    from test_student_gpu import build_student
    from vqcpc_bach_amd import hip
    cfg = S.make_cfg(ticks=16, d=64, H=4, ff=128, enc_layers=[1, 1], K=8, teacher_layers=2, dec_layers=[1, 1],
                     num_events_masked=2, B=4, vocab=[23, 19, 30, 14], emb=16)
    sd = S.init_state(cfg, seed=5)
    batches = [S.synthetic_batch(cfg, seed=60 + i) for i in range(7)]
    events = [3, 9, 3, 3, 9, 12, 3]
    res = {}
    hip.set_gemm_mode(1)
    try:
        for graph in (False, True):
            tr = build_student(cfg, sd, lr=1e-3)
            tr.train()
            tr.enable_step_graph(graph)
            losses = []
            for b, m in zip(batches, events):
                out = tr.train_step({k: v.cuda() for k, v in b.items()}, train=True, masked_event_index=m)
                losses.append(float(out['loss_encdec']) + float(out['loss_teacher']))
            res[graph] = (losses, tr.flat.flat.detach().cpu().clone(),
                          (tr._graph.replays, len(tr._graph.graphs)) if tr._graph is not None else None)
            tr.enable_step_graph(False)
    finally:
        hip.set_gemm_mode(0)
    assert np.allclose(res[False][0], res[True][0], rtol=1e-6)
    assert float((res[False][1] - res[True][1]).abs().max()) < 1e-6 * float(res[False][1].abs().max())
    assert res[False][2] is None and res[True][2] == (5, 3)                 # events 3, 9 (after warm-up) and 12: 3 graphs


def test_decoder_step_graph():
    from oracle import decoder_oracle as DO
    from test_decoder_gpu import seeded_decoder
    from vqcpc_bach_amd import hip
    cfg = DO.make_cfg(emb=16, vocab=[20] * 4, d=64, H=4, layers=[1, 1], ff=128, D=4, K=8, ncb=1, zdim=16, up_hidden=32,
                      events=16, dec_emb=16, dec_d=64, dec_H=4, dec_enc_layers=1, dec_dec_layers=1, dec_ff=128, B=4, Kl=2, Kr=2)
    hip.set_gemm_mode(1)
    res = {}
    try:
        for graph in (False, True):
            dec, _ = seeded_decoder(cfg, seed=3)
            dec.schedule_lr = True
            dec.train()
            dec.enable_step_graph(graph)
            gen = torch.Generator().manual_seed(9)
            losses = []
            for _ in range(5):
                x = torch.stack([torch.randint(0, nv, (cfg['B'], cfg['events']), generator=gen) for nv in cfg['vocab']], dim=2)
                losses.append(float(dec.train_step({'x': x.cuda()}, train=True)))
            res[graph] = (losses, dec.flat.flat.detach().cpu().clone())
            dec.enable_step_graph(False)
    finally:
        hip.set_gemm_mode(0)
    assert np.allclose(res[False][0], res[True][0], rtol=1e-6)
    assert float((res[False][1] - res[True][1]).abs().max()) < 1e-6 * float(res[False][1].abs().max())


@pytest.mark.parametrize('graph_a,graph_b', [(True, False), (True, True), (False, False)])
def test_two_trainers_interleaved_in_one_process_equal_each_run_apart(graph_a, graph_b):
    """the library keeps process-wide state (GEMM mode, the device-side RNG salt of graph replays,
    the deferred weight-gradient lists and the transposed-weight cache of a gradient scope, the dropout-seed source).
    None of it may leak from one trainer into another at STEP granularity: two trainers of different shapes whose steps
    alternate in one process -- eager and replayed, both with dropout -- end with parameters bit-identical to the ones
    each reaches when it runs alone."""
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.utils import SEEDS
    hip.load()
    cfg_a, sd_a, batches_a = _cpc_setup(0.2)
    cfg_b = O.make_cfg(emb=32, vocab=[56] * 4, d=128, H=4, layers=[1, 1], ff=256, D=32, K=16, ncb=1, zdim=32, up_hidden=64,
                       cdim=32, gru_hidden=64, B=4, N=5, Kl=2, Kr=2)
    sd_b = O.init_state(cfg_b, seed=33)
    batches_b = [O.synthetic_batch(cfg_b, seed=90 + i) for i in range(6)]
    st = {}
    O.encoder_forward(batches_b[0]['negative_samples'].reshape(-1, 4, 4), sd_b, cfg_b, stages=st)
    sd_b['encoder.quantizer.embeddings.0'] = st['z'].reshape(-1, cfg_b['D'])[:cfg_b['K']].clone() + 0.01

    def make(cfg, sd, seed, dropout, graph):
        SEEDS.manual_seed(seed)
        tr = build_trainer(cfg, sd, lr=2e-3, dropout=dropout)
        tr.train()
        tr.enable_step_graph(graph)
        return tr

    def steps(tr, batches):
        for b in batches:
            yield tr.train_step({k: v.cuda() for k, v in b.items()}, train=True)

    hip.set_gemm_mode(1)
    try:
        apart = []
        for cfg, sd, batches, seed, p, graph in ((cfg_a, sd_a, batches_a, 7, 0.2, graph_a), (cfg_b, sd_b, batches_b, 8, 0.1, graph_b)):
            tr = make(cfg, sd, seed, p, graph)
            losses = [float(o['loss']) for o in steps(tr, batches)]
            apart.append((losses, tr.flat.flat.detach().clone()))
            tr.enable_step_graph(False)
        ta = make(cfg_a, sd_a, 7, 0.2, graph_a)
        # trainer B is seeded and built AFTER trainer A took its first steps: its stream forks from its own seeding
        la, lb = [], []
        ga = steps(ta, batches_a)
        la.append(float(next(ga)['loss']))
        tb = make(cfg_b, sd_b, 8, 0.1, graph_b)
        gb = steps(tb, batches_b)
        for _ in range(len(batches_a) - 1):
            lb.append(float(next(gb)['loss']))
            la.append(float(next(ga)['loss']))
        lb.append(float(next(gb)['loss']))
        assert la == apart[0][0] and lb == apart[1][0], (la, apart[0][0], lb, apart[1][0])
        assert torch.equal(ta.flat.flat, apart[0][1]) and torch.equal(tb.flat.flat, apart[1][1])
        if graph_a:
            assert ta._graph.replays == len(batches_a) - ta.graph_warmup_steps
        ta.enable_step_graph(False)
        tb.enable_step_graph(False)
    finally:
        hip.set_gemm_mode(0)


@pytest.mark.parametrize('graph_a,graph_b', [(True, False), (True, True)])
def test_two_trainers_stepping_from_two_threads_equal_each_run_apart(graph_a, graph_b):
    """SURVEY.md section 8(b) "re-entrant", the part the round-4 review found untested: two trainers of different shapes driven from
    TWO Python threads, each on its own HIP stream, started together -- with dropout, eager and replayed steps, and
    torch.autograd's single engine thread running both backward passes.  utils.STEP_LOCK serialises their steps on the host and
    chains them on the device (the next step's stream waits for the previous step's last kernel: the device-side RNG salt and
    the gradient scope are never shared): losses and parameters bit-identical to each trainer running alone."""
    import threading
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.utils import SEEDS
    hip.load()
    cfg_a, sd_a, batches_a = _cpc_setup(0.2)
    cfg_b = O.make_cfg(emb=32, vocab=[56] * 4, d=128, H=4, layers=[1, 1], ff=256, D=32, K=16, ncb=1, zdim=32, up_hidden=64,
                       cdim=32, gru_hidden=64, B=4, N=5, Kl=2, Kr=2)
    sd_b = O.init_state(cfg_b, seed=33)
    batches_b = [O.synthetic_batch(cfg_b, seed=90 + i) for i in range(6)]
    st = {}
    O.encoder_forward(batches_b[0]['negative_samples'].reshape(-1, 4, 4), sd_b, cfg_b, stages=st)
    sd_b['encoder.quantizer.embeddings.0'] = st['z'].reshape(-1, cfg_b['D'])[:cfg_b['K']].clone() + 0.01

    def make(cfg, sd, seed, dropout, graph):
        SEEDS.manual_seed(seed)
        tr = build_trainer(cfg, sd, lr=2e-3, dropout=dropout)
        tr.train()
        tr.seed_dropout(seed)                       # each trainer's own stream, whatever the other thread seeds meanwhile
        tr.enable_step_graph(graph)
        return tr

    def run(tr, batches, losses, stream=None, gate=None):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            if gate is not None:
                gate.wait()
            for b in batches:
                losses.append(float(tr.train_step({k: v.cuda() for k, v in b.items()}, train=True)['loss']))
            torch.cuda.current_stream().synchronize()

    hip.set_gemm_mode(1)
    try:
        apart = []
        for cfg, sd, batches, seed, p, graph in ((cfg_a, sd_a, batches_a, 7, 0.2, graph_a), (cfg_b, sd_b, batches_b, 8, 0.1, graph_b)):
            tr = make(cfg, sd, seed, p, graph)
            losses = []
            run(tr, batches, losses)
            apart.append((losses, tr.flat.flat.detach().clone()))
            tr.enable_step_graph(False)
        ta, tb = make(cfg_a, sd_a, 7, 0.2, graph_a), make(cfg_b, sd_b, 8, 0.1, graph_b)
        la, lb, errors = [], [], []
        gate = threading.Barrier(2)

        def worker(tr, batches, losses):
            try:
                run(tr, batches, losses, stream=torch.cuda.Stream(), gate=gate)
            except BaseException as e:              # surfaced by the main thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(ta, batches_a, la)), threading.Thread(target=worker, args=(tb, batches_b, lb))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not errors, errors
        torch.cuda.synchronize()
        assert la == apart[0][0] and lb == apart[1][0], (la, apart[0][0], lb, apart[1][0])
        assert torch.equal(ta.flat.flat, apart[0][1]) and torch.equal(tb.flat.flat, apart[1][1])
        ta.enable_step_graph(False)
        tb.enable_step_graph(False)
    finally:
        hip.set_gemm_mode(0)
