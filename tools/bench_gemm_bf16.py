"""bf16 NT GEMM (configs[4] path) per epilogue / output form at the C4 layer shapes (a quarter of the rows).
    python tools/bench_gemm_bf16.py [rows]"""
import os, sys, statistics, torch
os.environ.setdefault('VQCPC_LAB', '1')        # the A/B switch between the bf16 kernels is a lab-build entry point
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
hip.set_gemm_mode(8)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 278528


def timeit(f, n=5, reps=5):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


VARIANTS = [int(v) for v in os.environ.get('VQCPC_BF16_VARIANTS', '0,1').split(',')]
SHAPES = [(2048, 512), (512, 2048), (512, 512), (1536, 512)]
if os.environ.get('VQCPC_BF16_SHAPES'):
    SHAPES = [tuple(int(x) for x in sh.split('x')) for sh in os.environ['VQCPC_BF16_SHAPES'].split(',')]
FORMS = os.environ.get('VQCPC_BF16_FORMS')
for N, K in SHAPES:
    a = ops.cast_bf16(torch.randn(M, K, device='cuda')); b = ops.cast_bf16(torch.randn(N, K, device='cuda') * 0.05)
    bias = torch.randn(N, device='cuda')
    gate_b = torch.randn(M, N, device='cuda').bfloat16()
    res = torch.randn(M, N, device='cuda')
    out32 = torch.empty(M, N, device='cuda')
    forms = [('none -> f32', dict(out=out32)),
             ('none -> bf16', dict(out_f32=False, out_bf16=True)),
             ('bias -> f32', dict(bias=bias, out=out32)),
             ('bias+relu -> bf16', dict(bias=bias, act=1, out_f32=False, out_bf16=True)),
             ('bias+relu+drop -> bf16', dict(bias=bias, act=1, drop_p=0.1, seed=5, out_f32=False, out_bf16=True)),
             ('gate_b -> bf16', dict(gate_b=gate_b, gate_scale=1.1, out_f32=False, out_bf16=True)),
             ('add -> f32', dict(add=res, out=out32))]
    for name, kw in forms:
        if FORMS and name not in FORMS.split(','):
            continue
        res_ = []
        outs = []
        for v in VARIANTS:
            hip.call('vqcpc_gemm_bf16_set_variant', v)
            r_ = ops.gemm_nt_bf16(a, b, **kw)
            outs.append([x.clone() for x in (r_ if isinstance(r_, (tuple, list)) else [r_]) if torch.is_tensor(x)])
            t = timeit(lambda: ops.gemm_nt_bf16(a, b, **kw))
            res_.append(f'v{v}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TFLOP/s')
        same = all(torch.equal(x, y) for o in outs[1:] for x, y in zip(outs[0], o))
        print(f'M={M} N={N} K={K} {name:24s} ' + ' | '.join(res_) + ('' if len(VARIANTS) < 2 else f'  bit-identical: {same}'), flush=True)
    hip.call('vqcpc_gemm_bf16_set_variant', 1)
hip.set_gemm_mode(0)

# weight-gradient (TN) bf16 kernel
hip.set_gemm_mode(8)
for N, K in ([] if os.environ.get('VQCPC_BF16_NO_TN') else [(2048, 512), (512, 2048), (1536, 512), (512, 512)]):
    a = ops.cast_bf16(torch.randn(M, N, device='cuda')); b = ops.cast_bf16(torch.randn(M, K, device='cuda'))
    res_, outs = [], []
    for v in VARIANTS:
        hip.call('vqcpc_gemm_bf16_set_variant', v)
        outs.append(ops.gemm_tn_bf16(a, b)[0].clone())
        t = timeit(lambda: ops.gemm_tn_bf16(a, b))
        res_.append(f'v{v}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TFLOP/s')
    hip.call('vqcpc_gemm_bf16_set_variant', 1)
    diff = float((outs[0] - outs[-1]).abs().max() / outs[0].abs().max()) if len(outs) > 1 else 0.0
    print(f'M={M} N={N} K={K} wgrad (TN, incl. reduction) ' + ' | '.join(res_) + f'  max rel diff {diff:.1e}', flush=True)
hip.set_gemm_mode(0)
