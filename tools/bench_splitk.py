"""Split-K NT GEMM (vqcpc_gemm_nt_splitk) against the single-launch path at the student / decoder step's under-filled
shapes, and the vectorised partial-sum reduction of the weight-gradient GEMM.    python tools/bench_splitk.py"""
import os, sys, statistics, torch
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')   # measurement switches live in the lab build (vqcpc_bach_amd/build.py)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
hip.set_gemm_mode(1)


def timeit(f, n=20, reps=6):
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


for M, N, K in [(3072, 512, 2048), (3072, 512, 1536), (3072, 512, 1024), (768, 512, 2048), (768, 512, 1536), (192, 512, 2048),
                (2048, 512, 2048), (4096, 512, 2048), (1536, 512, 2048), (3072, 512, 512), (768, 512, 512), (768, 1536, 512),
                (768, 2048, 512), (768, 512, 1024), (256, 512, 1536), (256, 512, 1024)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda') * 0.05
    bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda')
    ref = a.double() @ b.double().t()
    out = {}
    for sk in (False, True):
        ops.SPLIT_K = sk
        t_b = timeit(lambda: ops.gemm_nt(a, b, bias=bias))
        t_a = timeit(lambda: ops.gemm_nt(a, b, add=res))
        y = ops.gemm_nt(a, b, bias=bias, add=res)
        err = float((y.double() - (ref + bias.double() + res.double())).abs().max() / ref.abs().max())
        out[sk] = (t_b, t_a, err)
    ws = hip.query('vqcpc_gemm_nt_splitk_workspace', M, N, K)
    print(f'M={M} N={N} K={K}: single launch {out[False][0]:.1f} / {out[False][1]:.1f} us (bias / add), split-K '
          f'{out[True][0]:.1f} / {out[True][1]:.1f} us [{ws // (4 * M * N)} planes]  rel err {out[False][2]:.2e} / {out[True][2]:.2e}  '
          f'({2.0 * M * N * K / out[True][0] / 1e6:.0f} TFLOP/s)', flush=True)
ops.SPLIT_K = True
hip.set_gemm_mode(0)

# feed-forward relu / dropout projection and its gate GEMM: bit-mask forms (256-tile ping-pong kernel only) against the
# fp32-gate forms (which may pick the 128-tile kernel) at few-tile shapes
hip.set_gemm_mode(1)
for M, N, K in [(3072, 2048, 512), (768, 2048, 512), (1536, 2048, 512), (6144, 2048, 512)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda') * 0.05; bias = torch.randn(N, device='cuda')
    dy = torch.randn(M, N // 4, device='cuda'); w2 = torch.randn(N, N // 4, device='cuda') * 0.05
    t_f = timeit(lambda: ops.gemm_nt(a, b, bias=bias, act=1, drop_p=0.1, seed=3))
    t_m = timeit(lambda: ops.gemm_nt_relu_mask(a, b, bias, 0.1, 3))
    h, mask = ops.gemm_nt_relu_mask(a, b, bias, 0.1, 3)
    t_g = timeit(lambda: ops.gemm_nt(dy, w2, gate=h, gate_scale=1 / 0.9))
    t_b = timeit(lambda: ops.gemm_nt_gatebits(dy, w2, mask, 1 / 0.9))
    hip.set_gemm_mode(3)
    t_f128 = timeit(lambda: ops.gemm_nt(a, b, bias=bias, act=1, drop_p=0.1, seed=3))
    t_g128 = timeit(lambda: ops.gemm_nt(dy, w2, gate=h, gate_scale=1 / 0.9))
    hip.set_gemm_mode(1)
    print(f'M={M} N={N} K={K}: relu+dropout fp32 {t_f:.1f} us (128-tile only {t_f128:.1f}), with mask-out {t_m:.1f} us; '
          f'gate fp32 {t_g:.1f} us (128-tile only {t_g128:.1f}), gate bits {t_b:.1f} us', flush=True)
hip.set_gemm_mode(0)
