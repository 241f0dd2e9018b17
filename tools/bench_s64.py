"""64 x 64-tile bf16x6 NT kernel (under-filled launches of the student / decoder steps) against the 128-tile kernel and the
split-K entry point, per shape and epilogue.  A/B inside one process is not possible (the tile limit is read once):
    VQCPC_S64_MAX_TILES=0 python tools/bench_s64.py ; python tools/bench_s64.py"""
import os, sys, statistics, torch
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')   # measurement switches live in the lab build (vqcpc_bach_amd/build.py)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
hip.set_gemm_mode(1)


def timeit(f, n=20, reps=7):
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


print('VQCPC_S64_MAX_TILES =', os.environ.get('VQCPC_S64_MAX_TILES', '(default 256)'))
for M in (768, 3072, 12288, 24576):
    for N, K in ((512, 512), (1536, 512), (2048, 512), (512, 2048), (256, 256), (1024, 256)):
        a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda') * 0.05
        bias = torch.randn(N, device='cuda'); res = torch.randn(M, N, device='cuda')
        t0 = timeit(lambda: ops.gemm_nt(a, b))
        t1 = timeit(lambda: ops.gemm_nt(a, b, bias=bias, act=1, drop_p=0.1, seed=3))
        t2 = timeit(lambda: ops.gemm_nt(a, b, add=res))
        wsb = hip.query('vqcpc_gemm_nt_splitk_workspace', M, N, K)
        print(f'M={M:6d} N={N:5d} K={K:5d} tiles128={(M // 128) * (N // 128):5d}  plain {t0:7.1f} us  bias+relu+drop {t1:7.1f}  add {t2:7.1f}'
              f'  {2.0 * M * N * K / t0 / 1e6:6.1f} TFLOP/s  splitk_ws={wsb}', flush=True)
