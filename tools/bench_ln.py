"""Residual + dropout + LayerNorm kernels (forward / backward) at the C1 and C4 row counts: time and achieved HBM rate.
    python tools/bench_ln.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip
hip.load()


def timeit(f, n=10, reps=6):
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


for M, d in [(557056, 256), (139264, 256), (278528, 512), (98304, 1024)]:
    x = torch.randn(M, d, device='cuda'); r = torch.randn(M, d, device='cuda'); dy = torch.randn(M, d, device='cuda')
    g = torch.randn(d, device='cuda'); b = torch.randn(d, device='cuda')
    y = torch.empty_like(x); mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
    ds = torch.empty_like(x); dr = torch.empty_like(x); dg = torch.empty(d, device='cuda'); db = torch.empty(d, device='cuda')
    nbytes = hip.query('vqcpc_add_layernorm_bwd_workspace', M, d)
    ws = torch.empty(nbytes // 4, device='cuda')
    tf = timeit(lambda: hip.call('vqcpc_add_layernorm_fwd', x, d, r, g, b, y, mean, rstd, M, d, 1e-5, 0.1, 7))
    tb = timeit(lambda: hip.call('vqcpc_add_layernorm_bwd', dy, x, d, r, g, mean, rstd, ds, dr, dg, db, M, d, 0.1, 7, ws, nbytes))
    # the form the encoder layers of C1 / C4 use: x IS the residual sum (r = None), the mask of d_r is regenerated
    ts = timeit(lambda: hip.call('vqcpc_add_layernorm_bwd', dy, x, d, None, g, mean, rstd, ds, dr, dg, db, M, d, 0.1, 7, ws, nbytes))
    tfs = timeit(lambda: hip.call('vqcpc_add_layernorm_fwd', x, d, None, g, b, y, mean, rstd, M, d, 1e-5, 0.0, 0))
    print(f'M={M} d={d}: s-form forward {tfs:.1f} us ({2 * 4.0 * M * d / tfs / 1e6:.2f} TB/s of 2 tensors)', flush=True)
    print(f'M={M} d={d}: s-form backward {ts:.1f} us ({4 * 4.0 * M * d / ts / 1e6:.2f} TB/s of 4 tensors)', flush=True)
    print(f'M={M} d={d}: forward {tf:.1f} us ({3 * 4.0 * M * d / tf / 1e6:.2f} TB/s of 3 tensors), backward {tb:.1f} us '
          f'({5 * 4.0 * M * d / tb / 1e6:.2f} TB/s of 5 tensors)', flush=True)
