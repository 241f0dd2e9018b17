"""Round 5: the tail-row launch of ragged f16x3 products (csrc/gemm_grad.hip gemm_nt_g3_tail_kernel, 64 x 128 tiles) through the C ABI:
the ragged C1 shapes as (a) six products (ops.gemm_nt outside any scope: two rounds + split-K / 128-tile remainder), (b) three
rounds of 256-tiles on the f16x3 kernel, (c) two rounds on the f16x3 kernel + the tail rows on the small tiles; and the tail
kernel alone on the remainder and on the under-filled 34 816-row shapes.

    python tools/bench_g3_tail.py        (needs a GPU)
"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
hip.set_gemm_mode(1)


def timed(fn, reps=10):
    """ms per call, from replays of a HIP graph of `reps` calls (no host time between the launches)"""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


def state_for(a, b):
    st = torch.zeros(4, device='cuda')
    hip.call('vqcpc_grad_amax', a, a.stride(0), a.shape[0], a.shape[1], st[0:1])
    hip.call('vqcpc_grad_amax', b, b.stride(0), b.shape[0], b.shape[1], st[1:2])
    return st


def main_rows(a, b, out, st, rows, add=None):
    hip.call('vqcpc_gemm_nt_grad', a, a.stride(0), b, b.stride(0), out, out.stride(0), rows, b.shape[0], a.shape[1], add,
             0 if add is None else add.stride(0), None, 0, None, 1.0, st)


def tail_rows(a, b, out, st, row0, add=None):
    rem = a.shape[0] - row0
    hip.call('vqcpc_gemm_nt_grad_tail', a[row0:], a.stride(0), b, b.stride(0), out[row0:], out.stride(0), rem, b.shape[0], a.shape[1],
             None, 0.0, 0, row0, None if add is None else add[row0:], 0 if add is None else add.stride(0), None, 0, st)


gen = torch.Generator(device='cuda').manual_seed(0)
print('shape                      six products | 3 rounds f16x3 | 2 rounds + tail (tail alone)   [us, TFLOP/s]')
for M, N, K in ((139264, 256, 1024), (139264, 256, 768), (139264, 256, 512), (139264, 256, 256)):
    a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3
    b = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    add = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
    out = torch.empty(M, N, device='cuda')
    st = state_for(a, b)
    fl = 2.0 * M * N * K
    t6 = timed(lambda: ops.gemm_nt(a, b, add=add, out=out))
    t3 = timed(lambda: main_rows(a, b, out, st, M, add))
    m_main = 131072
    tt = timed(lambda: tail_rows(a, b, out, st, m_main, add))
    t2 = timed(lambda: (main_rows(a, b, out, st, m_main, add), tail_rows(a, b, out, st, m_main, add)))
    print(f'{M:7d} x {N:4d} x {K:4d} + add   {t6 * 1e3:7.1f} {fl / t6 / 1e9:6.1f} | {t3 * 1e3:7.1f} {fl / t3 / 1e9:6.1f} | '
          f'{t2 * 1e3:7.1f} {fl / t2 / 1e9:6.1f} ({tt * 1e3:6.1f} us, {2.0 * (M - m_main) * N * K / tt / 1e9:6.1f})', flush=True)

print('the small-tile kernel as the whole launch (under-filled shapes: 136 tiles of 256 x 256)')
for M, N, K in ((34816, 256, 1024), (34816, 256, 256), (34816, 1024, 256), (69632, 256, 512)):
    a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3
    b = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    out = torch.empty(M, N, device='cuda')
    st = state_for(a, b)
    fl = 2.0 * M * N * K
    t6 = timed(lambda: ops.gemm_nt(a, b, out=out))
    tt = timed(lambda: tail_rows(a, b, out, st, 0))
    line = f'{M:7d} x {N:4d} x {K:4d}         six {t6 * 1e3:7.1f} {fl / t6 / 1e9:6.1f} | small tiles {tt * 1e3:7.1f} {fl / tt / 1e9:6.1f}'
    if M % 256 == 0:
        t3 = timed(lambda: main_rows(a, b, out, st, M))
        line += f' | 256-tiles f16x3 {t3 * 1e3:7.1f} {fl / t3 / 1e9:6.1f}'
    print(line, flush=True)

print('fixed cost per launch of the 256-tile kernels: time against whole rounds (256 tiles each), least-squares line a + b * rounds')
for N, K in ((256, 256), (256, 1024), (1024, 256)):
    pts = []
    for rounds in (1, 2, 3, 4, 6, 8):
        M = rounds * 65536 * 256 // N
        a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3
        b = torch.randn(N, K, device='cuda', generator=gen) * 0.05
        out = torch.empty(M, N, device='cuda')
        st = state_for(a, b)
        pts.append((rounds, timed(lambda: main_rows(a, b, out, st, M)) * 1e3))
        del a, out
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    bb = (n * sxy - sx * sy) / (n * sxx - sx * sx); aa = (sy - bb * sx) / n
    print(f'NT N = {N:4d} K = {K:4d}: ' + ' '.join(f'{r}:{t:.1f}' for r, t in pts) + f'  -> a = {aa:.1f} us, b = {bb:.1f} us per round', flush=True)
for N, K in ((256, 256), (1024, 256)):
    pts = []
    for M in (65536, 131072, 262144, 524288):
        a = torch.randn(M, N, device='cuda', generator=gen) * 1e-3
        b = torch.randn(M, K, device='cuda', generator=gen) * 0.05
        dw = torch.empty(N, K, device='cuda'); db = torch.empty(N, device='cuda')
        st = state_for(a, b)
        nb = hip.query('vqcpc_gemm_tn_grad_workspace', M, N, K)
        ws = hip.workspace(nb, a.device)
        pts.append((M // 65536, timed(lambda: hip.call('vqcpc_gemm_tn_grad', a, N, b, K, dw, db, M, N, K, 0, ws, nb, st)) * 1e3))
        del a, b
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    bb = (n * sxy - sx * sy) / (n * sxx - sx * sx); aa = (sy - bb * sx) / n
    print(f'TN (+ reduction) N = {N:4d} K = {K:4d}: ' + ' '.join(f'{r}x64k:{t:.1f}' for r, t in pts) + f'  -> a = {aa:.1f} us, b = {bb:.1f} us per 65 536 rows', flush=True)
