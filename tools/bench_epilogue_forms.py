"""Cost of the epilogue forms of the f16x3 NT kernel at the FFN1 shape (557 056 x 1024 x 256; 139 264 rows): bias | bias + relu + mask |
bias + relu + dropout + mask, B from planes.  GPU; `python tools/bench_epilogue_forms.py`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqcpc_bach_amd import hip
hip.load(); hip.set_gemm_mode(1)
gen = torch.Generator(device='cuda').manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M in (557056, 139264):
    N, K = 1024, 256
    a = torch.randn(M, K, device='cuda', generator=gen)
    w = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    bias = torch.randn(N, device='cuda', generator=gen) * 0.1
    desc = torch.tensor([(0, N, K, 0)], dtype=torch.int64).cuda(); tiles = (N // 32) * (K // 32)
    pl, plt, amax, ws = torch.empty_like(w), torch.empty_like(w), torch.zeros(1, device='cuda'), torch.empty(tiles, device='cuda')
    hip.call('vqcpc_weight_planes_many', w, desc, 1, tiles, amax, pl, plt, ws, 4 * tiles)
    st = torch.zeros(4, device='cuda'); hip.call('vqcpc_grad_amax', a, K, M, K, st[0:1])
    out = torch.empty(M, N, device='cuda'); mask = torch.empty(M * (N // 32), dtype=torch.int32, device='cuda')
    res = []
    for label, act, p, mk in (('bias', 0, 0.0, None), ('bias + relu + mask', 1, 0.0, mask), ('bias + relu + dropout + mask', 1, 0.1, mask)):
        t = timeit(lambda: hip.call('vqcpc_gemm_nt_g3_pl', a, K, pl, K, out, N, M, N, K, bias, act, p, 7, None, 0, None, 0, None, 1.0, mk, st, None, amax))
        res.append(f'{label} {t:.3f} ms')
    print(M, '|', ' | '.join(res))
