import os, sys, torch
sys.path.insert(0, '/root/repo')
from vqcpc_bach_amd import hip
hip.load(); hip.set_gemm_mode(1)
gen = torch.Generator(device='cuda').manual_seed(0)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (139264, 34816):
    N = K = 256
    a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3
    w = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    add = torch.randn(M, N, device='cuda', generator=gen)
    big = torch.randn(4 * M, N, device='cuda', generator=gen)
    big2 = big.clone()
    c, other = big[::4], big2[::4]           # strided rows, as d x[::f]
    st = torch.zeros(4, device='cuda')
    hip.call('vqcpc_grad_amax', a, K, M, K, st[0:1]); hip.call('vqcpc_grad_amax', w, K, N, K, st[1:2])
    Mm = M - M % 256
    def old(): hip.call('vqcpc_gemm_nt_grad', a, K, w, K, c, 4 * N, Mm, N, K, add, N, other, 4 * N, None, 1.0, st)
    def new(): hip.call('vqcpc_gemm_nt_grad', a, K, w, K, c, 4 * N, Mm, N, K, add, N, c, 4 * N, None, 1.0, st)
    print(M, 'add + add2 (load, store)', round(timeit(old), 4), 'ms | add2 == C (atomic accumulate)', round(timeit(new), 4), 'ms')
