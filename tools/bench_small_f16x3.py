"""Sub-256-tile NT products of the student / decoder steps (3 072 - 12 288 rows, d_model 512): what the step runs today (ops.gemm_nt in
the bf16x6 mode: 128-tile / 64-tile six-product kernels, split-K for long K) against the 64 x 128-tile three-product kernel
(vqcpc_gemm_nt_grad_tail, written for the tail rows of ragged C1 launches).  GPU; `python tools/bench_small_f16x3.py`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqcpc_bach_amd import hip, ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    hip.load()
    hip.set_gemm_mode(1)
    gen = torch.Generator(device='cuda').manual_seed(0)
    for M, N, K in ((3072, 512, 512), (3072, 1536, 512), (3072, 2048, 512), (3072, 512, 2048), (768, 512, 512), (768, 2048, 512),
                    (12288, 512, 512), (12288, 1536, 512), (12288, 1024, 512), (12288, 512, 1024), (192, 512, 512)):
        a = torch.randn(M, K, device='cuda', generator=gen)
        w = torch.randn(N, K, device='cuda', generator=gen) * 0.05
        bias = torch.randn(N, device='cuda', generator=gen)
        out0, out1 = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
        st = torch.zeros(4, device='cuda')
        hip.call('vqcpc_grad_amax', a, K, M, K, st[0:1])
        hip.call('vqcpc_grad_amax', w, K, N, K, st[1:2])
        t0 = timeit(lambda: ops.gemm_nt(a, w, bias=bias, out=out0))
        if not hip.query('vqcpc_gemm_nt_grad_tail_supported', M, N, K):
            print(f'{M} x {N} x {K}: six products {t0:.1f} us; the small-tile kernel does not take this shape')
            continue
        t1 = timeit(lambda: hip.call('vqcpc_gemm_nt_grad_tail', a, K, w, K, out1, N, M, N, K, bias, 0.0, 0, 0, None, 0, None, 0, st))
        err = float((out0 - out1).abs().max() / out0.abs().max())
        fl = 2.0 * M * N * K / 1e6
        print(f'{M} x {N} x {K}: six products (today) {t0:.1f} us ({fl / t0:.0f} TFLOP/s) | f16x3 64 x 128 tiles {t1:.1f} us ({fl / t1:.0f}) '
              f'x{t0 / t1:.2f}, max diff {err:.1e}')


if __name__ == '__main__':
    main()
