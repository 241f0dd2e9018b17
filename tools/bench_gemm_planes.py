"""Micro-benchmark: bf16x6 NT GEMM on pre-split (P3) operands vs the fp32-in kernel (mode 1) on the C1 shapes (GPU only).
    python tools/bench_gemm_planes.py [reps]"""
import os
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')   # measurement switches live in the lab build (vqcpc_bach_amd/build.py)
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops  # noqa: E402


def time_it(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    hip.load()
    hip.set_gemm_mode(1)
    M1, M2 = 557056, 139264
    shapes = [(M1, 768, 256, 'bias'), (M1, 256, 256, 'bias'), (M1, 1024, 256, 'relu_drop'), (M1, 256, 1024, 'bias'),
              (M1, 256, 768, 'none'), (M1, 1024, 256, 'gate'), (M1, 256, 1024, 'add'), (M2, 1024, 256, 'relu_drop'),
              (M2, 256, 1024, 'bias'), (34816, 256, 256, 'bias')]
    print(f'{"shape":>26s} {"epilogue":10s} {"fp32-in TF":>11s} {"planes TF":>10s} {"split A us":>11s}')
    for M, N, K, epi in shapes:
        a = torch.randn(M, K, device='cuda')
        b = torch.randn(N, K, device='cuda')
        bias = torch.randn(N, device='cuda')
        out = torch.empty(M, N, device='cuda')
        aux = torch.randn(M, N, device='cuda') if epi in ('gate', 'add') else None
        kw = dict(none={}, bias=dict(bias=bias), relu_drop=dict(bias=bias, act=1, drop_p=0.1, seed=3),
                  gate=dict(gate=aux, gate_scale=1.1), add=dict(add=aux))[epi]
        ap, bp = ops.split3_planes(a), ops.split3_planes(b)
        r0, r1 = [], []
        for r in range(6):
            t0 = time_it(lambda: ops.gemm_nt(a, b, out=out, **kw), reps)
            t1 = time_it(lambda: ops.gemm_nt_planes(ap, bp, M, N, K, out=out, **kw), reps)
            if r:
                r0.append(2.0 * M * N * K / t0 / 1e12)
                r1.append(2.0 * M * N * K / t1 / 1e12)
        ts = time_it(lambda: ops.split3_planes(a), reps)
        print(f'{str((M, N, K)):>26s} {epi:10s} {statistics.median(r0):11.1f} {statistics.median(r1):10.1f} {ts * 1e6:11.1f}',
              flush=True)
        del a, b, out, aux, ap, bp
    hip.set_gemm_mode(0)


if __name__ == '__main__':
    main()
