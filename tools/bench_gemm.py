"""Micro-benchmark of the fp32 MFMA GEMMs on the shapes of the C1 training step (HIP-event timed, GPU only).
    python tools/bench_gemm.py [reps]
Prints TFLOP/s and % of the 157.3 TFLOP/s fp32 MFMA peak per shape; used to A/B kernel variants."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops  # noqa: E402

PEAK = 157.3


def time_it(fn, reps):
    fn()
    torch.cuda.synchronize()
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
    if len(sys.argv) > 2:
        hip.set_gemm_mode(int(sys.argv[2]))
    print('gemm mode', hip.get_gemm_mode())
    dev = 'cuda'
    M1, M2 = 557056, 139264
    nt_shapes = [(M1, 768, 256, 'qkv fwd'), (M1, 256, 256, 'out-proj'), (M1, 1024, 256, 'ffn1 fwd'), (M1, 256, 1024, 'ffn2 fwd'),
                 (M1, 256, 768, 'qkv dgrad'), (M2, 768, 256, 'qkv fwd s2'), (M2, 256, 1024, 'ffn2 fwd s2'), (34816, 32, 256, 'out_linear')]
    tn_shapes = [(M1, 768, 256, 'qkv wgrad'), (M1, 256, 256, 'out wgrad'), (M1, 1024, 256, 'ffn1 wgrad'), (M1, 256, 1024, 'ffn2 wgrad'),
                 (M2, 768, 256, 'qkv wgrad s2')]
    print(f'{"kernel":8s} {"shape":>26s} {"what":14s} {"ms":>8s} {"TFLOP/s":>8s} {"%peak":>6s}')
    for M, N, K, what in nt_shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        t = time_it(lambda: ops.gemm_nt(a, b, bias=bias, out=out), reps)
        tf = 2.0 * M * N * K / t / 1e12
        print(f'{"gemm_nt":8s} {str((M, N, K)):>26s} {what:14s} {t * 1e3:8.3f} {tf:8.1f} {100 * tf / PEAK:6.1f}')
        del a, b, out
    for M, N, K, what in tn_shapes:
        a = torch.randn(M, N, device=dev)
        b = torch.randn(M, K, device=dev)
        t = time_it(lambda: ops.gemm_tn(a, b), reps)
        tf = 2.0 * M * N * K / t / 1e12
        print(f'{"gemm_tn":8s} {str((M, N, K)):>26s} {what:14s} {t * 1e3:8.3f} {tf:8.1f} {100 * tf / PEAK:6.1f}')
        del a, b


if __name__ == '__main__':
    main()
