"""Micro-benchmark of the GEMM kernels on the shapes of the C1 training step (HIP-event timed, GPU only).
    python tools/bench_gemm.py [reps] [modeA,modeB,...]
Modes are vqcpc_gemm_set_mode() values (1 = bf16x6 with the register-staged ping-pong 256-tile NT kernel, 17 = bf16x6 with
the LDS-DMA kernel, 0 = fp32 MFMA ...); variants are timed interleaved in ONE process (rounds of all modes),
median over rounds, random operands.  Prints algorithmic TFLOP/s per shape / epilogue and mode."""
import os
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
    modes = [int(m) for m in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 17]
    rounds = 5
    hip.load()
    dev = 'cuda'
    M1, M2 = 557056, 139264
    nt_shapes = [(M1, 768, 256, 'bias', 'qkv fwd'), (M1, 256, 256, 'bias', 'out-proj'), (M1, 1024, 256, 'relu_drop', 'ffn1 fwd'),
                 (M1, 256, 1024, 'bias', 'ffn2 fwd'), (M1, 256, 768, 'none', 'qkv dgrad'), (M1, 1024, 256, 'gate', 'ffn2 dgrad'),
                 (M1, 256, 1024, 'add', 'ffn1 dgrad'), (M2, 768, 256, 'bias', 'qkv fwd s2'), (M2, 1024, 256, 'relu_drop', 'ffn1 s2'),
                 (M2, 256, 1024, 'bias', 'ffn2 fwd s2'), (34816, 256, 256, 'bias', 'out-proj s2/4')]
    tn_shapes = [(M1, 768, 256, 'qkv wgrad'), (M1, 256, 256, 'out wgrad'), (M1, 1024, 256, 'ffn1 wgrad'), (M1, 256, 1024, 'ffn2 wgrad'),
                 (M2, 768, 256, 'qkv wgrad s2')]
    print(f'{"kernel":8s} {"shape":>24s} {"what":14s} ' + ' '.join(f'{"mode " + str(m) + " TF":>12s}' for m in modes))
    for M, N, K, epi, what in nt_shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        aux = torch.randn(M, N, device=dev) if epi in ('gate', 'add') else None
        kw = dict(none={}, bias=dict(bias=bias), relu_drop=dict(bias=bias, act=1, drop_p=0.1, seed=3),
                  gate=dict(gate=aux, gate_scale=1.1), add=dict(add=aux))[epi]
        res = {m: [] for m in modes}
        for r in range(rounds + 1):
            for m in modes:
                hip.set_gemm_mode(m)
                t = time_it(lambda: ops.gemm_nt(a, b, out=out, **kw), reps)
                if r:
                    res[m].append(2.0 * M * N * K / t / 1e12)
        print(f'{"gemm_nt":8s} {str((M, N, K)):>24s} {what + "/" + epi:14s} ' +
              ' '.join(f'{statistics.median(res[m]):12.1f}' for m in modes), flush=True)
        del a, b, out, aux
    for M, N, K, what in tn_shapes:
        a = torch.randn(M, N, device=dev)
        b = torch.randn(M, K, device=dev)
        res = {m: [] for m in modes}
        for r in range(rounds + 1):
            for m in modes:
                hip.set_gemm_mode(m)
                t = time_it(lambda: ops.gemm_tn(a, b), reps)
                if r:
                    res[m].append(2.0 * M * N * K / t / 1e12)
        print(f'{"gemm_tn":8s} {str((M, N, K)):>24s} {what:14s} ' + ' '.join(f'{statistics.median(res[m]):12.1f}' for m in modes),
              flush=True)
        del a, b
    hip.set_gemm_mode(0)


if __name__ == '__main__':
    main()
