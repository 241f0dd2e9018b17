"""Do the output tiles of the persistent f16x3 NT kernel cost less when the 256 workgroups do not store them at the same moment?
Lab build (VQCPC_G3_STAGGER = n: workgroup b starts (b & 7) * n * 512 clocks late).     python tools/bench_g3_stagger.py"""
import os, subprocess, sys
os.environ.setdefault('VQCPC_LAB', '1')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def one():
    import torch
    from vqcpc_bach_amd import hip
    hip.load(); hip.set_gemm_mode(1)
    M = 557056
    for N, K in ((1024, 256), (256, 256), (256, 1024)):
        a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); out = torch.empty(M, N, device='cuda')
        st = torch.zeros(4, device='cuda'); st[0] = 4.0; st[1] = 4.0
        f = lambda: hip.call('vqcpc_gemm_nt_grad', a, K, b, K, out, N, M, N, K, None, 0, None, 0, None, 1.0, st)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f'stagger {os.environ.get("VQCPC_G3_STAGGER", "0"):>3s}  {M} x {N} x {K}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s', flush=True)
        del a, b, out


if __name__ == '__main__':
    if len(sys.argv) > 1:
        one()
    else:
        for v in ('0', '4', '8', '15', '30', '0'):
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, VQCPC_G3_STAGGER=v))
