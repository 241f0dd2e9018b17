"""Ablations of the three-product fp16 dgrad kernel (gemm_nt_g3_kernel<0, ABL>, lab build): what paces it?
    python tools/ablate_g3.py            (needs a GPU; builds / loads libvqcpc_hip_lab.so)
ABL bits: 1 no operand requests, 2 no split / LDS stores, 4 no MFMAs, 8 no output stores, 16 no fragment reads."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(%r, '..'))
from vqcpc_bach_amd import hip
hip.load(); hip.set_gemm_mode(1)
M = 557056
for N, K in ((1024, 256), (256, 1024), (256, 256)):
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); out = torch.empty(M, N, device='cuda')
    st = torch.zeros(4, device='cuda'); st[0] = 4.0; st[1] = 4.0
    fn = lambda: hip.call('vqcpc_gemm_nt_grad', a, K, b, K, out, N, M, N, K, None, 0, None, 0, None, 1.0, st)
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): fn()
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) / 8)
    print(f'{N}x{K}: {2.0 * M * N * K / (statistics.median(ts) * 1e-3) / 1e12:6.1f}', end='   ')
print()
''' % HERE
LABELS = {32: 'whole-line requests (timing only)', 0: 'as shipped', 1: 'no operand requests', 2: 'no split / LDS stores', 3: 'neither (1 + 2)', 4: 'no MFMAs', 8: 'no output stores',
          16: 'no fragment reads', 19: 'MFMAs + epilogue only (1 + 2 + 16)', 9: 'no requests, no output stores', 11: 'no requests, split, stores'}
for abl in (0, 32, 1, 2, 16, 19, 4, 8):
    env = dict(os.environ, VQCPC_LAB='1', VQCPC_G3_ABL=str(abl))
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=600)
    print(f'ABL {abl:3d} {LABELS[abl]:38s}: {r.stdout.strip() or r.stderr[-400:]}', flush=True)
