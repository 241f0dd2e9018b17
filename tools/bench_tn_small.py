"""wgrad (TN) GEMM + its split-K reduction at the student step's shapes (M = 3072 rows): 256-tile vs 128-tile kernels.
    python tools/bench_tn_small.py [M]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
for N, K in [(2048, 512), (512, 2048), (1536, 512), (512, 512), (256, 512)]:
    a = torch.randn(M, N, device='cuda'); b = torch.randn(M, K, device='cuda')
    res = {}
    for mode in (1, 3):
        hip.set_gemm_mode(mode)
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm_tn(a, b)
            e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[mode] = statistics.median(ts)
    print(f'M={M} N={N} K={K}: 256-tile path {res[1]:.1f} us, 128-tile path {res[3]:.1f} us  ({2.0*M*N*K/res[1]/1e6:.0f} / {2.0*M*N*K/res[3]/1e6:.0f} TFLOP/s incl. reduction)', flush=True)
hip.set_gemm_mode(0)
