"""Weight-gradient (TN) GEMM at the C1 step's shapes: ping-pong kernel (mode 1) against the lockstep kernel (mode 5),
bitwise comparison of the results.    python tools/bench_tn.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
for M, N, K in [(557056, 1024, 256), (557056, 256, 1024), (557056, 512, 256), (557056, 256, 256), (139264, 1024, 256),
                (139264, 256, 1024), (139264, 768, 256), (139264, 256, 256), (34816, 1024, 256)]:
    a = torch.randn(M, N, device='cuda'); b = torch.randn(M, K, device='cuda')
    res, outs = {}, {}
    for mode in (1, 5):
        hip.set_gemm_mode(mode)
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dw, db = ops.gemm_tn(a, b)
            e1.record(); torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        res[mode] = statistics.median(ts); outs[mode] = (dw.clone(), db.clone())
    same = torch.equal(outs[1][0], outs[5][0]) and torch.equal(outs[1][1], outs[5][1])
    print(f'M={M} N={N} K={K}: ping-pong {res[1]:.1f} us ({2.0*M*N*K/res[1]/1e6:.0f} TFLOP/s), lockstep {res[5]:.1f} us '
          f'({2.0*M*N*K/res[5]/1e6:.0f}); bitwise equal: {same}', flush=True)
hip.set_gemm_mode(0)
