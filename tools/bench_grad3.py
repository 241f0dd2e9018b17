"""Gradient arithmetic of the bf16x6 mode, 6 vs 3 products (hip.set_gradient_products): speed at the C1 backward shapes and
error against an fp64 product of the same operands, next to the error of the exact-fp32 MFMA mode (= an fp32 GEMM).

    python tools/bench_grad3.py            (needs a GPU)
"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()


def timed(fn, reps=8):
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


def err(x, ref):
    d = (x.double() - ref).abs()
    return float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


M = 557056
print('== speed (C1 backward shapes, M = 557056; TFLOP/s = 2 M N K / t)')
for kind, N, K in (('nt', 1024, 256), ('nt', 256, 1024), ('nt', 256, 256), ('nt', 256, 768), ('tn', 1024, 256), ('tn', 256, 1024), ('tn', 256, 256), ('tn', 768, 256)):
    hip.set_gemm_mode(1)
    if kind == 'nt':
        a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); out = torch.empty(M, N, device='cuda')
        fn = lambda: ops.gemm_nt(a, b, out=out)
    else:
        a = torch.randn(M, N, device='cuda'); b = torch.randn(M, K, device='cuda')
        fn = lambda: ops.gemm_tn(a, b)
    res = {}
    for prod in (6, 3):
        hip.set_gradient_products(prod); hip.gradient_scope(True)
        try:
            res[prod] = 2.0 * M * N * K / (timed(fn) * 1e-3) / 1e12
        finally:
            hip.gradient_scope(False)
    print(f'{kind} {M} x {N} x {K}: 6 products {res[6]:6.1f}   3 products {res[3]:6.1f} TFLOP/s   x{res[3] / res[6]:.2f}', flush=True)
    del a, b

print('== error vs fp64 (max |err| / max |ref|, rms err / rms ref); operands ~ N(0,1)')
gen = torch.Generator(device='cuda').manual_seed(0)
for kind, Mv, N, K in (('nt', 65536, 256, 1024), ('nt', 65536, 1024, 256), ('tn', 557056, 256, 256), ('tn', 65536, 1024, 256)):
    if kind == 'nt':
        a = torch.randn(Mv, K, device='cuda', generator=gen); b = torch.randn(N, K, device='cuda', generator=gen)
        ref = a.double() @ b.double().t()
        fn = lambda: ops.gemm_nt(a, b)
    else:
        a = torch.randn(Mv, N, device='cuda', generator=gen); b = torch.randn(Mv, K, device='cuda', generator=gen)
        ref = a.double().t() @ b.double()
        fn = lambda: ops.gemm_tn(a, b, want_bias=False)[0]
    row = []
    for label, mode, prod in (('fp32 MFMA', 0, 6), ('bf16x6', 1, 6), ('bf16x3 (gradient)', 1, 3), ('bf16 (1 product)', 8, 6)):
        hip.set_gemm_mode(mode); hip.set_gradient_products(prod); hip.gradient_scope(True)
        try:
            row.append((label, err(fn(), ref)))
        finally:
            hip.gradient_scope(False)
    print(f'{kind} {Mv} x {N} x {K}: ' + '   '.join(f'{l}: max {e[0]:.2e} rms {e[1]:.2e}' for l, e in row), flush=True)
    del a, b, ref
hip.set_gemm_mode(0); hip.set_gradient_products(6)
