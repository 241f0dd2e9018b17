"""A/B of the two LDS images of the bf16x6 weight-gradient (TN) kernel: VQCPC_TN_PQ=1 (quad-row image, two ds_read_b64 per
fragment) vs 0 (row-pair image, two ds_read2_b32); prints TFLOP/s and a digest of the result (must be identical).
    VQCPC_TN_PQ=0 python tools/bench_tn_pq.py; VQCPC_TN_PQ=1 python tools/bench_tn_pq.py"""
import hashlib, os, statistics, sys, torch
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')   # measurement switches live in the lab build (vqcpc_bach_amd/build.py)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load(); hip.set_gemm_mode(1)
M = 557056
torch.manual_seed(0)
for N, K in ((1024, 256), (256, 1024), (256, 256), (768, 256), (512, 256)):
    a = torch.randn(M, N, device='cuda'); b = torch.randn(M, K, device='cuda')
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): dw, db = ops.gemm_tn(a, b)
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(2.0 * M * N * K / (e0.elapsed_time(e1) / 6 * 1e-3) / 1e12)
    h = hashlib.sha1(dw.cpu().numpy().tobytes() + db.cpu().numpy().tobytes()).hexdigest()[:12]
    for prod in (3,):
        hip.set_gradient_products(prod); hip.gradient_scope(True)
        dw3, db3 = ops.gemm_tn(a, b); hip.gradient_scope(False); hip.set_gradient_products(6)
        h3 = hashlib.sha1(dw3.cpu().numpy().tobytes() + db3.cpu().numpy().tobytes()).hexdigest()[:12]
    print(os.environ.get('VQCPC_TN_PQ', '1'), (M, N, K), round(statistics.median(ts), 1), h, 'x3:', h3, flush=True)
    del a, b
