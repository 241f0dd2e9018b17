import os, sys, torch
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')   # measurement switches live in the lab build (vqcpc_bach_amd/build.py)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load(); hip.set_gemm_mode(int(os.environ.get('VQCPC_ONE_GEMM_MODE', '1')))
M, N, K = 557056, int(sys.argv[1]), int(sys.argv[2])
a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); bias = torch.randn(N, device='cuda'); out = torch.empty(M, N, device='cuda')
grad = int(os.environ.get('VQCPC_ONE_GEMM_GRAD', '0'))     # 3: the opt-in three-product gradient arithmetic (no bias epilogue)
if grad:
    hip.set_gradient_products(grad); hip.gradient_scope(True)
for _ in range(6):
    if grad:
        ops.gemm_nt(a, b, out=out)
    else:
        ops.gemm_nt(a, b, bias=None if os.environ.get('VQCPC_ONE_GEMM_NOBIAS') else bias, out=out)
torch.cuda.synchronize()
