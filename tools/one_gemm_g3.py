"""One shape of the three-product fp16 gradient GEMMs (csrc/gemm_grad.hip), six launches: the target of rocprofv3 counter passes.
    python tools/one_gemm_g3.py nt|tn N K [p4]        (p4: B of the NT product from pre-split planes, vqcpc_gemm_nt_g3_pl)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip
hip.load(); hip.set_gemm_mode(1)
M, kind, N, K = 557056, sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
st = torch.zeros(4, device='cuda'); st[0] = 4.0; st[1] = 4.0
if kind == 'nt':
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); out = torch.empty(M, N, device='cuda')
    if len(sys.argv) > 4 and sys.argv[4] == 'p4':
        desc = torch.tensor([(0, N, K, 0)], dtype=torch.int64).cuda(); tiles = (N // 32) * (K // 32)
        pl, plt, amax, ws = torch.empty_like(b), torch.empty_like(b), torch.zeros(1, device='cuda'), torch.empty(tiles, device='cuda')
        hip.call('vqcpc_weight_planes_many', b, desc, 1, tiles, amax, pl, plt, ws, 4 * tiles)
        for _ in range(6):
            hip.call('vqcpc_gemm_nt_g3_pl', a, K, pl, K, out, N, M, N, K, None, 0, 0.0, 0, None, 0, None, 0, None, 1.0, None, st, None, amax)
    else:
        for _ in range(6):
            hip.call('vqcpc_gemm_nt_grad', a, K, b, K, out, N, M, N, K, None, 0, None, 0, None, 1.0, st)
else:
    a = torch.randn(M, N, device='cuda'); b = torch.randn(M, K, device='cuda'); dw = torch.empty(N, K, device='cuda'); db = torch.empty(N, device='cuda')
    nb = hip.query('vqcpc_gemm_tn_grad_workspace', M, N, K); ws = hip.workspace(nb, a.device)
    for _ in range(6):
        hip.call('vqcpc_gemm_tn_grad', a, N, b, K, dw, db, M, N, K, 0, ws, nb, st)
torch.cuda.synchronize()
