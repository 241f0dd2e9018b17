import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
torch.manual_seed(0)
for (M, N, K) in [(256 * 300, 256, 64), (256 * 300, 256, 256), (256*257, 256, 64)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda')
    hip.set_gemm_mode(1); ref = ops.gemm_nt(a, b)
    hip.set_gemm_mode(17); out = ops.gemm_nt(a, b)
    torch.cuda.synchronize()
    bad = (out != ref)
    print(M, N, K, 'mismatch', int(bad.sum()), 'of', bad.numel())
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        tiles = torch.unique(rows // 256)
        print(' bad row tiles:', tiles[:40].tolist(), 'count', len(tiles))
        t0 = int(tiles[0])
        sub = bad[t0 * 256:(t0 + 1) * 256]
        print(' within first bad tile: bad rows', sub.any(1).nonzero().flatten()[:64].tolist())
        print(' bad cols', sub.any(0).nonzero().flatten()[:64].tolist())
        r = int(sub.any(1).nonzero().flatten()[0]); c = int(sub[r].nonzero().flatten()[0])
        print(' sample', float(out[t0*256+r, c]), float(ref[t0*256+r, c]))
        # is the wrong value equal to some other tile's value?
        val = out[t0*256+r, c]
        hit = (ref == val).nonzero()
        print(' value found at', hit[:5].tolist())
hip.set_gemm_mode(0)
