"""Tuning probe for vqcpc_block_table_segsum at the C1 size (557 056 rows x 768 columns): prints ms and GB/s."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqcpc_bach_amd import hip, ops
hip.load()
M, L, vmax, C = 34816 * 16, 16, 57, int(sys.argv[1]) if len(sys.argv) > 1 else 768
g = torch.randn(M, C, device='cuda')
tok = torch.randint(0, vmax, (M,), device='cuda')
out = torch.empty(vmax * L, C, device='cuda')
nb = hip.query('vqcpc_block_table_segsum_workspace', M, L, vmax, C)
ws = hip.workspace(nb, 'cuda')
for _ in range(3):
    hip.call('vqcpc_block_table_segsum', g, tok, out, M, L, vmax, C, ws, nb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    hip.call('vqcpc_block_table_segsum', g, tok, out, M, L, vmax, C, ws, nb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f'variant {os.environ.get("VQCPC_SEGSUM_VARIANT", "0")}: {dt * 1e3:.3f} ms  {M * C * 4 / dt / 1e9:.0f} GB/s')
