"""L = 16 attention backward, all-bf16 form, at the configs[4] size (69 632 blocks x 8 heads of 64): the fp32-MFMA contractions
(relattn16_bwd_kernel) against every contraction on the bf16 matrix pipe (relattn16_bwd_mm16_kernel); lab build:
VQCPC_RELATTN16_MM16=0 / 1.     python tools/bench_relattn16_bwd.py [hd]"""
import os, sys, statistics, subprocess
os.environ.setdefault('VQCPC_LAB', '1')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def one(hd):
    import torch
    from vqcpc_bach_amd import hip
    hip.load()
    hip.set_gemm_mode(8)
    torch.manual_seed(0)
    nblk, L, H = 69632, 16, 8
    d = H * hd
    qkv = (torch.randn(nblk * L, 3 * d, device='cuda') * 0.5).bfloat16()
    dctx = (torch.randn(nblk * L, d, device='cuda') * 0.1).bfloat16()
    e1, e2 = torch.randn(H * 16, hd, device='cuda') * 0.3, torch.randn(H * 16, hd, device='cuda') * 0.3
    ctx = torch.empty(nblk * L, d, device='cuda', dtype=torch.bfloat16)
    probs = torch.empty(nblk, H, L, L, device='cuda')
    dqkv = torch.empty(nblk * L, 3 * d, device='cuda', dtype=torch.bfloat16)
    de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
    nbytes = hip.query('vqcpc_relattn_bwd_workspace', nblk, L, H, hd)
    ws = hip.workspace(nbytes, 'cuda')
    hip.call('vqcpc_relattn16_fwd_b16io', qkv, 3 * d, e1, e2, ctx, d, probs, nblk, H, hd, 0.1, 77)
    f = lambda: hip.call('vqcpc_relattn16_bwd_b16io', dctx, d, qkv, 3 * d, probs, e1, e2, dqkv, 3 * d, de1, de2, nblk, H, hd, 0.1, 77,
                         ws, nbytes)
    f(); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 5 * 1e3)
    gb = nblk * H * (4 * 16 * hd * 2 + 1024 + 3 * 16 * hd * 2) / 1e9
    t = statistics.median(ts)
    print(f'hd={hd} MM16={os.environ.get("VQCPC_RELATTN16_MM16", "1")}  {t:8.1f} us  {gb / t * 1e3:.2f} TB/s  checksum {float(dqkv.float().abs().mean()):.6e} '
          f'{float(de1.abs().mean()):.6e}', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2:
        one(int(sys.argv[2]))
    else:
        hd = sys.argv[1] if len(sys.argv) > 1 else '64'
        for v in ('0', '1', '0', '1'):
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child', hd], env=dict(os.environ, VQCPC_RELATTN16_MM16=v))
