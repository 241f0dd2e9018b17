"""L = 16 attention forward at the C1 size (34 816 blocks x 8 heads of 32): the fp32-MFMA kernel against the bf16x6 variant of its
q . k / q . Erel contractions (lab build: VQCPC_RELATTN16_X6=1).     python tools/bench_relattn16.py"""
import os, sys, statistics, subprocess
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def one():
    import torch
    from vqcpc_bach_amd import hip
    hip.load()
    torch.manual_seed(0)
    nblk, L, H, hd = 34816, 16, 8, 32
    d = H * hd
    qkv = torch.randn(nblk * L, 3 * d, device='cuda') * 0.5
    e1, e2 = torch.randn(H, 16, hd, device='cuda') * 0.3, torch.randn(H, 16, hd, device='cuda') * 0.3
    att = torch.empty(nblk * L, d, device='cuda')
    probs = torch.empty(nblk, H, L, L, device='cuda')
    f = lambda: hip.call('vqcpc_relattn_fwd', qkv, 3 * d, e1, e2, att, d, probs, nblk, L, H, hd, 0.1, 77)
    f(); torch.cuda.synchronize()
    ts = []
    for r in range(6):
        e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1_.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1_) / 10 * 1e3)
    # reference in fp64 on a slice
    n0 = 64
    q = qkv[:n0 * L].double().view(n0, L, 3, H, hd)
    qs, k, v = q[:, :, 0] / hd ** 0.5, q[:, :, 1], q[:, :, 2]
    sc = torch.einsum('nihd,njhd->nhij', qs, k)
    er = torch.cat([e1.double(), e2.double()[:, 1:]], 1)                     # (H, 31, hd)
    qe = torch.einsum('nihd,hxd->nhix', qs, er)
    idx = (torch.arange(L).view(1, L) - torch.arange(L).view(L, 1) + 15).cuda()
    sc = sc + torch.gather(qe, 3, idx.view(1, 1, L, L).expand(n0, H, L, L))
    pr = torch.softmax(sc, -1)
    err = float((probs[:n0].double() - pr).abs().max())
    print(f'X6={os.environ.get("VQCPC_RELATTN16_X6", "0")}  {statistics.median(ts):8.1f} us   max |probs - fp64| on 64 blocks: {err:.2e}', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        one()
    else:
        for x6 in ('0', '1', '0', '1'):
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, VQCPC_RELATTN16_X6=x6))
