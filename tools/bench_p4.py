"""f16x3 NT products on PRE-SPLIT (P4) operands against the fp32-operand kernel (GPU; `python tools/bench_p4.py`): bit-identity and
time per launch at the C1 shapes.  B (weights) as planes = what vqcpc_weight_planes_many provides once per step; A as planes = what a
producer epilogue would have to write (here made by the same kernel, for timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqcpc_bach_amd import hip  # noqa: E402


def planes_of(mats):
    """[(rows, cols) fp32 matrices] -> (flat, planes, planes_t, amax, offsets) through vqcpc_weight_planes_many."""
    offs, n = [], 0
    for m in mats:
        offs.append(n)
        n += m.numel()
    flat = torch.cat([m.reshape(-1) for m in mats]).contiguous()
    rows, tiles = [], 0
    for m, o in zip(mats, offs):
        rows.append((o, m.shape[0], m.shape[1], tiles))
        tiles += ((m.shape[0] + 31) // 32) * ((m.shape[1] + 31) // 32)
    desc = torch.tensor(rows, dtype=torch.int64).cuda()
    planes, planes_t = torch.empty_like(flat), torch.empty_like(flat)
    amax = torch.zeros(len(mats), device='cuda')
    ws = torch.empty(tiles, device='cuda')
    hip.call('vqcpc_weight_planes_many', flat, desc, len(mats), tiles, amax, planes, planes_t, ws, 4 * tiles)
    return flat, planes, planes_t, amax, offs


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    hip.load()
    hip.set_gemm_mode(1)
    gen = torch.Generator(device='cuda').manual_seed(0)
    for M, N, K, form in ((557056, 1024, 256, 'bias'), (557056, 256, 1024, 'bias'), (557056, 256, 256, 'bias'), (557056, 256, 1024, 'none'),
                          (557056, 768, 256, 'bias'), (139264, 256, 1024, 'accum')):
        a = torch.randn(M, K, device='cuda', generator=gen)
        w = torch.randn(N, K, device='cuda', generator=gen) * 0.05
        bias = torch.randn(N, device='cuda', generator=gen) if form == 'bias' else None
        _, pl, _, amax, offs = planes_of([w, a])
        wp, ap = pl[:N * K].view(N, K), pl[N * K:].view(M, K)
        st = torch.zeros(4, device='cuda')
        st[0], st[1] = amax[1], amax[0]
        out0, out2, out3 = (torch.zeros(M, N, device='cuda') for _ in range(3))

        def ref():
            if form == 'bias':
                hip.call('vqcpc_gemm_nt_f16x3', a, K, w, K, out0, N, M, N, K, bias, 0, 0.0, 0, None, 0, None, st)
            elif form == 'accum':
                hip.call('vqcpc_gemm_nt_grad', a, K, w, K, out0, N, M, N, K, out0, N, None, 0, None, 1.0, st)
            else:
                hip.call('vqcpc_gemm_nt_grad', a, K, w, K, out0, N, M, N, K, None, 0, None, 0, None, 1.0, st)

        def pl_(out, a_, amax_a):
            add = out if form == 'accum' else None
            hip.call('vqcpc_gemm_nt_g3_pl', a_, K, wp, K, out, N, M, N, K, bias, 0, 0.0, 0, add, N if add is not None else 0, None, 0, None,
                     1.0, None, st, amax_a, amax[0:1])
        ref(); pl_(out2, a, None); pl_(out3, ap, amax[1:2])
        torch.cuda.synchronize()
        same2, same3 = torch.equal(out0, out2), torch.equal(out0, out3)
        t0, t2, t3 = timeit(ref), timeit(lambda: pl_(out2, a, None)), timeit(lambda: pl_(out3, ap, amax[1:2]))
        fl = 2.0 * M * N * K / 1e9
        print(f'{M} x {N} x {K} {form:6s}: fp32 operands {t0:.3f} ms ({fl / t0:.0f} TFLOP/s) | B planes {t2:.3f} ms ({fl / t2:.0f}, x{t0 / t2:.3f}, '
              f'bit-identical {same2}) | A and B planes {t3:.3f} ms ({fl / t3:.0f}, x{t0 / t3:.3f}, bit-identical {same3})')


if __name__ == '__main__':
    main()
