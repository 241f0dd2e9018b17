"""Round 5: the three-product fp16 gradient GEMMs (csrc/gemm_grad.hip) called directly through the C ABI: error against fp64
on N(0,1) operands, on operands with gradient-like magnitudes / a wide row dynamic range, saturation behaviour of a stale
(too small) previous-step amax, and speed at the C1 backward shapes next to the six-product and bf16-pair kernels.

    python tools/bench_grad_f16.py [quick]        (needs a GPU)
"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load()
hip.set_gemm_mode(1)
QUICK = len(sys.argv) > 1 and sys.argv[1] == 'quick'


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


def state_for(a, b, stale=1.0):
    """scale state of one call site, primed with the operands' own amax (times `stale` for the A operand)"""
    st = torch.zeros(4, device='cuda')
    hip.call('vqcpc_grad_amax', a, a.stride(0), a.shape[0], a.shape[1], st[0:1])
    hip.call('vqcpc_grad_amax', b, b.stride(0), b.shape[0], b.shape[1], st[1:2])
    if stale != 1.0:
        st[0] *= stale
    return st


def nt_grad(a, b, st, add=None, add2=None, mask=None, gate_scale=1.0, out=None):
    M, K = a.shape; N = b.shape[0]
    out = torch.empty(M, N, device='cuda') if out is None else out
    hip.call('vqcpc_gemm_nt_grad', a, a.stride(0), b, b.stride(0), out, out.stride(0), M, N, K, add, 0 if add is None else add.stride(0),
             add2, 0 if add2 is None else add2.stride(0), mask, float(gate_scale), st)
    return out


def tn_grad(a, b, st, want_bias=True):
    M, N = a.shape; K = b.shape[1]
    dw = torch.empty(N, K, device='cuda'); db = torch.empty(N, device='cuda') if want_bias else None
    nb = hip.query('vqcpc_gemm_tn_grad_workspace', M, N, K)
    ws = hip.workspace(nb, a.device)
    hip.call('vqcpc_gemm_tn_grad', a, a.stride(0), b, b.stride(0), dw, db, M, N, K, 0, ws, nb, st)
    return dw, db


gen = torch.Generator(device='cuda').manual_seed(0)
print('== error vs fp64 (max |err| / max |ref|, rms err / rms ref)', flush=True)
for kind, Mv, N, K in (('nt', 65536, 256, 1024), ('nt', 65536, 1024, 256), ('tn', 557056, 256, 256), ('tn', 65536, 1024, 256)):
    for dist in ('N(0,1)', 'grad-like'):
        if kind == 'nt':
            a = torch.randn(Mv, K, device='cuda', generator=gen); b = torch.randn(N, K, device='cuda', generator=gen)
        else:
            a = torch.randn(Mv, N, device='cuda', generator=gen); b = torch.randn(Mv, K, device='cuda', generator=gen)
        if dist == 'grad-like':      # rows of very different magnitude (1e-3 .. 1e-9), a weight-like / activation-like partner
            a = a * torch.exp(torch.empty(Mv, 1, device='cuda').uniform_(-20.7, -6.9, generator=gen))
            b = b * (0.05 if kind == 'nt' else 1.0)
        ref = (a.double() @ b.double().t()) if kind == 'nt' else (a.double().t() @ b.double())
        row = []
        for label, mode, prod in (('fp32 MFMA', 0, 6), ('bf16x6', 1, 6), ('bf16x3', 1, 3)):
            hip.set_gemm_mode(mode); hip.set_gradient_products(prod); hip.gradient_scope(True)
            try:
                out = ops.gemm_nt(a, b) if kind == 'nt' else ops.gemm_tn(a, b, want_bias=False)[0]
                row.append((label, err(out, ref)))
            finally:
                hip.gradient_scope(False)
        hip.set_gemm_mode(1); hip.set_gradient_products(6)
        st = state_for(a, b)
        out = nt_grad(a, b, st) if kind == 'nt' else tn_grad(a, b, st)[0]
        row.append(('f16x3', err(out, ref)))
        # the kernel's own amax must equal the primed one
        torch.cuda.synchronize()
        assert torch.equal(st[0:2], st[2:4]), (st,)
        if kind == 'nt' and dist == 'grad-like':
            # per-ROW relative error: small-magnitude rows must not lose accuracy to the tensor-wide scale
            e_row = ((out.double() - ref).pow(2).mean(1).sqrt() / ref.pow(2).mean(1).sqrt())
            mag = a.abs().amax(1)
            small = mag < mag.max() * 1e-4
            row.append(('f16x3 rows<1e-4 amax: worst row rms', (float(e_row[small].max()) if bool(small.any()) else 0.0, float(e_row.max()))))
        if kind == 'tn':
            db_ref = a.double().sum(0)
            db = tn_grad(a, b, state_for(a, b))[1]
            row.append(('f16x3 bias', err(db, db_ref)))
        print(f'{kind} {Mv} x {N} x {K} {dist:9s}: ' + '   '.join(f'{l}: max {e[0]:.2e} rms {e[1]:.2e}' for l, e in row), flush=True)
        del a, b, ref, out

print('== stale scale: previous-step amax smaller than this step\'s by the factor shown (head-room, then saturation)', flush=True)
a = torch.randn(8192, 256, device='cuda', generator=gen) * 1e-4; b = torch.randn(256, 256, device='cuda', generator=gen) * 0.05
ref = a.double() @ b.double().t()
for stale in (1.0, 0.5, 0.125, 1 / 16, 1 / 32, 1 / 64, 1 / 1024, 4.0, 1024.0, 2.0 ** 20):
    st = state_for(a, b, stale=stale)
    out = nt_grad(a, b, st)
    e = err(out, ref)
    print(f'  previous amax = {stale:g} x actual: max {e[0]:.2e} rms {e[1]:.2e} finite {bool(torch.isfinite(out).all())}', flush=True)

print('== epilogues of the dgrad kernel vs the six-product kernel (max |diff| / max |ref|)', flush=True)
M, N, K = 65536, 1024, 256
a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3; b = torch.randn(N, K, device='cuda', generator=gen) * 0.05
add = torch.randn(M, N, device='cuda', generator=gen) * 1e-4; add2 = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
st = state_for(a, b)
for label, kw in (('none', {}), ('add', dict(add=add)), ('add + add2', dict(add=add, add2=add2))):
    six = ops.gemm_nt(a, b, **kw)
    g3 = nt_grad(a, b, st, **kw)
    print(f'  {label}: {float((six - g3).abs().max() / six.abs().max()):.2e}', flush=True)
h, mask = ops.gemm_nt_relu_mask(torch.randn(M, K, device='cuda', generator=gen), torch.randn(N, K, device='cuda', generator=gen),
                                torch.zeros(N, device='cuda'))
six = ops.gemm_nt_gatebits(a, b, mask, gate_scale=1.25)
g3 = nt_grad(a, b, st, mask=mask, gate_scale=1.25)
print(f'  gate bits: {float((six - g3).abs().max() / six.abs().max()):.2e}   zero pattern equal: {bool(torch.equal(six == 0, g3 == 0))}', flush=True)
del a, b, add, add2, six, g3, h, mask

M = 557056
print('== + residual epilogue: out-of-place (add operand loaded) vs in place (atomic accumulate), M = 557056', flush=True)
for N, K in ((256, 1024), (256, 768), (256, 256)):
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); add = torch.randn(M, N, device='cuda'); out = torch.empty(M, N, device='cuda')
    st = state_for(a, b)
    t_none = timed(lambda: nt_grad(a, b, st, out=out))
    t_add = timed(lambda: nt_grad(a, b, st, add=add, out=out))
    ref = nt_grad(a, b, st, add=add).clone()
    acc = add.clone()
    nt_grad(a, b, st, add=acc, out=acc)
    same = float((acc - ref).abs().max() / ref.abs().max())
    t_acc = timed(lambda: nt_grad(a, b, st, add=acc, out=acc))
    fl = 2.0 * M * N * K / 1e9
    print(f'  {N} x {K}: none {fl / t_none:6.1f}   + add {fl / t_add:6.1f}   in place {fl / t_acc:6.1f} TFLOP/s   (in place vs out of place: max rel diff {same:.1e})', flush=True)
    del a, b, add, out, acc, ref

print('== speed (C1 backward shapes, M = 557056; TFLOP/s = 2 M N K / t)', flush=True)
shapes = (('nt', 1024, 256), ('nt', 256, 1024), ('nt', 256, 256), ('nt', 256, 768), ('tn', 1024, 256), ('tn', 256, 1024), ('tn', 256, 256), ('tn', 768, 256))
for kind, N, K in (shapes[:2] + shapes[4:5] if QUICK else shapes):
    if kind == 'nt':
        a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda'); out = torch.empty(M, N, device='cuda')
        fn = lambda: ops.gemm_nt(a, b, out=out)
        st = state_for(a, b)
        fg = lambda: nt_grad(a, b, st, out=out)
    else:
        a = torch.randn(M, N, device='cuda'); b = torch.randn(M, K, device='cuda')
        fn = lambda: ops.gemm_tn(a, b)
        st = state_for(a, b)
        fg = lambda: tn_grad(a, b, st)
    res = {}
    for prod in (6, 3):
        hip.set_gradient_products(prod); hip.gradient_scope(True)
        try:
            res[prod] = 2.0 * M * N * K / (timed(fn) * 1e-3) / 1e12
        finally:
            hip.gradient_scope(False)
    hip.set_gradient_products(6)
    res['f16'] = 2.0 * M * N * K / (timed(fg) * 1e-3) / 1e12
    print(f'{kind} {M} x {N} x {K}: six {res[6]:6.1f}   bf16x3 {res[3]:6.1f}   f16x3 (new) {res["f16"]:6.1f} TFLOP/s   x{res["f16"] / res[6]:.2f}', flush=True)
    del a, b
