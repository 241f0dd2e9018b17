"""Per-kernel parity: every libvqcpc_hip.so entry point (called through the C ABI via ctypes) against the CPU oracle on
seeded inputs, and against the reference-generated golden vectors.  Integer outputs are bit-exact; fp32 tolerances are
written next to each assert (relative to the reference tensor's max magnitude)."""
import numpy as np
import pytest
import torch

from conftest import lab_only  # noqa: F401
from conftest import load_golden, rel_err, sub_state
from oracle import vqcpc_oracle as O

pytestmark = pytest.mark.gpu

FWD_TOL = 3e-5
GRAD_TOL = 3e-4
T = torch.from_numpy


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from vqcpc_bach_amd import hip, ops as _ops
    hip.load()
    return _ops


def dev(t):
    return t.detach().to('cuda').contiguous()


# ----------------------------------------------------------------------------------------------------------------
# VQ
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['vq_ncb1', 'vq_ncb2', 'vq_ncb2_d32', 'vq_wide_d64', 'vq_ties', 'vq_l2norm'])
def test_vq_golden(ops, name):
    g = load_golden(name)
    z = T(g['z'])
    shape = z.shape
    D = shape[-1]
    cb = T(g['codebooks'])
    zd = dev(z.reshape(-1, D)).requires_grad_(True)
    cbd = dev(cb).requires_grad_(True)
    zq, idx, loss = ops.VQFn.apply(zd, cbd, float(g['beta']), bool(g['squared']))
    assert idx.dtype == torch.int64
    assert torch.equal(idx.cpu().reshape(g['idx'].shape), T(g['idx'])), 'index assignment must be bit-exact'
    assert torch.equal(zq.detach().cpu().reshape(shape), T(g['zq'])), 'z + (q - z) is exact fp32 arithmetic'
    assert rel_err(loss.detach().cpu().reshape(g['loss'].shape), g['loss']) < FWD_TOL
    gz, gl = dev(T(g['g_zq']).reshape(-1, D)), dev(T(g['g_loss']).reshape(-1))
    ((zq * gz).sum() + (loss * gl).sum()).backward()
    assert rel_err(zd.grad.cpu().reshape(shape), g['dz']) < GRAD_TOL
    assert rel_err(cbd.grad.cpu(), g['dE']) < GRAD_TOL


@pytest.mark.parametrize('R,ncb,K,dsub', [(5000, 2, 512, 16), (3001, 1, 64, 16), (777, 4, 1024, 16), (1500, 1, 32, 3),
                                          (1024, 2, 100, 5), (34816, 2, 512, 16)])
def test_vq_index_bit_exact_vs_oracle(ops, R, ncb, K, dsub):
    gen = torch.Generator().manual_seed(R + K)
    D = ncb * dsub
    z = torch.randn(R, D, generator=gen)
    cb = torch.randn(ncb, K, dsub, generator=gen)
    cb[:, K // 2] = cb[:, K // 3]                      # an exact tie inside every codebook
    z[: K // 4, :dsub] = cb[0, : K // 4]               # inputs sitting exactly on codes
    zq, idx, loss = ops.VQFn.apply(dev(z), dev(cb), 0.25, True)
    ref_idx = O.vq_assign(z, list(cb))
    assert torch.equal(idx.cpu(), ref_idx)
    zq_ref, _, loss_ref = O.vq_forward(z, list(cb), 0.25, True, idx=ref_idx)
    assert torch.equal(zq.cpu(), zq_ref)
    assert rel_err(loss.cpu(), loss_ref) < FWD_TOL


@pytest.mark.parametrize('K,dsub', [(512, 16), (37, 16), (6, 3), (3, 8), (1, 4)])
def test_vq_non_finite_rows_and_tiny_codebooks(ops, K, dsub):
    """Rows that are entirely NaN / +inf select code 0 (every comparison with them is false: torch.argmin on the
    reference's distance row gives 0 as well), finite rows around them are unaffected, and codebooks with fewer codes
    than lanes per row (the four lanes of a row split the codes) still pick the first minimum."""
    gen = torch.Generator().manual_seed(K * 31 + dsub)
    ncb, R = 2, 300
    z = torch.randn(R, ncb * dsub, generator=gen)
    cb = torch.randn(ncb, K, dsub, generator=gen)
    if K > 2:
        cb[:, K - 1] = cb[:, 1]                            # tie between a code of lane 1's list and the last code
    z[7] = float('nan')
    z[64] = float('inf')
    z[130, :dsub] = float('-inf')                         # only the first codebook's sub-vector
    idx = ops.vq_assign(dev(z), dev(cb))
    finite = torch.ones(R, dtype=torch.bool)
    finite[[7, 64, 130]] = False
    ref = O.vq_assign(z[finite], list(cb))
    assert torch.equal(idx.cpu()[finite], ref)
    assert idx[7].tolist() == [0, 0] and idx[64].tolist() == [0, 0] and int(idx[130, 0]) == 0
    assert int(idx[130, 1]) == int(O.vq_assign(z[130:131], list(cb))[0, 1])


def test_vq_backward_vs_oracle(ops):
    gen = torch.Generator().manual_seed(3)
    R, ncb, K, dsub = 2000, 2, 128, 16
    z = torch.randn(R, ncb * dsub, generator=gen)
    cb = torch.randn(ncb, K, dsub, generator=gen)
    gz, gl = torch.randn(R, ncb * dsub, generator=gen), torch.randn(R, generator=gen)
    zc = z.clone().requires_grad_(True)
    cbs = [c.clone().requires_grad_(True) for c in cb]
    zq, _, loss = O.vq_forward(zc, cbs, 0.25, True)
    ((zq * gz).sum() + (loss * gl).sum()).backward()
    zd, cbd = dev(z).requires_grad_(True), dev(cb).requires_grad_(True)
    zq2, _, loss2 = ops.VQFn.apply(zd, cbd, 0.25, True)
    ((zq2 * dev(gz)).sum() + (loss2 * dev(gl)).sum()).backward()
    assert rel_err(zd.grad.cpu(), zc.grad) < GRAD_TOL
    assert rel_err(cbd.grad.cpu(), torch.stack([c.grad for c in cbs])) < GRAD_TOL


# ----------------------------------------------------------------------------------------------------------------
# GEMMs
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(256, 128, 64), (1000, 96, 32), (130, 33, 36), (4096, 768, 256), (64, 1024, 256),
                                   (513, 256, 1024), (7, 5, 4)])
def test_gemm_nt_plain(ops, M, N, K):
    gen = torch.Generator().manual_seed(M + N + K)
    a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias))
    ref = (a.double() @ b.double().t() + bias.double())
    assert rel_err(out.cpu(), ref) < 2e-6 * max(1, K ** 0.5)


@pytest.mark.parametrize('M,N,K,with_add', [(256, 1536, 512, False), (256, 512, 1536, True), (8, 56, 512, False),
                                            (5, 33, 32, True), (1000, 40, 64, False), (32, 32, 4096, True)])
@pytest.mark.parametrize('mode', [0, 1])
def test_gemm_nt_skinny_shapes(ops, M, N, K, with_add, mode):
    """Small-M products (GRU recurrence, per-event heads) take the wave-split-K 32x32 kernel: bias / add epilogues,
    ragged edges, in-place add, both GEMM modes (it computes in exact fp32 MFMA either way)."""
    from vqcpc_bach_amd import hip
    hip.set_gemm_mode(mode)
    try:
        gen = torch.Generator().manual_seed(M + N + K)
        a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
        add = torch.randn(M, N, generator=gen) if with_add else None
        ref = a.double() @ b.double().t() + bias.double() + (add.double() if with_add else 0)
        out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias), add=dev(add) if with_add else None)
        assert rel_err(out.cpu(), ref) < 2e-6 * max(1, K ** 0.5)
        if with_add:                                       # C aliases add
            buf = dev(add)
            ops.gemm_nt(dev(a), dev(b), bias=dev(bias), add=buf, out=buf)
            assert torch.equal(buf, out)
        # strided A rows (every 3rd row of a taller matrix) and strided output
        tall = torch.randn(3 * M, K, generator=gen)
        wide = torch.zeros(M, N + 8, device='cuda')
        ops.gemm_nt(dev(tall)[::3], dev(b), out=wide[:, 4:4 + N])
        assert rel_err(wide[:, 4:4 + N].cpu(), tall[::3].double() @ b.double().t()) < 2e-6 * max(1, K ** 0.5)
        assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + N:].abs().max()) == 0.0
    finally:
        hip.set_gemm_mode(0)


@pytest.fixture()
def bf16_mode(ops):
    from vqcpc_bach_amd import hip
    hip.set_gemm_mode(8)
    assert hip.get_gemm_mode() == 2
    yield
    hip.set_gemm_mode(0)


@pytest.mark.parametrize('M,N,K', [(2048, 256, 128), (4096, 768, 256), (1300, 96, 64), (513, 256, 1024), (2000, 44, 36)])
def test_gemm_bf16_mode_is_a_bf16_product_with_fp32_accumulation(ops, bf16_mode, M, N, K):
    """mode 8 (BASELINE configs[4] names bf16): operands rounded to bf16 (nearest even, as a torch cast), one bf16 MFMA per
    product, fp32 accumulation, fp32 epilogue -> equals the product of the bf16-cast operands up to summation order."""
    gen = torch.Generator().manual_seed(M + N + K)
    a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias), act=1)
    ref = torch.relu(ar @ br.t() + bias.double())
    assert rel_err(out.cpu(), ref) < 2e-6 * max(1, K ** 0.5)
    assert rel_err(out.cpu(), torch.relu(a.double() @ b.double().t() + bias.double())) > 1e-4       # and it IS bf16
    # weight gradient: dW = g^T x, db = column sums (exact fp32 sums of the unrounded g)
    g = torch.randn(M, N, generator=gen)
    dw, db = ops.gemm_tn(dev(g), dev(a))
    assert rel_err(dw.cpu(), g.bfloat16().double().t() @ ar) < 2e-6 * max(1, M ** 0.5)
    assert rel_err(db.cpu(), g.double().sum(0)) < 2e-6 * max(1, M ** 0.5)


@pytest.fixture()
def bf16x6(ops):
    from vqcpc_bach_amd import hip
    hip.set_gemm_mode(1)
    yield
    hip.set_gemm_mode(0)


@pytest.mark.parametrize('M,N,K', [(256, 128, 64), (1000, 96, 32), (130, 33, 36), (4096, 768, 256), (513, 256, 1024), (7, 5, 4)])
def test_gemm_nt_bf16x6_has_fp32_class_accuracy(ops, bf16x6, M, N, K):
    """mode 1: fp32 operands split exactly into 3 bf16 pieces, 6 bf16 MFMAs per product, fp32 accumulation."""
    gen = torch.Generator().manual_seed(M + N + K)
    a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    a[0, :] *= 1e-3                                   # mixed magnitudes inside one contraction
    b[:, 0] *= 1e3
    out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias))
    ref = (a.double() @ b.double().t() + bias.double())
    err_x6 = rel_err(out.cpu(), ref)
    err_f32 = rel_err(a @ b.t() + bias, ref)          # plain fp32 matmul on the host, for scale
    assert err_x6 < 2e-6 * max(1, K ** 0.5), (err_x6, err_f32)
    # element-wise: error relative to the magnitude sum |a|.|b| of each output (the natural fp32 error scale)
    scale = (a.abs().double() @ b.abs().double().t()) + bias.abs().double()
    err_el = float(((out.cpu().double() - ref).abs() / scale).max())
    ref_el = float((((a @ b.t() + bias).double() - ref).abs() / scale).max())      # a host fp32 matmul, same measure
    assert err_el < max(2e-7, 3 * ref_el), (err_el, ref_el)


def test_gemm_bf16x6_componentwise_bound_over_40_orders_of_magnitude(ops, bf16x6):
    """The exact 3-way split keeps the fp32 componentwise error bound |C - AB^T| <= c K 2^-24 (|A| |B|^T) when rows are
    scaled by 10^U(-10, 10) (a plain bf16 GEMM would be off by 2^-8 of that bound); NT (256- and 128-tile) and TN."""
    gen = torch.Generator().manual_seed(31)
    M, N, K = 1024, 512, 256
    sa = 10.0 ** (torch.rand(M, 1, generator=gen) * 20 - 10)
    sb = 10.0 ** (torch.rand(N, 1, generator=gen) * 20 - 10)
    a, b = torch.randn(M, K, generator=gen) * sa, torch.randn(N, K, generator=gen) * sb
    ref = a.double() @ b.double().t()
    bound = a.double().abs() @ b.double().abs().t()
    for rows in (M, 300):                                   # full 256-tiles / ragged 128-tile kernel
        out = ops.gemm_nt(dev(a[:rows]), dev(b))
        assert float(((out.cpu().double() - ref[:rows]).abs() / bound[:rows]).max()) < 4 * K * 2.0 ** -24
    g = torch.randn(M, N, generator=gen) * sa               # TN: dW = g^T a, contraction over the scaled rows
    dw, db = ops.gemm_tn(dev(g), dev(a))
    ref_w, bound_w = g.double().t() @ a.double(), g.double().abs().t() @ a.double().abs()
    assert float(((dw.cpu().double() - ref_w).abs() / bound_w).max()) < 4 * M * 2.0 ** -24
    assert rel_err(db.cpu(), g.double().sum(0)) < 1e-5


def test_gemm_nt_bf16x6_exact_on_bf16_representable_inputs(ops, bf16x6):
    """Inputs with <= 8 significant bits have m = l = 0: the result must equal the exact integer matmul."""
    n = 128
    a = torch.randint(-7, 8, (n, 64)).float()
    b = torch.arange(n * 64, dtype=torch.float32).reshape(n, 64) % 13 - 6      # asymmetric: transpose-detecting
    out = ops.gemm_nt(dev(a), dev(b))
    assert torch.equal(out.cpu(), a @ b.t())
    eye = torch.eye(n)
    bb = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251)
    assert torch.equal(ops.gemm_nt(dev(eye), dev(bb)).cpu(), bb.t().contiguous())


@pytest.mark.parametrize('M,N,K', [(384, 256, 64), (512, 512, 48), (1024, 256, 256)])
def test_gemm_nt_bf16x6_epilogues(ops, bf16x6, M, N, K):
    """(384, 256, .) runs the 128x128-tile kernel, the others the 256x256-tile kernel."""
    gen = torch.Generator().manual_seed(15)
    a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    gate, add = torch.randn(M, N, generator=gen), torch.randn(M, N, generator=gen)
    out = ops.gemm_nt(dev(a), dev(b), gate=dev(gate), gate_scale=1.25)
    assert rel_err(out.cpu(), (a.double() @ b.double().t()) * (gate.double() > 0) * 1.25) < 1e-5
    out = ops.gemm_nt(dev(a), dev(b), add=dev(add))
    assert rel_err(out.cpu(), a.double() @ b.double().t() + add.double()) < 1e-5
    p, seed = 0.3, 99
    out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias), act=1, drop_p=p, seed=seed)
    mask = ops.dropout_mask(M * N, p, seed, 'cuda').cpu().reshape(M, N)
    assert rel_err(out.cpu(), torch.relu(a.double() @ b.double().t() + bias.double()) * mask.double() / (1 - p)) < 1e-5
    big = torch.randn(4 * M, K, generator=gen)          # strided A rows and strided C rows
    dst = torch.zeros(4 * M, N, device='cuda')
    ops.gemm_nt(dev(big)[::4], dev(b), bias=dev(bias), out=dst[::4])
    assert rel_err(dst[::4].cpu(), big[::4].double() @ b.double().t() + bias.double()) < 1e-5
    assert float(dst[1::4].abs().max()) == 0.0


@lab_only
@pytest.mark.parametrize('M,N,K', [(512, 256, 64), (1024, 768, 256), (256 * 300, 256, 128), (2048, 1024, 1024), (256 * 130, 512, 64)])
def test_gemm_nt_bf16x6_dma_kernel_is_bit_identical_to_the_register_staged_kernel(ops, M, N, K):
    """gemm_nt_x6_dma_kernel (operands global -> LDS by DMA, fragments split per wave) against gemm_nt_x6_pp_kernel
    (global -> registers -> split -> LDS): same exact split, same MFMA order per K tile -> bitwise-equal outputs, for every
    epilogue of the 256-tile path, with several output tiles per persistent workgroup (300 / 260 tiles on 256 CUs) and
    strided operand / output rows; and against the fp64 product."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + N + K)
    a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    gate, add = torch.randn(M, N, generator=gen), torch.randn(M, N, generator=gen)
    ad, bd, biasd, gated, addd = dev(a), dev(b), dev(bias), dev(gate), dev(add)
    big = dev(torch.randn(2 * M, K, generator=gen))
    outs = {}
    try:
        for mode in (17, 1):                      # 17: LDS-DMA kernel, 1: register-staged ping-pong kernel
            hip.set_gemm_mode(mode)
            dst = torch.zeros(2 * M, N, device='cuda')
            ops.gemm_nt(big[::2], bd, bias=biasd, out=dst[::2])
            outs[mode] = [ops.gemm_nt(ad, bd), ops.gemm_nt(ad, bd, bias=biasd), ops.gemm_nt(ad, bd, bias=biasd, act=1),
                          ops.gemm_nt(ad, bd, bias=biasd, act=1, drop_p=0.25, seed=5),
                          ops.gemm_nt(ad, bd, gate=gated, gate_scale=1.5), ops.gemm_nt(ad, bd, add=addd), dst]
    finally:
        hip.set_gemm_mode(0)
    for x, y in zip(outs[1], outs[17]):
        assert torch.equal(x, y)
    ref = a.double() @ b.double().t()
    assert rel_err(outs[17][0].cpu(), ref) < 2e-6 * max(1, K ** 0.5)
    assert rel_err(outs[17][5].cpu(), ref + add.double()) < 2e-6 * max(1, K ** 0.5)
    assert float(outs[17][6][1::2].abs().max()) == 0.0


@pytest.fixture(params=[1, 0], ids=['k64-dma', 'k32-pingpong'])
def bf16_nt_variant(request):
    """Both kernels behind vqcpc_gemm_nt_bf16: 1 = K tiles of 64 by LDS-DMA (default, K % 128 == 0), 0 = ping-pong, K tiles of 32."""
    from vqcpc_bach_amd import hip
    hip.load()
    if not hip.is_lab():
        if request.param == 0:
            pytest.skip('the A/B switch between the bf16 kernels is a lab-build entry point (VQCPC_LAB=1)')
        yield request.param                       # the product library always prefers the DMA kernels
        return
    hip.call('vqcpc_gemm_bf16_set_variant', request.param)
    yield request.param
    hip.call('vqcpc_gemm_bf16_set_variant', 1)


@pytest.mark.parametrize('M,N,K', [(512, 256, 64), (1024, 768, 512), (256 * 260, 256, 128), (2048, 512, 2048), (256 * 300, 512, 256)])
def test_gemm_nt_bf16_native_kernel(ops, bf16_nt_variant, M, N, K):
    """vqcpc_gemm_nt_bf16 (bf16 operands in HBM, fp32 accumulation) against the fp64 product of the bf16-rounded operands:
    every epilogue / output combination the training step uses, several output tiles per persistent workgroup."""
    gen = torch.Generator().manual_seed(M + N + K)
    a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    gate, add = torch.randn(M, N, generator=gen), torch.randn(M, N, generator=gen)
    ab, bb = ops.cast_bf16(dev(a)), ops.cast_bf16(dev(b))
    assert ab.dtype == torch.bfloat16 and torch.equal(ab.cpu(), a.bfloat16()), 'cast == torch round-to-nearest-even'
    big = torch.randn(2 * M, K, generator=gen)
    assert torch.equal(ops.cast_bf16(dev(big)[::2]).cpu(), big[::2].bfloat16()), 'row-strided source'
    ref = ab.cpu().double() @ bb.cpu().double().t()
    tol = 2e-6 * max(1, K ** 0.5)
    out = ops.gemm_nt_bf16(ab, bb)
    assert rel_err(out.cpu(), ref) < tol
    out = ops.gemm_nt_bf16(dev(a), dev(b), bias=dev(bias))                       # fp32 operands are cast on the way in
    assert rel_err(out.cpu(), ref + bias.double()) < tol
    p, seed = 0.25, 77
    mask = ops.dropout_mask(M * N, p, seed, 'cuda').cpu().reshape(M, N).double()
    h_ref = torch.relu(ref + bias.double()) * mask / (1 - p)
    h32, h16 = ops.gemm_nt_bf16(ab, bb, bias=dev(bias), act=1, drop_p=p, seed=seed, out_f32=True, out_bf16=True)
    assert rel_err(h32.cpu(), h_ref) < tol
    assert h16.dtype == torch.bfloat16 and torch.equal(h16.cpu(), h32.cpu().bfloat16()), 'bf16 output = rounded fp32 output'
    only16 = ops.gemm_nt_bf16(ab, bb, bias=dev(bias), act=1, out_f32=False, out_bf16=True)
    assert torch.equal(only16.cpu(), ops.gemm_nt_bf16(ab, bb, bias=dev(bias), act=1).cpu().bfloat16())
    g32 = ops.gemm_nt_bf16(ab, bb, gate=dev(gate), gate_scale=1.5)
    assert rel_err(g32.cpu(), ref * (gate.double() > 0) * 1.5) < tol
    gb = dev(gate).bfloat16()
    d32, d16 = ops.gemm_nt_bf16(ab, bb, gate_b=gb, gate_scale=1.5, out_f32=True, out_bf16=True)
    assert rel_err(d32.cpu(), ref * (gb.cpu().double() > 0) * 1.5) < tol and torch.equal(d16.cpu(), d32.cpu().bfloat16())
    res = dev(add)
    ops.gemm_nt_bf16(ab, bb, add=res, out=res)                                   # in-place residual
    assert rel_err(res.cpu(), ref + add.double()) < tol
    # transpose-detecting: A = I against an asymmetric B
    n = 512
    eye, bb2 = torch.eye(n), (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251)
    assert torch.equal(ops.gemm_nt_bf16(dev(eye), dev(bb2)).cpu(), bb2.t().contiguous())


@lab_only
@pytest.mark.parametrize('M,N,K', [(512, 256, 128), (256 * 260, 256, 256), (2048, 512, 2048), (256 * 300, 512, 512)])
def test_gemm_nt_bf16_kernels_are_bit_identical(ops, M, N, K):
    """The LDS-DMA kernel (K tiles of 64) sums every output element over k in the order of the ping-pong kernel (K tiles of
    32): both give the same bits for every epilogue / output form of the training step (persistent workgroups with several
    output tiles each at the larger shapes: tile-boundary handling of the DMA ring)."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + K)
    a, b = ops.cast_bf16(dev(torch.randn(M, K, generator=gen))), ops.cast_bf16(dev(torch.randn(N, K, generator=gen)))
    bias, add = dev(torch.randn(N, generator=gen)), dev(torch.randn(M, N, generator=gen))
    gate, gb = dev(torch.randn(M, N, generator=gen)), dev(torch.randn(M, N, generator=gen)).bfloat16()
    forms = [dict(), dict(out_f32=False, out_bf16=True), dict(bias=bias), dict(bias=bias, out_f32=True, out_bf16=True),
             dict(bias=bias, act=1, out_f32=False, out_bf16=True), dict(bias=bias, act=1, drop_p=0.25, seed=7, out_f32=False, out_bf16=True),
             dict(bias=bias, act=1, drop_p=0.25, seed=7, out_f32=True, out_bf16=True), dict(gate=gate, gate_scale=1.5),
             dict(gate_b=gb, gate_scale=1.5, out_f32=False, out_bf16=True), dict(gate_b=gb, gate_scale=1.5, out_f32=True, out_bf16=True),
             dict(add=add), dict(bias=bias, add=add), dict(bias=bias, drop_p=0.1, seed=3, add=add), dict(bias=bias, out_f32=False, out_bf16=True)]
    try:
        for kw in forms:
            outs = []
            for v in (0, 1):
                hip.call('vqcpc_gemm_bf16_set_variant', v)
                r = ops.gemm_nt_bf16(a, b, **kw)
                outs.append([x.clone() for x in (r if isinstance(r, tuple) else (r,))])
            assert all(torch.equal(x, y) for x, y in zip(*outs)), sorted(kw)
    finally:
        hip.call('vqcpc_gemm_bf16_set_variant', 1)


@pytest.mark.parametrize('M,N,K', [(512, 512, 256), (256 * 260, 512, 2048)])
def test_gemm_nt_bf16_residual_operand_in_bf16(ops, M, N, K):
    """Round 5: the residual of the feed-forward block on the bf16 path is read from the LayerNorm's bf16 output (`add_bf16`; that
    output is then the LayerNorm's ONLY output: vqcpc_add_layernorm_fwd_b16 with y == NULL).  bf16 -> fp32 is exact, so the result
    equals the fp32-residual form on the same values bit for bit; the LayerNorm's bf16 output does not depend on whether the fp32
    one is written."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + K)
    a, b = ops.cast_bf16(dev(torch.randn(M, K, generator=gen))), ops.cast_bf16(dev(torch.randn(N, K, generator=gen)))
    bias = dev(torch.randn(N, generator=gen))
    res_b = dev(torch.randn(M, N, generator=gen)).bfloat16()
    for kw in (dict(bias=bias), dict(bias=bias, drop_p=0.1, seed=3)):
        assert torch.equal(ops.gemm_nt_bf16(a, b, add_b=res_b, **kw), ops.gemm_nt_bf16(a, b, add=res_b.float(), **kw)), sorted(kw)
    s_in = dev(torch.randn(M, N, generator=gen))
    g, be = dev(torch.randn(N, generator=gen)), dev(torch.randn(N, generator=gen))
    outs = []
    for with_f32 in (True, False):
        y = torch.empty(M, N, device='cuda') if with_f32 else None
        yb = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
        hip.call('vqcpc_add_layernorm_fwd_b16', s_in, N, None, g, be, y, yb, mean, rstd, M, N, 1e-5, 0.0, 0)
        outs.append((yb, mean, rstd))
    assert all(torch.equal(x, y) for x, y in zip(*outs))


@pytest.mark.parametrize('M,N,K', [(1024, 512, 512), (4352, 256, 256), (2560, 1024, 128)])
def test_residual_sum_in_bf16_and_layernorm_on_it(ops, M, N, K):
    """Round 5: on the bf16 path the residual sum s = x + dropout(a W^T + b) leaves the GEMM epilogue in bf16 ONLY and the LayerNorm
    kernels read it so (vqcpc_layernorm_fwd_xb16 / _bwd_xb16).  The bf16 output is the rounding of the fp32 output of the same
    epilogue, and the LayerNorm kernels on the bf16 sum equal the fp32-input kernels on the upcast values bit for bit (forward:
    y, bf16 copy, mean, rstd; backward: d_s, the bf16 d_r with the regenerated mask, d_gamma, d_beta)."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + K)
    a, b = ops.cast_bf16(dev(torch.randn(M, K, generator=gen))), ops.cast_bf16(dev(torch.randn(N, K, generator=gen)))
    bias = dev(torch.randn(N, generator=gen))
    res = dev(torch.randn(M, N, generator=gen))
    res_b = res.bfloat16()
    for kw in (dict(bias=bias), dict(bias=bias, drop_p=0.1, seed=3)):
        for r in (dict(add=res), dict(add_b=res_b)):
            f32 = ops.gemm_nt_bf16(a, b, **kw, **r)
            b16 = ops.gemm_nt_bf16(a, b, out_f32=False, out_bf16=True, **kw, **r)
            assert b16.dtype == torch.bfloat16 and torch.equal(b16, f32.bfloat16()), (sorted(kw), sorted(r))
    sb = b16
    sf = sb.float()
    g, be = dev(torch.randn(N, generator=gen)), dev(torch.randn(N, generator=gen))

    def fwd(xb16):
        y, yb = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
        if xb16:
            hip.call('vqcpc_layernorm_fwd_xb16', sb, N, g, be, y, yb, mean, rstd, M, N, 1e-5)
        else:
            hip.call('vqcpc_add_layernorm_fwd_b16', sf, N, None, g, be, y, yb, mean, rstd, M, N, 1e-5, 0.0, 0)
        return y, yb, mean, rstd

    ref, got = fwd(False), fwd(True)
    assert all(torch.equal(x, y) for x, y in zip(ref, got))
    yb_only = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    m2, r2 = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    hip.call('vqcpc_layernorm_fwd_xb16', sb, N, g, be, None, yb_only, m2, r2, M, N, 1e-5)       # the norm1 form: no fp32 output
    assert torch.equal(yb_only, ref[1]) and torch.equal(m2, ref[2])
    _, _, mean, rstd = ref
    dy = dev(torch.randn(M, N, generator=gen))
    nbytes = hip.query('vqcpc_add_layernorm_bwd_workspace', M, N)

    def bwd(xb16, p):
        ds, drb = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        dg, db = torch.empty(N, device='cuda'), torch.empty(N, device='cuda')
        ws = torch.empty(nbytes // 4, device='cuda')
        if xb16:
            hip.call('vqcpc_layernorm_bwd_xb16', dy, sb, N, g, mean, rstd, ds, None, None, drb, dg, db, M, N, p, 5, ws, nbytes)
        else:
            hip.call('vqcpc_add_layernorm_bwd_b16', dy, sf, N, None, g, mean, rstd, ds, None, drb, dg, db, M, N, p, 5, ws, nbytes)
        return ds, drb, dg, db

    for p in (0.0, 0.1):
        ref_b = bwd(False, p)
        assert all(torch.equal(x, y) for x, y in zip(ref_b, bwd(True, p))), p
        # the gradient of the residual branch in bf16 only (d_s == NULL): the rounding of the fp32 one, everything else unchanged
        dsb, drb = (torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(2))
        dg, db = torch.empty(N, device='cuda'), torch.empty(N, device='cuda')
        ws = torch.empty(nbytes // 4, device='cuda')
        hip.call('vqcpc_layernorm_bwd_xb16', dy, sb, N, g, mean, rstd, None, dsb, None, drb, dg, db, M, N, p, 5, ws, nbytes)
        assert torch.equal(dsb, ref_b[0].bfloat16()) and torch.equal(drb, ref_b[1]) and torch.equal(dg, ref_b[2]), p
    a2 = ops.cast_bf16(dev(torch.randn(M, K, generator=gen)))                 # ... and its consumer: dgrad + bf16 residual
    dx32 = ops.gemm_nt_bf16(a2, b, add_b=dsb)
    assert torch.equal(dx32, ops.gemm_nt_bf16(a2, b, add=dsb.float()))
    # the main-stream gradient in bf16 (VQCPC_BF16_GRAD_STREAM): that sum leaves the epilogue in bf16 only == the rounding of the fp32
    # one, and the LayerNorm backward on a bf16 dy (vqcpc_layernorm_bwd_b16io) == the fp32-dy kernel on the upcast values, bit for bit
    dxb = ops.gemm_nt_bf16(a2, b, add_b=dsb, out_f32=False, out_bf16=True)
    assert dxb.dtype == torch.bfloat16 and torch.equal(dxb, dx32.bfloat16())
    dy_saved, dyb = dy, dxb
    dy = dyb.float()
    for p in (0.0, 0.1):
        ref_b = bwd(True, p)
        ds, dsb2, drb = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda', dtype=torch.bfloat16), \
            torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        dg, db = torch.empty(N, device='cuda'), torch.empty(N, device='cuda')
        ws = torch.empty(nbytes // 4, device='cuda')
        hip.call('vqcpc_layernorm_bwd_b16io', dyb, sb, N, g, mean, rstd, ds, dsb2, None, drb, dg, db, M, N, p, 5, ws, nbytes)
        assert all(torch.equal(x, y) for x, y in zip(ref_b, (ds, drb, dg, db))), p
        assert torch.equal(dsb2, ds.bfloat16())
    dy = dy_saved
    with pytest.raises(hip.VqcpcHipError):           # an unaligned bf16 gradient is refused
        hip.call('vqcpc_layernorm_bwd_b16io', dyb.view(-1)[1:1 + (M - 1) * N].view(M - 1, N), sb, N, g, mean, rstd, ds, None, None, drb,
                 dg, db, M - 1, N, 0.0, 5, ws, nbytes)
    with pytest.raises(hip.VqcpcHipError):           # an unaligned bf16 stream is refused, not read
        hip.call('vqcpc_layernorm_fwd_xb16', sb.view(-1)[1:1 + (M - 1) * N].view(M - 1, N), N, g, be, None, yb_only, m2, r2, M - 1, N,
                 1e-5)


@pytest.mark.parametrize('M,N,K', [(256 * 300, 512, 512), (128 * 1024, 256, 1024)])
def test_dma_gemm_kernels_under_memory_contention(ops, M, N, K):
    """Race screen of the two kernels that order their LDS-DMA deliveries with COUNTED `s_waitcnt vmcnt(n)` instead of barriers
    around every transfer (gemm_nt_bf16_k64_kernel, gemm_tn_bf16_tr_kernel; SURVEY.md section 5 "race detection"): 200 launches
    each while a second stream streams 1 GB copies through HBM / L2 (which stretches and reorders the latency of every DMA), bit
    for bit against the result of a quiet launch -- and, on a lab build, against the register-staged kernels, which have no
    counted waits.  A counted wait that is one transfer short shows up as a stale fragment under exactly this load."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + K)
    a, b = ops.cast_bf16(dev(torch.randn(M, K, generator=gen))), ops.cast_bf16(dev(torch.randn(N, K, generator=gen)))
    bias = dev(torch.randn(N, generator=gen))
    g, x = ops.cast_bf16(dev(torch.randn(M, N, generator=gen))), a
    quiet_nt = ops.gemm_nt_bf16(a, b, bias=bias).clone()
    quiet_tn = [t.clone() for t in ops.gemm_tn_bf16(g, x)]
    if hip.is_lab():
        hip.call('vqcpc_gemm_bf16_set_variant', 0)
        try:
            assert torch.equal(ops.gemm_nt_bf16(a, b, bias=bias), quiet_nt), 'DMA vs register-staged NT kernel'
        finally:
            hip.call('vqcpc_gemm_bf16_set_variant', 1)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    src, dst = torch.empty(1 << 28, dtype=torch.float32, device='cuda'), torch.empty(1 << 28, dtype=torch.float32, device='cuda')
    stop = torch.zeros(1, device='cuda')
    bad_nt = bad_tn = 0
    with torch.cuda.stream(side):
        for _ in range(400):                       # ~1.6 TB of copy / add traffic queued beside the GEMMs
            dst.copy_(src, non_blocking=True)
            src.add_(stop)
    for i in range(200):
        if not torch.equal(ops.gemm_nt_bf16(a, b, bias=bias), quiet_nt):
            bad_nt += 1
        dw, db = ops.gemm_tn_bf16(g, x)
        if not (torch.equal(dw, quiet_tn[0]) and torch.equal(db, quiet_tn[1])):
            bad_tn += 1
    torch.cuda.synchronize()
    assert bad_nt == 0 and bad_tn == 0, (bad_nt, bad_tn)


@pytest.mark.parametrize('M,N,K', [(512, 256, 256), (4096, 512, 256), (128 * 700, 256, 768), (33280, 1024, 512), (128, 256, 256),
                                   (128 * 3, 512, 512)])
def test_gemm_tn_bf16_native_kernel(ops, bf16_nt_variant, M, N, K):
    """vqcpc_gemm_tn_bf16: dW = A^T B and db = column sums of A on bf16 operands, against fp64 on the rounded operands;
    plain, accumulating into existing buffers, and A = I-like transpose detection -- for both kernels (variant 1: operands by
    LDS-DMA, fragments by ds_read_b64_tr_b16; variant 0: row pairs interleaved in registers, ds_read_b32 fragments)."""
    gen = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
    ab, bb = ops.cast_bf16(dev(a)), ops.cast_bf16(dev(b))
    ref_w = ab.cpu().double().t() @ bb.cpu().double()
    ref_b = ab.cpu().double().sum(0)
    dw, db = ops.gemm_tn_bf16(ab, bb)
    assert rel_err(dw.cpu(), ref_w) < 2e-6 * max(1, M ** 0.5)
    assert rel_err(db.cpu(), ref_b) < 1e-5
    dw2, db2 = torch.ones(N, K, device='cuda'), torch.ones(N, device='cuda')
    ops.gemm_tn_bf16(dev(a), dev(b), into=(dw2, db2))                   # fp32 operands are cast on the way in; accumulate
    assert rel_err(dw2.cpu(), ref_w + 1) < 2e-6 * max(1, M ** 0.5) and rel_err(db2.cpu(), ref_b + 1) < 1e-5
    # exact and transpose-detecting: one-hot rows of A pick rows of an integer-valued B
    sel = torch.zeros(M, N)
    sel[torch.arange(N) * (M // N), torch.arange(N)] = 1.0
    bi = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 127) - 63
    dw3, _ = ops.gemm_tn_bf16(dev(sel), dev(bi))
    assert torch.equal(dw3.cpu(), bi[torch.arange(N) * (M // N)])


@pytest.mark.parametrize('mode', [1, 8])
def test_grouped_weight_gradients_equal_the_single_launches(ops, mode):
    """vqcpc_gemm_tn_grouped (the deferred small weight gradients of a trainer's backward pass) against one vqcpc_gemm_tn per
    product and against fp64: 70 products of mixed shapes (full and ragged tiles, strided operands, no bias, a row block of a
    larger gradient) incl. a gradient buffer that occurs three times -- accumulated in problem order on top of what the buffers
    held."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(11 + mode)
    shapes = [(3072, 512, 512), (3072, 2048, 512), (3072, 512, 2048), (768, 512, 512), (3072, 1536, 512), (1000, 132, 260),
              (256, 32, 1536), (3072, 512, 512)]
    hip.set_gemm_mode(mode)
    try:
        with torch.no_grad():
            probs = []
            for i in range(70):
                M, N, K = shapes[i % len(shapes)]
                assert hip.query('vqcpc_gemm_tn_groupable', M, N, K)
                a = torch.randn(M, N, generator=gen).cuda()
                b = torch.randn(M, K + 8, generator=gen).cuda()[:, :K] if i % 5 == 0 else torch.randn(M, K, generator=gen).cuda()
                probs.append((a, b, i % 7 != 3))
            shared_w = torch.randn(512, 512, generator=gen).cuda()
            shared_b = torch.randn(512, generator=gen).cuda()
            big = torch.randn(1536 + 512, 512, generator=gen).cuda()         # problem 4's gradient is rows 512.. of this one
            def buffers():
                out = []
                for i, (a, b, wb) in enumerate(probs):
                    N, K = a.shape[1], b.shape[1]
                    if i in (0, 7, 8):                                       # the same (512, 512) gradient three times
                        out.append((shared_w_c, shared_b_c if wb else None))
                    elif i == 4:
                        out.append((big_c[512:], None))
                    else:
                        out.append((torch.full((N, K), 0.5, device='cuda'), torch.full((N,), -1.0, device='cuda') if wb else None))
                return out
            # reference: one launch pair per product
            shared_w_c, shared_b_c, big_c = shared_w.clone(), shared_b.clone(), big.clone()
            ref = buffers()
            real = ops.GROUP_WGRADS
            for (a, b, wb), (dw, db) in zip(probs, ref):
                ops.gemm_tn(a, b, into=(dw, db))
            ref_shared, ref_big = (shared_w_c, shared_b_c), big_c
            # grouped: deferred inside a gradient scope, issued at its end
            shared_w_c, shared_b_c, big_c = shared_w.clone(), shared_b.clone(), big.clone()
            got = buffers()
            with ops.direct_weight_gradients():
                for (a, b, wb), (dw, db) in zip(probs, got):
                    ops.gemm_tn(a, b, into=(dw, db))
                    assert ops.LAST_TN_DEFERRED
                assert float((got[1][0] - 0.5).abs().max()) == 0.0          # nothing has been issued yet
            torch.cuda.synchronize()
            for i, ((dw0, db0), (dw1, db1)) in enumerate(zip(ref, got)):
                scale = float(dw0.abs().max())
                assert float((dw0 - dw1).abs().max()) <= 2e-5 * scale, i
                if db0 is not None:
                    assert float((db0 - db1).abs().max()) <= 2e-5 * float(db0.abs().max()), i
            a0, b0, _ = probs[1]
            w64 = a0.double().t() @ b0.double() + 0.5
            tol = 1e-5 if mode == 1 else 2e-2
            assert rel_err(got[1][0].cpu(), w64.cpu()) < tol
            assert rel_err(got[1][1].cpu(), (a0.double().sum(0) - 1.0).cpu()) < 1e-5
            tri = sum(probs[i][0].double().t() @ probs[i][1].double() for i in (0, 7, 8)) + shared_w.double()
            assert rel_err(shared_w_c.cpu(), tri.cpu()) < tol
            assert torch.equal(big_c[:512], big[:512])                       # rows outside the block untouched
    finally:
        hip.set_gemm_mode(0)


def test_encoder_layer_bf16_native_path_matches_the_rounded_operand_path(ops):
    """hip.set_gemm_mode(8): a layer whose shapes fit the 256-tile bf16 kernel takes bf16 operands from HBM (cast passes,
    bf16 FFN hidden activation); a layer that does not fit rounds fp32 operands inside the 128-tile kernels.  Same
    arithmetic (operands rounded to bf16, fp32 accumulation): outputs and gradients agree to fp32 summation order."""
    from vqcpc_bach_amd import hip
    from vqcpc_bach_amd.transformer.transformer_custom import TransformerEncoderLayerCustom
    torch.manual_seed(3)
    L, H, d, ff, nblk = 16, 8, 256, 512, 256      # M = 4096: enough tiles that the rounded-operand path is not the fp32 skinny kernel
    layer = TransformerEncoderLayerCustom(d_model=d, nhead=H, attention_bias_type='relative_attention', num_channels=1,
                                          num_events=L, dim_feedforward=ff, dropout=0.0).cuda()
    x = torch.randn(nblk * L, d, device='cuda', requires_grad=True)
    g = torch.randn(nblk * L, d, device='cuda')
    res = {}
    real, real_in, real_res = ops.bf16_native, ops.ATT_B16_IN, ops.BF16_RESIDUAL
    try:
        hip.set_gemm_mode(8)
        # 'io': the product's native path (q | k | v and d ctx bf16 into the attention kernels as well, and -- round 5 -- the first
        # LayerNorm's output in bf16 only, residual of the feed-forward block included); True: native GEMMs with fp32 attention
        # inputs and an fp32 residual stream; False: rounded-operand path
        for native in ('io', True, False):
            ops.bf16_native = real if native else (lambda *shapes: False)
            ops.ATT_B16_IN = native == 'io'
            ops.BF16_RESIDUAL = native == 'io'
            assert ops.bf16_native((1024, 512, 256)) == bool(native)
            for p_ in layer.parameters():
                p_.grad = None
            x.grad = None
            y, _ = layer.forward_rows(x)
            (y * g).sum().backward()
            res[native] = [y.detach().clone(), x.grad.clone()] + [p_.grad.clone() for p_ in layer.parameters()]
    finally:
        ops.bf16_native, ops.ATT_B16_IN, ops.BF16_RESIDUAL = real, real_in, real_res
        hip.set_gemm_mode(0)
    for a_, b_ in zip(res[True], res[False]):
        assert rel_err(a_, b_) < 5e-3, rel_err(a_, b_)          # bf16 roundings of near-identical fp32 values may flip
    assert rel_err(res[True][0], res[False][0]) < 2e-4
    # bf16 q | k | v / d ctx: one more rounding (2^-9 relative) of the attention operands, as in any bf16 training stack.  With
    # this test's N(0, 1) activations the logits reach +-10, so 2^-9 on q and k moves single probabilities by a few per cent
    # and the gradients that pass through the softmax follow (5.8 % max-norm measured): bounded here, switched off by
    # VQCPC_ATT_B16_IN=0, and pinned statistically at the model level by test_c4_bf16_mode_vs_bf16_cast_oracle
    for a_, b_ in zip(res['io'], res[False]):
        assert rel_err(a_, b_) < 0.12, rel_err(a_, b_)
    assert rel_err(res['io'][0], res[False][0]) < 2e-2


def test_gemm_nt_bf16x6_256_tile_is_transpose_detecting(ops, bf16x6):
    n = 512
    a = torch.randint(-7, 8, (n, 64)).float()
    b = torch.arange(n * 64, dtype=torch.float32).reshape(n, 64) % 13 - 6
    assert torch.equal(ops.gemm_nt(dev(a), dev(b)).cpu(), a @ b.t())
    eye = torch.eye(n)
    bb = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251)
    assert torch.equal(ops.gemm_nt(dev(eye), dev(bb)).cpu(), bb.t().contiguous())


def test_gemm_nt_is_transpose_detecting(ops):
    # A = I with an ASYMMETRIC B catches a swapped C-write
    n = 128
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    out = ops.gemm_nt(dev(torch.eye(n)), dev(b))
    assert torch.equal(out.cpu(), b.t().contiguous())


def test_gemm_nt_two_residuals_in_place(ops):
    gen = torch.Generator().manual_seed(8)
    M, N, K = 256, 128, 64
    a, b = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen)
    add, base = torch.randn(M, N, generator=gen), torch.randn(4 * M, N, generator=gen)
    based = dev(base)
    view = based[::4]
    ops.gemm_nt(dev(a), dev(b), add=dev(add), add2=view, out=view)            # out aliases add2 element-wise
    ref = a.double() @ b.double().t() + add.double() + base[::4].double()
    assert rel_err(based[::4].cpu(), ref) < 1e-5
    assert torch.equal(based[1::4].cpu(), base[1::4])


@pytest.mark.parametrize('M,N,K', [(1024, 256, 256), (139264, 256, 256), (2560, 512, 64)])
def test_gemm_nt_two_residuals_on_the_256_tile_kernel(ops, M, N, K):
    """bf16x6 mode, full 256-tiles: `add + add2` runs on the ping-pong kernel (second residual fetched per 32 x 32 tile),
    in place over add2 like the trainer's call."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + N)
    a, b = dev(torch.randn(M, K, generator=gen)), dev(torch.randn(N, K, generator=gen))
    add, base = dev(torch.randn(M, N, generator=gen)), dev(torch.randn(M, 2 * N, generator=gen))
    keep = base.clone()
    view = base[:, N:]                                                         # row stride 2 N
    hip.set_gemm_mode(1)
    try:
        ops.gemm_nt(a, b, add=add, add2=view, out=view)
        plain = ops.gemm_nt(a, b, add=add)
    finally:
        hip.set_gemm_mode(0)
    ref = a.double() @ b.double().t() + add.double() + keep[:, N:].double()
    assert rel_err(base[:, N:].cpu(), ref.cpu()) < 2e-6
    assert torch.equal(base[:, :N], keep[:, :N])
    assert rel_err(base[:, N:].cpu(), (plain + keep[:, N:]).cpu()) < 2e-6
    if M >= 65536:
        # same accumulation, one more addend: bit-identical to the single-residual launch + the second residual added
        # afterwards (small shapes run the single-residual form on the skinny fp32-MFMA kernel: other arithmetic)
        assert torch.equal(base[:, N:], plain + keep[:, N:])


def test_gemm_nt_strided_rows_and_epilogue(ops):
    gen = torch.Generator().manual_seed(5)
    M, N, K = 300, 200, 64
    big = torch.randn(4 * M, K, generator=gen)
    a = big[::4]                                    # the [::4] subsample as a row stride
    b, bias = torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    gate, add = torch.randn(M, N, generator=gen), torch.randn(M, N, generator=gen)
    bigd = dev(big)
    out = ops.gemm_nt(bigd[::4], dev(b), bias=dev(bias), act=1, gate=dev(gate), gate_scale=1.25, add=dev(add))
    ref = torch.relu(a.double() @ b.double().t() + bias.double()) * (gate.double() > 0) * 1.25 + add.double()
    assert rel_err(out.cpu(), ref) < 1e-5
    # write into a strided destination
    dst = torch.zeros(4 * M, N, device='cuda')
    ops.gemm_nt(bigd[::4], dev(b), out=dst[::4])
    assert rel_err(dst[::4].cpu(), a.double() @ b.double().t()) < 1e-5
    assert float(dst[1::4].abs().max()) == 0.0


def test_gemm_nt_dropout_epilogue(ops):
    gen = torch.Generator().manual_seed(6)
    M, N, K, p, seed = 257, 130, 32, 0.3, 12345
    a, b = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen)
    out = ops.gemm_nt(dev(a), dev(b), act=1, drop_p=p, seed=seed)
    mask = ops.dropout_mask(M * N, p, seed, 'cuda').cpu().reshape(M, N)
    assert 0.25 < 1 - float(mask.mean()) < 0.35
    ref = torch.relu(a.double() @ b.double().t()) * mask.double() / (1 - p)
    assert rel_err(out.cpu(), ref) < 1e-5


@pytest.mark.parametrize('epi', ['bias_relu_drop', 'gate', 'add_in_place'])
def test_gemm_nt_row_split_launch(ops, bf16x6, epi):
    """300 row tiles x 1 column tile of 256 = 1.17 rounds of the 256-tile kernel: the launch is cut by rows into the full
    round (256-tile kernel) and the remaining 44 tiles (128-tile kernel); epilogue operands, in-place residual and the
    dropout element index must follow the row offset."""
    gen = torch.Generator().manual_seed(8)
    M, N, K, p, seed = 300 * 256, 256, 64, 0.25, 777
    a, b, bias = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(N, generator=gen)
    prod = a.double() @ b.double().t()
    if epi == 'bias_relu_drop':
        out = ops.gemm_nt(dev(a), dev(b), bias=dev(bias), act=1, drop_p=p, seed=seed)
        mask = ops.dropout_mask(M * N, p, seed, 'cuda').cpu().reshape(M, N)
        ref = torch.relu(prod + bias.double()) * mask.double() / (1 - p)
    elif epi == 'gate':
        gate = torch.randn(M, N, generator=gen)
        out = ops.gemm_nt(dev(a), dev(b), gate=dev(gate), gate_scale=1.5)
        ref = prod * (gate > 0).double() * 1.5
    else:
        res = torch.randn(M, N, generator=gen)
        out = dev(res)
        ops.gemm_nt(dev(a), dev(b), add=out, out=out)
        ref = prod + res.double()
    err = (out.cpu().double() - ref).abs()
    assert float(err.max()) < 1e-5 * float(ref.abs().max())
    assert float(err[-44 * 256:].max()) < 1e-5 * float(ref.abs().max())        # the rows of the second launch


@pytest.mark.parametrize('M,N,K', [(1000, 128, 128), (4097, 96, 36), (50000, 768, 256), (333, 4, 256), (20000, 32, 512)])
def test_gemm_tn(ops, M, N, K):
    gen = torch.Generator().manual_seed(M + N)
    a, b = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
    dw, db = ops.gemm_tn(dev(a), dev(b))
    assert rel_err(dw.cpu(), a.double().t() @ b.double()) < 2e-6 * max(1, M ** 0.5)
    assert rel_err(db.cpu(), a.double().sum(0)) < 2e-6 * max(1, M ** 0.5)


@pytest.mark.parametrize('M,N,K', [(1000, 128, 128), (4097, 96, 36), (50000, 768, 256), (333, 4, 256), (20000, 32, 512)])
def test_gemm_tn_bf16x6(ops, bf16x6, M, N, K):
    gen = torch.Generator().manual_seed(M + N)
    a, b = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
    a[:, 0] *= 1e3
    b[0] *= 1e-3
    dw, db = ops.gemm_tn(dev(a), dev(b))
    ref = a.double().t() @ b.double()
    assert rel_err(dw.cpu(), ref) < 2e-6 * max(1, M ** 0.5)
    scale = a.abs().double().t() @ b.abs().double()
    err_el = float(((dw.cpu().double() - ref).abs() / scale).max())
    ref_el = float((((a.t() @ b).double() - ref).abs() / scale).max())
    assert err_el < max(2e-7, 3 * ref_el), (err_el, ref_el)
    assert rel_err(db.cpu(), a.double().sum(0)) < 2e-6 * max(1, M ** 0.5)


def test_gemm_tn_bf16x6_exact_and_transpose_detecting(ops, bf16x6):
    M, N, K = 512, 128, 256
    a = (torch.arange(M * N, dtype=torch.float32).reshape(M, N) % 7) - 3
    b = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 11) - 5
    dw, db = ops.gemm_tn(dev(a), dev(b))
    assert torch.equal(dw.cpu(), a.t() @ b)
    assert torch.equal(db.cpu(), a.sum(0))


def test_gemm_tn_strided_b(ops):
    gen = torch.Generator().manual_seed(9)
    M, N, K = 2000, 64, 32
    a, big = torch.randn(M, N, generator=gen), torch.randn(4 * M, K, generator=gen)
    dw, _ = ops.gemm_tn(dev(a), dev(big)[::4])
    assert rel_err(dw.cpu(), a.double().t() @ big[::4].double()) < 1e-4


def test_transpose(ops):
    w = torch.randn(100, 37)
    assert torch.equal(ops.transpose(dev(w)).cpu(), w.t().contiguous())


# ----------------------------------------------------------------------------------------------------------------
# LayerNorm, embedding
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,d,p', [(1000, 256, 0.0), (77, 32, 0.0), (513, 512, 0.0), (400, 128, 0.2), (64, 1024, 0.0)])
def test_add_layernorm(ops, M, d, p):
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + d)
    big = torch.randn(4 * M, d, generator=gen)
    x, r = big[::4], torch.randn(M, d, generator=gen)
    gamma, beta, dy = torch.randn(d, generator=gen), torch.randn(d, generator=gen), torch.randn(M, d, generator=gen)
    seed = 99
    mask = ops.dropout_mask(M * d, p, seed, 'cuda').cpu().reshape(M, d) / (1 - p)
    xc, rc, gc, bc = (t.clone().requires_grad_(True) for t in (x, r, gamma, beta))
    y_ref = O.layer_norm(xc + rc * mask, gc, bc)
    (y_ref * dy).sum().backward()

    bigd = dev(big)
    y = torch.empty(M, d, device='cuda')
    mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    hip.call('vqcpc_add_layernorm_fwd', bigd[::4], 4 * d, dev(r), dev(gamma), dev(beta), y, mean, rstd, M, d, 1e-5, p, seed)
    assert rel_err(y.cpu(), y_ref.detach()) < FWD_TOL
    ds, dr = torch.empty(M, d, device='cuda'), torch.empty(M, d, device='cuda')
    dg, db = torch.empty(d, device='cuda'), torch.empty(d, device='cuda')
    nbytes = hip.query('vqcpc_add_layernorm_bwd_workspace', M, d)
    ws = hip.workspace(nbytes, 'cuda')
    hip.call('vqcpc_add_layernorm_bwd', dev(dy), bigd[::4], 4 * d, dev(r), dev(gamma), mean, rstd, ds, dr, dg, db, M, d, p,
             seed, ws, nbytes)
    assert rel_err(ds.cpu(), xc.grad) < GRAD_TOL
    assert rel_err(dr.cpu(), rc.grad) < GRAD_TOL
    assert rel_err(dg.cpu(), gc.grad) < GRAD_TOL
    assert rel_err(db.cpu(), bc.grad) < GRAD_TOL


# d = 76 / 72 / 332: the last wave of the backward's column loop owns fewer than 16 columns (d % 64 in [1, 15]) and some of
# them are table columns: the token ids handed out with v_readlane must come from lanes that executed the load
@pytest.mark.parametrize('nblk,d,V', [(300, 256, 57), (17, 32, 12), (1000, 128, 57), (40, 76, 23), (200, 72, 57), (64, 332, 12)])
def test_embed_pos(ops, nblk, d, V):
    gen = torch.Generator().manual_seed(nblk)
    pos, nv, tpb = 8, 4, 16
    dlin = d - 2 * pos
    tokens = torch.randint(0, V, (nblk * tpb,), generator=gen)
    table = torch.randn(nv, V, dlin, generator=gen)
    chan, event = torch.randn(nv, pos, generator=gen), torch.randn(tpb // nv, pos, generator=gen)
    gout = torch.randn(nblk * tpb, d, generator=gen)
    tc, cc, ec = (t.clone().requires_grad_(True) for t in (table, chan, event))
    p = torch.arange(nblk * tpb) % tpb
    ref = torch.cat([tc[p % nv, tokens], cc[p % nv], ec[p // nv]], dim=1)
    (ref * gout).sum().backward()
    td, cd, ed = (dev(t).requires_grad_(True) for t in (table, chan, event))
    out = ops.EmbedPosFn.apply(dev(tokens), td, cd, ed, tpb)
    assert torch.equal(out.detach().cpu(), ref.detach())
    (out * dev(gout)).sum().backward()
    assert rel_err(td.grad.cpu(), tc.grad) < GRAD_TOL
    assert rel_err(cd.grad.cpu(), cc.grad) < GRAD_TOL
    assert rel_err(ed.grad.cpu(), ec.grad) < GRAD_TOL


@pytest.mark.parametrize('nblk,L,vmax,C', [(300, 16, 57, 768), (5000, 16, 57, 768), (33, 4, 12, 96), (1000, 16, 23, 260)])
def test_block_table_gather_and_segment_sum(ops, nblk, L, vmax, C):
    """First-layer QKV lookup: gather is exact; its backward (deterministic segment sum) matches index_add in fp64 and
    is bit-reproducible run to run."""
    gen = torch.Generator().manual_seed(nblk + C)
    table = torch.randn(vmax * L, C, generator=gen)
    tokens = torch.randint(0, vmax, (nblk * L,), generator=gen)
    g = torch.randn(nblk * L, C, generator=gen)
    ids = tokens * L + torch.arange(nblk * L) % L
    td = dev(table).requires_grad_(True)
    out = ops.BlockTableGatherFn.apply(td, tokens.cuda(), L)
    assert torch.equal(out.cpu(), table[ids])
    out.backward(dev(g))
    ref = torch.zeros(vmax * L, C, dtype=torch.float64).index_add_(0, ids, g.double())
    assert rel_err(td.grad.cpu(), ref) < 1e-6
    first = td.grad.clone()
    td.grad = None
    ops.BlockTableGatherFn.apply(td, tokens.cuda(), L).backward(dev(g))
    assert torch.equal(td.grad, first)
    # the bf16-input form (configs[4] bf16 path): the same sums of the upcast values, bit for bit
    from vqcpc_bach_amd import hip
    gb = dev(g).bfloat16()
    nb = hip.query('vqcpc_block_table_segsum_workspace', nblk * L, L, vmax, C)
    outs = []
    for name, gin in (('vqcpc_block_table_segsum', gb.float()), ('vqcpc_block_table_segsum_b16', gb)):
        dt = torch.empty(vmax * L, C, device='cuda')
        ws = hip.workspace(nb, 'cuda')
        hip.call(name, gin, tokens.cuda(), dt, nblk * L, L, vmax, C, ws, nb)
        outs.append(dt.clone())
    assert torch.equal(outs[0], outs[1])


# ----------------------------------------------------------------------------------------------------------------
# relative attention
# ----------------------------------------------------------------------------------------------------------------
def _attn_ref(qkv, e1, e2, L, H, hd, mask=None):
    """Oracle attention core on an (n*L, 3d) qkv matrix -> ctx (n*L, d), probs (n, H, L, L)."""
    d = H * hd
    n = qkv.shape[0] // L
    q, k, v = qkv.reshape(n, L, 3 * d).split(d, dim=-1)
    q = q * (float(hd) ** -0.5)
    q, k, v = (t.reshape(n, L, H, hd).transpose(1, 2) for t in (q, k, v))
    scores = q @ k.transpose(-1, -2) + O.relative_bias(q, e1, e2)
    probs = torch.softmax(scores, dim=-1)
    pd = probs if mask is None else probs * mask
    return (pd @ v).transpose(1, 2).reshape(n * L, d), probs


@pytest.mark.parametrize('n,L,H,hd,p', [(37, 16, 2, 16, 0.0), (64, 16, 8, 32, 0.0), (129, 4, 8, 32, 0.0), (40, 16, 8, 64, 0.0),
                                        (33, 4, 4, 16, 0.0), (50, 16, 4, 32, 0.15), (70, 4, 2, 16, 0.15), (3000, 16, 8, 32, 0.0),
                                        (1, 16, 8, 32, 0.0), (2, 4, 8, 32, 0.1), (5, 16, 3, 32, 0.1), (1, 4, 2, 16, 0.0)])
def test_relattn(ops, n, L, H, hd, p):
    _relattn_case(ops, n, L, H, hd, p)


# general-L strip kernels (student path: teacher L = 384, auxiliary decoder L = 24 / 96); ragged L and hd < 32 included
@pytest.mark.parametrize('n,L,H,hd,p', [(5, 24, 2, 16, 0.0), (3, 96, 8, 64, 0.0), (2, 384, 8, 64, 0.0), (7, 40, 4, 32, 0.0),
                                        (4, 33, 2, 64, 0.0), (9, 1, 2, 16, 0.0), (3, 96, 4, 32, 0.1), (2, 384, 2, 64, 0.1),
                                        (70, 24, 8, 64, 0.1), (2, 200, 1, 16, 0.2)])
def test_relattn_general_L(ops, n, L, H, hd, p):
    _relattn_case(ops, n, L, H, hd, p)


@pytest.mark.parametrize('n,L,H,hd,p', [(37, 16, 2, 16, 0.0), (64, 16, 8, 32, 0.15), (129, 4, 8, 32, 0.0), (40, 16, 8, 64, 0.1)])
def test_relattn_general_kernels_match_small_L_kernels(ops, n, L, H, hd, p):
    from vqcpc_bach_amd import hip
    hip.force_general_attention(True)
    try:
        _relattn_case(ops, n, L, H, hd, p)
    finally:
        hip.force_general_attention(False)


@pytest.mark.parametrize('n,L,H,hd,p', [(300, 16, 8, 32, 0.1), (77, 4, 2, 16, 0.0), (50, 16, 4, 64, 0.0)])
def test_relattn_table_indirection_is_bit_identical_to_gathered_qkv(ops, n, L, H, hd, p):
    """First-layer variant: q | k | v read from the block table through the tokens == the plain kernels on the gathered
    (n L, 3d) copy, bit for bit, forward and backward."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(n + L)
    d, vmax, seed = H * hd, 13, 99
    table = torch.randn(vmax * L, 3 * d, generator=gen).cuda()
    tokens = torch.randint(0, vmax, (n * L,), generator=gen).cuda()
    e1, e2 = torch.randn(H * L, hd, generator=gen).cuda(), torch.randn(H * L, hd, generator=gen).cuda()
    dctx = torch.randn(n * L, d, generator=gen).cuda()
    qkv = ops.BlockTableGatherFn.apply(table, tokens, L)
    outs = []
    for tab in (False, True):
        ctx = torch.empty(n * L, d, device='cuda')
        probs = torch.empty(n, H, L, L, device='cuda')
        dqkv = torch.empty(n * L, 3 * d, device='cuda')
        de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
        nbytes = hip.query('vqcpc_relattn_bwd_workspace', n, L, H, hd)
        ws = hip.workspace(nbytes, 'cuda')
        if tab:
            hip.call('vqcpc_relattn_tab_fwd', table, 3 * d, tokens, e1, e2, ctx, d, probs, n, L, H, hd, p, seed)
            hip.call('vqcpc_relattn_tab_bwd', dctx, d, table, 3 * d, tokens, probs, e1, e2, dqkv, 3 * d, de1, de2, n, L, H, hd, p,
                     seed, ws, nbytes)
        else:
            hip.call('vqcpc_relattn_fwd', qkv, 3 * d, e1, e2, ctx, d, probs, n, L, H, hd, p, seed)
            hip.call('vqcpc_relattn_bwd', dctx, d, qkv, 3 * d, probs, e1, e2, dqkv, 3 * d, de1, de2, n, L, H, hd, p, seed, ws,
                     nbytes)
        outs.append((ctx, probs, dqkv, de1, de2))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _relattn_case(ops, n, L, H, hd, p):
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(n + L + H)
    d = H * hd
    qkv = torch.randn(n * L, 3 * d, generator=gen)
    e1, e2 = torch.randn(H * L, hd, generator=gen), torch.randn(H * L, hd, generator=gen)
    dctx = torch.randn(n * L, d, generator=gen)
    seed = 4242
    mask = None
    if p > 0:
        mask = ops.dropout_mask(n * H * L * L, p, seed, 'cuda').cpu().reshape(n, H, L, L) / (1 - p)
    qc, e1c, e2c = (t.clone().requires_grad_(True) for t in (qkv, e1, e2))
    ctx_ref, probs_ref = _attn_ref(qc, e1c, e2c, L, H, hd, mask)
    (ctx_ref * dctx).sum().backward()

    qd, e1d, e2d = dev(qkv), dev(e1), dev(e2)
    ctx = torch.empty(n * L, d, device='cuda')
    probs = torch.empty(n, H, L, L, device='cuda')
    hip.call('vqcpc_relattn_fwd', qd, 3 * d, e1d, e2d, ctx, d, probs, n, L, H, hd, p, seed)
    assert rel_err(probs.cpu(), probs_ref.detach()) < FWD_TOL
    assert rel_err(ctx.cpu(), ctx_ref.detach()) < FWD_TOL
    dqkv = torch.empty(n * L, 3 * d, device='cuda')
    de1, de2 = torch.empty_like(e1d), torch.empty_like(e2d)
    nbytes = hip.query('vqcpc_relattn_bwd_workspace', n, L, H, hd)
    ws = hip.workspace(nbytes, 'cuda')
    hip.call('vqcpc_relattn_bwd', dev(dctx), d, qd, 3 * d, probs, e1d, e2d, dqkv, 3 * d, de1, de2, n, L, H, hd, p, seed, ws,
             nbytes)
    assert rel_err(dqkv.cpu(), qc.grad) < GRAD_TOL
    assert rel_err(de1.cpu(), e1c.grad) < GRAD_TOL
    assert rel_err(de2.cpu(), e2c.grad) < GRAD_TOL
    assert float(de2.cpu().reshape(H, L, hd)[:, 0].abs().max()) == 0.0   # e2 row 0 is never used (j > i strictly)


@pytest.mark.parametrize('n,L,H,hd,p', [(37, 16, 2, 16, 0.0), (64, 16, 8, 32, 0.0), (129, 4, 8, 32, 0.0), (40, 16, 8, 64, 0.0),
                                        (33, 4, 4, 16, 0.0), (50, 16, 4, 32, 0.15), (70, 4, 2, 16, 0.15), (5000, 16, 8, 32, 0.0),
                                        (4100, 4, 8, 32, 0.0), (1, 16, 8, 32, 0.0), (2, 4, 8, 32, 0.1), (3, 16, 2, 32, 0.1)])
def test_relattn_query_subsampled(ops, n, L, H, hd, p):
    """Last-layer variant: only the queries at positions 0, 4, 8, .. -- must equal the full attention's rows [::4]."""
    from vqcpc_bach_amd import hip
    F = 4
    LQ = L // F
    gen = torch.Generator().manual_seed(n + L + H + 1)
    d = H * hd
    qkv = torch.randn(n * L, 3 * d, generator=gen)
    e1, e2 = torch.randn(H * L, hd, generator=gen), torch.randn(H * L, hd, generator=gen)
    dctx = torch.randn(n * LQ, d, generator=gen)
    seed = 777
    mask_full = None
    if p > 0:
        m = ops.dropout_mask(n * H * LQ * L, p, seed, 'cuda').cpu().reshape(n, H, LQ, L) / (1 - p)
        mask_full = torch.ones(n, H, L, L)
        mask_full[:, :, ::F] = m
    qc, e1c, e2c = (t.clone().requires_grad_(True) for t in (qkv, e1, e2))
    ctx_ref, probs_ref = _attn_ref(qc, e1c, e2c, L, H, hd, mask_full)
    ctx_sel = ctx_ref.reshape(n, L, d)[:, ::F].reshape(n * LQ, d)
    (ctx_sel * dctx).sum().backward()
    g = qc.grad.reshape(n, L, 3 * d)

    q_in = dev(qkv.reshape(n, L, 3 * d)[:, ::F, :d].reshape(n * LQ, d))
    kv_in = dev(qkv[:, d:])
    e1d, e2d = dev(e1), dev(e2)
    ctx = torch.empty(n * LQ, d, device='cuda')
    probs = torch.empty(n, H, LQ, L, device='cuda')
    hip.call('vqcpc_relattn_sub_fwd', q_in, d, kv_in, 2 * d, e1d, e2d, ctx, d, probs, n, L, F, H, hd, p, seed)
    assert rel_err(probs.cpu(), probs_ref.detach()[:, :, ::F]) < FWD_TOL
    assert rel_err(ctx.cpu(), ctx_sel.detach()) < FWD_TOL
    dq = torch.empty(n * LQ, d, device='cuda')
    dkv = torch.empty(n * L, 2 * d, device='cuda')
    de1, de2 = torch.empty_like(e1d), torch.empty_like(e2d)
    nbytes = hip.query('vqcpc_relattn_sub_bwd_workspace', n, L, F, H, hd)
    ws = hip.workspace(nbytes, 'cuda')
    hip.call('vqcpc_relattn_sub_bwd', dev(dctx), d, q_in, d, kv_in, 2 * d, probs, e1d, e2d, dq, d, dkv, 2 * d, de1, de2, n, L,
             F, H, hd, p, seed, ws, nbytes)
    assert rel_err(dq.cpu(), g[:, ::F, :d].reshape(n * LQ, d)) < GRAD_TOL
    assert rel_err(dkv.cpu(), g[:, :, d:].reshape(n * L, 2 * d)) < GRAD_TOL
    assert rel_err(de1.cpu(), e1c.grad) < GRAD_TOL
    assert rel_err(de2.cpu(), e2c.grad) < GRAD_TOL
    assert float(g[:, :, :d].reshape(n, L // F, F, d)[:, :, 1:].abs().max()) == 0.0   # dropped queries: zero gradient


@pytest.mark.parametrize('n,L,H,hd,p', [(300, 16, 8, 64, 0.1), (77, 4, 8, 64, 0.1), (129, 4, 2, 16, 0.0), (50, 16, 4, 32, 0.0),
                                        (64, 16, 2, 16, 0.1)])
def test_relattn_bf16_output_forms_are_the_rounded_fp32_results(ops, n, L, H, hd, p):
    """bf16 training path (configs[4]): ctx / d qkv written as bf16 by the attention kernels == the fp32 kernels' results rounded
    to nearest even (what the removed cast pass produced), bit for bit; everything else (probs, d e1 / d e2) unchanged."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(n + L + hd)
    d, seed = H * hd, 31
    qkv = torch.randn(n * L, 3 * d, generator=gen).cuda()
    e1, e2 = torch.randn(H * L, hd, generator=gen).cuda(), torch.randn(H * L, hd, generator=gen).cuda()
    dctx = torch.randn(n * L, d, generator=gen).cuda()
    assert hip.query('vqcpc_relattn_b16_supported', L, H, hd)
    nbytes = hip.query('vqcpc_relattn_bwd_workspace', n, L, H, hd)
    ws = hip.workspace(nbytes, 'cuda')
    outs = []
    for b16 in (False, True):
        dt = torch.bfloat16 if b16 else torch.float32
        ctx = torch.empty(n * L, d, device='cuda', dtype=dt)
        probs = torch.empty(n, H, L, L, device='cuda')
        dqkv = torch.empty(n * L, 3 * d, device='cuda', dtype=dt)
        de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
        sfx = '_b16' if b16 else ''
        hip.call('vqcpc_relattn_fwd' + sfx, qkv, 3 * d, e1, e2, ctx, d, probs, n, L, H, hd, p, seed)
        hip.call('vqcpc_relattn_bwd' + sfx, dctx, d, qkv, 3 * d, probs, e1, e2, dqkv, 3 * d, de1, de2, n, L, H, hd, p, seed, ws,
                 nbytes)
        outs.append((ctx, probs, dqkv, de1, de2))
    (c0, p0, g0, a0, b0), (c1, p1, g1, a1, b1) = outs
    assert torch.equal(c0.bfloat16(), c1) and torch.equal(g0.bfloat16(), g1)
    assert torch.equal(p0, p1) and torch.equal(a0, a1) and torch.equal(b0, b1)


@pytest.mark.parametrize('n,H,hd,p', [(300, 8, 64, 0.1), (65, 4, 32, 0.0), (33, 2, 16, 0.1)])
def test_relattn16_all_bf16_form_equals_fp32_inputs_holding_the_same_values(ops, n, H, hd, p):
    """q | k | v and d ctx handed over as bf16 (vqcpc_relattn16_*_b16io) == the bf16-output kernels on fp32 tensors that hold
    the same (bf16-representable) values: the conversion on load is exact, so every output is bit-identical."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(n + hd)
    L, d, seed = 16, H * hd, 77
    qkv_b = torch.randn(n * L, 3 * d, generator=gen).cuda().bfloat16()
    dctx_b = torch.randn(n * L, d, generator=gen).cuda().bfloat16()
    e1, e2 = torch.randn(H * L, hd, generator=gen).cuda(), torch.randn(H * L, hd, generator=gen).cuda()
    nbytes = hip.query('vqcpc_relattn_bwd_workspace', n, L, H, hd)
    ws = hip.workspace(nbytes, 'cuda')
    outs = []
    for io in (False, True):
        ctx = torch.empty(n * L, d, device='cuda', dtype=torch.bfloat16)
        probs = torch.empty(n, H, L, L, device='cuda')
        dqkv = torch.empty(n * L, 3 * d, device='cuda', dtype=torch.bfloat16)
        de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
        if io:
            hip.call('vqcpc_relattn16_fwd_b16io', qkv_b, 3 * d, e1, e2, ctx, d, probs, n, H, hd, p, seed)
            hip.call('vqcpc_relattn16_bwd_b16io', dctx_b, d, qkv_b, 3 * d, probs, e1, e2, dqkv, 3 * d, de1, de2, n, H, hd, p, seed,
                     ws, nbytes)
        else:
            qkv, dctx = qkv_b.float(), dctx_b.float()
            hip.call('vqcpc_relattn16_fwd_b16', qkv, 3 * d, None, e1, e2, ctx, d, probs, n, H, hd, p, seed)
            hip.call('vqcpc_relattn16_bwd_b16', dctx, d, qkv, 3 * d, None, probs, e1, e2, dqkv, 3 * d, de1, de2, n, H, hd, p, seed,
                     ws, nbytes)
        outs.append((ctx, probs, dqkv, de1, de2))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    if hd not in (32, 64):
        return
    # outside the exact GEMM mode 0 the all-bf16 backward runs EVERY contraction on the bf16 matrix pipe (relattn16_bwd_mm16_kernel:
    # bf16 operands as loaded, Pd / dS / Erel as two bf16 pieces): other summation orders and 2^-17-relative operand pieces, i.e.
    # the bf16 outputs agree except for roundings that flip by one unit in the last place, the fp32 d Erel to ~1e-5
    ref_dqkv, ref_de1, ref_de2 = outs[1][2], outs[1][3], outs[1][4]
    dqkv = torch.empty(n * L, 3 * d, device='cuda', dtype=torch.bfloat16)
    de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
    hip.set_gemm_mode(8)
    try:
        hip.call('vqcpc_relattn16_bwd_b16io', dctx_b, d, qkv_b, 3 * d, outs[1][1], e1, e2, dqkv, 3 * d, de1, de2, n, H, hd, p, seed,
                 ws, nbytes)
    finally:
        hip.set_gemm_mode(0)
    a, b = dqkv.float(), ref_dqkv.float()
    diff = (a - b).abs()
    assert bool((diff <= 2.0 ** -7 * b.abs() + 3e-5 * b.abs().max()).all()), float(diff.max())     # one bf16 ulp (+ cancellation noise)
    assert float((a != b).float().mean()) < 0.02
    for part in range(3):                                   # d q, d k, d v each: a layout slip in one of them cannot hide in the others
        sl = slice(part * d, (part + 1) * d)
        assert float((a[:, sl] != b[:, sl]).float().mean()) < 0.03, part
    assert rel_err(de1, ref_de1) < 1e-4 and rel_err(de2, ref_de2) < 1e-4, (rel_err(de1, ref_de1), rel_err(de2, ref_de2))


@pytest.mark.parametrize('n,L,H,hd,p', [(300, 16, 8, 64, 0.1), (129, 4, 8, 64, 0.1), (40, 16, 2, 16, 0.0), (33, 4, 4, 32, 0.0),
                                        (50, 16, 4, 32, 0.1)])
def test_relattn_query_subsampled_bf16_output_forms(ops, n, L, H, hd, p):
    """Query-subsampled kernels: ctx and d k | v as bf16 == the rounded fp32 results; d q stays fp32 and is unchanged."""
    from vqcpc_bach_amd import hip
    F = 4
    LQ = L // F
    gen = torch.Generator().manual_seed(n + L + H + 5)
    d, seed = H * hd, 123
    q_in = torch.randn(n * LQ, d, generator=gen).cuda()
    kv_in = torch.randn(n * L, 2 * d, generator=gen).cuda()
    e1, e2 = torch.randn(H * L, hd, generator=gen).cuda(), torch.randn(H * L, hd, generator=gen).cuda()
    dctx = torch.randn(n * LQ, d, generator=gen).cuda()
    assert hip.query('vqcpc_relattn_sub_b16_supported', L, F, H, hd)
    nbytes = hip.query('vqcpc_relattn_sub_bwd_workspace', n, L, F, H, hd)
    ws = hip.workspace(nbytes, 'cuda')
    outs = []
    for b16 in (False, True):
        dt = torch.bfloat16 if b16 else torch.float32
        sfx = '_b16' if b16 else ''
        ctx = torch.empty(n * LQ, d, device='cuda', dtype=dt)
        probs = torch.empty(n, H, LQ, L, device='cuda')
        dq = torch.empty(n * LQ, d, device='cuda')
        dkv = torch.empty(n * L, 2 * d, device='cuda', dtype=dt)
        de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
        hip.call('vqcpc_relattn_sub_fwd' + sfx, q_in, d, kv_in, 2 * d, e1, e2, ctx, d, probs, n, L, F, H, hd, p, seed)
        hip.call('vqcpc_relattn_sub_bwd' + sfx, dctx, d, q_in, d, kv_in, 2 * d, probs, e1, e2, dq, d, dkv, 2 * d, de1, de2, n, L, F,
                 H, hd, p, seed, ws, nbytes)
        outs.append((ctx, probs, dq, dkv, de1, de2))
    (c0, p0, q0, k0, a0, b0), (c1, p1, q1, k1, a1, b1) = outs
    assert torch.equal(c0.bfloat16(), c1) and torch.equal(k0.bfloat16(), k1)
    assert torch.equal(p0, p1) and torch.equal(q0, q1) and torch.equal(a0, a1) and torch.equal(b0, b1)


@pytest.mark.parametrize('L,H,d,ff', [(16, 2, 32, 64), (4, 2, 32, 48), (16, 8, 256, 512)])
def test_encoder_layer_query_stride_equals_full_then_select(ops, L, H, d, ff):
    """EncoderLayerFn(qstride=4) == oracle layer followed by [::4]: outputs, input gradient, parameter gradients."""
    gen = torch.Generator().manual_seed(L + d)
    n = 24
    sdl = {}
    cfg = O.make_cfg(d=d, H=H, layers=[1, 1], ff=ff)
    full = O.init_state(cfg, seed=2)
    pre = 'encoder.downscaler.transformers.0.layers.0.'
    order = ['self_attn.in_proj_weight', 'self_attn.in_proj_bias', 'self_attn.out_proj.weight', 'self_attn.out_proj.bias',
             'self_attn.attn_bias.e1', 'self_attn.attn_bias.e2', 'linear1.weight', 'linear1.bias', 'linear2.weight',
             'linear2.bias', 'norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias']
    for k in order:
        t = full[pre + k].clone()
        if k.endswith('attn_bias.e1') or k.endswith('attn_bias.e2'):
            t = torch.randn(H * L, d // H, generator=gen)
        if t.dim() == 1:
            t = t + 0.1 * torch.randn(t.shape, generator=gen)
        sdl[k] = t
    x = torch.randn(n, L, d, generator=gen)
    gy = torch.randn(n, L // 4, d, generator=gen)
    P = {k: v.clone().requires_grad_(True) for k, v in sdl.items()}
    xc = x.clone().requires_grad_(True)
    y_ref, _ = O.encoder_layer(xc, P, '', H)
    (y_ref[:, ::4] * gy).sum().backward()
    params = [dev(sdl[k]).requires_grad_(True) for k in order]
    xd = dev(x.reshape(n * L, d)).requires_grad_(True)
    y, probs = ops.EncoderLayerFn.apply(xd, L, H, 0.0, 0, 4, None, None, False, *params)
    assert y.shape == (n * L // 4, d)
    assert rel_err(y.detach().cpu(), y_ref.detach()[:, ::4].reshape(-1, d)) < FWD_TOL
    (y * dev(gy.reshape(-1, d))).sum().backward()
    assert rel_err(xd.grad.cpu(), xc.grad.reshape(n * L, d)) < GRAD_TOL
    for k, prm in zip(order, params):
        assert rel_err(prm.grad.cpu(), P[k].grad) < GRAD_TOL, k


@pytest.mark.parametrize('name,L', [('layer_L16', 16), ('layer_L4', 4)])
def test_encoder_layer_golden(ops, name, L):
    """One TransformerEncoderLayerCustom forward/backward against the reference's own output."""
    g = load_golden(name)
    H = int(g['H'])
    sd = sub_state(g, 'sd')
    order = ['self_attn.in_proj_weight', 'self_attn.in_proj_bias', 'self_attn.out_proj.weight', 'self_attn.out_proj.bias',
             'self_attn.attn_bias.e1', 'self_attn.attn_bias.e2', 'linear1.weight', 'linear1.bias', 'linear2.weight',
             'linear2.bias', 'norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias']
    params = [dev(sd[k]).requires_grad_(True) for k in order]
    x = T(g['x']).transpose(0, 1).contiguous()                   # (n, L, d) block-major
    n, _, d = x.shape
    xd = dev(x.reshape(n * L, d)).requires_grad_(True)
    y, probs = ops.EncoderLayerFn.apply(xd, L, H, 0.0, 0, 1, None, None, False, *params)
    y_ref = T(g['y']).transpose(0, 1).reshape(n * L, d)
    assert rel_err(y.detach().cpu(), y_ref) < FWD_TOL
    assert rel_err(probs.cpu(), g['attn']) < FWD_TOL
    gy = T(g['g']).transpose(0, 1).reshape(n * L, d)
    (y * dev(gy)).sum().backward()
    assert rel_err(xd.grad.cpu(), T(g['dx']).transpose(0, 1).reshape(n * L, d)) < GRAD_TOL
    for k, prm in zip(order, params):
        assert rel_err(prm.grad.cpu(), g['grad/' + k]) < GRAD_TOL, k


# ----------------------------------------------------------------------------------------------------------------
# CPC heads, upscaler activation, optimiser
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,K,N,zdim,cdim', [(5, 4, 6, 8, 6), (256, 8, 15, 32, 32), (33, 16, 15, 32, 32)])
def test_nce(ops, B, K, N, zdim, cdim):
    gen = torch.Generator().manual_seed(B + K)
    c, W = torch.randn(B, cdim, generator=gen), torch.randn(zdim, cdim, K, generator=gen) * 0.3
    zp, zn = torch.randn(B, K, zdim, generator=gen), torch.randn(B, N, K, zdim, generator=gen)
    ref_in = [t.clone().requires_grad_(True) for t in (c, W, zp, zn)]
    f_pos, f_neg = O.fks_scores(*ref_in)
    loss_ref = O.nce_loss(f_pos, f_neg)
    loss_ref.backward()
    din = [dev(t).requires_grad_(True) for t in (c, W, zp, zn)]
    loss_b, hits, fp, fn = ops.NCEFn.apply(*din)
    loss = loss_b.mean()
    loss.backward()
    assert rel_err(fp.cpu(), f_pos.detach()) < FWD_TOL and rel_err(fn.cpu(), f_neg.detach()) < FWD_TOL
    assert abs(float(loss) - float(loss_ref)) < FWD_TOL * abs(float(loss_ref))
    assert torch.equal(hits.cpu(), (f_pos > f_neg.max(2)[0]).float())
    for a, b in zip(din, ref_in):
        assert rel_err(a.grad.cpu(), b.grad) < GRAD_TOL


def test_nce_golden(ops):
    g = load_golden('cpc_heads')
    c, W, zr, zn = (dev(T(g[k])) for k in ('c', 'fks_module/W', 'z_right', 'z_neg'))
    loss_b, hits, fp, fn = ops.NCEFn.apply(c, W, zr, zn)
    assert rel_err(fp.cpu(), g['f_pos']) < FWD_TOL and rel_err(fn.cpu(), g['f_neg']) < FWD_TOL
    assert abs(float(loss_b.mean()) - float(g['loss'])) < FWD_TOL * abs(float(g['loss']))
    assert torch.equal(hits.mean(0).cpu(), T(g['acc']))


@pytest.mark.parametrize('B,T,inp,hid,layers,p', [(5, 3, 8, 12, 2, 0.0), (256, 8, 32, 512, 2, 0.0), (7, 1, 16, 32, 1, 0.0),
                                                  (33, 4, 32, 64, 3, 0.3), (40, 5, 32, 128, 2, 0.2), (64, 2, 16, 192, 2, 0.0)])
def test_gru_context_network(ops, B, T, inp, hid, layers, p):
    """CModule on the library's own GRU (GEMMs + gate kernels; hid % 64 == 0: one fused launch per step, csrc/gru.hip)
    against the oracle's explicit recurrence, forward and every gradient; with p > 0 the inter-layer dropout masks are
    reproduced through vqcpc_dropout_mask."""
    from vqcpc_bach_amd.utils import SEEDS
    from vqcpc_bach_amd.vqcpc_helper import CModule
    gen = torch.Generator().manual_seed(B + T + hid)
    cm = CModule(input_dim=inp, hidden_size=hid, output_dim=6, num_layers=layers, dropout=p)
    P = {'c.' + k: v.detach().clone().requires_grad_(True) for k, v in cm.state_dict().items()}
    zs = torch.randn(B, T, inp, generator=gen)
    gout = torch.randn(B, 6, generator=gen)
    SEEDS.manual_seed(4321)
    seeds = []
    if p > 0:                         # one seed per layer but the last, drawn in forward order
        probe = type(SEEDS)(4321)
        seeds = [probe.next() for _ in range(layers - 1)]

    # oracle with the kernel's masks: layer l output (B, T, hid) is dropped with index (t*B + b)*hid + c
    class MaskGen:
        def __init__(self):
            self.l = 0
    def oracle_forward(zs_):
        x = zs_
        for l in range(layers):
            w_ih, w_hh = P[f'c.g_ar_fwd.weight_ih_l{l}'], P[f'c.g_ar_fwd.weight_hh_l{l}']
            b_ih, b_hh = P[f'c.g_ar_fwd.bias_ih_l{l}'], P[f'c.g_ar_fwd.bias_hh_l{l}']
            h = torch.zeros(B, hid)
            outs = []
            for t in range(T):
                gi = O.linear(x[:, t], w_ih, b_ih)
                gh = O.linear(h, w_hh, b_hh)
                r = torch.sigmoid(gi[:, :hid] + gh[:, :hid])
                u = torch.sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
                n = torch.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
                h = (1 - u) * n + u * h
                outs.append(h)
            x = torch.stack(outs, dim=1)
            if l < layers - 1 and p > 0:
                mask = ops.dropout_mask(T * B * hid, p, seeds[l], 'cuda').cpu().view(T, B, hid).transpose(0, 1) / (1 - p)
                x = x * mask
        return O.linear(x[:, -1], P['c.output_linear.weight'], P['c.output_linear.bias'])

    ref = oracle_forward(zs.clone().requires_grad_(True))
    zr = zs.clone().requires_grad_(True)
    ref = oracle_forward(zr)
    (ref * gout).sum().backward()
    if p == 0.0:                      # and the oracle's own packaged version agrees
        assert rel_err(O.gru_context(zs, {k: v.detach() for k, v in P.items()}, 'c.', layers), ref.detach()) < 1e-6

    cm = cm.cuda().train()
    zd = zs.clone().cuda().requires_grad_(True)
    out = cm(zd, None)
    assert rel_err(out.cpu(), ref.detach()) < FWD_TOL
    (out * gout.cuda()).sum().backward()
    assert rel_err(zd.grad.cpu(), zr.grad) < GRAD_TOL
    for n_, prm in cm.named_parameters():
        assert rel_err(prm.grad.cpu(), P['c.' + n_].grad) < GRAD_TOL, n_


@pytest.mark.parametrize('p', [0.0, 0.25])
def test_dropout_selu(ops, p):
    gen = torch.Generator().manual_seed(11)
    h, g = torch.randn(5000, generator=gen) * 2, torch.randn(5000, generator=gen)
    seed = 31337
    mask = ops.dropout_mask(5000, p, seed, 'cuda').cpu() / (1 - p)
    hc = h.clone().requires_grad_(True)
    ref = O.selu(hc * mask)
    (ref * g).sum().backward()
    hd_ = dev(h).requires_grad_(True)
    out = ops.DropoutSeluFn.apply(hd_, p, seed)
    (out * dev(g)).sum().backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < FWD_TOL
    assert rel_err(hd_.grad.cpu(), hc.grad) < GRAD_TOL


@pytest.mark.parametrize('scale', [1.0, 300.0])
def test_flat_adam_with_clip(ops, scale):
    gen = torch.Generator().manual_seed(12)
    n = 100003
    p0, g0 = torch.randn(n, generator=gen), torch.randn(n, generator=gen) * scale / (n ** 0.5)
    P = {'w': p0.clone()}
    grads = {'w': g0.clone()}
    state = {}
    for _ in range(3):
        gg = {'w': grads['w'].clone()}
        total = O.clip_grad_norm(list(gg.values()), 5.0)
        O.adam_step(P, gg, state, 1e-3)
    pd, gd = dev(p0), dev(g0)
    opt = ops.FlatAdam(pd, gd, lr=1e-3)
    for _ in range(3):
        gd.copy_(dev(g0))
        opt.step()
    assert abs(opt.grad_norm() - float(total)) < 1e-5 * float(total)
    assert (scale > 100) == (float(total) > 5.0)
    assert rel_err(pd.cpu(), P['w']) < 1e-6


# ----------------------------------------------------------------------------------------------------------------
# student-step kernels
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('R,V', [(8, 56), (3, 5), (100, 130), (1, 1)])
def test_softmax_ce_hard_and_soft_targets(ops, R, V):
    gen = torch.Generator().manual_seed(R + V)
    logits = (torch.randn(R, V, generator=gen) * 3)
    teacher = torch.randn(R, V, generator=gen) * 2
    target = torch.randint(0, V, (R,), generator=gen)
    w = torch.randn(R, generator=gen)
    for soft in (False, True):
        lc = logits.clone().requires_grad_(True)
        lp = torch.log_softmax(lc, dim=1)
        ref = -(torch.softmax(teacher, 1) * lp).sum(1) if soft else -lp.gather(1, target.view(-1, 1)).squeeze(1)
        (ref * w).sum().backward()
        ld = dev(logits).requires_grad_(True)
        out = ops.SoftmaxCEFn.apply(ld, None if soft else target.cuda(), dev(teacher) if soft else None)
        assert rel_err(out.cpu(), ref.detach()) < FWD_TOL
        (out * w.cuda()).sum().backward()
        assert rel_err(ld.grad.cpu(), lc.grad) < GRAD_TOL
    # strided logit rows (the masked event of a (B, E, V) tensor)
    big = torch.randn(R, 7, V, generator=gen)
    out = ops.SoftmaxCEFn.apply(dev(big)[:, 3], target.cuda(), None)
    ref = -torch.log_softmax(big[:, 3], 1).gather(1, target.view(-1, 1)).squeeze(1)
    assert rel_err(out.cpu(), ref) < FWD_TOL


@pytest.mark.parametrize('rows,f,d', [(24, 4, 512), (7, 3, 32), (1000, 8, 64), (1, 1, 4)])
def test_upscale(ops, rows, f, d):
    gen = torch.Generator().manual_seed(rows + f)
    x, emb, g = torch.randn(rows, d, generator=gen), torch.randn(f, d, generator=gen), torch.randn(rows * f, d, generator=gen)
    xd, ed = dev(x).requires_grad_(True), dev(emb).requires_grad_(True)
    out = ops.UpscaleFn.apply(xd, ed)
    ref = (x.unsqueeze(1) + emb.unsqueeze(0)).reshape(rows * f, d)            # repeat_interleave + tiled embeddings
    assert torch.equal(out.cpu(), ref)
    out.backward(dev(g))
    assert rel_err(xd.grad.cpu(), g.view(rows, f, d).sum(1)) < 1e-6
    assert rel_err(ed.grad.cpu(), g.view(rows, f, d).double().sum(0)) < 1e-5


def test_embed_pos_without_event_part(ops):
    """Teacher input rows: [table(voice, token) | channel embedding] (teacher_relative.py:63-75)."""
    gen = torch.Generator().manual_seed(9)
    nv, vmax, dlin, pos, rows = 4, 13, 24, 8, 4 * 50
    table = torch.randn(nv, vmax, dlin, generator=gen)
    chan = torch.randn(nv, pos, generator=gen)
    tokens = torch.randint(0, vmax, (rows,), generator=gen)
    g = torch.randn(rows, dlin + pos, generator=gen)
    td, cd = dev(table).requires_grad_(True), dev(chan).requires_grad_(True)
    out = ops.EmbedPosFn.apply(tokens.cuda(), td, cd, None, nv)
    v = torch.arange(rows) % nv
    tr, cr = table.clone().requires_grad_(True), chan.clone().requires_grad_(True)
    ref = torch.cat([tr[v, tokens], cr[v]], dim=1)
    assert torch.equal(out.cpu(), ref.detach())
    out.backward(dev(g))
    ref.backward(g)
    assert rel_err(td.grad.cpu(), tr.grad) < 1e-5 and rel_err(cd.grad.cpu(), cr.grad) < 1e-5


def test_empty_inputs_are_no_ops(ops):
    """Zero rows / zero blocks: every forward entry point returns success without touching memory."""
    from vqcpc_bach_amd import hip
    z = torch.empty(0, 16, device='cuda')
    w = torch.randn(8, 16, device='cuda')
    assert ops.gemm_nt(z, w).shape == (0, 8)
    cb = torch.randn(1, 4, 16, device='cuda')
    hip.call('vqcpc_vq_fwd', z, cb, 0, 1, 4, 16, 0.25, 1, 1, torch.empty(0, 1, dtype=torch.int64, device='cuda'),
             torch.empty(0, 16, device='cuda'), torch.empty(0, device='cuda'))
    e = torch.randn(2 * 16, 8, device='cuda')
    hip.call('vqcpc_relattn_fwd', torch.empty(0, 48, device='cuda'), 48, e, e, torch.empty(0, 16, device='cuda'), 16,
             torch.empty(0, 2, 16, 16, device='cuda'), 0, 16, 2, 8, 0.0, 0)
    hip.call('vqcpc_softmax_ce', torch.empty(0, 5, device='cuda'), 5, torch.empty(0, dtype=torch.int64, device='cuda'), None, 0,
             torch.empty(0, device='cuda'), torch.empty(0, 5, device='cuda'), 0, 5)
    tok = torch.empty(0, dtype=torch.int64, device='cuda')
    hip.call('vqcpc_same_sequence_negatives', tok, tok, tok, 0, 2, 2, 16)
    torch.cuda.synchronize()


def test_bad_arguments_raise_with_the_library_message(ops):
    from vqcpc_bach_amd import hip
    a = torch.randn(4, 6, device='cuda')                   # K = 6 is not a multiple of 4
    with pytest.raises(hip.VqcpcHipError, match='gemm_nt'):
        hip.call('vqcpc_gemm_nt', a, 6, a, 6, torch.empty(4, 4, device='cuda'), 4, 4, 4, 6, None, 0, 0.0, 0, None, 0, 1.0, None,
                 0, None, 0)
    e = torch.randn(2 * 2000, 16, device='cuda')
    with pytest.raises(hip.VqcpcHipError, match='unsupported L'):
        hip.call('vqcpc_relattn_fwd', torch.empty(2000, 96, device='cuda'), 96, e, e, torch.empty(2000, 32, device='cuda'), 32,
                 torch.empty(1, device='cuda'), 1, 2000, 2, 16, 0.0, 0)
    with pytest.raises(hip.VqcpcHipError, match='upscale'):
        g = torch.randn(9 * 4, 8, device='cuda')
        hip.call('vqcpc_upscale_bwd', g, torch.empty(4, 8, device='cuda'), torch.empty(9, 8, device='cuda'), 4, 9, 8,
                 torch.empty(16, device='cuda'), 16)


def test_grouped_weight_gradient_argument_checks(ops, bf16x6):
    """vqcpc_gemm_tn_grouped refuses what it cannot serve (a product of the 256-tile class, a short workspace, the exact fp32
    mode) with the library's message instead of computing something else; n = 0 is a no-op."""
    import ctypes
    from vqcpc_bach_amd import hip
    a, b = torch.randn(3072, 512, device='cuda'), torch.randn(3072, 512, device='cuda')
    dw = torch.zeros(512, 512, device='cuda')

    def call(M, N, K, ws_bytes=None):
        vp, i64, i32 = ctypes.c_void_p * 1, ctypes.c_int64 * 1, ctypes.c_int * 1
        Ms, Ns, Ks = i64(M), i32(N), i32(K)
        need = hip.query('vqcpc_gemm_tn_grouped_workspace', 1, Ms, Ns, Ks)
        ws = hip.workspace(need, 'cuda')
        hip.call('vqcpc_gemm_tn_grouped', 1, vp(a.data_ptr()), i64(N), vp(b.data_ptr()), i64(K), vp(dw.data_ptr()), vp(None), Ms, Ns,
                 Ks, 0, ws, need if ws_bytes is None else ws_bytes)

    assert hip.query('vqcpc_gemm_tn_groupable', 3072, 512, 512) == 1
    assert hip.query('vqcpc_gemm_tn_groupable', 557056, 1024, 256) == 0        # fills the chip alone: 256-tile kernel
    assert hip.query('vqcpc_gemm_tn_groupable', 3072, 510, 512) == 0           # N % 4
    call(3072, 512, 512)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), (a.double().t() @ b.double()).cpu()) < 1e-5
    with pytest.raises(hip.VqcpcHipError, match='workspace too small'):
        call(3072, 512, 512, ws_bytes=16)
    with pytest.raises(hip.VqcpcHipError, match='not groupable'):
        call(557056, 1024, 256)
    hip.call('vqcpc_gemm_tn_grouped', 0, None, None, None, None, None, None, None, None, None, 0, None, 0)
    hip.set_gemm_mode(0)
    try:
        assert hip.query('vqcpc_gemm_tn_groupable', 3072, 512, 512) == 0       # exact fp32 MFMA mode: single launches only
        with ops.direct_weight_gradients():
            ops.gemm_tn(a, b, into=(dw, None))
            assert not ops.LAST_TN_DEFERRED
    finally:
        hip.set_gemm_mode(1)


# ----------------------------------------------------------------------------------------------------------------
# bf16x6 GEMM on pre-split (P3) operands: csrc/gemm_planes.hip (lab build only)
# ----------------------------------------------------------------------------------------------------------------
@lab_only
def test_split3_planes_round_trip_is_exact(ops):
    gen = torch.Generator().manual_seed(11)
    for rows, cols in [(256, 64), (1000, 256), (37, 1024), (16, 16)]:
        x = torch.randn(rows, cols, generator=gen) * torch.logspace(-20, 20, rows).reshape(-1, 1)
        x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, -3.4e38])               # (pieces below 2^-126 would be flushed by the VALU)
        xd = dev(x)
        pl = ops.split3_planes(xd)
        assert bool((ops.join3_planes(pl, rows, cols) == xd).all()), 'high + mid + low must reproduce every fp32 value'
        wide = dev(torch.randn(rows, cols + 32, generator=gen))              # strided source rows
        assert torch.equal(ops.join3_planes(ops.split3_planes(wide[:, 16:16 + cols]), rows, cols), wide[:, 16:16 + cols])


@lab_only
@pytest.mark.parametrize('M,N,K,epi', [(512, 256, 64, 'bias'), (2048, 768, 256, 'bias'), (1536, 256, 1024, 'none'),
                                       (4096, 1024, 256, 'relu_drop'), (2560, 1024, 256, 'gate'), (1280, 256, 1024, 'add'),
                                       (65536 + 256, 512, 160, 'bias')])
def test_gemm_nt_planes_is_the_x6_gemm(ops, M, N, K, epi):
    """Same products as the fp32-in bf16x6 GEMM (mode 1); bit-identical where that one runs its 256-tile ping-pong kernel
    (same MFMA order), within fp32 rounding of the accumulation order otherwise (small shapes take its 128-tile kernel).
    The large-M case runs many persistent rounds (the run-ahead LDS-DMA crosses output tiles)."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + N + K)
    a, b = dev(torch.randn(M, K, generator=gen)), dev(torch.randn(N, K, generator=gen))
    bias = dev(torch.randn(N, generator=gen))
    aux = dev(torch.randn(M, N, generator=gen))
    kw = dict(none={}, bias=dict(bias=bias), relu_drop=dict(bias=bias, act=1, drop_p=0.1, seed=77),
              gate=dict(gate=aux, gate_scale=1.25), add=dict(add=aux))[epi]
    hip.set_gemm_mode(1)
    try:
        ref = ops.gemm_nt(a, b, **kw)
    finally:
        hip.set_gemm_mode(0)
    out = ops.gemm_nt_planes(ops.split3_planes(a), ops.split3_planes(b), M, N, K, **kw)
    if M >= 65536 and epi != 'gate':                      # whole rounds of 256 tiles: rows [0, 65536) at N = 512
        assert torch.equal(out[:65536], ref[:65536])
    assert rel_err(out.cpu(), ref.cpu()) < 2e-6
    exact = a.double() @ b.double().t()
    if epi in ('none',):
        assert rel_err(out.cpu(), exact.cpu()) < 2e-6 * max(1, K ** 0.5)


# ----------------------------------------------------------------------------------------------------------------
# relu / dropout gate as a bit mask (vqcpc_gemm_nt_relu_mask / vqcpc_gemm_nt_gatebits)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K,p', [(512, 256, 64, 0.0), (4096, 1024, 256, 0.1), (2560, 512, 160, 0.3),
                                     (65536 + 512, 1024, 256, 0.1)])
def test_gate_bits_equal_the_fp32_gate(ops, M, N, K, p):
    """Forward: the same output as gemm_nt(act=1, drop_p) and a mask that is exactly [out > 0] in the documented word
    layout.  Backward: bit-identical to the dgrad GEMM gated by the fp32 activation, where that one runs the same
    256-tile kernel order (whole rounds), within fp32 rounding of the accumulation order otherwise."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + N + K)
    x, w1 = dev(torch.randn(M, K, generator=gen)), dev(torch.randn(N, K, generator=gen))
    b1 = dev(torch.randn(N, generator=gen))
    dy, w2t = dev(torch.randn(M, K, generator=gen)), dev(torch.randn(N, K, generator=gen))
    hip.set_gemm_mode(1)
    try:
        assert ops.gatebits_supported(M, N, K)
        h_ref = ops.gemm_nt(x, w1, bias=b1, act=1, drop_p=p, seed=321)
        h, mask = ops.gemm_nt_relu_mask(x, w1, b1, drop_p=p, seed=321)
        da_ref = ops.gemm_nt(dy, w2t, gate=h_ref, gate_scale=1.0 / (1.0 - p))
        da = ops.gemm_nt_gatebits(dy, w2t, mask, gate_scale=1.0 / (1.0 - p))
    finally:
        hip.set_gemm_mode(0)
    assert rel_err(h.cpu(), h_ref.cpu()) < 2e-6
    # the mask, unpacked on the host: word ((row >> 2) * N/32 + col/32) * 4 + (row & 3), bit col % 32
    words = mask.cpu().numpy().astype(np.uint32).reshape(M // 4, N // 32, 4)
    bits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)          # (M/4, N/32, 4, 32)
    unpacked = torch.from_numpy(np.ascontiguousarray(bits.transpose(0, 2, 1, 3)).reshape(M, N))
    assert torch.equal(unpacked, (h > 0).cpu())
    assert 0.3 < float(unpacked.float().mean()) / (1.0 - p) < 0.7                           # relu keeps about half
    gate_ok = ((h > 0) == (h_ref > 0))                     # identical up to elements whose pre-activation is ~0
    assert float((~gate_ok).float().mean()) < 1e-5
    err = ((da - da_ref).abs() * gate_ok).max() / da_ref.abs().max()
    assert float(err) < 2e-6
    assert not ops.gatebits_supported(M + 128, N, K) and not ops.gatebits_supported(M, N + 32, K)


def test_ffn_uses_the_bit_gate_in_bf16x6_mode_and_matches_the_fp32_gate(ops):
    """FFNFn end to end in mode 1 (bit gate) against the same function with the bit path disabled."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(5)
    M, d, ff = 1024, 256, 1024
    x = torch.randn(M, d, generator=gen)
    w1, b1 = torch.randn(ff, d, generator=gen) * 0.05, torch.randn(ff, generator=gen) * 0.1
    w2, b2 = torch.randn(d, ff, generator=gen) * 0.05, torch.randn(d, generator=gen) * 0.1
    gy = torch.randn(M, d, generator=gen)
    res = {}
    hip.set_gemm_mode(1)
    try:
        for bits in (True, False):
            saved = ops.gatebits_supported, ops.GATEBITS_MIN_TILES
            ops.GATEBITS_MIN_TILES = 0                     # 16 tiles here: below the size at which ops prefers the bit forms
            if not bits:
                ops.gatebits_supported = lambda *a: False
            try:
                t = [dev(v).requires_grad_(True) for v in (x, w1, b1, w2, b2)]
                y = ops.FFNFn.apply(t[0], t[1], t[2], t[3], t[4], 0.1, 99)
                (y * dev(gy)).sum().backward()
                res[bits] = [y.detach().cpu()] + [v.grad.cpu() for v in t]
            finally:
                ops.gatebits_supported, ops.GATEBITS_MIN_TILES = saved
    finally:
        hip.set_gemm_mode(0)
    for a, b in zip(res[True], res[False]):
        assert rel_err(a, b) < 5e-6


@lab_only
@pytest.mark.parametrize('M,N,K,with_bias', [(512, 256, 64, True), (65536, 768, 256, True), (66048, 256, 1024, False),
                                             (131072, 512, 96, True)])
def test_gemm_nt_one_wave_per_simd_kernel_is_bitwise_the_ping_pong_kernel(ops, M, N, K, with_bias):
    """gemm_sw.hip (mode +32): same LDS image, same fragments, same MFMA order as gemm_nt_x6_pp_kernel."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + N + K)
    a, b = dev(torch.randn(M, K, generator=gen)), dev(torch.randn(N, K, generator=gen))
    bias = dev(torch.randn(N, generator=gen)) if with_bias else None
    try:
        hip.set_gemm_mode(1)
        ref = ops.gemm_nt(a, b, bias=bias)
        hip.set_gemm_mode(33)
        out = ops.gemm_nt(a, b, bias=bias)
    finally:
        hip.set_gemm_mode(0)
    assert torch.equal(out, ref)
    exact = a.double() @ b.double().t() + (bias.double() if with_bias else 0)
    assert rel_err(out.cpu(), exact.cpu()) < 2e-6 * max(1, K ** 0.5)


# ----------------------------------------------------------------------------------------------------------------
# split-K NT GEMM (vqcpc_gemm_nt_splitk): the under-filled d_model-wide projections of the student / decoder steps
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(3072, 512, 2048), (768, 512, 1536), (128, 128, 1024), (4096, 512, 1024)])
def test_gemm_nt_split_k_matches_the_single_launch(ops, M, N, K):
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(M + K)
    a, b = dev(torch.randn(M, K, generator=gen)), dev(torch.randn(N, K, generator=gen) * 0.05)
    bias, res = dev(torch.randn(N, generator=gen)), dev(torch.randn(M, N, generator=gen))
    exact = a.double() @ b.double().t()
    assert hip.query('vqcpc_gemm_nt_splitk_workspace', M, N, K) == 0          # fp32-MFMA mode: never split
    hip.set_gemm_mode(1)
    try:
        ws_bytes = hip.query('vqcpc_gemm_nt_splitk_workspace', M, N, K)
        planes = ws_bytes // (4 * M * N)
        assert planes >= 2 and K % (32 * planes) == 0 and K // planes >= (64 if (M // 128) * (N // 128) <= 16 else 256)
        assert hip.query('vqcpc_gemm_nt_splitk_workspace', M, N, 512) == 0      # short K
        assert hip.query('vqcpc_gemm_nt_splitk_workspace', 65536, N, K) == 0    # enough tiles already
        assert hip.query('vqcpc_gemm_nt_splitk_workspace', M + 32, N, K) == 0   # partial tiles
        for kw in ({}, {'bias': bias}, {'add': res}, {'bias': bias, 'add': res}):
            ops.SPLIT_K = False
            single = ops.gemm_nt(a, b, **kw)
            ops.SPLIT_K = True
            split = ops.gemm_nt(a, b, **kw)
            again = ops.gemm_nt(a, b, **kw)
            ref = exact + (bias.double() if 'bias' in kw else 0) + (res.double() if 'add' in kw else 0)
            assert torch.equal(split, again)                                    # fixed summation order
            assert not torch.equal(split, single)                               # ... and really another path
            assert rel_err(split.cpu(), ref.cpu()) < 2e-6 and rel_err(single.cpu(), ref.cpu()) < 2e-6
        # strided output / residual (a column block of a wider buffer), workspace too small, wrong shape
        wide = dev(torch.zeros(M, 2 * N))
        ops.gemm_nt(a, b, bias=bias, add=res, out=wide[:, N:])
        assert torch.equal(wide[:, N:], split) and float(wide[:, :N].abs().max()) == 0.0
        ws = torch.empty(ws_bytes // 4, device='cuda')
        out = torch.empty(M, N, device='cuda')
        with pytest.raises(RuntimeError, match='workspace too small'):
            hip.call('vqcpc_gemm_nt_splitk', a, K, b, K, out, N, M, N, K, None, None, 0, ws, ws_bytes - 4)
        with pytest.raises(RuntimeError, match='not a split-K shape'):
            hip.call('vqcpc_gemm_nt_splitk', a, K, b, K, out, N, M, N, 512, None, None, 0, ws, ws_bytes)
    finally:
        ops.SPLIT_K = True
        hip.set_gemm_mode(0)


# ----------------------------------------------------------------------------------------------------------------
# distinct product codes of a step (vqcpc_count_distinct_codes): epoch()'s num_codewords metrics without a sort
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('ncb,K,rows_a,rows_b', [(1, 64, 1000, 0), (2, 512, 4096, 30720), (2, 512, 7, 0), (4, 32, 5000, 123),
                                                  (1, 1 << 20, 50000, 50000)])
def test_count_distinct_codes_equals_torch_unique(ops, ncb, K, rows_a, rows_b):
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(ncb * K + rows_a)
    hi = min(K, 40) if ncb > 1 else K                          # few distinct values per codebook: plenty of repeats
    a = dev(torch.randint(0, hi, (rows_a, ncb), generator=gen))
    b = dev(torch.randint(0, hi, (rows_b, ncb), generator=gen)) if rows_b else None
    assert hip.query('vqcpc_count_distinct_codes_supported', ncb, K) == 1
    out = torch.full((1,), -1.0, device='cuda')
    hip.call('vqcpc_count_distinct_codes', a, rows_a, b, rows_b, ncb, K, out)
    both = a if b is None else torch.cat([a, b])
    merged = sum(both[:, c] * K ** c for c in range(ncb))
    assert float(out[0]) == float(torch.unique(merged).numel())
    # the largest code is representable and counted
    top = dev(torch.full((3, ncb), K - 1, dtype=torch.int64))
    hip.call('vqcpc_count_distinct_codes', top, 3, None, 0, ncb, K, out)
    assert float(out[0]) == 1.0


def test_count_distinct_codes_rejects_spaces_beyond_the_bitmap(ops):
    from vqcpc_bach_amd import hip
    assert hip.query('vqcpc_count_distinct_codes_supported', 4, 1024) == 0          # configs[4]: the trainer sorts instead
    assert hip.query('vqcpc_count_distinct_codes_supported', 3, 128) == 0
    a = dev(torch.zeros(4, 4, dtype=torch.int64))
    out = torch.zeros(1, device='cuda')
    with pytest.raises(RuntimeError, match='exceed'):
        hip.call('vqcpc_count_distinct_codes', a, 4, None, 0, 4, 1024, out)


def test_gemm_nt_remainder_rows_go_through_split_k(ops):
    """139 264 x 256 x 1024 is 2.125 rounds of 256-tiles: vqcpc_gemm_nt cuts it by rows (whole rounds + a 128-tile
    remainder launch); ops.gemm_nt sends the remainder rows through the split-K entry point instead."""
    from vqcpc_bach_amd import hip
    M, N, K = 139264, 256, 1024
    gen = torch.Generator().manual_seed(3)
    a, b = dev(torch.randn(M, K, generator=gen)), dev(torch.randn(N, K, generator=gen) * 0.05)
    bias, res = dev(torch.randn(N, generator=gen)), dev(torch.randn(M, N, generator=gen))
    hip.set_gemm_mode(1)
    try:
        main = hip.query('vqcpc_gemm_nt_main_rows', M, N, K)
        assert main == 131072 and hip.query('vqcpc_gemm_nt_main_rows', 131072, N, K) == 131072
        assert hip.query('vqcpc_gemm_nt_splitk_workspace', M - main, N, K) > 0
        for kw in ({'bias': bias}, {'add': res}, {}):
            ops.SPLIT_K = False
            single = ops.gemm_nt(a, b, **kw)
            ops.SPLIT_K = True
            cut = ops.gemm_nt(a, b, **kw)
            assert torch.equal(cut[:main], single[:main])                     # the same whole rounds of the same kernel
            assert not torch.equal(cut[main:], single[main:])                 # ... and another path for the rest
            ref = (a[main:].double() @ b.double().t() + (bias.double() if 'bias' in kw else 0)
                   + (res[main:].double() if 'add' in kw else 0))
            assert rel_err(cut[main:].cpu(), ref.cpu()) < 2e-6
    finally:
        ops.SPLIT_K = True
        hip.set_gemm_mode(0)
    assert hip.query('vqcpc_gemm_nt_main_rows', M, N, K) == M                 # fp32-MFMA mode: no 256-tile kernel, no cut


# ----------------------------------------------------------------------------------------------------------------
# batched weight transposes (vqcpc_transpose_many, ops.WEIGHT_T)
# ----------------------------------------------------------------------------------------------------------------
def test_transpose_many_equals_single_transposes(ops):
    from vqcpc_bach_amd import hip
    gen = torch.Generator().manual_seed(1)
    shapes = [(256, 64), (33, 7), (1, 100), (512, 2048), (96, 96), (5, 1)]
    flat = dev(torch.randn(sum(r * c for r, c in shapes) + 64, generator=gen))
    arena = torch.full_like(flat, -7.0)
    rows, off, tiles = [], 16, 0                                  # first matrix at a non-zero offset
    for r, c in shapes:
        rows.append((off, r, c, tiles))
        tiles += ((r + 31) // 32) * ((c + 31) // 32)
        off += r * c
    desc = torch.tensor(rows, dtype=torch.int64).cuda()
    hip.call('vqcpc_transpose_many', flat, arena, desc, len(shapes), tiles)
    for (o, r, c, _) in rows:
        assert torch.equal(arena[o:o + r * c].view(c, r), flat[o:o + r * c].view(r, c).t())
    assert float(arena[:16].max()) == -7.0 and float(arena[off:].max()) == -7.0        # nothing outside the matrices


def test_weight_transposes_are_served_from_the_arena_inside_a_backward_pass(ops):
    """ops.transpose inside `direct_weight_gradients(flat)`: first pass learns the weights, later passes serve them from
    the arena (one launch), always equal to the plain transpose -- also after the weights changed; tensors outside the
    flat buffer and overlapping sub-blocks are never cached."""
    from vqcpc_bach_amd import hip
    flat = dev(torch.randn(4 * 64 * 48 + 8))
    w1, w2 = flat[:64 * 48].view(64, 48), flat[64 * 48:2 * 64 * 48].view(48, 64)
    sub = flat[:32 * 48].view(32, 48)                              # overlaps w1
    outside = dev(torch.randn(16, 24))
    calls = []
    raw = hip.call
    hip.call = lambda name, *a: (calls.append(name), raw(name, *a))[1]
    try:
        for step in range(3):
            calls.clear()
            with ops.direct_weight_gradients(flat):
                got = [ops.transpose(t) for t in (w1, w2, sub, outside)]
            for g, t in zip(got, (w1, w2, sub, outside)):
                assert torch.equal(g, t.t())
            if step == 0:
                assert calls.count('vqcpc_transpose') == 4 and 'vqcpc_transpose_many' not in calls
            else:
                assert calls.count('vqcpc_transpose_many') == 1 and calls.count('vqcpc_transpose') == 2   # sub, outside
            flat.mul_(1.5)                                          # the "optimiser step"
        with ops.direct_weight_gradients():                         # no flat buffer: plain transposes
            calls.clear()
            assert torch.equal(ops.transpose(w1), w1.t()) and calls == ['vqcpc_transpose']
        calls.clear()
        assert torch.equal(ops.transpose(w1), w1.t()) and calls == ['vqcpc_transpose']       # outside a backward pass
    finally:
        hip.call = raw


# ----------------------------------------------------------------------------------------------------------------
# glue nodes: stacked embedding tables, row split of the encoder output
# ----------------------------------------------------------------------------------------------------------------
def test_stack_tables_fn_equals_pad_and_stack(ops):
    gen = torch.Generator().manual_seed(2)
    sizes = [57, 49, 56, 33]
    for direct in (False, True):
        ws = [dev(torch.randn(n, 32, generator=gen)).requires_grad_(True) for n in sizes]
        ref_ws = [w.detach().clone().requires_grad_(True) for w in ws]
        gout = dev(torch.randn(4, 57, 32, generator=gen))
        ref = torch.stack([torch.nn.functional.pad(w, (0, 0, 0, 57 - w.shape[0])) for w in ref_ws], dim=0)
        (ref * gout).sum().backward()
        if direct:                                   # the trainers' situation: .grad buffers exist, gradients are added in place
            for w in ws:
                w.grad = torch.full_like(w, 0.5)
        out = ops.StackTablesFn.apply(*ws)
        assert torch.equal(out, ref.detach())
        if direct:
            with ops.direct_weight_gradients():
                (out * gout).sum().backward()
        else:
            (out * gout).sum().backward()
        for w, r in zip(ws, ref_ws):
            assert torch.equal(w.grad, r.grad + (0.5 if direct else 0.0))


def test_split_rows_fn_equals_slicing(ops):
    gen = torch.Generator().manual_seed(4)
    x = dev(torch.randn(100, 8, generator=gen)).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    a, b, c = ops.SplitRowsFn.apply(x, 30, 50, 20)
    ar, br, cr = xr[:30], xr[30:80], xr[80:]
    assert torch.equal(a, ar) and torch.equal(b, br) and torch.equal(c, cr)
    ga, gc = dev(torch.randn(30, 8, generator=gen)), dev(torch.randn(20, 8, generator=gen))
    ((a * ga).sum() + (c * gc).sum()).backward()                  # the middle block receives no gradient
    ((ar * ga).sum() + (cr * gc).sum()).backward()
    assert torch.equal(x.grad, xr.grad)


# ----------------------------------------------------------------------------------------------------------------
# opt-in gradient arithmetic of the bf16x6 mode (hip.set_gradient_products(3): two rounded planes, three products)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kind,M,N,K,epi', [('nt', 65536, 256, 1024, 'add'), ('nt', 32768, 1024, 256, 'none'),
                                            ('nt', 65536, 256, 256, 'add2'), ('tn', 131072, 256, 256, ''),
                                            ('tn', 65536, 1024, 256, '')])
def test_three_product_gradient_arithmetic(ops, bf16x6, kind, M, N, K, epi):
    """Inside a gradient scope with hip.set_gradient_products(3) the 256-tile NT / TN kernels evaluate hh + hm + mh on two
    ROUNDED planes per operand: ~2^-17 per product, i.e. rms error ~5e-6 of an N(0,1) product's rms -- an order above the
    six-product split (5e-7) and 500 x below a one-product bf16 GEMM (2.3e-3).  Outside a scope, and with the default of 6,
    nothing changes (bit-identical to the six-product result)."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M + N + K)
    if kind == 'nt':
        a, b = torch.randn(M, K, device='cuda', generator=gen), torch.randn(N, K, device='cuda', generator=gen)
        add = torch.randn(M, N, device='cuda', generator=gen) if epi in ('add', 'add2') else None
        add2 = torch.randn(M, N, device='cuda', generator=gen) if epi == 'add2' else None
        ref = a.double() @ b.double().t()
        for t in (add, add2):
            if t is not None:
                ref = ref + t.double()
        fn = lambda: ops.gemm_nt(a, b, add=add, add2=add2)
    else:
        a, b = torch.randn(M, N, device='cuda', generator=gen), torch.randn(M, K, device='cuda', generator=gen)
        ref = a.double().t() @ b.double()
        fn = lambda: ops.gemm_tn(a, b, want_bias=False)[0]

    def rms_err(x):
        return float((x.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())

    six = fn()
    assert hip.get_gradient_products() == 6
    try:
        hip.set_gradient_products(3)
        assert torch.equal(fn(), six), 'outside a gradient scope the setting must not change anything'
        hip.gradient_scope(True)
        try:
            three = fn()
            assert torch.equal(fn(), three), 'deterministic'
        finally:
            hip.gradient_scope(False)
        assert torch.equal(fn(), six)
    finally:
        hip.set_gradient_products(6)
    e6, e3 = rms_err(six), rms_err(three)
    assert e6 < 2.5e-6, e6                       # fp32-class (the TN contraction over 10^5 rows adds accumulation noise)
    assert 1e-6 < e3 < 1.2e-5, e3                # ~2^-17 per product: visibly not the six-product result, 500 x below bf16
    assert float((three.double() - ref).abs().max() / ref.abs().max()) < 2e-5


# ----------------------------------------------------------------------------------------------------------------
# f16x3 gradient arithmetic (round 5, csrc/gemm_grad.hip): two fp16 planes per operand under a per-tensor power-of-two scale
# ----------------------------------------------------------------------------------------------------------------
def _grad_state(a, b, stale=1.0):
    from vqcpc_bach_amd import hip
    st = torch.zeros(4, device='cuda')
    hip.call('vqcpc_grad_amax', a, a.stride(0), a.shape[0], a.shape[1], st[0:1])
    hip.call('vqcpc_grad_amax', b, b.stride(0), b.shape[0], b.shape[1], st[1:2])
    if stale != 1.0:
        st[0] *= stale
    return st


def _nt_grad(a, b, st, add=None, add2=None, mask=None, gate_scale=1.0):
    from vqcpc_bach_amd import hip
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, device='cuda')
    hip.call('vqcpc_gemm_nt_grad', a, a.stride(0), b, b.stride(0), out, N, M, N, K, add, 0 if add is None else add.stride(0), add2,
             0 if add2 is None else add2.stride(0), mask, float(gate_scale), st)
    return out


def _rms(x, ref):
    return float((x.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


@pytest.mark.parametrize('M,N,K,scale_a,scale_b', [(2048, 256, 64, 1.0, 1.0), (65536, 256, 1024, 1e-5, 0.05), (8192, 1024, 256, 3e-8, 1.0),
                                                    (512, 512, 32, 1e3, 1e-3)])
def test_f16x3_input_gradient_gemm_vs_fp64(ops, bf16x6, M, N, K, scale_a, scale_b):
    """vqcpc_gemm_nt_grad against an fp64 product: fp32-class (rms <= 1e-6 of the result's rms, within 2 x of the exact fp32 MFMA
    kernel's own error) for operands of any magnitude -- the per-tensor power-of-two scale makes the arithmetic scale free --,
    deterministic, and the kernel reports the operands' amax (what the next step's scale is taken from)."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn(M, K, device='cuda', generator=gen) * scale_a
    b = torch.randn(N, K, device='cuda', generator=gen) * scale_b
    ref = a.double() @ b.double().t()
    st = _grad_state(a, b)
    out = _nt_grad(a, b, st)
    assert torch.equal(out, _nt_grad(a, b, st)), 'deterministic'
    torch.cuda.synchronize()
    assert float(st[2]) == float(a.abs().max()) and float(st[3]) == float(b.abs().max()), st
    hip.set_gemm_mode(0)
    try:
        e32 = _rms(ops.gemm_nt(a, b), ref)
    finally:
        hip.set_gemm_mode(1)
    e = _rms(out, ref)
    assert e < 1e-6 and e < 2.0 * e32 + 1e-7, (e, e32)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 3e-6
    # the scale may lag: a previous-step amax 8 x smaller or 1000 x larger than this step's changes nothing / little
    for stale, tol in ((0.125, 1e-6), (1024.0, 1e-6), (2.0 ** 18, 2e-4)):
        assert _rms(_nt_grad(a, b, _grad_state(a, b, stale)), ref) < tol, stale
    # roll: this step's amax becomes the next step's scale, the written slots are cleared
    hip.call('vqcpc_grad_scale_roll', st, 1)
    torch.cuda.synchronize()
    assert float(st[0]) == float(a.abs().max()) and float(st[2]) == 0.0 and float(st[3]) == 0.0


def test_f16x3_saturates_instead_of_overflowing(ops, bf16x6):
    """A tensor that grew beyond the head-room of its previous-step scale: the largest elements saturate at 65504 / scale -- the
    result stays finite (no inf - inf) and is exact again once the scale has followed (one roll later); a NaN operand element
    stays visible."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(3)
    a = torch.randn(1024, 256, device='cuda', generator=gen) * 1e-4
    b = torch.randn(256, 256, device='cuda', generator=gen) * 0.05
    ref = a.double() @ b.double().t()
    count = torch.zeros(1, dtype=torch.int32, device='cuda')
    for stale in (1 / 64, 1 / 4096):
        st = _grad_state(a, b, stale)
        out = _nt_grad(a, b, st)
        assert bool(torch.isfinite(out).all()), stale
        hip.call('vqcpc_grad_scale_roll_counted', st, 1, count)     # the kernel saw the true amax; the roll counts the clamped operand
        assert _rms(_nt_grad(a, b, st), ref) < 1e-6
    assert int(count) == 2                                          # one operand (A) of one site, twice
    # a previous-step amax of ZERO (the tensor was all zeros when the site was primed) says nothing about the magnitude: the neutral
    # scale 1 (round 6; 2^60 zeroed the operand for that step) -- reduced precision for one step (these 1e-4 values sit in fp16's
    # subnormal range), no clamp, exact again after the roll
    st = _grad_state(a, b, 0.0)
    out = _nt_grad(a, b, st)
    assert bool(torch.isfinite(out).all()) and _rms(out, ref) < 5e-3, _rms(out, ref)
    hip.call('vqcpc_grad_scale_roll_counted', st, 1, count)
    assert int(count) == 2 and _rms(_nt_grad(a, b, st), ref) < 1e-6
    # the logged roll (what the trainers use): step index of the saturated steps
    mon = torch.zeros(3 + 4, dtype=torch.int32, device='cuda')
    for i, stale in enumerate((1.0, 1 / 4096, 1.0, 1 / 64)):
        st = _grad_state(a, b, stale)
        _nt_grad(a, b, st)
        hip.call('vqcpc_grad_scale_roll_logged', st, 1, mon, 4)
    assert mon.tolist() == [2, 4, 2, 1, 3, 0, 0], mon.tolist()
    hip.call('vqcpc_grad_scale_roll_counted', st, 1, count)         # a step within the head-room adds nothing
    assert int(count) == 2
    a2 = a.clone()
    a2[5, 7] = float('nan')
    out = _nt_grad(a2, b, _grad_state(a, b))
    assert bool(torch.isnan(out[5]).all()) and bool(torch.isfinite(out[6]).all())


@pytest.mark.parametrize('M,N,K', [(4096, 1024, 256), (2048, 256, 512)])
def test_f16x3_input_gradient_epilogues(ops, bf16x6, M, N, K):
    """The epilogue forms of the input-gradient GEMMs -- + residual, + two residuals, relu / dropout gate as a bit mask --
    against the six-product kernels' (both fp32-class: they differ by rounding noise only; the gate's zero pattern is equal)."""
    gen = torch.Generator(device='cuda').manual_seed(M + K)
    a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3
    b = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    add = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
    add2 = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
    st = _grad_state(a, b)
    for kw in ({}, dict(add=add), dict(add=add, add2=add2)):
        six = ops.gemm_nt(a, b, **kw)
        assert float((six - _nt_grad(a, b, st, **kw)).abs().max() / six.abs().max()) < 3e-6, list(kw)
    # the residual already in C (add == C): accumulated in place by fp32 atomic adds -- one add per element, the same bits as
    # the out-of-place form
    from vqcpc_bach_amd import hip
    acc = add.clone()
    hip.call('vqcpc_gemm_nt_grad', a, K, b, K, acc, N, M, N, K, acc, N, None, 0, None, 1.0, st)
    assert torch.equal(acc, _nt_grad(a, b, st, add=add))
    # ... and the second residual already in C (add2 == C): C += A . B^T + add, same bits as the out-of-place two-residual form
    acc = add2.clone()
    hip.call('vqcpc_gemm_nt_grad', a, K, b, K, acc, N, M, N, K, add, N, acc, N, None, 1.0, st)
    assert torch.equal(acc, _nt_grad(a, b, st, add=add, add2=add2))
    h, mask = ops.gemm_nt_relu_mask(torch.randn(M, K, device='cuda', generator=gen), torch.randn(N, K, device='cuda', generator=gen),
                                    torch.zeros(N, device='cuda'))
    six = ops.gemm_nt_gatebits(a, b, mask, gate_scale=1.25)
    g3 = _nt_grad(a, b, st, mask=mask, gate_scale=1.25)
    assert torch.equal(six == 0, g3 == 0) and float((six - g3).abs().max() / six.abs().max()) < 3e-6
    assert bool(((h > 0) == (g3 != 0)).float().mean() > 0.999)


def _nt_f16x3(a, b, st, bias, act=0, drop_p=0.0, seed=0, add=None, want_mask=False):
    from vqcpc_bach_amd import hip
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, device='cuda')
    mask = torch.empty(M * (N // 32), dtype=torch.int32, device='cuda') if want_mask else None
    hip.call('vqcpc_gemm_nt_f16x3', a, a.stride(0), b, b.stride(0), out, N, M, N, K, bias, int(act), float(drop_p), int(seed), add,
             0 if add is None else add.stride(0), mask, st)
    return (out, mask) if want_mask else out


@pytest.mark.parametrize('M,N,K', [(4096, 1024, 256), (2048, 256, 1024), (1024, 768, 256)])
def test_f16x3_forward_epilogues(ops, bf16x6, M, N, K):
    """vqcpc_gemm_nt_f16x3, the forward forms of the three-product kernel (opt-in FWD_ARITH = 'f16x3'): bias, bias + residual, bias +
    dropout + residual, bias + relu (+ dropout) with the bit mask -- against fp64 (rms in the class of the six-product and the
    exact fp32-MFMA kernels) and against vqcpc_gemm_nt: the SAME dropout pattern (same element index), the same mask bits wherever
    the pre-activation is not within rounding noise of zero, and the mask the backward reads equals `out > 0` exactly."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M + K)
    a = torch.randn(M, K, device='cuda', generator=gen)
    b = torch.randn(N, K, device='cuda', generator=gen) * K ** -0.5
    bias = torch.randn(N, device='cuda', generator=gen) * 0.1
    add = torch.randn(M, N, device='cuda', generator=gen)
    st = _grad_state(a, b)
    pre = a.double() @ b.double().t() + bias.double()
    hip.set_gemm_mode(0)
    try:
        e32 = _rms(ops.gemm_nt(a, b, bias=bias), pre)
    finally:
        hip.set_gemm_mode(1)
    e6 = _rms(ops.gemm_nt(a, b, bias=bias), pre)
    e3 = _rms(_nt_f16x3(a, b, st, bias), pre)
    print(f'rms error vs fp64: fp32 MFMA {e32:.2e}  bf16x6 {e6:.2e}  f16x3 {e3:.2e}')
    assert e3 < 1e-6 and e3 < 1.5 * max(e32, e6) + 5e-8, (e3, e6, e32)
    assert _rms(_nt_f16x3(a, b, st, bias, add=add), pre + add.double()) < 1e-6
    for p in (0.1,):
        six = ops.gemm_nt(a, b, bias=bias, drop_p=p, seed=11, add=add)
        g3 = _nt_f16x3(a, b, st, bias, drop_p=p, seed=11, add=add)
        assert torch.equal((six - add) == 0, (g3 - add) == 0), 'dropout pattern'
        assert float((six - g3).abs().max() / six.abs().max()) < 3e-6
    for p in (0.0, 0.1):
        h6, m6 = ops.gemm_nt_relu_mask(a, b, bias, drop_p=p, seed=5)
        h3, m3 = _nt_f16x3(a, b, st, bias, act=1, drop_p=p, seed=5, want_mask=True)
        assert float((h6 - h3).abs().max() / h6.abs().max()) < 3e-6, p
        # the mask is the bit pattern of out > 0, in the layout vqcpc_gemm_nt_gatebits reads
        g = torch.ones(M, 256, device='cuda')
        eye = torch.zeros(N, 256, device='cuda')          # (g @ eye^T) is irrelevant: only the zero pattern of the gated product counts
        eye[:, 0] = 1.0
        gated = ops.gemm_nt_gatebits(g, eye, m3)
        assert torch.equal(gated != 0, h3 > 0), p
        # against the six-product kernel's mask: differs only where the pre-activation is within rounding noise of zero
        diff = (h6 > 0) != (h3 > 0)
        assert float(diff.float().mean()) < 1e-5 and bool((pre[diff].abs() < 1e-5).all()), (p, int(diff.sum()))


@pytest.mark.parametrize('M,N,K,splits', [(8192, 256, 1024, 8), (4096, 512, 512, 4), (256, 256, 768, 8)])
def test_f16x3_splitk_remainder_launch(ops, bf16x6, M, N, K, splits):
    """vqcpc_gemm_nt_grad_splitk (the remainder rows of a ragged launch: few tiles, K cut into slices that run as one launch of the
    three-product kernel, planes summed in ascending order): equals the unsplit kernel up to fp32 summation order for every
    epilogue it serves -- none, + add, + add + add2, in place, + bias, + bias + dropout + add with the dropout pattern of the
    unsplit launch at the same global rows (row0) -- fp32-class against fp64, deterministic."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M + K)
    a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3
    b = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    add = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
    add2 = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
    bias = torch.randn(N, device='cuda', generator=gen) * 1e-4
    st = _grad_state(a, b)
    nbytes = hip.query('vqcpc_gemm_nt_grad_splitk_workspace', M, N, splits)
    ws = torch.empty(nbytes // 4, device='cuda')

    def sk(out=None, bias=None, drop_p=0.0, seed=0, row0=0, add=None, add2=None):
        out = torch.empty(M, N, device='cuda') if out is None else out
        hip.call('vqcpc_gemm_nt_grad_splitk', a, K, b, K, out, N, M, N, K, splits, bias, float(drop_p), int(seed), int(row0), add,
                 0 if add is None else N, add2, 0 if add2 is None else N, ws, nbytes, st)
        return out

    ref = a.double() @ b.double().t()
    plain = sk()
    assert torch.equal(plain, sk()), 'deterministic'
    assert _rms(plain, ref) < 1e-6 and float((plain - _nt_grad(a, b, st)).abs().max() / plain.abs().max()) < 3e-6
    for kw in (dict(add=add), dict(add=add, add2=add2)):
        assert float((sk(**kw) - _nt_grad(a, b, st, **kw)).abs().max() / plain.abs().max()) < 3e-6, list(kw)
    acc = add.clone()
    sk(out=acc, add=acc)                                          # the residual already in C
    assert torch.equal(acc, sk(add=add))
    assert torch.equal(sk(bias=bias), plain + bias)               # the epilogue pass adds the bias to the summed planes
    # dropout: the element index is (row0 + row) * N + col -- rows [row0, row0 + M) of a taller unsplit launch
    row0 = 512
    tall_a = torch.cat([torch.zeros(row0, K, device='cuda'), a])
    tall_add = torch.cat([torch.zeros(row0, N, device='cuda'), add])
    st2 = _grad_state(a, b)
    full = _nt_f16x3(tall_a, b, st2, bias, drop_p=0.1, seed=9, add=tall_add)[row0:]
    part = sk(bias=bias, drop_p=0.1, seed=9, row0=row0, add=add)
    assert torch.equal((full - add) == 0, (part - add) == 0), 'dropout pattern'
    assert float((full - part).abs().max() / full.abs().max()) < 3e-6


@pytest.mark.parametrize('M,N,K', [(8192, 256, 1024), (8192, 256, 256), (256, 512, 96), (64, 128, 32), (192, 384, 160)])
def test_f16x3_tail_rows_kernel(ops, bf16x6, M, N, K):
    """vqcpc_gemm_nt_grad_tail (the last rows of a ragged launch on 64 x 128 tiles): bit for bit the 256-tile kernel's rows for every
    epilogue without dropout -- none, + add, + add + add2, in place, + bias, + bias + add --, the 256-tile kernel's dropout pattern at
    the same global rows (row0), fp32-class against fp64, deterministic, and it reports the operands' amax."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn(M, K, device='cuda', generator=gen) * 1e-3
    b = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    add = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
    add2 = torch.randn(M, N, device='cuda', generator=gen) * 1e-4
    bias = torch.randn(N, device='cuda', generator=gen) * 1e-4
    st = _grad_state(a, b)
    assert hip.query('vqcpc_gemm_nt_grad_tail_supported', M, N, K) == 1
    assert hip.query('vqcpc_gemm_nt_grad_tail_supported', M + 32, N, K) == 0 and hip.query('vqcpc_gemm_nt_grad_tail_supported', M, N + 64, K) == 0

    def tail(out=None, bias=None, drop_p=0.0, seed=0, row0=0, add=None, add2=None, state=st):
        out = torch.empty(M, N, device='cuda') if out is None else out
        hip.call('vqcpc_gemm_nt_grad_tail', a, K, b, K, out, N, M, N, K, bias, float(drop_p), int(seed), int(row0), add,
                 0 if add is None else N, add2, 0 if add2 is None else N, state)
        return out

    ref = a.double() @ b.double().t()
    plain = tail()
    assert torch.equal(plain, tail()), 'deterministic'
    assert _rms(plain, ref) < 1e-6
    assert float(st[2]) == float(a.abs().max()) and float(st[3]) == float(b.abs().max())
    acc = add.clone()
    tail(out=acc, add=acc)                                        # the residual already in C
    assert torch.equal(acc, tail(add=add))
    assert torch.equal(tail(add=add), plain + add) and torch.equal(tail(add=add, add2=add2), plain + add + add2)
    assert torch.equal(tail(bias=bias), plain + bias) and torch.equal(tail(bias=bias, add=add), plain + bias + add)
    # against the 256-tile kernel on the same rows (zero rows below, so that M and N are whole 256-tiles)
    Mp, Np = -(-M // 256) * 256, -(-N // 256) * 256
    ap = torch.zeros(Mp, K, device='cuda'); ap[:M] = a
    bp = torch.zeros(Np, K, device='cuda'); bp[:N] = b
    big = _nt_grad(ap, bp, st.clone())
    assert torch.equal(big[:M, :N], plain), 'the 256-tile kernel gives these rows the same bits'
    if N == Np:
        # dropout: the element index is (row0 + row) * N + col -- rows [row0, row0 + M) of a taller launch of the 256-tile kernel
        row0 = 512
        tall_a = torch.cat([torch.zeros(row0, K, device='cuda'), ap])
        tall_add = torch.zeros(row0 + Mp, N, device='cuda'); tall_add[row0:row0 + M] = add
        full = _nt_f16x3(tall_a, b, st.clone(), bias, drop_p=0.1, seed=9, add=tall_add)[row0:row0 + M]
        part = tail(bias=bias, drop_p=0.1, seed=9, row0=row0, add=add)
        assert torch.equal((full - add) == 0, (part - add) == 0), 'dropout pattern'
        assert 0.05 < float(((part - add) == 0).float().mean()) < 0.15
        assert float((full - part).abs().max() / full.abs().max()) < 1e-6


def test_f16x3_ragged_rounds_take_whole_rounds_plus_the_tail_rows(ops, bf16x6):
    """ops._g3_plan at the C1 shapes that do not fill whole rounds (139 264 x 256 x K: 544 tiles): 512 tiles on the 256-tile kernel,
    the last 8 192 rows on vqcpc_gemm_nt_grad_tail, inside the forward scope and inside the gradient scope -- the dropout pattern
    of the six-product kernel, fp32-class agreement with it, and the in-place residual form equal to the out-of-place one."""
    from vqcpc_bach_amd import hip
    M, N = 139264, 256
    saved = ops.GRAD_SPLITK, ops.GRAD_TAIL
    ops.GRAD_SPLITK, ops.GRAD_TAIL = False, True
    prev_f, prev_g = ops.set_forward_arithmetic('f16x3'), ops.set_gradient_arithmetic('f16x3')
    raw = hip.call
    try:
        for K in (1024, 256):
            assert ops._g3_plan(M, N, K) == (131072, -1)
            gen = torch.Generator(device='cuda').manual_seed(5 + K)
            a = torch.randn(M, K, device='cuda', generator=gen)
            w = torch.randn(N, K, device='cuda', generator=gen) * K ** -0.5
            bias = torch.randn(N, device='cuda', generator=gen) * 0.1
            res = torch.randn(M, N, device='cuda', generator=gen)

            class Flat:
                flat = torch.zeros(4, device='cuda')
            owner = Flat()
            calls = []
            hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
            six = ops.gemm_nt(a, w, bias=bias, drop_p=0.1, seed=4, add=res)
            with torch.enable_grad(), ops.forward_arithmetic(owner):
                fwd = ops.gemm_nt(a, w, bias=bias, drop_p=0.1, seed=4, add=res)
            assert calls.count('vqcpc_gemm_nt_f16x3') == 1 and calls.count('vqcpc_gemm_nt_grad_tail') == 1
            with ops.direct_weight_gradients(owner):
                g = ops.gemm_nt(a, w, add=res)
                acc = res.clone()
                g2 = ops.gemm_nt_residual(a, w, acc)
            hip.call = raw
            assert calls.count('vqcpc_gemm_nt_grad_tail') == 3 and g2.data_ptr() == acc.data_ptr()
            # the dropout pattern of the six-product kernel (a kept product below half an ulp of its residual looks dropped: a few in 35 M)
            bad = ((six - res) == 0) != ((fwd - res) == 0)
            assert int(bad.sum()) <= 4 and (not bad.any() or float(torch.maximum((six - res).abs(), (fwd - res).abs())[bad].max()) < 1e-5)
            assert 0.09 < float(((fwd - res)[131072:] == 0).float().mean()) < 0.11
            assert float((six - fwd).abs().max() / six.abs().max()) < 3e-6
            assert torch.equal(g, g2)
            rows = torch.cat([torch.arange(0, 512), torch.arange(131072 - 256, 131072 + 256), torch.arange(M - 256, M)]).cuda()
            ref = a[rows].double() @ w.double().t() + res[rows].double()
            assert _rms(g[rows], ref) < 1e-6
    finally:
        hip.call = raw
        ops.GRAD_SPLITK, ops.GRAD_TAIL = saved
        ops.set_forward_arithmetic(prev_f)
        ops.set_gradient_arithmetic(prev_g)


def test_f16x3_ragged_rounds_take_whole_rounds_plus_a_splitk_remainder(ops, bf16x6):
    """ops._g3_plan at the C1 shape that does not fill whole rounds (139 264 x 256 x 1024: 544 tiles): 512 tiles on one launch, 32
    tiles x 8 K slices on the remainder launch, inside the forward scope and inside the gradient scope -- against fp64 on a row
    sample, and with the dropout pattern of the six-product kernel."""
    from vqcpc_bach_amd import hip
    M, N, K = 139264, 256, 1024
    saved_sk, ops.GRAD_SPLITK = ops.GRAD_SPLITK, True             # opt-in (VQCPC_GRAD_SPLITK=1)
    saved_tail, ops.GRAD_TAIL = ops.GRAD_TAIL, False              # (the default plan of these shapes: the tail-row launch, tested above)
    assert ops._g3_plan(M, N, K) == (131072, 8) and ops._g3_plan(M, N, 256) is None and ops._g3_plan(557056, N, K) == (557056, 0)
    gen = torch.Generator(device='cuda').manual_seed(5)
    a = torch.randn(M, K, device='cuda', generator=gen)
    w = torch.randn(N, K, device='cuda', generator=gen) * K ** -0.5
    bias = torch.randn(N, device='cuda', generator=gen) * 0.1
    res = torch.randn(M, N, device='cuda', generator=gen)

    class Flat:
        flat = torch.zeros(4, device='cuda')
    owner = Flat()
    calls, raw = [], hip.call
    prev_f, prev_g = ops.set_forward_arithmetic('f16x3'), ops.set_gradient_arithmetic('f16x3')
    hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
    try:
        six = ops.gemm_nt(a, w, bias=bias, drop_p=0.1, seed=4, add=res)
        with torch.enable_grad(), ops.forward_arithmetic(owner):
            fwd = ops.gemm_nt(a, w, bias=bias, drop_p=0.1, seed=4, add=res)
        assert calls.count('vqcpc_gemm_nt_f16x3') == 1 and calls.count('vqcpc_gemm_nt_grad_splitk') == 1
        with ops.direct_weight_gradients(owner):
            g = ops.gemm_nt(a, w, add=res)
            acc = res.clone()
            g2 = ops.gemm_nt_residual(a, w, acc)
        assert calls.count('vqcpc_gemm_nt_grad_splitk') == 3 and g2.data_ptr() == acc.data_ptr()
    finally:
        hip.call = raw
        ops.GRAD_SPLITK = saved_sk
        ops.GRAD_TAIL = saved_tail
        ops.set_forward_arithmetic(prev_f)
        ops.set_gradient_arithmetic(prev_g)
    assert torch.equal((six - res) == 0, (fwd - res) == 0)
    assert float((six - fwd).abs().max() / six.abs().max()) < 3e-6
    assert torch.equal(g, g2)
    rows = torch.cat([torch.arange(0, 512), torch.arange(131072 - 256, 131072 + 512), torch.arange(M - 256, M)]).cuda()
    ref = a[rows].double() @ w.double().t() + res[rows].double()
    assert _rms(g[rows], ref) < 1e-6


@pytest.mark.parametrize('M,N,K,scale_a', [(4096, 256, 256, 1.0), (131072, 256, 256, 1e-6), (65536, 1024, 256, 1e-4), (1056, 256, 512, 1.0)])
def test_f16x3_weight_gradient_gemm_vs_fp64(ops, bf16x6, M, N, K, scale_a):
    """vqcpc_gemm_tn_grad: dW = A^T B within the fp32 class of an fp64 product (the contraction over 10^5 rows carries fp32
    accumulation noise, as the fp32 MFMA kernel's does), db = column sums of A in exact fp32 arithmetic, accumulation into an
    existing gradient (accumulate = 1), amax of both operands reported."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M + N)
    a = torch.randn(M, N, device='cuda', generator=gen) * scale_a
    b = torch.randn(M, K, device='cuda', generator=gen)
    ref, ref_b = a.double().t() @ b.double(), a.double().sum(0)
    st = _grad_state(a, b)
    nb = hip.query('vqcpc_gemm_tn_grad_workspace', M, N, K)
    ws = hip.workspace(nb, a.device)
    dw, db = torch.empty(N, K, device='cuda'), torch.empty(N, device='cuda')
    hip.call('vqcpc_gemm_tn_grad', a, N, b, K, dw, db, M, N, K, 0, ws, nb, st)
    torch.cuda.synchronize()
    assert float(st[2]) == float(a.abs().max()) and float(st[3]) == float(b.abs().max())
    hip.set_gemm_mode(0)
    try:
        e32 = _rms(ops.gemm_tn(a, b, want_bias=False)[0], ref)
    finally:
        hip.set_gemm_mode(1)
    e = _rms(dw, ref)
    assert e < 1.5e-6 and e < 2.0 * e32 + 1e-7, (e, e32)
    assert _rms(db, ref_b) < 1e-6
    base_w, base_b = torch.randn(N, K, device='cuda', generator=gen) * scale_a, torch.randn(N, device='cuda', generator=gen) * scale_a
    acc_w, acc_b = base_w.clone(), base_b.clone()
    hip.call('vqcpc_gemm_tn_grad', a, N, b, K, acc_w, acc_b, M, N, K, 1, ws, nb, st)
    assert float((acc_w - (base_w + dw)).abs().max()) <= 1e-6 * float(dw.abs().max())
    assert float((acc_b - (base_b + db)).abs().max()) <= 1e-6 * float(db.abs().max())


def test_f16x3_arithmetic_is_taken_inside_a_backward_scope_only(ops, bf16x6):
    """ops.set_gradient_arithmetic('f16x3'): gemm_nt / gemm_nt_gatebits / gemm_tn(into=...) inside ops.direct_weight_gradients(flat)
    go through vqcpc_gemm_nt_grad / _tn_grad with one scale site each (in call order, primed on first use, rolled when the scope
    closes); outside a scope, and for the forward's epilogues, nothing changes; a ragged last round of tiles is cut by rows."""
    from vqcpc_bach_amd import hip

    class Flat:                                    # stands in for parallel.FlatParameters
        flat = torch.zeros(4, device='cuda')

    gen = torch.Generator(device='cuda').manual_seed(9)
    M, N, K = 2048 + 128, 256, 512
    a, b = torch.randn(M, K, device='cuda', generator=gen) * 1e-3, torch.randn(N, K, device='cuda', generator=gen) * 0.1
    ga, xb = torch.randn(4096, 256, device='cuda', generator=gen) * 1e-3, torch.randn(4096, 512, device='cuda', generator=gen)
    six = ops.gemm_nt(a, b)
    six_w = ops.gemm_tn(ga, xb)
    calls, raw = [], hip.call
    prev = ops.set_gradient_arithmetic('f16x3')
    saved = ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS
    ops.GRAD_MIN_TILES = ops.GRAD_TN_MIN_ROWS = 0
    try:
        assert torch.equal(ops.gemm_nt(a, b), six), 'outside a backward scope: the six-product kernels'
        owner = Flat()
        hip.call = lambda name, *args: (calls.append(name), raw(name, *args))[1]
        for step in range(2):
            calls.clear()
            with ops.direct_weight_gradients(owner):
                g3 = ops.gemm_nt(a, b)
                fwd_like = ops.gemm_nt(a, b, bias=torch.zeros(N, device='cuda'))        # a bias epilogue is not a gradient GEMM
                dw = torch.zeros(256, 512, device='cuda')
                dbias = torch.zeros(256, device='cuda')
                ops.gemm_tn(ga, xb, into=(dw, dbias))
            tab = owner._grad_scales[None]
            assert tab.keys == [('nt', M, N, K), ('tn', 4096, 256, 512)]
            assert calls.count('vqcpc_gemm_nt_grad') == 1 and calls.count('vqcpc_gemm_tn_grad') == 1
            assert calls.count('vqcpc_grad_amax') == (4 if step == 0 else 0), 'primed once'
            assert calls.count('vqcpc_grad_scale_roll_logged') == 1
            assert calls.count('vqcpc_gemm_nt') == 2, 'the 128 ragged rows + the bias GEMM'
            assert torch.equal(fwd_like, six)
            assert not torch.equal(g3, six) and float((g3 - six).abs().max() / six.abs().max()) < 3e-6
            assert float((dw - six_w[0]).abs().max() / six_w[0].abs().max()) < 3e-6
            assert float((dbias - six_w[1]).abs().max() / six_w[1].abs().max()) < 3e-6
            torch.cuda.synchronize()
            assert float(tab.state[0]) == float(a[:2048].abs().max()) and float(tab.state[2]) == 0.0
    finally:
        hip.call = raw
        ops.GRAD_MIN_TILES, ops.GRAD_TN_MIN_ROWS = saved
        ops.set_gradient_arithmetic(prev)


def test_gru_fused_steps_equal_the_gemm_plus_gate_launches(ops):
    """vqcpc_gru_step_fwd / _bwd (one launch per time step) against the same layer run as GEMM + gate launches: same values
    up to the summation order of the fp32 recurrent product (both are exact-fp32 MFMA chains, split differently)."""
    from vqcpc_bach_amd import hip
    assert hip.query('vqcpc_gru_step_supported', 256, 512) == 1 and hip.query('vqcpc_gru_step_supported', 8, 96) == 0
    gen = torch.Generator().manual_seed(5)
    B, T, inp, hid = 200, 6, 32, 256
    x0 = torch.randn(T * B, inp, generator=gen)
    ws = [torch.randn(3 * hid, inp, generator=gen) * 0.2, torch.randn(3 * hid, hid, generator=gen) * 0.06,
          torch.randn(3 * hid, generator=gen) * 0.1, torch.randn(3 * hid, generator=gen) * 0.1]
    gout = torch.randn(T * B, hid, generator=gen)
    res = {}
    calls = []
    raw = hip.call
    hip.call = lambda name, *a: (calls.append(name), raw(name, *a))[1]
    try:
        for fused in (True, False):
            ops.GRU_FUSED_STEPS = fused
            calls.clear()
            x = dev(x0).requires_grad_(True)
            prm = [dev(w).requires_grad_(True) for w in ws]
            y = ops.GRULayerFn.apply(x, prm[0], prm[1], prm[2], prm[3], T, 0.25, 77, False)
            (y * dev(gout)).sum().backward()
            res[fused] = [y.detach(), x.grad] + [w.grad for w in prm]
            if fused:
                assert calls.count('vqcpc_gru_step_fwd') == T and calls.count('vqcpc_gru_step_bwd') == T - 1
                assert calls.count('vqcpc_gru_cell_fwd') == 0 and calls.count('vqcpc_gru_cell_bwd') == 1
            else:
                assert calls.count('vqcpc_gru_step_fwd') == 0 and calls.count('vqcpc_gru_cell_fwd') == T
    finally:
        hip.call = raw
        ops.GRU_FUSED_STEPS = True
    for a, b in zip(res[True], res[False]):
        assert rel_err(a.cpu(), b.cpu()) < 2e-6


@lab_only
@pytest.mark.parametrize('M,N,K', [(65536, 256, 256), (131072, 1024, 256), (557056, 256, 512)])
def test_gemm_tn_quad_row_lds_image_is_bit_identical_to_the_row_pair_image(ops, bf16x6, M, N, K):
    """gemm_tn_x6_pq_kernel (four consecutive rows per staging thread, fragments by two ds_read_b64) against
    gemm_tn_x6_pp_kernel (row pairs, two ds_read2_b32): the same products in the same order, so the weight gradient is
    bit-identical (the bias sums are accumulated four rows at a time instead of two: fp32-rounding apart); also in the
    opt-in three-product gradient arithmetic."""
    import os
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M // 1000 + N + K)
    a, b = torch.randn(M, N, device='cuda', generator=gen), torch.randn(M, K, device='cuda', generator=gen)
    saved = os.environ.get('VQCPC_TN_PQ')
    res = {}
    try:
        for pq in ('0', '1'):
            os.environ['VQCPC_TN_PQ'] = pq
            dw, db = ops.gemm_tn(a, b)
            hip.set_gradient_products(3)
            hip.gradient_scope(True)
            try:
                dw3, _ = ops.gemm_tn(a, b)
            finally:
                hip.gradient_scope(False)
                hip.set_gradient_products(6)
            res[pq] = (dw, db, dw3)
    finally:
        if saved is None:
            os.environ.pop('VQCPC_TN_PQ', None)
        else:
            os.environ['VQCPC_TN_PQ'] = saved
    assert torch.equal(res['0'][0], res['1'][0]), 'weight gradient: same products, same order'
    assert torch.equal(res['0'][2], res['1'][2])
    assert rel_err(res['1'][1].cpu(), res['0'][1].cpu()) < 1e-6
    ref = a.double().t() @ b.double()
    assert rel_err(res['1'][0].cpu(), ref.cpu()) < 3e-6
    assert rel_err(res['1'][1].cpu(), a.double().sum(0).cpu()) < 3e-6


@pytest.mark.parametrize('mode', [1, 8])
def test_deferred_grouped_reduction_of_weight_gradient_partials_is_bit_identical(ops, mode):
    """Inside a trainer's gradient scope the partial sums of the LARGE weight gradients stay in their workspaces and ONE
    vqcpc_reduce_grouped_vec launch sums them when the scope closes (ops.DEFER_TN_REDUCTIONS): same bits as the reduction each
    product would have run on its own -- also for a gradient buffer that receives two products (accumulation order = issue order)
    and next to a small product that keeps its immediate reduction."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(5)
    shapes = [(65536, 768, 256), (65536, 256, 256), (32768, 256, 1024), (4096, 64, 64)]
    hip.set_gemm_mode(mode)
    try:
        res = {}
        for defer in (False, True):
            ops.DEFER_TN_REDUCTIONS = defer
            outs = []
            gen.manual_seed(5)
            with ops.direct_weight_gradients():
                for M, N, K in shapes:
                    a, b = torch.randn(M, N, device='cuda', generator=gen), torch.randn(M, K, device='cuda', generator=gen)
                    dw, db = torch.full((N, K), 0.5, device='cuda'), torch.full((N,), -1.0, device='cuda')
                    tn = ops.gemm_tn_bf16 if (mode == 8 and hip.query('vqcpc_gemm_tn_bf16_supported', M, N, K)) else ops.gemm_tn
                    tn(a, b, into=(dw, db))
                    if N == 768:                     # a second product into the same buffers
                        tn(b.new_ones(M, N), b, into=(dw, db))
                    outs += [dw, db]
                pending = len(ops._PENDING_VEC_REDUCTIONS)
            assert (pending > 0) == defer and not ops._PENDING_VEC_REDUCTIONS
            res[defer] = [o.clone() for o in outs]
        for x, y in zip(res[False], res[True]):
            assert torch.equal(x, y)
    finally:
        ops.DEFER_TN_REDUCTIONS = False
        hip.set_gemm_mode(0)


# ----------------------------------------------------------------------------------------------------------------
# round 6: PRE-SPLIT ("P4") operands of the f16x3 products -- weights split once per step (csrc/gemm_grad.hip)
# ----------------------------------------------------------------------------------------------------------------
def _p4_images(mats):
    """[(rows, cols) fp32 matrices] -> (flat, planes, planes_t, amax, offsets) through vqcpc_weight_planes_many."""
    from vqcpc_bach_amd import hip
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
    planes, planes_t = torch.full_like(flat, float('nan')), torch.full_like(flat, float('nan'))
    amax = torch.full((len(mats),), 7.0, device='cuda')              # stale contents: the entry point clears them itself
    ws = torch.empty(tiles, device='cuda')
    hip.call('vqcpc_weight_planes_many', flat, desc, len(mats), tiles, amax, planes, planes_t, ws, 4 * tiles)
    return flat, planes, planes_t, amax, offs


def _p4_decode(img, rows, cols, amax):
    """P4 image -> h + m in fp64 (descaled): what the matrix pipe multiplies."""
    e = 11 + 127 - ((amax.view(torch.int32) >> 23) & 0xFF)
    halves = img.view(torch.float16).view(rows, cols // 4, 8).double()
    return ((halves[:, :, :4] + halves[:, :, 4:]) * 2.0 ** (-float(e))).reshape(rows, cols)


def test_p4_weight_planes_hold_the_two_plane_split(ops, bf16x6):
    """vqcpc_weight_planes_many: per matrix the exact amax, the P4 image of W (groups along the columns) and of W^T (groups along the
    rows) under that amax's power-of-two scale: h + m reproduces every element to 2^-21 (relative; 2^-36 of the amax below)."""
    gen = torch.Generator(device='cuda').manual_seed(11)
    mats = [torch.randn(768, 256, device='cuda', generator=gen) * 0.05, torch.randn(100, 36, device='cuda', generator=gen) * 3.0,
            torch.randn(256, 1024, device='cuda', generator=gen) * 1e-3]
    flat, planes, planes_t, amax, offs = _p4_images(mats)
    for i, (m, o) in enumerate(zip(mats, offs)):
        assert float(amax[i]) == float(m.abs().max())
        r, c = m.shape
        for img, ref in ((planes[o:o + r * c], m), (planes_t[o:o + r * c], m.t().contiguous())):
            got = _p4_decode(img, ref.shape[0], ref.shape[1], amax[i])
            err = (got - ref.double()).abs()
            assert bool((err <= 2.0 ** -21 * ref.double().abs() + 2.0 ** -36 * float(amax[i])).all()), (i, float(err.max()))


@pytest.mark.parametrize('M,N,K', [(4096, 1024, 256), (2048, 256, 1024), (1024, 768, 256)])
def test_p4_products_are_bit_identical_to_the_fp32_operand_kernels(ops, bf16x6, M, N, K):
    """vqcpc_gemm_nt_g3_pl with B (and A) as P4 images == vqcpc_gemm_nt_f16x3 / vqcpc_gemm_nt_grad on the fp32 operands under the same
    amax, bit for bit, for every epilogue form of the training step -- and the scale state still receives the amax of an fp32 A."""
    from vqcpc_bach_amd import hip
    gen = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn(M, K, device='cuda', generator=gen)
    w = torch.randn(N, K, device='cuda', generator=gen) * 0.05
    bias = torch.randn(N, device='cuda', generator=gen)
    add = torch.randn(M, N, device='cuda', generator=gen)
    add2 = torch.randn(M, N, device='cuda', generator=gen)
    _, pl, _, amax, _ = _p4_images([w, a])
    wp, ap = pl[:N * K], pl[N * K:]

    def state():
        st = torch.zeros(4, device='cuda')
        st[0], st[1] = amax[1], amax[0]
        return st

    def run_pl(a_, amax_a, **kw):
        out = kw.pop('out', None)
        out = torch.empty(M, N, device='cuda') if out is None else out
        st = state()
        mask_out = torch.empty(M * (N // 32), dtype=torch.int32, device='cuda') if kw.get('act') else None
        hip.call('vqcpc_gemm_nt_g3_pl', a_, K, wp, K, out, N, M, N, K, kw.get('bias'), int(kw.get('act', 0)), float(kw.get('drop_p', 0.0)),
                 int(kw.get('seed', 0)), kw.get('add'), N if kw.get('add') is not None else 0, kw.get('add2'),
                 N if kw.get('add2') is not None else 0, kw.get('gate'), float(kw.get('gate_scale', 1.0)), mask_out, st, amax_a, amax[0:1])
        return out, mask_out, st

    # forward forms
    for kw in (dict(bias=bias), dict(bias=bias, add=add), dict(bias=bias, drop_p=0.1, seed=77, add=add)):
        ref = _nt_f16x3(a, w, state(), kw['bias'], drop_p=kw.get('drop_p', 0.0), seed=kw.get('seed', 0), add=kw.get('add'))
        for a_, am in ((a, None), (ap, amax[1:2])):
            out, _, st = run_pl(a_, am, **kw)
            assert torch.equal(out, ref), (list(kw), am is not None)
            assert float(st[3]) == 0.0 and float(st[2]) == (0.0 if am is not None else float(a.abs().max()))
    ref, ref_mask = _nt_f16x3(a, w, state(), bias, act=1, drop_p=0.1, seed=5, want_mask=True)
    out, mask, _ = run_pl(a, None, bias=bias, act=1, drop_p=0.1, seed=5)
    assert torch.equal(out, ref) and torch.equal(mask, ref_mask)
    # input-gradient forms
    for kw in (dict(), dict(add=add), dict(add=add, add2=add2)):
        ref = _nt_grad(a, w, state(), **kw)
        out, _, _ = run_pl(a, None, **kw)
        assert torch.equal(out, ref), list(kw)
    ref = _nt_grad(a, w, state(), mask=ref_mask, gate_scale=1.25)
    out, _, _ = run_pl(a, None, gate=ref_mask, gate_scale=1.25)
    assert torch.equal(out, ref)
    for a_, am in ((a, None), (ap, amax[1:2])):                       # in place: add == C
        acc = add.clone()
        run_pl(a_, am, add=acc, out=acc)
        assert torch.equal(acc, _nt_grad(a, w, state(), add=add)), am is not None
    # tail rows (64 x 128 tiles)
    rows = 192
    ref = torch.empty(rows, N, device='cuda')
    hip.call('vqcpc_gemm_nt_grad_tail', a[:rows], K, w, K, ref, N, rows, N, K, bias, 0.1, 9, 4096, add[:rows], N, None, 0, state())
    for a_, am in ((a, None), (ap, amax[1:2])):
        out = torch.empty(rows, N, device='cuda')
        hip.call('vqcpc_gemm_nt_g3_small', a_[:rows * K], K, wp, K, out, N, rows, N, K, bias, 0, 0.1, 9, 4096, None, 0, 1.0, add[:rows], N,
                 None, 0, state(), am, amax[0:1])
        assert torch.equal(out, ref), am is not None
