"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports exactly what include/vqcpc.h
declares (no compute calls: there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import ROOT


def header_symbols(lab=False):
    """Entry points declared by include/vqcpc.h: the product ABI, or (lab=True) the `#ifdef VQCPC_LAB` section only."""
    text = open(os.path.join(ROOT, 'include', 'vqcpc.h')).read()
    m = re.search(r'#ifdef VQCPC_LAB\n(.*?)#endif /\* VQCPC_LAB \*/', text, flags=re.S)
    assert m, 'include/vqcpc.h must keep its lab section'
    text = m.group(1) if lab else text[:m.start()] + text[m.end():]
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vqcpc_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    import torch  # noqa: F401  (loads the process-wide libamdhip64.so.7 first)
    from vqcpc_bach_amd import build, hip
    if not os.path.exists(hip.LIB_PATH):
        build.build(verbose=False)
    return hip.load()


def test_every_declared_symbol_is_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/vqcpc.h but not exported'


def test_binding_covers_header_exactly():
    from vqcpc_bach_amd import hip
    assert sorted(hip.SIGNATURES) == header_symbols()
    assert sorted(hip.LAB_SIGNATURES) == header_symbols(lab=True)


def test_product_library_is_not_the_lab_bench(lib):
    """Rejected kernel designs and measurement switches live in the lab build only (`VQCPC_LAB=1 python -m
    vqcpc_bach_amd.build`): the product library exports none of their entry points, contains none of their kernels and
    does not know the tools' environment variables."""
    from vqcpc_bach_amd import hip
    if os.environ.get('VQCPC_LAB', '0') == '1':
        pytest.skip('the lab library is loaded')
    for s in header_symbols(lab=True):
        assert not hasattr(lib, s), f'{s} is a lab entry point'
    blob = open(hip.LIB_PATH, 'rb').read()
    for needle in (b'gemm_nt_x6_dma_kernel', b'gemm_nt_x6_sw', b'planes_kernel', b'VQCPC_PP_ABL', b'VQCPC_TN_PQ', b'VQCPC_PP_GRID',
                   b'VQCPC_S64_MAX_TILES', b'VQCPC_BF16_STAGGER', b'VQCPC_GEMM_ABL', b'VQCPC_G3_ABL'):
        assert needle not in blob, needle
    # ... and it has no kernel-selection switch: the A/B bits of vqcpc_gemm_set_mode are refused, the arithmetic modes are not
    before = lib.vqcpc_gemm_get_mode()
    try:
        for m in (3, 5, 17, 33):
            assert lib.vqcpc_gemm_set_mode(m) != 0, m
        assert lib.vqcpc_gemm_set_mode(1) == 0 and lib.vqcpc_gemm_get_mode() == 1
    finally:
        lib.vqcpc_gemm_set_mode({0: 0, 1: 1, 2: 8}[before])


def test_abi_version_and_error_string(lib):
    from vqcpc_bach_amd import hip
    assert lib.vqcpc_abi_version() == hip.ABI_VERSION == 2
    assert isinstance(lib.vqcpc_last_error(), bytes)


def test_workspace_queries_are_host_only(lib):
    # pure host functions: callable without a GPU
    assert lib.vqcpc_gemm_tn_workspace(557056, 768, 256) > 0
    assert lib.vqcpc_vq_bwd_workspace(34816, 2, 512, 16) == 136 * 2 * 512 * 16 * 4
    assert lib.vqcpc_add_layernorm_bwd_workspace(1000, 256) == 250 * 2 * 256 * 4
    assert lib.vqcpc_relattn_bwd_workspace(34816, 16, 8, 32) > 0
    assert lib.vqcpc_sumsq_workspace(10) == 1024 * 8


def test_argument_validation_without_gpu(lib):
    # bad arguments are rejected before any HIP call, with a message
    rc = lib.vqcpc_gemm_nt(None, 0, None, 0, None, 0, 4, 4, 4, None, 0, 0.0, 0, None, 0, 1.0, None, 0, None, 0, None)
    assert rc == -1 and b'gemm_nt' in lib.vqcpc_last_error()
    rc = lib.vqcpc_vq_fwd(None, None, 1, 1, 1, 1, 0.25, 1, 1, None, None, None, None)
    assert rc == -1 and b'vq_fwd' in lib.vqcpc_last_error()


def test_product_has_no_cpu_fallback():
    """The package must not import the oracle, and must raise when the HIP library is absent."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import vqcpc_bach_amd.hip as h; h.LIB_PATH='/nonexistent.so'; "
            "import os; os.environ.pop('VQCPC_HIP_LIB', None)\n"
            "try:\n    h.load('/nonexistent.so'); print('LOADED')\nexcept h.VqcpcHipError as e:\n    print('RAISED')\n"
            "print('oracle' in ' '.join(sys.modules))") % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True).stdout.split()
    assert out == ['RAISED', 'False'], out
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'vqcpc_bach_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
