"""Builds libvqcpc_hip.so (the C-ABI library of include/vqcpc.h) with hipcc for gfx950, in-tree.

    python -m vqcpc_bach_amd.build            # or: from vqcpc_bach_amd.build import build; build()

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under vqcpc_bach_amd/csrc/_obj/.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libvqcpc_hip.so')
SOURCES = ['util.hip', 'vq.hip', 'nce.hip', 'embed_ln.hip', 'relattn.hip', 'relattn_sub.hip', 'relattn16.hip', 'relattn_x.hip', 'gemm.hip', 'gemm_dma.hip', 'gemm_bf16.hip', 'gemm_planes.hip', 'gemm_sw.hip', 'student.hip', 'gru.hip']
# the VQ argmin must reproduce separately-rounded sub/mul/add: no FMA contraction in that file
EXTRA = {'vq.hip': ['-ffp-contract=off']}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'gemm_common.h'), os.path.join(HERE, '..', 'include', 'vqcpc.h')]
    hdr_m = max(os.path.getmtime(h) for h in headers)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            jobs.append([hipcc] + FLAGS + EXTRA.get(src, []) + ['-c', s, '-o', o])
    rebuilt = bool(jobs)
    if jobs:                                   # independent translation units: compile them concurrently
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(run, jobs))
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
