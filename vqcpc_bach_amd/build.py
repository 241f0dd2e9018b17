"""Builds libvqcpc_hip.so (the C-ABI library of include/vqcpc.h) with hipcc for gfx950, in-tree.

    python -m vqcpc_bach_amd.build            # or: from vqcpc_bach_amd.build import build; build()
    VQCPC_LAB=1 python -m vqcpc_bach_amd.build   # the LAB build: libvqcpc_hip_lab.so (see below)

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under vqcpc_bach_amd/csrc/_obj/.

Product and lab: the product library holds the kernels the training steps run and reads no tuning variable.  Rejected
kernel designs kept for A/B measurements (gemm_dma.hip, gemm_planes.hip, gemm_sw.hip), the ablation instantiations of the
ping-pong GEMM and the environment switches of the tools under tools/ are compiled only with -DVQCPC_LAB=1 into a SECOND
library, libvqcpc_hip_lab.so (objects under csrc/_obj_lab/), which `hip.load()` picks when VQCPC_LAB=1 is set in the
environment.  The lab library exports a superset of include/vqcpc.h (its `#ifdef VQCPC_LAB` section).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libvqcpc_hip.so')
LAB_OBJ = os.path.join(CSRC, '_obj_lab')
LAB_LIB = os.path.join(HERE, 'libvqcpc_hip_lab.so')
SOURCES = ['util.hip', 'vq.hip', 'nce.hip', 'embed_ln.hip', 'relattn.hip', 'relattn_sub.hip', 'relattn16.hip', 'relattn_x.hip', 'gemm.hip', 'gemm_bf16.hip', 'gemm_grad.hip', 'student.hip', 'gru.hip']
LAB_SOURCES = ['gemm_dma.hip', 'gemm_planes.hip', 'gemm_sw.hip']      # measurement-only translation units
# the VQ argmin must reproduce separately-rounded sub/mul/add: no FMA contraction in that file
EXTRA = {'vq.hip': ['-ffp-contract=off']}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _variant(lab):
    if lab:
        return LAB_OBJ, LAB_LIB, SOURCES + LAB_SOURCES, FLAGS + ['-DVQCPC_LAB=1']
    return OBJ, LIB, SOURCES, FLAGS


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def build(force=False, verbose=True, lab=None):
    """lab=None: the product library, or the lab one when VQCPC_LAB=1 is set; lab=True / False: explicitly."""
    if lab is None:
        lab = os.environ.get('VQCPC_LAB', '0') == '1'
    OBJ, LIB, SOURCES, FLAGS = _variant(lab)
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
