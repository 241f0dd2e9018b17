#!/bin/bash
# Measurement builds of the kernel library that differ in ONE compile-time choice of gemm_bf16.hip (never shipped, never loaded
# by default; select with VQCPC_HIP_LIB=<path>):   tools/_lab/libvqcpc_hip_<tag>.so
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
OBJ=$REPO/vqcpc_bach_amd/csrc/_obj
mkdir -p $REPO/tools/_lab
python -c "import sys; sys.path.insert(0, '$REPO'); from vqcpc_bach_amd import build; build.build(verbose=False)"
for spec in "nt: nt" "sc1: sc1" "sc0sc1: sc0 sc1"; do
    tag=${spec%%:*}; pol=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "-DVQ_BF16_STORE_POL=\"$pol\"" \
        -c $REPO/vqcpc_bach_amd/csrc/gemm_bf16.hip -o $REPO/tools/_lab/gemm_bf16_$tag.o
    objs=$(ls $OBJ/*.o | grep -v gemm_bf16.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/_lab/libvqcpc_hip_$tag.so $objs $REPO/tools/_lab/gemm_bf16_$tag.o
    echo built tools/_lab/libvqcpc_hip_$tag.so
done
