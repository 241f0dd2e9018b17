#!/bin/bash
# K = 256 closure experiment (lab build): the LDS-DMA bf16x6 kernel -- whose epilogue already stores LDS-transposed 16-byte rows --
# with its persistent workgroups started 8 phases apart (VQCPC_GEMM_STAGGER = 1024-cycle units per phase), against the same kernel
# in lockstep and the register-staged ping-pong kernel (mode 1).   bash tools/bench_stagger.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export VQCPC_LAB=1
for st in 0 3 6 12; do
    echo "== mode 1 (ping-pong) | mode 17 (LDS-DMA, transposed 16-byte stores) with VQCPC_GEMM_STAGGER=$st"
    VQCPC_GEMM_STAGGER=$st python $REPO/tools/bench_gemm.py 10 1,17 2>&1 | grep "^gemm_nt" | head -7
done
