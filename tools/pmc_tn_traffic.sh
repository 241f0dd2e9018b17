#!/bin/bash
# HBM fetch bytes of one weight-gradient (TN) launch, ping-pong kernel (XCD-aware tile/split mapping) vs lockstep kernel
# (dispatch-order mapping):   bash tools/pmc_tn_traffic.sh N K
cd /tmp; export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; K=$2
for mode in 1 5; do
rm -rf /tmp/ptt
VQCPC_TN_MODE=$mode timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d /tmp/ptt -- python $REPO/tools/one_gemm_tn.py $N $K > /tmp/ptt.log 2>&1
python - <<PY
import csv, glob
v = []
for f in glob.glob('/tmp/ptt/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_tn_x6' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            v.append(float(r['Counter_Value']))
M, N, K = 557056, $N, $K
alg = 4.0 * M * (N + K)
kb = sum(v) / len(v)
print(f'mode $mode: FETCH_SIZE {kb:.0f} KB raw per launch -> x2 (gfx950 correction) = {2 * kb * 1024 / 1e9:.3f} GB; algorithmic operand bytes {alg / 1e9:.3f} GB; ratio {2 * kb * 1024 / alg:.2f}')
PY
done
