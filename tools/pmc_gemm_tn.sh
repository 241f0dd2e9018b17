#!/bin/bash
# Effective clock and MFMA-pipe occupancy of the bf16x6 weight-gradient (TN) kernel: bash tools/pmc_gemm_tn.sh N K
cd /tmp; export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; K=$2
rm -rf /tmp/pgt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -f csv -d /tmp/pgt -- python $REPO/tools/one_gemm_tn.py $N $K > /tmp/pgt.log 2>&1
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter(); dur = []
for f in glob.glob('/tmp/pgt/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_tn_x6_p' in r['Kernel_Name'] or 'gemm_tn_x6_256' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for f in glob.glob('/tmp/pgt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_tn_x6_p' in r['Kernel_Name'] or 'gemm_tn_x6_256' in r['Kernel_Name']:
            dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
M, N, K = 557056, $N, $K
waves = 256 * 8
cyc = 4.0 * tot['SQ_WAVE_CYCLES'] / n['SQ_WAVE_CYCLES'] / waves
t = sum(dur) / len(dur) * 1e-9
mfma_cyc = 6.0 * M * N * K / (32 * 32 * 16) * 32 / 1024
print(f'gemm_tn (256-tile) {M}x{N}x{K}: {t * 1e6:.1f} us, cycles per wave {cyc:.0f} -> effective clock {cyc / t / 1e9:.3f} GHz; MFMA pipe busy '
      f'{100 * mfma_cyc / cyc:.1f} % of the cycles; {2.0 * M * N * K / t / 1e12:.1f} TFLOP/s (profiled pass); VALU insts '
      f'{tot["SQ_INSTS_VALU"] / n["SQ_INSTS_VALU"]:.0f}, LDS insts {tot["SQ_INSTS_LDS"] / n["SQ_INSTS_LDS"]:.0f}')
PY
