#!/bin/bash
# PMC passes over one isolated bf16x6 NT GEMM (557056 x N x K, bias epilogue): bash tools/pmc_gemm.sh N K [ABL] [gemm mode: 1 = ping-pong, 33 = one wave per SIMD]
cd /tmp; export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; K=$2; export VQCPC_PP_ABL=${3:-0}; export VQCPC_ONE_GEMM_MODE=${4:-1}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  rm -rf /tmp/pg
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -f csv -d /tmp/pg -- python $REPO/tools/one_gemm.py $N $K > /tmp/pg.log 2>&1
  python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('/tmp/pg/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_nt_x6_' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
dur = []
for f in glob.glob('/tmp/pg/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_nt_x6_' in r['Kernel_Name']:
            dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
if dur: print(f'kernel duration in this pass: {sum(dur) / len(dur) / 1e3:.1f} us (avg of {len(dur)})')
if dur and 'SQ_WAVE_CYCLES' in tot:
    # SQ_WAVE_CYCLES counts quad-cycles summed over waves; every wave of these persistent kernels lives for the whole launch
    import os
    M, N, K = 557056, $N, $K
    waves = 256 * (4 if os.environ.get('VQCPC_ONE_GEMM_MODE') == '33' else 8)
    cyc = 4.0 * tot['SQ_WAVE_CYCLES'] / n['SQ_WAVE_CYCLES'] / waves
    t = sum(dur) / len(dur) * 1e-9
    mfma_cyc = 6.0 * M * N * K / (32 * 32 * 16) * 32 / 1024          # 32 cycles each, 1024 SIMDs
    print(f'cycles per wave {cyc:.0f} -> effective shader clock {cyc / t / 1e9:.3f} GHz; MFMA pipe busy {100 * mfma_cyc / cyc:.1f} % '
          f'of the cycles; {2.0 * M * N * K / t / 1e12:.1f} TFLOP/s in this (profiled) pass')
for k in tot: print(f'{k:32s} {tot[k] / n[k]:16.0f}  per launch ({n[k]} launches)')
PY
done
