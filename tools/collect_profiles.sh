#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03
# kernel-trace summaries of the bench command for every measured configuration + the two PMC passes (FETCH_SIZE, WRITE_SIZE;
# counters in their own runs, kernel trace only) for the HBM traffic of the GEMM launches.  Raw output: gpurun_out/<tag>/.
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary --no-long"
run_trace() {   # name, bench args
    local name=$1; shift
    rm -rf /tmp/prof_$name
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python $REPO/bench.py "$@" $COMMON > $OUT/${name}_bench.log 2>&1
    local db=$(find /tmp/prof_$name -name "*.db" | head -1)
    python $REPO/tools/rocpd_stats.py $db 0.3 > $OUT/${name}_kernel_stats.csv
    python $REPO/tools/rocpd_stats.py $db 0.3 --by-grid > $OUT/${name}_kernel_stats_by_grid.csv
    python $REPO/tools/rocpd_timeline.py $db 0.5 > $OUT/${name}_timeline.txt
    grep '"metric"' $OUT/${name}_bench.log | tail -1 | cut -c1-300
    tail -3 $OUT/${name}_kernel_stats.csv
}
run_trace c1_step --steps 20 --warmup 4
run_trace c3_step --config C3 --steps 20 --warmup 4
run_trace dec_step --config DEC --steps 20 --warmup 4
run_trace c4_bf16_step --config C4 --gemm-mode bf16 --steps 6 --warmup 3
# PMC passes (eager launches: counters are collected per dispatch)
for c in FETCH_SIZE WRITE_SIZE; do
    d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    rm -rf /tmp/pmc/$d
    timeout 900 rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pmc/$d -- python $REPO/bench.py --steps 2 --warmup 1 --no-graph $COMMON > $OUT/pmc_$d.log 2>&1
done
# bench.py --steps 2 --warmup 1 --no-graph = 1 warm-up + 2 bare train_step + 2 epoch steps = 5 steps
python $REPO/tools/pmc_hbm_traffic.py /tmp/pmc 5 > $OUT/gemm_hbm_traffic.json
cat $OUT/gemm_hbm_traffic.json | head -20
