#!/bin/bash
# kernel-trace summary of the C1 bench command only: bash tools/profile_c1.sh <tag> [extra bench args]
set -u
TAG=${1:-tmp}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -- python $REPO/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary "$@" > $OUT/c1_step_bench.log 2>&1
db=$(find /tmp/prof_c1 -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py $db 0.3 > $OUT/c1_step_kernel_stats.csv
python $REPO/tools/rocpd_stats.py $db 0.3 --by-grid > $OUT/c1_step_kernel_stats_by_grid.csv
python $REPO/tools/rocpd_timeline.py $db 0.5 > $OUT/c1_step_timeline.txt
grep '"metric"' $OUT/c1_step_bench.log | tail -1 | cut -c1-200
