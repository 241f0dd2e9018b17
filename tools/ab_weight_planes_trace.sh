set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary --no-long"
for p in 1 0; do
  rm -rf /tmp/prof_p$p
  VQCPC_WEIGHT_PLANES=$p timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p$p -- python $REPO/bench.py --steps 20 --warmup 4 $COMMON > $OUT/p${p}_bench.log 2>&1
  db=$(find /tmp/prof_p$p -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py $db 0.3 > $OUT/p${p}_kernel_stats.csv
  grep -i "g3\|planes\|TOTAL\|span" $OUT/p${p}_kernel_stats.csv
done
