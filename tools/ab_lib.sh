# A/B of two builds of the kernel library on one box: tools/ab_lib.sh <bench args...>   (old = vqcpc_bach_amd/libvqcpc_hip_old.so)
set -u
COMMON="--no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary"
for i in 1 2; do
for v in old new; do
  if [ $v = old ]; then export VQCPC_HIP_LIB=$PWD/vqcpc_bach_amd/libvqcpc_hip_old.so; else unset VQCPC_HIP_LIB; fi
  python bench.py "$@" $COMMON 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('final_loss'))"
done
done
