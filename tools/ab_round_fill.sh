set -u
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary"
for i in 1 2; do
for v in 0.8 0.7; do
  VQCPC_GRAD_ROUND_FILL=$v python bench.py $COMMON 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FILL=$v', d['ms_per_step'], d.get('final_loss'))"
done
done
