# A/B of one environment switch on one box: tools/ab_env.sh NAME VALUE_A VALUE_B [bench args...]
set -u
NAME=$1; A=$2; B=$3; shift 3
COMMON="--no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary"
for i in 1 2; do
for v in $A $B; do
  env $NAME=$v python bench.py "$@" $COMMON 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$NAME=$v', d['ms_per_step'], d.get('final_loss'))"
done
done
