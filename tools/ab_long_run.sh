# 300 training steps from the same initial state and the same batches: six products everywhere vs the training defaults (f16x3)
set -u
COMMON="--steps 300 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary"
python bench.py $COMMON --grad-arith six --fwd-arith six 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('six  ', d['ms_per_step'], 'final loss', d.get('final_loss'))"
python bench.py $COMMON 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16x3', d['ms_per_step'], 'final loss', d.get('final_loss'), 'saturations', d.get('f16x3_scale_saturations'))"
python bench.py $COMMON --gemm-mode f32 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 MFMA', d['ms_per_step'], 'final loss', d.get('final_loss'))"
