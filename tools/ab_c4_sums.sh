set -u
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "residual_sum_in_bf16 or residual_operand_in_bf16 or encoder_layer_bf16" 2>&1 | tail -3
COMMON="--config C4 --gemm-mode bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-live-pmc --no-extras --no-secondary"
for i in 1 2; do
for v in 0 1; do
  VQCPC_BF16_SUMS=$v python bench.py $COMMON 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SUMS=$v', d['ms_per_step'], d.get('final_loss'))"
done
done
python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_trainer_gpu.py -m gpu -q -k "bf16" 2>&1 | tail -3
