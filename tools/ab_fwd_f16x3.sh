set -u
python -m pytest tests/test_kernels_gpu.py tests/test_graphs_gpu.py tests/test_dropin_gpu.py -m gpu -q -x -k "f16x3 or defaults_are" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_fwd_f16x3.log 2>&1; grep '"metric"' gpurun_out/r05_bench_fwd_f16x3.log | cut -c1-300
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
