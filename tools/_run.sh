mkdir -p gpurun_out/r04aa
python -m pytest tests/test_kernels_gpu.py tests/test_x3_gpu.py tests/test_parity_gpu.py -q -m gpu -k "conv or x3 or guided_calls_golden" 2>&1 | tail -1 > gpurun_out/r04aa/t.log
for i in 1 2; do python bench.py --dtype bf16x3 --steps 20 --warmup 2 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done >> gpurun_out/r04aa/t.log
