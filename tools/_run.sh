mkdir -p gpurun_out/r04w
python -m pytest tests/test_x3_gpu.py tests/test_parity_gpu.py -q -m gpu -s -k "x3 or churn or sampler_golden" 2>&1 | grep -E "wide-range|per-launch|churn.*x3|passed|failed|assert|Error" | cut -c1-260 > gpurun_out/r04w/t.log
python bench.py --dtype bf16x3 --steps 20 --warmup 2 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" > gpurun_out/r04w/bench.log
