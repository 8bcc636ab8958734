mkdir -p gpurun_out/r04h
python -m pytest tests/test_parity_gpu.py tests/test_harness_gpu.py tests/test_mid_gpu.py tests/test_x3_gpu.py -q -m gpu -s -k "guided_calls_golden or sampler_golden or churn or harness_parity or mid or x3 or graph or second_vjp or custom_mat" 2>&1 | grep -E "^\[|ode, tiny|churn|harness.*x3|mid |passed|failed|assert|Error" | cut -c1-300 > gpurun_out/r04h/parity.log
python -m pytest tests/test_fullsize_gpu.py -q -m gpu -s -k "e2e or configs_fullsize" 2>&1 | grep -E "^e2e|^cfg|passed|failed|assert" | cut -c1-600 > gpurun_out/r04h/fullsize.log
python bench.py --dtype bf16x3 --steps 10 --warmup 2 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg > gpurun_out/r04h/x3_bench.json 2> gpurun_out/r04h/x3_bench.err
