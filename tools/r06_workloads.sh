#!/bin/bash
# round 6: every BASELINE config at its per-GPU batch on one GPU (bench.py --workload), bf16x3, 10 timed steps each
mkdir -p gpurun_out/r06
python -m pytest tests/test_configs_b16_gpu.py -x -q -m gpu -s > gpurun_out/r06/configs_b16_tests.log 2>&1
tail -3 gpurun_out/r06/configs_b16_tests.log
for w in cfg0 cfg2 cfg3 cfg4; do
  st=10; [ $w = cfg0 ] && st=20
  KDIP_PROFILE_DUMP=gpurun_out/r06/dispatch_$w.csv timeout 900 python bench.py --workload $w --dtype ${DTYPE:-bf16x3} --steps $st --warmup 2 > gpurun_out/r06/bench_$w.json 2> gpurun_out/r06/bench_$w.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r06/bench_$w.json').read().strip().splitlines()[-1])
    print('$w', d['dtype'], 'fallbacks', d.get('x3_fallbacks'), d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['achieved_tflops_whole_step'], 'TFLOP/s', d.get('workspace_gb'), d['roofline']['kernel'][-40:], d['roofline']['achieved'])
except Exception as e:
    print('$w FAILED', e); print(open('gpurun_out/r06/bench_$w.err').read()[-1500:])
PY
done
