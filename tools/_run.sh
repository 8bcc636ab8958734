mkdir -p gpurun_out/r04ab
python -m pytest tests/test_parity_gpu.py -q -m gpu -k "hipgraph" 2>&1 | tail -1 > gpurun_out/r04ab/t.log
python bench.py --no-large-batch --no-cpu-baseline > gpurun_out/r04ab/bench.json 2> gpurun_out/r04ab/bench.err
