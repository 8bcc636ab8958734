# stream-count sweep of the bf16x3 headline workload (20 steps): bash tools/r05_streams.sh
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
for rep in 1 2; do for s in 2 1 3 4; do
  timeout 400 python bench.py --steps 20 --warmup 2 --streams $s --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[streams $s]', d['ms_per_step'], d['value'], d.get('power',{}).get('package_power_w_median'), d.get('power',{}).get('shader_clock_mhz_median'), d.get('x3_saturated'))"
done; done 2>&1 | tee $R/gpurun_out/r05/streams.log
