# phase offset between the two part-batch streams of the bf16x3 headline workload (20 steps): bash tools/r05_stagger.sh "0 3 6 9 12 18"
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
for rep in 1 2; do for s in $1; do
  timeout 400 python bench.py --steps ${STEPS:-20} --warmup 2 --stagger-ms $s --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[stagger $s ms]', d['ms_per_step'], d['value'], d.get('power',{}).get('package_power_w_median'), d.get('power',{}).get('shader_clock_mhz_median'))"
done; done 2>&1 | tee -a $R/gpurun_out/r05/stagger.log
