# round 6 A/B of bench.py flags on the headline workload (20 steps, interleaved twice): bash tools/r06_ab_flags.sh NAME "" "--stream-prio" ...
R=${GRAFT_REPO_ROOT:-.}; mkdir -p $R/gpurun_out/r06
TAG=$1; shift
for rep in 1 2; do for v in "$@"; do
  timeout 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg --no-roofline $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[flags $v]', d['ms_per_step'], d.get('power',{}).get('shader_clock_mhz_median'), d.get('power',{}).get('package_power_w_median'), 'fallbacks', d.get('x3_fallbacks'))"
done; done 2>&1 | tee $R/gpurun_out/r06/ab_$TAG.log
