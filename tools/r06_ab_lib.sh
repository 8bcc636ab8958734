# round 6 A/B of library variants (build: VFILES=conv tools/variants.sh TAG "-D..." ...) on the bf16x3 headline workload, 20 steps, interleaved twice
# usage: bash tools/r06_ab_lib.sh NAME "" _tagA _tagB   -> gpurun_out/r06/ab_NAME.log
R=${GRAFT_REPO_ROOT:-.}; mkdir -p $R/gpurun_out/r06
TAG=$1; shift
for rep in 1 2; do for v in "$@"; do
  KDIP_LIB_PATH=$R/k-diffusion-inverse-problems_amd/libkdip_hip$v.so timeout 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['all_conv_classes']; h=d['roofline'].get('hbm_bound_classes',{}); g=d['roofline']['top_conv_launch_groups'][0]; print('[lib$v]', d['ms_per_step'], d.get('power',{}).get('shader_clock_mhz_median'), g['avg_launch_us'], {k:(v['ms'],v['launches']) for k,v in c.items()}, {k:(v['ms'],v['launches']) for k,v in h.items() if k.startswith('gn')})"
done; done 2>&1 | tee $R/gpurun_out/r06/ab_$TAG.log
