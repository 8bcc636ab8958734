# round 5 A/B: (row pitch | natural pitch) x (fixed-order reductions | atomics) on the bf16x3 headline workload, 20 steps, interleaved
# usage: bash tools/r05_ab.sh  -> gpurun_out/r05/ab.log
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
for rep in 1 2; do for v in "" _norowpad; do for det in 1 0; do
  KDIP_DET=$det KDIP_LIB_PATH=$R/k-diffusion-inverse-problems_amd/libkdip_hip$v.so timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['all_conv_classes']; h=d['roofline'].get('hbm_bound_classes',{}); g=d['roofline']['top_conv_launch_groups'][0]; print('[lib$v det=$det]', d['ms_per_step'], g['avg_launch_us'], {k:(v['ms'],v['tflops']) for k,v in c.items()}, {k:v['ms'] for k,v in h.items() if k.startswith('gn')})"
done; done; done 2>&1 | tee $R/gpurun_out/r05/ab.log
