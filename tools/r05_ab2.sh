# round 5 A/B of kernel-variant builds on the bf16x3 headline workload (20 steps, interleaved twice): bash tools/r05_ab2.sh "" _kc16 ...
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05
for rep in 1 2; do for v in "$@"; do
  KDIP_LIB_PATH=$R/k-diffusion-inverse-problems_amd/libkdip_hip$v.so timeout 400 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['all_conv_classes']; h=d['roofline'].get('hbm_bound_classes',{}); print('[lib$v]', d['ms_per_step'], [(g['HW'],g['Cin'],g['Cout'],g['avg_launch_us']) for g in d['roofline']['top_conv_launch_groups'][:6]], {k:(v['ms'],v['tflops']) for k,v in c.items()})"
done; done 2>&1 | tee -a $R/gpurun_out/r05/ab2.log
