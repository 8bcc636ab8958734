# usage: bash tools/ab_variants.sh "" _tw16 ...   (A/B of kernel-variant builds, interleaved twice)
for rep in 1 2; do for v in "$@"; do
  KDIP_LIB_PATH=$GRAFT_REPO_ROOT/k-diffusion-inverse-problems_amd/libkdip_hip$v.so timeout 300 python bench.py --steps ${KDIP_AB_STEPS:-4} --warmup 2 --no-cpu-baseline --batch ${KDIP_AB_BATCH:-128} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['roofline']['all_conv_classes']; h=d['roofline'].get('hbm_bound_classes',{}); print('[$v]', d['ms_per_step'], {k:v['tflops'] for k,v in c.items()}, {k:v['GBps'] for k,v in h.items()})"
done; done
