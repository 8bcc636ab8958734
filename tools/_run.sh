mkdir -p gpurun_out/r04o
for v in "" _x3tw32 _x3ew3 _x3ew7 ""; do
  for sh in "8 128 128 256 256 9" "8 256 128 256 256 9" "8 256 256 64 64 9"; do echo -n "[$v] "; KDIP_LIB_PATH=$GRAFT_REPO_ROOT/k-diffusion-inverse-problems_amd/libkdip_hip$v.so python tools/conv_micro.py $sh 20 2 2>&1 | grep -v amdgpu.ids; done
done > gpurun_out/r04o/micro.log 2>&1
for v in "" _x3tw32 "" _x3tw32; do echo -n "[$v] "; KDIP_LIB_PATH=$GRAFT_REPO_ROOT/k-diffusion-inverse-problems_amd/libkdip_hip$v.so python bench.py --dtype bf16x3 --steps 20 --warmup 2 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > gpurun_out/r04o/ab.log 2>&1
