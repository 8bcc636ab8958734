mkdir -p gpurun_out/r04ac
KDIP_LIB_PATH=$GRAFT_REPO_ROOT/k-diffusion-inverse-problems_amd/libkdip_hip_x3b3.so python -m pytest tests/test_x3_gpu.py -q -m gpu -k "ladder or dgrad" 2>&1 | tail -1 > gpurun_out/r04ac/t.log
for v in "" _x3b3 "" _x3b3; do
  for sh in "8 128 128 256 256 9" "8 256 256 64 64 9" "8 512 512 16 16 9"; do echo -n "[$v] "; KDIP_LIB_PATH=$GRAFT_REPO_ROOT/k-diffusion-inverse-problems_amd/libkdip_hip$v.so python tools/conv_micro.py $sh 20 2 2>&1 | grep -v amdgpu.ids | cut -c1-120; done
  echo -n "[$v] bench "; KDIP_LIB_PATH=$GRAFT_REPO_ROOT/k-diffusion-inverse-problems_amd/libkdip_hip$v.so python bench.py --dtype bf16x3 --steps 20 --warmup 2 --no-cpu-baseline --no-large-batch --no-f32-leg --no-graph-leg --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done >> gpurun_out/r04ac/t.log 2>&1
