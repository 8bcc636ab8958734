mkdir -p gpurun_out/r04k
for i in 1 2 3; do python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" 2>&1 | tail -1; done > gpurun_out/r04k/unit.log
python -m pytest tests/test_parity_gpu.py -q -m gpu -k "guided_calls_golden or sampler_golden" 2>&1 | tail -1 >> gpurun_out/r04k/unit.log
for sh in "8 512 512 8 8 9" "8 512 512 16 16 9" "8 256 256 16 16 9" "8 1024 512 8 8 9"; do for v in "" _sk0; do echo -n "[$v] "; KDIP_LIB_PATH=$GRAFT_REPO_ROOT/k-diffusion-inverse-problems_amd/libkdip_hip$v.so python tools/conv_micro.py $sh 30 1 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r04k/micro.log 2>&1
KDIP_AB_BATCH=16 KDIP_AB_STEPS=20 bash tools/ab_variants.sh "" _sk0 > gpurun_out/r04k/ab.log 2>&1
