mkdir -p gpurun_out/r04q
for sh in "8 512 512 8 8 9" "8 512 512 16 16 9" "8 256 256 16 16 9" "8 256 256 32 32 9" "8 512 512 32 32 9" "8 256 256 64 64 9" "8 128 128 64 64 9" "8 128 128 128 128 9"; do
  for f in 0 1 2 3; do for dt in 2 1; do echo -n "[force $f] "; KDIP_TILE_FORCE=$f python tools/conv_micro.py $sh 20 $dt 2>&1 | grep -v amdgpu.ids | cut -c1-130; done; done
done > gpurun_out/r04q/tiles.log 2>&1
