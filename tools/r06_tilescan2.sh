# round 6: tile scan of the 128^2 / 64^2 split-precision 3x3 shapes (tail effect of 1024 / 2048-block launches on 768 block slots)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r06
for s in "8 128 128 128 128" "8 256 256 128 128" "8 128 256 128 128" "8 256 128 128 128" "8 384 128 128 128" "8 256 256 64 64" "8 512 256 64 64"; do for tf in 0 1 2; do
  echo -n "[tile=$tf] "; KDIP_TILE_FORCE=$tf timeout 120 python $R/tools/conv_micro.py $s 9 20 2 2>&1 | tail -1
done; done 2>&1 | tee $R/gpurun_out/r06/tilescan2.log
