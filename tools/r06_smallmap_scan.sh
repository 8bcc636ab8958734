# round 6: split-precision small-map / 1x1 micro scan (tools/conv_micro.py through the C-ABI test hook): tile forcing x library variants
# usage (GPU box): bash tools/r06_smallmap_scan.sh "" _bd4 _subs ...   -> gpurun_out/r06/smallmap_scan.log
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r06
SH3="8 256 256 32 32|8 512 512 8 8|8 512 512 16 16|8 512 512 32 32|8 128 128 64 64|8 256 256 16 16|8 768 256 32 32|8 1024 512 8 8|8 256 256 64 64"
SH1="8 1024 512 16 16|8 512 512 16 16|8 1536 512 16 16|8 512 1536 16 16|8 1024 512 8 8|8 768 256 32 32|8 512 256 64 64|8 256 128 256 256|8 128 256 256 256"
for v in "$@"; do
  export KDIP_LIB_PATH=$R/k-diffusion-inverse-problems_amd/libkdip_hip$v.so
  S3=$SH3; S1=$SH1; [ "$v" = "_subs" ] && S3=""; [ "$v" = "_bd4" ] && S1=""
  IFS='|'; for s in $S3; do IFS=' '; for tf in 0 1 2 3; do
    echo -n "[lib$v tile=$tf] "; KDIP_TILE_FORCE=$tf timeout 120 python $R/tools/conv_micro.py $s 9 20 2 2>&1 | tail -1
  done; IFS='|'; done
  IFS='|'; for s in $S1; do IFS=' '; for tf in 0 2 3; do
    echo -n "[lib$v tile=$tf] "; KDIP_TILE_FORCE=$tf timeout 120 python $R/tools/conv_micro.py $s 1 20 2 2>&1 | tail -1
  done; IFS='|'; done
  IFS=' '
done 2>&1 | tee $R/gpurun_out/r06/smallmap_scan.log
