# conv3 A/B of variant libraries (build.py --variant TAG -D...); usage: bash tools/conv3_abl.sh "" _tagA _tagB ...
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$R/k-diffusion-inverse-problems_amd
m() { KDIP_LIB_PATH=$L/libkdip_hip$1.so python $R/tools/conv3_micro.py ${@:2} 2>&1 | tail -1 | sed "s/^/[$1] /"; }
for rep in 1 2; do for v in "$@"; do
m "$v" 8 128 128 256 256 0 0 0
m "$v" 8 128 128 256 256 1 1 1
m "$v" 8 128 128 256 256 0 2 0
m "$v" 8 256 128 256 256 1 1 0
m "$v" 8 128 128 128 128 1 1 0
done; done
