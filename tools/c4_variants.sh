#!/bin/bash
# build conv4 A/B variants: tools/c4_variants.sh TAG "-DFLAG=1 ..." [TAG2 "..."] ...   (-> libkdip_hip_TAG.so; only conv4.hip / conv3.hip are recompiled)
set -e
cd "$(dirname "$0")/../k-diffusion-inverse-problems_amd"
python build.py >/dev/null
while [ $# -gt 1 ]; do
  tag=$1; flags=$2; shift 2
  (
    mkdir -p build_c4$tag
    for f in ${C4V_FILES:-conv4}; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -x hip -c csrc/$f.hip -o build_c4$tag/$f.o
    done
    objs=""
    for o in build/*.o; do b=$(basename $o); if [ -f build_c4$tag/$b ]; then objs="$objs build_c4$tag/$b"; else objs="$objs $o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libkdip_hip_$tag.so $objs
    echo built $tag
  ) &
done
wait
