#!/bin/bash
# build kernel A/B variants: VFILES="conv conv3" tools/variants.sh TAG "-DFLAG=1 ..." [TAG2 "..."] ...   (-> libkdip_hip_TAG.so; only the
# translation units named in VFILES (default: conv) are recompiled with the flags, the rest comes from the current build/)
set -e
cd "$(dirname "$0")/../k-diffusion-inverse-problems_amd"
python build.py >/dev/null
while [ $# -gt 1 ]; do
  tag=$1; flags=$2; shift 2
  (
    mkdir -p build_v$tag
    for f in ${VFILES:-conv}; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -x hip -c csrc/$f.hip -o build_v$tag/$f.o
    done
    objs=""
    for o in build/*.o; do b=$(basename $o); if [ -f build_v$tag/$b ]; then objs="$objs build_v$tag/$b"; else objs="$objs $o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libkdip_hip_$tag.so $objs
    echo built $tag
  ) &
done
wait
