# build the library from a git revision into libkdip_hip_<tag>.so for same-run A/B:  bash tools/build_ref_variant.sh HEAD prev
set -e
REV=${1:-HEAD}; TAG=${2:-prev}
rm -rf /tmp/kdip_$TAG && mkdir -p /tmp/kdip_$TAG
git archive $REV k-diffusion-inverse-problems_amd/csrc include | tar -x -C /tmp/kdip_$TAG
cd /tmp/kdip_$TAG/k-diffusion-inverse-problems_amd/csrc
for f in common.cpp conv.hip gemm.hip norm.hip elementwise.hip unet.hip fft.hip ops.hip solver.hip api.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -x hip -c $f -o ${f%.*}.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/k-diffusion-inverse-problems_amd/libkdip_hip_$TAG.so *.o
echo built libkdip_hip_$TAG.so from $REV
