"""Per-kernel register / spill / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py csrc/conv.hip [filter-substring] [-DFLAG ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if os.path.isabs(sys.argv[1]) else os.path.join(ROOT, "k-diffusion-inverse-problems_amd", sys.argv[1])
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
flags = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o", os.path.join(td, "o.o"),
                        "-Rpass-analysis=kernel-resource-usage"] + flags, capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPR Spill): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
if r.returncode:
    sys.stderr.write(r.stderr[-3000:])
for k, v in rows.items():
    if flt and not all(f in k for f in flt):
        continue
    print(f"{k[:110]:110s} VGPR {v.get('VGPRs', -1):3d} AGPR {v.get('AGPRs', 0):3d} scratch {v.get('ScratchSize', 0):4d} occ {v.get('Occupancy', 0)}")
