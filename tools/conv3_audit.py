"""Audit of the compiled conv3 kernels (inline-asm load pipeline, csrc/conv3.hip): compiles conv3.hip with -save-temps and, per
kernel instantiation, reports VGPR count, spills, and every scratch access or compiler-inserted vmcnt(0) INSIDE the stage loop
(between the first and the last stage barrier) -- either would break / drain the hand-counted load pipeline.
usage: python tools/conv3_audit.py [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "k-diffusion-inverse-problems_amd", "csrc", "conv3.hip")
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-x", "hip", "-c", src,
                "-o", os.path.join(tmp, "c.o"), "-save-temps"] + sys.argv[1:], cwd=tmp, check=True, capture_output=True)
asm = open([os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]).read()
bad = 0
for m in re.finditer(r"^(_ZN4kdip\S*conv3_kernel\S*):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    # the rare table-reload path (block's next tile in another image) uses compiler-managed loads by design: skip it
    keep, skip = [], False
    for l in body:
        if "C3_RARE_BEGIN" in l: skip = True
        if not skip: keep.append(l)
        if "C3_RARE_END" in l: skip = False
    body = keep
    bars = [i for i, l in enumerate(body) if "s_barrier" in l]
    meta = re.search(r"\.name:\s+%s\n(?:(?!\.name:).)*?\.vgpr_count:\s+(\d+)(?:(?!\.name:).)*?\.vgpr_spill_count:\s+(\d+)" % re.escape(name), asm, re.S)
    # stage loop = the longest run of barriers whose gaps are < 400 lines
    lo, hi = bars[0], bars[-1]
    runs, cur = [], [bars[0]]
    for a, b in zip(bars, bars[1:]):
        if b - a < 400: cur.append(b)
        else: runs.append(cur); cur = [b]
    runs.append(cur)
    run = max(runs, key=len)
    lo, hi = run[0], run[-1]
    inloop = [(i, l.strip()) for i, l in enumerate(body[lo:hi], lo) if "scratch_" in l or re.search(r"s_waitcnt vmcnt\(0\)", l)]
    tag = re.search(r"conv3_kernelILi(\d)ELi(\d)ELb(\d)", name).groups()
    print(f"TF={tag[0]} STM={tag[1]} RES={tag[2]}: vgpr {meta.group(1) if meta else '?'} spilled {meta.group(2) if meta else '?'}; stage barriers {len(run)}; in-loop scratch / vmcnt(0): {len(inloop)}")
    for i, l in inloop[:6]:
        print("     line", i, l)
    bad += len(inloop)
print("AUDIT", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
