"""Build libkdip_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkdip_hip.so")
SOURCES = ["common.cpp", "conv.hip", "conv3.hip", "attention.hip", "gemm.hip", "norm.hip", "elementwise.hip", "unet.hip", "fft.hip", "ops.hip",
           "solver.hip", "api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _newer(src_paths, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in src_paths)


def build(force=False, verbose=True, extra_flags=(), out=None, tag=""):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build" + tag)
    global OUT
    out = out or OUT
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(HERE, "..", "include", h) for h in ("kdip.h", "kdip_internal.h")]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        if force or _newer([src] + headers, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r

    with ThreadPoolExecutor(max_workers=8) as ex:
        for src, r in ex.map(cc, jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed on {src}")
            if verbose:
                print("compiled", os.path.basename(src))
    objs = [os.path.join(objdir, s.rsplit(".", 1)[0] + ".o") for s in SOURCES]
    if force or jobs or not os.path.exists(out):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print("linked", out)
    return out


if __name__ == "__main__":
    # python build.py [--force] [--variant TAG -DFLAG ...]  (variants go to libkdip_hip_TAG.so)
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        tag = sys.argv[i + 1]
        build(force=True, extra_flags=sys.argv[i + 2:], out=os.path.join(HERE, f"libkdip_hip_{tag}.so"), tag="_" + tag)
    else:
        build(force="--force" in sys.argv)
