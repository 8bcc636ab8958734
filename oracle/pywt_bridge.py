"""DEV-ONLY: the real PyWavelets for the imported reference.

TEST INFRASTRUCTURE (build container only; never imported by the product, by tests, by bench.py or by smoke()).

The reference imports `pywt` (condition/utils.py:6) and calls four of its functions in DiscreteWaveletTransform (:116-132):
wavedec2, coeffs_to_array, array_to_coeffs, waverec2.  The system interpreter the reference is imported with has no PyWavelets; a
second interpreter of this image, /opt/conda/bin/python3.9, does (1.1.1).  This module starts that interpreter once as a worker and
forwards those four calls to the REAL package over a pipe (arrays travel as (dtype, shape, bytes): the two interpreters carry
different numpy major versions, whose pickles are not interchangeable).  It is a wire to the real library, not a stand-in for it: when
the worker cannot be started, `available()` is False and oracle/refimport.py keeps its empty stub (the DWT path of the reference
then cannot run, as before)."""
import os
import pickle
import struct
import subprocess

import numpy as np

CONDA_PY = os.environ.get("KDIP_PYWT_PYTHON", "/opt/conda/bin/python3.9")

_WORKER = r'''
import pickle, struct, sys, warnings
warnings.filterwarnings("ignore")
import numpy as np, pywt
def dec(o):
    if isinstance(o, dict) and "__nd__" in o:
        d, s, b = o["__nd__"]; return np.frombuffer(b, dtype=np.dtype(d)).reshape(s).copy()
    if isinstance(o, dict): return {k: dec(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)): return type(o)(dec(v) for v in o)
    return o
def enc(o):
    if isinstance(o, np.ndarray): return {"__nd__": (o.dtype.str, o.shape, np.ascontiguousarray(o).tobytes())}
    if isinstance(o, dict): return {k: enc(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)): return type(o)(enc(v) for v in o)
    return o
inp, out = sys.stdin.buffer, sys.stdout.buffer
out.write(struct.pack("<Q", 0)); out.flush()
while True:
    h = inp.read(8)
    if len(h) < 8: break
    fn, args, kw = pickle.loads(inp.read(struct.unpack("<Q", h)[0]))
    try: res = ("ok", enc(getattr(pywt, fn)(*dec(args), **dec(kw))))
    except Exception as e: res = ("err", repr(e))
    b = pickle.dumps(res, protocol=2)
    out.write(struct.pack("<Q", len(b))); out.write(b); out.flush()
'''

_proc = None


def _enc(o):
    if isinstance(o, np.ndarray):
        return {"__nd__": (o.dtype.str, tuple(o.shape), np.ascontiguousarray(o).tobytes())}
    if isinstance(o, dict):
        return {k: _enc(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_enc(v) for v in o)
    return o


def _dec(o):
    if isinstance(o, dict) and "__nd__" in o:
        d, s, b = o["__nd__"]
        return np.frombuffer(b, dtype=np.dtype(d)).reshape(s).copy()
    if isinstance(o, dict):
        return {k: _dec(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_dec(v) for v in o)
    return o


def available():
    global _proc
    if _proc is not None:
        return True
    if not os.path.exists(CONDA_PY):
        return False
    try:
        p = subprocess.Popen([CONDA_PY, "-u", "-c", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        if len(p.stdout.read(8)) < 8:          # the worker's hello (sent after `import pywt` succeeded)
            return False
        _proc = p
        return True
    except Exception:
        return False


def _call(fn, *args, **kw):
    assert available(), "PyWavelets worker is not available"
    b = pickle.dumps((fn, _enc(args), _enc(kw)), protocol=2)
    _proc.stdin.write(struct.pack("<Q", len(b))); _proc.stdin.write(b); _proc.stdin.flush()
    n = struct.unpack("<Q", _proc.stdout.read(8))[0]
    tag, res = pickle.loads(_proc.stdout.read(n))
    if tag != "ok":
        raise RuntimeError("pywt." + fn + ": " + res)
    return _dec(res)


def wavedec2(*a, **k):
    return _call("wavedec2", *a, **k)


def coeffs_to_array(*a, **k):
    return _call("coeffs_to_array", *a, **k)


def array_to_coeffs(*a, **k):
    return _call("array_to_coeffs", *a, **k)


def waverec2(*a, **k):
    return _call("waverec2", *a, **k)
