"""Which CU-mask bit addresses which XCD / CU: for a few masks, launch a census kernel on a masked stream and print the set of
(XCC, SE, CU) ids its blocks ran on.  usage: python tools/cu_census.py"""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, kdip_amd._lib as L
lib = L.load(); L.require_gpu()
def census(words, blocks=1024):
    m = (C.c_uint * len(words))(*words)
    st = C.c_void_p()
    L.check(lib.kdip_stream_create_cu_mask(0, m, len(words), C.byref(st)))
    out = (C.c_uint * (2 * blocks))()
    L.check(lib.kdip_debug_cu_census(st, blocks, out))
    L.check(lib.kdip_stream_destroy(st))
    ids = collections.Counter()
    for b in range(blocks):
        hw, xcc = out[2 * b], out[2 * b + 1] & 15
        cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        ids[(xcc, se, sh, cu)] += 1
    return ids
full = census([0xffffffff] * 8)
print("full mask: distinct (xcc, se, sh, cu):", len(full), "xccs:", sorted({k[0] for k in full}))
for name, words in (("bits 0-31", [0xffffffff, 0, 0, 0, 0, 0, 0, 0]), ("bits 32-63", [0, 0xffffffff, 0, 0, 0, 0, 0, 0]), ("even bits", [0x55555555] * 8),
                    ("bits 0-127", [0xffffffff] * 4 + [0] * 4), ("bits 128-255", [0] * 4 + [0xffffffff] * 4), ("bits = 0 mod 8", [0x01010101] * 8),
                    ("bits 0-3 mod 8", [0x0f0f0f0f] * 8)):
    ids = census(words)
    by = collections.Counter(k[0] for k in ids)
    print(f"{name:16s}: {len(ids):3d} CUs; per XCC: {dict(sorted(by.items()))}")
