"""Micro-benchmark of one implicit-GEMM conv shape through the C-ABI test hook (for rocprofv3 --pmc
runs on the dominant kernel; storage-dtype output = the UNet-internal bf16 epilogue).  usage: python tools/conv_micro.py B Cin Cout H W ntaps reps [dtype 0 f32 | 1 bf16 | 2 bf16x3]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdip_amd._lib as L
B, Cin, Cout, H, W, ntaps, reps = [int(a) for a in sys.argv[1:8]]
dtype = int(sys.argv[8]) if len(sys.argv) > 8 else 1
lib = L.load()
k = 3 if ntaps == 9 else 1
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, W, generator=g).cuda()
w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * ntaps) ** 0.5).contiguous()
b = torch.randn(Cout, generator=g)
y = torch.empty(B, Cout, H, W, device="cuda")
def run(n):
    for _ in range(n):
        L.check(lib.kdip_test_conv(L.stream(), dtype, ntaps, L.ptr(x), B, Cin, H, W, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), Cout, 0, L.ptr(y), 1))
run(3)
# the hook allocates / packs per call: time the conv launches themselves with the library's per-launch HIP-event records
import tempfile, csv
L.check(lib.kdip_profile_enable(1)); run(reps)
dump = os.path.join(tempfile.gettempdir(), f"conv_micro_{os.getpid()}.csv")
L.check(lib.kdip_profile_dump(dump.encode())); L.check(lib.kdip_profile_enable(0))
rows = [r for r in csv.DictReader(open(dump)) if r["class"].startswith("conv")]
us = sum(float(r["us"]) for r in rows) / max(len(rows), 1)
fl = 2.0 * B * H * W * Cin * Cout * ntaps
print(f"conv dtype={dtype} B={B} {Cin}->{Cout} @{H}x{W} taps={ntaps} [{rows[0]['class'] if rows else '?'}]: {us:.1f} us/launch ({len(rows)} launches), {fl / us / 1e6:.1f} TFLOP/s, mean |y| {float(y.abs().mean()):.4f}")
