"""Folds the rocprofv3 counter CSVs of tools/calib/run_calib.sh into a table: counter bytes vs known bytes per access pattern."""
import csv, glob, os, sys, collections
O = sys.argv[1]
GiB = float(1 << 30)
known = {"read_wide": GiB, "read_rows<4, 8>": GiB, "read_rows<4, 1>": GiB / 8, "read_rows<8, 1>": GiB / 4, "read_rows<16, 2>": GiB,
         "write_wide": GiB, "write_rows256": GiB}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(O, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        name = next((k for k in known if k.replace(" ", "") in kn.replace(" ", "")), None)
        if name:
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("FETCH_SIZE / WRITE_SIZE calibration on known byte counts (1 GiB buffer = 4 x Infinity Cache; mean of 3 launches; counters in KiB as rocprofv3 reports them)")
print("pattern | known MiB | FETCH_SIZE MiB | known/FETCH | WRITE_SIZE MiB | known/WRITE | EA RDREQ (32B) | bytes/RDREQ | EA WRREQ (64B) | bytes/WRREQ")
for name, kb in known.items():
    m = {k: sum(v) / len(v) for k, v in vals[name].items()}
    fs = m.get("FETCH_SIZE"); ws = m.get("WRITE_SIZE")
    rq = m.get("TCC_EA0_RDREQ_sum"); r32 = m.get("TCC_EA0_RDREQ_32B_sum"); wq = m.get("TCC_EA0_WRREQ_sum"); w64 = m.get("TCC_EA0_WRREQ_64B_sum")
    rd = name.startswith("read")
    def f(x, d=1): return "-" if x is None else f"{x:.{d}f}"
    print(f"{name} | {kb / 2**20:.0f} | {f(fs / 1024 if fs is not None else None)} | {f(kb / (fs * 1024) if fs and rd else None, 3)} | {f(ws / 1024 if ws is not None else None)} | "
          f"{f(kb / (ws * 1024) if ws and not rd else None, 3)} | {f(rq, 0)} ({f(r32, 0)}) | {f(kb / rq if rq and rd else None, 1)} | {f(wq, 0)} ({f(w64, 0)}) | {f(kb / wq if wq and not rd else None, 1)}")
