"""Fold separate `rocprofv3 --kernel-trace --pmc <group>` passes (directories <dir>/{a..e}) of one micro-benchmark into a JSON of
per-launch means for the kernel whose name contains KERNEL: HBM bytes (FETCH_SIZE KiB x 2 on gfx950 for wide coalesced reads,
WRITE_SIZE KiB as reported: MI355X_MICROARCH.md), MFMA busy fraction, LDS conflict share, L2 hit rate, shader clock, SQ wait shares.
usage: python tools/pmc_fold.py <dir> <kernel-substring> <algorithmic-bytes> <algorithmic-gflop> <out.json> [note]"""
import csv, glob, json, os, sys, collections
d0, KERNEL, alg_bytes, alg_gflop, dst = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
note = sys.argv[6] if len(sys.argv) > 6 else ""
vals, durs = collections.defaultdict(list), []
for d in "abcde":
    for f in glob.glob(os.path.join(d0, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    durs.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9)
if not vals:
    sys.exit("no counter rows for " + KERNEL)
m = {k: sum(v) / len(v) for k, v in vals.items()}
dur = sum(durs) / len(durs) if durs else None
rd, wr = m.get("FETCH_SIZE", 0) * 1024 * 2, m.get("WRITE_SIZE", 0) * 1024
out = {"kernel_substring": KERNEL, "note": note, "launches_per_pass": len(durs), "mean_launch_us_under_pmc": dur * 1e6 if dur else None,
       "algorithmic_bytes": alg_bytes, "algorithmic_gflop": alg_gflop,
       "algorithmic_tflops_under_pmc": alg_gflop / dur / 1e3 if dur else None,
       "FETCH_SIZE_KiB_raw": m.get("FETCH_SIZE"), "WRITE_SIZE_KiB_raw": m.get("WRITE_SIZE"), "hbm_bytes_per_launch": rd + wr,
       "traffic_over_algorithmic": (rd + wr) / alg_bytes if alg_bytes else None,
       "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8) if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m else None,
       "shader_clock_ghz": (m["GRBM_GUI_ACTIVE"] / 8) / dur / 1e9 if dur and "GRBM_GUI_ACTIVE" in m else None,
       "lds_bank_conflict_frac_of_lds_active": m["SQ_LDS_BANK_CONFLICT"] / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1) if "SQ_LDS_BANK_CONFLICT" in m else None,
       "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
       "sq_wait_any_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
       "sq_wait_inst_any_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
       "sq_active_inst_any_frac": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in m else None,
       "insts_mfma_per_launch": m.get("SQ_INSTS_MFMA"), "insts_valu_per_launch": m.get("SQ_INSTS_VALU"), "raw_counter_means": m}
os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items() if k not in ("raw_counter_means", "note")}))
