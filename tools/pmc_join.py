"""Join the rocprofv3 --pmc passes of tools/pmc_innetwork.sh with the library's own launch records (same process, same order):
per (conv3 fusion mode, layer shape) mean counters of the in-network launches -> profiles/<PMC_TAG>_pmc_innetwork.json (default r03).
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE (KiB) x 2 on gfx950 for wide coalesced reads; WRITE_SIZE (KiB) as reported."""
import csv, glob, json, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.environ.get("PMC_DIR") or os.path.join(ROOT, "gpurun_out", "pmcnet")      # PMC_DIR: re-join a saved set of passes
TAG = os.environ.get("PMC_TAG", "r03")          # round tag of the output file: profiles/<TAG>_pmc_innetwork[_<dtype>].json
DT = os.environ.get("PMC_DTYPE", "bf16")        # bf16: the conv3 kernels (records tagged conv3*); bf16x3 / f32: every conv_igemm_kernel launch (class conv*)
KSUB = "conv3_kernel" if DT == "bf16" else "conv_igemm_kernel"
SUFFIX = "" if DT == "bf16" else "_" + DT
shapes = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in "abcde":
    dump = os.path.join(O, f"dump_{d}.csv")
    files = glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True)
    if not os.path.exists(dump) or not files:
        continue
    recs = [r for r in csv.DictReader(open(dump)) if (r["tag"].startswith("conv3") if DT == "bf16" else r["class"].startswith("conv"))]
    disp = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            if KSUB in r["Kernel_Name"]:
                disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
    ids = sorted(disp)
    assert len(ids) == len(recs), (d, len(ids), len(recs))
    for did, rec in zip(ids, recs):
        key = "|".join((rec["tag"], rec["d0"], rec["d1"], rec["d2"], rec["d3"]))
        if DT != "bf16":     # one tag ("conv") for every launch of the first-generation kernel: tell the instantiations apart by their template arguments
            kn = rows0 = disp[did][0]["Kernel_Name"]
            key += "|" + kn[kn.find("<") + 1:kn.rfind(">")].replace("kdip::", "").replace(" ", "")
        meta[key] = rec
        rows = disp[did]
        for r in rows:
            shapes[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        shapes[key]["_dur_us_" + d].append((float(rows[0]["End_Timestamp"]) - float(rows[0]["Start_Timestamp"])) * 1e-3)
        shapes[key]["_alg_mb"].append(float(rec["mbytes"])); shapes[key]["_alg_gflop"].append(float(rec["gflop"]))
        shapes[key]["_vgpr"].append(float(rows[0]["VGPR_Count"]) + float(rows[0].get("Accum_VGPR_Count", 0) or 0))
out = {"source": "rocprofv3 --kernel-trace --pmc <group> -- python tools/pmc_step.py (tools/pmc_innetwork.sh), one pass per counter group; "
                 "launch i of the library's profile dump = dispatch i of the conv3 kernels", "shapes": {}}
for key, c in shapes.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    rec = meta[key]
    B, H, Cin, Cout = int(rec["d0"]), int(rec["d1"]), int(rec["d2"]), int(rec["d3"])
    # (launches of one key differ in what they must move -- residual or not, backward-statistics input or not: mean over the launches)
    e = {"launches_per_pass": len(c.get("FETCH_SIZE", c.get("GRBM_GUI_ACTIVE", []))), "algorithmic_gflop": m["_alg_gflop"],
         "algorithmic_bytes": m["_alg_mb"] * 1e6}
    durs = [m[k] for k in m if k.startswith("_dur_us_")]
    e["mean_launch_us_under_pmc"] = sum(durs) / len(durs)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["FETCH_SIZE_KiB_raw"] = m["FETCH_SIZE"]; e["WRITE_SIZE_KiB_raw"] = m["WRITE_SIZE"]
        # gfx950: FETCH_SIZE = (128-byte line fetches of the L2) x 64 B -- HALF the bytes -- for EVERY access pattern of this library, including the
        # 64-byte-per-pixel staging of the split-precision 128 x 128 tile (the other half of the line is used by the next 16-channel chunk and
        # hits in L2).  Calibrated in round 6 on known byte counts past the Infinity Cache (tools/calib/, profiles/r06/fetch_calibration.txt):
        # wide 16 B / lane reads 2.000, the 64 B-per-512 B-row staging pattern 1.996, 256 B rows 2.000; WRITE_SIZE 1.000.  (Rounds 4 - 5 assumed 64-byte
        # requests for that kernel and under-reported its reads by 2 x.)
        e["fetch_bytes_per_counted_64B"] = 128
        e["hbm_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
        e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8)
        e["shader_clock_ghz"] = (m["GRBM_GUI_ACTIVE"] / 8) / (m["_dur_us_e"] * 1e-6) / 1e9
    if "SQ_LDS_BANK_CONFLICT" in m:
        e["lds_bank_conflict_frac_of_lds_active"] = m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)
    if "TCC_HIT_sum" in m:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in m:
        e["sq_wait_any_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]; e["sq_wait_inst_any_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
        e["sq_active_inst_any_frac"] = m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    e["vgpr_alloc"] = m.get("_vgpr")
    out["shapes"][key] = e
json.dump(out, open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_innetwork{SUFFIX}.json"), "w"), indent=1)
json.dump(out, open(os.path.join(O, f"{TAG}_pmc_innetwork{SUFFIX}.json"), "w"), indent=1)      # (gpurun merges only gpurun_out/ back: copy this one into profiles/)
for k, e in sorted(out["shapes"].items(), key=lambda kv: -kv[1]["mean_launch_us_under_pmc"] * kv[1]["launches_per_pass"])[:12]:
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a in ("launches_per_pass", "mean_launch_us_under_pmc", "traffic_over_algorithmic", "mfma_busy_frac", "lds_bank_conflict_frac_of_lds_active", "l2_hit_rate", "shader_clock_ghz")})
