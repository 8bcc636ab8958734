"""GPU busy / idle analysis of a rocprofv3 kernel trace CSV: union of kernel intervals over the densest phase, idle gaps by the
kernel that FOLLOWS them, and per-kernel totals.  usage: python tools/trace_gaps.py kernel_trace.csv"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kdip::", "").replace("(anonymous namespace)::", "")) for r in rows)
ph = [[iv[0]]]
for x in iv[1:]:
    if x[0] - max(e for _, e, _ in ph[-1][-50:]) > 20e6: ph.append([x])
    else: ph[-1].append(x)
p = max(ph, key=len)
s0, e0 = p[0][0], max(e for _, e, _ in p)
busy, cs, ce = 0, p[0][0], p[0][1]
gaps = collections.defaultdict(lambda: [0, 0])
for s, e, n in p[1:]:
    if s > ce:
        busy += ce - cs; gaps[n][0] += 1; gaps[n][1] += s - ce; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in p: tot[n][0] += 1; tot[n][1] += e - s
span = e0 - s0
print(f"dense phase: {len(p)} kernels, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({busy / span:.3f}), sum of durations {sum(v[1] for v in tot.values()) / 1e6:.1f} ms")
print("idle time by the kernel that follows the gap (top 12):")
for n, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]: print(f"  {n[:70]:70s} gaps {c:5d}  idle {t / 1e6:7.2f} ms  ({t / c / 1e3:5.1f} us each)")
print("kernel totals (top 25):")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]: print(f"  {n[:70]:70s} n {c:5d}  {t / 1e6:7.2f} ms  {100 * t / span:5.1f}% of span  avg {t / c / 1e3:7.1f} us")
