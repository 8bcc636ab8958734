"""Per (kernel class, tag, shape) table from a kdip_profile_dump CSV (bench.py writes one when KDIP_PROFILE_DUMP is set).
usage: python tools/shape_table.py dump.csv"""
import csv, sys, collections
g = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["class"], r["tag"], r["d0"], r["d1"], r["d2"], r["d3"])
    g[k][0] += 1; g[k][1] += float(r["us"]); g[k][2] += float(r["gflop"]); g[k][3] += float(r["mbytes"])
tot = sum(v[1] for v in g.values())
print(f"{'class':24s} {'tag':10s} {'B':>3s} {'H|HW':>6s} {'Cin|C':>6s} {'Cout':>5s} {'n':>4s} {'us/launch':>10s} {'total ms':>9s} {'%':>5s} {'TFLOP/s':>8s} {'GB/s':>7s}")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:24s} {k[1]:10s} {k[2]:>3s} {k[3]:>6s} {k[4]:>6s} {k[5]:>5s} {v[0]:4d} {v[1] / v[0]:10.1f} {v[1] / 1e3:9.3f} {100 * v[1] / tot:5.1f} {v[2] / v[1] * 1e3 if v[1] else 0:8.1f} {v[3] / v[1] * 1e3 if v[1] else 0:7.0f}")
print("total profiled ms:", tot / 1e3)
