#!/usr/bin/env python3
"""Top kernels of a rocprofv3 kernel_stats.csv, per step:  python tests/stats_top.py file.csv <steps incl. warm-up> [n]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    t = float(r["TotalDurationNs"])
    print(f"{t / 1e6 / steps:7.3f} ms/step {t / tot * 100:5.1f}%  calls/step {int(r['Calls']) / steps:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
print(f"total kernel time {tot / 1e6 / steps:.3f} ms/step")
