#!/usr/bin/env python3
"""Durations (us) of the wavefront step launches of the LAST training step in a rocprofv3 kernel trace, in launch order, per kernel name:
   python tests/step_durations.py gpurun_out/prof_x/x_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("clip_adam")]
a, b = ends[-2] + 1, ends[-1] + 1
by = collections.OrderedDict()
for r in rows[a:b]:
    n = r["Kernel_Name"]
    if "lstm_step" in n or "lstm_bwd_epi" in n:
        by.setdefault(n[:70], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in by.items():
    print(n, len(v), "launches, sum %.3f ms" % (sum(v) / 1e3))
    print("   ", " ".join(f"{x:.0f}" for x in v))
