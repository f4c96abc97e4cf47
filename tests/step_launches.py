#!/usr/bin/env python3
"""Launch inventory of ONE steady-state training step from a rocprofv3 kernel trace: the kernels between the last two clip_adam launches,
per kernel name (count, total us, average us) in order of total time, the number of launches shorter than 20 us, the span of the step and the
sum of kernel time on the device.  (The --stats summary of a whole process also counts the first step's workspace zero-fills: 700 fills in a
13-step run are 650 from step 1 and 3 per step after -- read per-step figures from here.)
   python tests/step_launches.py gpurun_out/prof_x/.../x_kernel_trace.csv [csv-out]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("clip_adam")]
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
by = collections.OrderedDict()
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e = by.setdefault(r["Kernel_Name"], [0, 0.0])
    e[0] += 1; e[1] += d
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
short = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in step]
n_short = sum(1 for d in short if d < 20.0)
print(f"launches in the step: {len(step)}   span {span / 1e3:.3f} ms   sum of kernel time {sum(short) / 1e3:.3f} ms   "
      f"launches < 20 us: {n_short} ({sum(d for d in short if d < 20.0) / 1e3:.3f} ms)")
out = sorted(by.items(), key=lambda kv: -kv[1][1])
for n, (c, t) in out:
    print(f"{c:4d} x {t / c:9.1f} us = {t:9.1f} us  {n[:150]}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "CallsPerStep", "TotalUsPerStep", "AverageUs"])
        for n, (c, t) in out:
            w.writerow([n, c, round(t, 1), round(t / c, 2)])
        w.writerow([f"TOTAL launches={len(step)} span_ms={span / 1e3:.3f} kernel_ms={sum(short) / 1e3:.3f} short(<20us)={n_short}", "", "", ""])
