#!/usr/bin/env python3
"""Per-queue timeline of the LAST training step in a rocprofv3 --kernel-trace CSV (one line per dispatch outside the decoder wavefront,
the wavefront launches collapsed):  python tests/prof_timeline.py gpurun_out/prof/x_kernel_trace.csv > profiles/rNN_timeline.txt"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("clip_adam")]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
print(f"# last step: {b - a} dispatches, {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e6:.3f} ms from first start to last end")
print("# queue   start_us   dur_us  kernel  grid x wg")
cnt, busy = 0, {}
for r in rows[a:b]:
    n = r["Kernel_Name"]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0.0) + (e - s)
    if "lstm_step" in n or "lstm_bwd_epi" in n:
        cnt += 1
        continue
    if cnt:
        print(f"          ... {cnt} decoder wavefront launches")
        cnt = 0
    print(f"q{r['Queue_Id']} {s:10.1f} {e - s:8.1f}  {n[:72]}  {r['Grid_Size_X']}x{r['Grid_Size_Y']} / {r['Workgroup_Size_X']}")
for q, v in sorted(busy.items()):
    print(f"# queue {q}: {v / 1e3:.3f} ms of kernel time")
