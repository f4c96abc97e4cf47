#!/usr/bin/env python3
"""Largest idle gaps on the device in a rocprofv3 kernel + memory-copy trace (csv directory as argv[1])."""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"COPY {r.get('Direction', '')} {r.get('Bytes', r.get('Size', ''))}"))
ev.sort()
t0 = ev[0][0]
gaps = []
busy_end = ev[0][1]
for i in range(1, len(ev)):
    s, e, n = ev[i]
    if s > busy_end:
        gaps.append((s - busy_end, busy_end - t0, ev[i - 1][2], n))
    busy_end = max(busy_end, e)
gaps.sort(reverse=True)
print(f"{len(ev)} events over {(busy_end - t0) / 1e6:.2f} ms")
for g, at, a, b in gaps[:14]:
    print(f"gap {g / 1e3:9.1f} us at {at / 1e6:8.2f} ms  after [{a}]  before [{b}]")
for s, e, n in ev:
    if n.startswith("COPY"):
        print(f"{(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:8.1f} us  {n}")
