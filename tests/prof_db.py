#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals, and a per-stream timeline of the LAST training step.
python tests/prof_db.py gpurun_out/prof5/v5_results.db [csv_out]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sy = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.display_name, d.start, d.end, d.queue_id, d.stream_id from {kd} d join {sy} s on d.kernel_id = s.id order by d.start").fetchall()


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")[:90]


tot = {}
for n, s, e, q, st in rows:
    k = short(n)
    a = tot.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
allt = sum(v[1] for v in tot.values())
print(f"{len(rows)} dispatches, {allt / 1e6:.2f} ms of kernel time")
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage"]
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    lines.append(f'"{k}",{n},{t},{t / n:.1f},{100 * t / allt:.2f}')
print("\n".join(lines[:32]))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
# timeline of the last step: find the last clip_adam, previous clip_adam
adam = [i for i, r in enumerate(rows) if "clip_adam" in r[0]]
if len(adam) >= 2:
    seg = rows[adam[-2] + 1: adam[-1] + 1]
    t0 = seg[0][1]
    print(f"\nlast step: {(seg[-1][2] - t0) / 1e6:.2f} ms wall, {sum(e - s for _, s, e, _, _ in seg) / 1e6:.2f} ms kernel time")
    # coarse phases: group consecutive kernels of the same short name per queue
    cur = None
    for n, s, e, q, st in seg:
        k = (short(n)[:50], q)
        if cur and cur[0] == k:
            cur[2] = e; cur[3] += 1; cur[4] += e - s
        else:
            if cur: print(f"  q{cur[0][1]} {(cur[1] - t0) / 1e3:9.1f} -> {(cur[2] - t0) / 1e3:9.1f} us  x{cur[3]:4d} busy {cur[4] / 1e3:8.1f}  {cur[0][0]}")
            cur = [k, s, e, 1, e - s]
    if cur: print(f"  q{cur[0][1]} {(cur[1] - t0) / 1e3:9.1f} -> {(cur[2] - t0) / 1e3:9.1f} us  x{cur[3]:4d} busy {cur[4] / 1e3:8.1f}  {cur[0][0]}")
