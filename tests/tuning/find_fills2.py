#!/usr/bin/env python3
"""ATen ops of one training step that launch fill / copy kernels, with shapes (torch.profiler): python tests/tuning/find_fills2.py [B]"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = bench.MolVaeWorkload(B, "bf16", torch.device("cuda"), 0, None, 120, 35)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    wl.step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::full", "aten::new_zeros") and ev.device_time_total > 0:
        st = [s for s in (ev.stack or []) if "molecular" in s or "bench" in s or "autograd" in s][:2]
        rows.append((ev.device_time_total, ev.name, str(ev.input_shapes)[:60], " <- ".join(x.split("/")[-1][:60] for x in st)))
for r in sorted(rows, reverse=True)[:40]:
    print(f"{r[0]:8.1f} us  {r[1]:16s} {r[2]:60s} {r[3]}")
