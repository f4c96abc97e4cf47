#!/usr/bin/env python3
"""mosesvae.VAE step: host enqueue time vs GPU-complete time per step, and a cProfile of the enqueue path.  python tests/tuning/host_issue_moses.py [B]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_extra  # noqa: E402

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = bench_extra.MosesWorkload(B, "bf16", dev, 0, None)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    wl.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, GPU-complete {1e3 * (t2 - t0) / n:.2f} ms/step", flush=True)
torch.autograd.set_multithreading_enabled(False)      # the backward's Python code on this thread, visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    wl.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
