#!/usr/bin/env python3
"""ms per MOSES training step (bench_extra.MosesWorkload, B = 1024, bf16), 3 x 20 steps.   python tests/tuning/host/moses_step.py"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench_extra   # noqa: E402

dev = torch.device("cuda")
wl = bench_extra.MosesWorkload(1024, "bf16", dev, 0, None)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
out = []
for r in range(3):
    gc.collect(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        wl.step()
    torch.cuda.synchronize()
    out.append(1e3 * (time.perf_counter() - t0) / 20)
print("ms/step:", " ".join(f"{x:.3f}" for x in out))
