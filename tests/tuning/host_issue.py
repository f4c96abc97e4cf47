#!/usr/bin/env python3
"""How long the HOST needs to enqueue one training step vs how long the GPU needs to run it (per-GPU batch given on the command line):
if the two are close the step is launch-bound on the host.  python tests/tuning/host_issue.py 128 256 1024"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda")
for B in [int(a) for a in sys.argv[1:]] or [128]:
    wl = bench.MolVaeWorkload(B, "bf16", dev, 0, None, 120, 35)
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
    del wl
    from molecular_vae_amd import ops
    ops.release_caches()
    torch.cuda.empty_cache()
