#!/usr/bin/env python3
"""Is the MOSES step (B = 1024, bf16) launch-bound?  Host enqueue time per step against the step time, and the launch count."""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench_extra   # noqa: E402
dev = torch.device("cuda")
wl = bench_extra.MosesWorkload(1024, "bf16", dev, 0, None)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
host = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); wl.step(); host.append(time.perf_counter() - t0)
host.sort()
gc.collect(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    wl.step()
te = time.perf_counter() - t0
torch.cuda.synchronize()
ta = time.perf_counter() - t0
print(f"host enqueue (queue empty) median {1e3 * host[10]:.3f} ms; back-to-back enqueue {1e3 * te / 30:.3f} ms/step, with final sync {1e3 * ta / 30:.3f} ms/step")
