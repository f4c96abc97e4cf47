#!/usr/bin/env python3
"""How far the host can run ahead of the GPU on one stream: enqueue N launches of a ~30 us kernel back to back and time the enqueue loop.
If the per-launch host time jumps from ~5 us to the kernel's duration beyond some N, the runtime's queue holds N launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda")
x = torch.empty(1 << 24, device=dev)       # 64 MB fill ~ 25-30 us
for _ in range(10):
    x.fill_(1.0)
torch.cuda.synchronize()
for N in (16, 32, 64, 128, 256, 512, 1024, 2048):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(N):
        x.fill_(1.0)
        if (i + 1) in (16, 32, 64, 128, 256, 512, 1024, 2048):
            marks.append((i + 1, time.perf_counter() - t0))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N:5d}: enqueue {1e6 * (t1 - t0) / N:6.1f} us/launch, GPU {1e6 * (t2 - t0) / N:6.1f} us/launch; cumulative enqueue at marks: " +
          ", ".join(f"{n}:{1e3 * t:.2f}ms" for n, t in marks), flush=True)
