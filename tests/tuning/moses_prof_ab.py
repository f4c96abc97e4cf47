import os, sys, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_extra
from molecular_vae_amd import ops
dev = torch.device("cuda")
wl = bench_extra.MosesWorkload(1024, "bf16", dev, 0, None)
for _ in range(5): wl.step()
torch.cuda.synchronize()
def run(n=20):
    gc.collect(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): wl.step()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for r in range(3):
    ops.PROFILE = None; a = run()
    ops.PROFILE = {}; b = run(); ops.PROFILE = None
    print(f"round {r}: PROFILE off {a:.3f} ms/step, on {b:.3f} ms/step", flush=True)
