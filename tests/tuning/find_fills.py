#!/usr/bin/env python3
"""Which host lines zero-fill or copy device tensors inside one training step (sizes + caller): python tests/tuning/find_fills.py [B]"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = bench.MolVaeWorkload(B, "bf16", torch.device("cuda"), 0, None, 120, 35)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
log = collections.Counter()
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "molecular" in fr.filename or "bench" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"
def wrap(obj, name, kind):
    orig = getattr(obj, name)
    def f(*a, **k):
        r = orig(*a, **k)
        t = r if isinstance(r, torch.Tensor) else (a[0] if a and isinstance(a[0], torch.Tensor) else None)
        if t is not None and t.is_cuda:
            log[(kind, where(), t.numel() * t.element_size())] += 1
        return r
    setattr(obj, name, f)
wrap(torch, "zeros", "zeros"); wrap(torch, "zeros_like", "zeros_like"); wrap(torch.Tensor, "zero_", "zero_"); wrap(torch.Tensor, "fill_", "fill_")
wrap(torch.Tensor, "copy_", "copy_"); wrap(torch, "empty_like", "empty_like")
torch.autograd.set_multithreading_enabled(False)
wl.step()
torch.cuda.synchronize()
for (kind, w, nbytes), c in sorted(log.items(), key=lambda kv: -kv[0][2] * kv[1]):
    print(f"{kind:10s} {w:28s} {nbytes / 1e6:10.3f} MB x{c}")
