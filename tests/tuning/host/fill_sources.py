#!/usr/bin/env python3
"""Where do the fill kernels of a training step come from?  Wraps the torch entry points that launch FillFunctor kernels and prints, for one
steady-state step at the given batch, a histogram of Python call sites (and the element counts)."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import molecular_vae_amd as mv   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = mv.MolecularVAE(i=120, o=292, c=35, dtype=torch.bfloat16).to(dev)
opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
loss_fn = mv.make_loss_function(120)
data = torch.randint(0, 35, (B, 120)).to(dev)
ohe = torch.nn.functional.one_hot(data, 35).float()
for _ in range(3):
    mv.train_step(model, opt, loss_fn, data, ohe)
torch.cuda.synchronize()
hist = collections.Counter()
sizes = collections.defaultdict(int)


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "molecular-vae_amd" in fr.filename or "molecular_vae_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}"
    return "outside the package"


def wrap(obj, name, numel):
    orig = getattr(obj, name)

    def f(*a, **k):
        r = orig(*a, **k)
        try:
            n = numel(r, a)
        except Exception:
            n = -1
        s = f"{name:10s} {site()}"
        hist[s] += 1
        sizes[s] += max(n, 0)
        return r
    setattr(obj, name, f)


wrap(torch, "zeros", lambda r, a: r.numel())
wrap(torch, "zeros_like", lambda r, a: r.numel())
wrap(torch, "full", lambda r, a: r.numel())
wrap(torch, "ones", lambda r, a: r.numel())
wrap(torch.Tensor, "zero_", lambda r, a: a[0].numel())
wrap(torch.Tensor, "fill_", lambda r, a: a[0].numel())
wrap(torch.Tensor, "new_zeros", lambda r, a: r.numel())
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    mv.train_step(model, opt, loss_fn, data, ohe)
    torch.cuda.synchronize()
print("Python-level fill call sites of ONE step:")
for s, c in hist.most_common():
    print(f"  {c:3d} x  {sizes[s] // max(c, 1):>10d} elements  {s}")
ev = [e for e in prof.events() if "FillFunctor" in e.name or "fill" in e.name.lower()]
dev_fills = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "FillFunctor" in e.name]
print("device FillFunctor kernels in the step:", len(dev_fills), " total us", sum(e.device_time for e in dev_fills))
memsets = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "emset" in e.name]
print("device memsets in the step:", len(memsets), " total us", sum(e.device_time for e in memsets))
# aten ops that launched fills, by op name
ops_ = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and
                           any(k in e.name for k in ("zero", "fill", "one_hot", "full")))
print("aten ops:", dict(ops_))
