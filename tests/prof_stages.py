#!/usr/bin/env python3
"""Per-op time breakdown of one training step (HIP events around every ops.* call, summed by op + GEMM shape).
Tuning aid, not a test.   python tests/prof_stages.py [B] [steps]"""
import collections
import inspect
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molecular_vae_amd as mv          # noqa: E402
from molecular_vae_amd import ops       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
REC = []
ON = [False]


def wrap(name, fn):
    def inner(*a, **k):
        if not ON[0]:
            return fn(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        s.record(st)
        r = fn(*a, **k)
        e.record(st)
        key = name
        try:
            if name in ("gemm_nt", "gemm_tn"):
                key += " M=%d N=%d K=%d %s" % (a[3], a[4], a[5], str(a[0].dtype).split(".")[-1])
            elif name in ("rnn_fwd", "rnn_bwd"):
                key += " H=%d" % a[4]
            elif name.startswith("conv1d"):
                key += " " + " ".join(str(x) for x in a if isinstance(x, int))
        except Exception:
            pass
        REC.append((key, s, e))
        return r
    return inner


for n, f in list(vars(ops).items()):
    if inspect.isfunction(f) and not n.startswith("_") and n not in ("join_pending", "stream_ptr", "ptr"):
        setattr(ops, n, wrap(n, f))

torch.manual_seed(0)
model = mv.MolecularVAE(i=120, c=35, o=292).to(dev)
opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
loss_fn = mv.make_loss_function(120)
idx = torch.randint(0, 35, (B, 120), device=dev)
ohe = torch.nn.functional.one_hot(idx, 35).float()
eps = 1e-2 * torch.randn(B, 292, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    recon, mu, lv = model(idx, eps=eps)
    loss = loss_fn(recon, ohe, mu, lv)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
ON[0] = True
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(steps):
    step()
t1.record()
torch.cuda.synchronize()
tot = collections.defaultdict(float); cnt = collections.Counter()
for k, s, e in REC:
    tot[k] += s.elapsed_time(e); cnt[k] += 1
print(f"step {t0.elapsed_time(t1) / steps:.2f} ms (with event overhead); per-step op times (ms, calls):")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {v / steps:8.3f}  {cnt[k] // steps:4d}  {k}")
print(f"  sum {sum(tot.values()) / steps:.3f}")
