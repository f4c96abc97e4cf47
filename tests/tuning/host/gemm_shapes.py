#!/usr/bin/env python3
"""Shapes of every dense contraction of one training step (which small GEMMs make up the dependent chain at the per-rank batch?)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import molecular_vae_amd as mv   # noqa: E402
from molecular_vae_amd import ops   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
model = mv.MolecularVAE(i=120, o=292, c=35, dtype=torch.bfloat16).to(dev)
opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
loss_fn = mv.make_loss_function(120)
data = torch.randint(0, 35, (B, 120)).to(dev)
ohe = torch.nn.functional.one_hot(data, 35).float()
for _ in range(3):
    mv.train_step(model, opt, loss_fn, data, ohe)
torch.cuda.synchronize()
log = []
for name in ("gemm_nt", "gemm_tn", "gemm_tn_f32_colsum", "gemm_tn_colsum", "conv1d_selu_fwd", "conv1d_act_bwd", "conv1d_selu_bwd", "colsum", "colsum_t", "rowsum", "timesum"):
    if not hasattr(ops, name):
        continue
    orig = getattr(ops, name)

    def f(*a, _o=orig, _n=name, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = _o(*a, **k); e.record()
        ints = [x for x in a if isinstance(x, int)][:4]
        dts = [str(x.dtype).replace("torch.", "") for x in a if isinstance(x, torch.Tensor)][:1]
        log.append((_n, tuple(ints), dts[0] if dts else "", s, e))
        return r
    setattr(ops, name, f)
mv.train_step(model, opt, loss_fn, data, ohe)
torch.cuda.synchronize()
tot = 0.0
for n, ints, dt, s, e in log:
    ms = s.elapsed_time(e)
    tot += ms
    print(f"{n:20s} {str(ints):28s} {dt:9s} {1e3 * ms:8.1f} us")
print(f"total {tot:.3f} ms in {len(log)} calls")
