#!/usr/bin/env python3
"""Soak of the persistent dataflow passes inside the training step: N steps at the per-rank batch, status checked every step, loss finite and
falling; prints the number of persistent launches and failures.   python tests/tuning/persist/soak.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import molecular_vae_amd as mv   # noqa: E402
from molecular_vae_amd import ops   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda", 0)
torch.manual_seed(1)
NOISE = sys.argv[3] if len(sys.argv) > 3 else "device"
model = mv.MolecularVAE(i=120, o=292, c=35, dtype=torch.bfloat16, noise=NOISE).to(dev)
if len(sys.argv) > 4:
    model.encoder.lmbd.seed_noise(int(sys.argv[4]))
opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
loss_fn = mv.make_loss_function(120)
g = torch.Generator().manual_seed(5)
corpus = torch.randint(0, 35, (64, 120), generator=g)          # few distinct molecules: the loss must fall
fails, losses = 0, []
t0 = time.perf_counter()
for i in range(N):
    idx = corpus[torch.randint(0, 64, (B,), generator=g)].to(dev)
    ohe = torch.nn.functional.one_hot(idx, 35).float()
    loss = mv.train_step(model, opt, loss_fn, idx, ohe)
    if i % 50 == 0:
        fails += ops.persist_check(sync=True)            # (round 5: a launch that gives up is counted and survived, not raised)
        losses.append(float(loss))
    if i % 1000 == 999:
        print(f"step {i + 1}: loss {float(loss):.3f}, failures so far {ops.PERSIST_STATS['failures']}", flush=True)
torch.cuda.synchronize()
fails += ops.persist_check(sync=True)
dt = time.perf_counter() - t0
print(f"B={B}: {N} steps in {dt:.1f} s ({1e3 * dt / N:.2f} ms/step incl. host batch assembly); persistent launches fwd {ops.PERSIST_STATS['launches']} "
      f"bwd {ops.PERSIST_STATS['bwd_launches']}, layer-concurrent encoder passes {ops.PERSIST_STATS['rowres_pipe']}; failures {ops.PERSIST_STATS['failures']}, "
      f"steps skipped by the optimiser {int(float(opt.skipped_steps))}, schedules disabled: {ops.PERSIST_STATS['disabled']}")
print("loss every 50 steps:", " ".join(f"{x:.3f}" for x in losses[:6]), "...", " ".join(f"{x:.3f}" for x in losses[-4:]))
ok = fails == 0 and all(x == x and abs(x) < 1e6 for x in losses) and losses[-1] < 0.7 * losses[0]
print("RESULT", "OK" if ok else "BAD")
sys.exit(0 if ok else 1)
