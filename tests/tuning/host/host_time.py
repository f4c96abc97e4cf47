#!/usr/bin/env python3
"""Host (enqueue) time of one training step against its device time: is the step launch-bound?   python tests/tuning/host/host_time.py [B] [model]"""
import os
import sys
import time

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
for _ in range(5):
    mv.train_step(model, opt, loss_fn, data, ohe)
torch.cuda.synchronize()
N = 30
# (a) host time with an EMPTY queue in front of every step (synchronise, then time the enqueue only)
host = []
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mv.train_step(model, opt, loss_fn, data, ohe)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
# (b) back-to-back steps
t0 = time.perf_counter()
for _ in range(N):
    mv.train_step(model, opt, loss_fn, data, ohe)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print(f"B={B}: host enqueue time per step (queue empty): median {1e3 * host[N // 2]:.3f} ms, min {1e3 * host[0]:.3f} ms")
print(f"      back-to-back: enqueue {1e3 * t_enq / N:.3f} ms/step, with the final synchronize {1e3 * t_all / N:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(10):
    mv.train_step(model, opt, loss_fn, data, ohe)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
