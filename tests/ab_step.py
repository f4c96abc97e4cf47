#!/usr/bin/env python3
"""In-situ A/B of schedule knobs on the full training step, interleaved in ONE process (cdna guide rule 24):
  python tests/ab_step.py --batch 512 --set MVAE_FWD_GM=0 --set MVAE_FWD_GM=1 [--rounds 4 --steps 10]
Each --set is a comma-separated list of KEY=VALUE pairs (the kernels read these knobs on every call); prints ms/step per setting and round."""
import argparse
import os
os.environ.setdefault("MVAE_TUNING", "1")   # schedule knobs are honoured only under this switch
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molecular_vae_amd as mv   # noqa: E402
from molecular_vae_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--model", default="molvae", choices=["molvae", "moses", "models2d"])
args = ap.parse_args()
settings = [dict(kv.split("=") for kv in s.split(",") if kv) for s in (args.set or [""])]
dev = torch.device("cuda")
torch.manual_seed(42)
if args.model != "molvae":
    # the secondary workloads of bench.py (bench_extra.py): same step, same data, knobs flipped between rounds
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_extra
    W = {"moses": bench_extra.MosesWorkload, "models2d": bench_extra.Models2dWorkload}[args.model]
    wl = W(args.batch, "bf16", dev, 0, None)
    keys = sorted({k for s in settings for k in s})
    res = {i: [] for i in range(len(settings))}
    for _ in range(3):
        wl.step()
    for r in range(args.rounds):
        for i, s in enumerate(settings):
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(s)
            wl.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                wl.step()
            torch.cuda.synchronize()
            res[i].append((time.perf_counter() - t0) / args.steps * 1e3)
            print(f"round {r} {s}: {res[i][-1]:.3f} ms/step", flush=True)
    for i, s in enumerate(settings):
        v = sorted(res[i])
        print(f"{s}: median {v[len(v) // 2]:.3f} min {v[0]:.3f} ms/step")
    sys.exit(0)
model = mv.MolecularVAE().to(dev)
opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
loss_fn = mv.make_loss_function(120)
g = torch.Generator().manual_seed(1234)
data = torch.randint(0, 35, (args.batch, 120), generator=g).to(dev)
ohe = torch.nn.functional.one_hot(data, 35).float()
keys = sorted({k for s in settings for k in s})


def apply(s):
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(s)


for _ in range(3):
    mv.train_step(model, opt, loss_fn, data, ohe)
res = {i: [] for i in range(len(settings))}
for r in range(args.rounds):
    for i, s in enumerate(settings):
        apply(s)
        mv.train_step(model, opt, loss_fn, data, ohe)
        torch.cuda.synchronize()
        ops.PROFILE = {}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mv.train_step(model, opt, loss_fn, data, ohe)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        prof, ops.PROFILE = ops.PROFILE, None
        tag = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in prof.items()}
        res[i].append(dt)
        print(f"round {r} {s}: {dt:.3f} ms/step  fwd {tag.get('dec_lstm_fwd', 0):.3f} bwd {tag.get('dec_lstm_bwd', 0):.3f} "
              f"wgrad {tag.get('dec_lstm_wgrad', 0):.3f}+{tag.get('dec_lstm_wgrad_deferred', 0):.3f} "
              f"enc {tag.get('enc_lstm_fwd', 0):.3f}/{tag.get('enc_lstm_bwd', 0):.3f}", flush=True)
for i, s in enumerate(settings):
    v = sorted(res[i])
    print(f"{s}: median {v[len(v) // 2]:.3f} min {v[0]:.3f} ms/step")
