#!/usr/bin/env python3
"""Weights-resident dataflow LSTM forward (rnn_persist.hip) against the wavefront schedule at the per-rank shape of configs[2]
(b = 128, 4 x LSTM(1024), bf16): same inputs, bit-level comparison of every output buffer, status record, time per pass and per diagonal.
   python tests/tuning/persist/ab_persist_fwd.py [T] [reps] [B]"""
import os
os.environ.setdefault("MVAE_TUNING", "1")
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import molecular_vae_amd as mv       # noqa: E402
from molecular_vae_amd import ops, _lib as L   # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 120
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
H, NL, PAD = 1024, 4, 64
dev = torch.device("cuda")
dt = torch.bfloat16
G4 = 4 * H
ldw, ldh = H + PAD, H + PAD
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.03)
Wih = [None] + [rnd(G4, ldw).to(dt) for _ in range(NL - 1)]
Whh = [rnd(G4, ldw).to(dt) for _ in range(NL)]
bias = [None] + [rnd(G4) * 10 for _ in range(NL - 1)]
gx0 = rnd(B, G4) * 30


def bufs():
    return dict(hs=[torch.zeros(T, B, ldh, device=dev, dtype=dt) for _ in range(NL)], cs=[torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)],
                gates=[torch.zeros(T, B, G4, device=dev, dtype=dt) for _ in range(NL)], cstate=[torch.zeros(2, B, H, device=dev) for _ in range(NL)])


def fwd(b, persist, save=True):
    ops.rnn_fwd(L.CELL_LSTM, dt, T, B, H, gx0, 0, Wih, [ldw] * NL, Whh, [ldw] * NL, bias, b["hs"], ldh, b["cs"] if save else None,
                b["gates"] if save else None, b["cstate"], persist=persist)


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


a, p = bufs(), bufs()
fwd(a, False); torch.cuda.synchronize()
fwd(p, True); torch.cuda.synchronize()
ops.persist_check(sync=True)
print("persistent launch status: ok")
worst, ulps = 0.0, 0.0
for k in ("hs", "cs", "gates"):
    for l in range(NL):
        x, y = a[k][l].float(), p[k][l].float()
        if k == "hs":
            x, y = x[:, :, :H], y[:, :, :H]
        d = (x - y).abs().max().item()
        worst = max(worst, d)
        ulps = max(ulps, d / (2.0 ** (torch.floor(torch.log2(x.abs().max())).item() - 7)))
        print(f"{k}[{l}] max |wavefront - dataflow| = {d:.3e}   (max |value| {x.abs().max().item():.3f}, equal bits: {bool(torch.equal(x, y))})")
dc = max((a["cstate"][l][(T - 1) & 1] - p["cstate"][l][(T - 1) & 1]).abs().max().item() for l in range(NL))
print(f"final cell state max diff {dc:.3e}")
nl = T + NL - 1
for label, persist in (("wavefront (T+3 launches)", False), ("dataflow  (one launch) ", True)):
    for save in ((False, True) if os.environ.get("LAST_SAVE", "0") == "1" else (True, False)):
        ms = timeit(lambda: fwd(a if not persist else p, persist, save), REPS)
        print(f"{label} save={int(save)}: {ms:.3f} ms per pass  {1e3 * ms / nl:.2f} us per diagonal  ({2 * B * G4 * 7 * H * T / ms / 1e9:.0f} TFLOP/s)")
ops.persist_check(sync=True)
print("RESULT", "OK" if ulps <= 2.0 else "MISMATCH", f"(largest difference {ulps:.2f} bf16 ulps of the buffer's largest value)")
if os.environ.get("MVAE_LIB"):
    # diagnostic build: clock samples (100 MHz wall clock) of workgroup j = 0 of every layer at 8 points of every step
    from molecular_vae_amd.ops import Scratch
    ws = Scratch.get(1, p["hs"][0].device, tag="rnn_persist")
    nbytes = NL * T * 8 * 8
    need = 64 + NL * T * 64 * 4
    st = ws[need:need + nbytes].view(torch.int64).view(NL, T, 8).cpu().double() * 0.01      # us
    names = ["start", "xflag ok", "hflag poll", "hflag ok", "half0 slots done", "half0 cells stored", "half1 slots done", "half1 cells stored"]
    for l in range(NL):
        d = st[l]
        step = (d[2:, 0] - d[1:-1, 0]).mean().item()
        print(f"layer {l}: step period {step:.2f} us; offsets from step start (mean over t >= 2):")
        for k in range(1, 8):
            if l == 0 and k == 1:
                continue
            print(f"    {names[k]:22s} {(d[2:, k] - d[2:, 0]).mean().item():7.2f} us")
        print(f"    next step start        {(d[2:, 0][1:] - d[2:, 0][:-1]).mean().item():7.2f} us")
