#!/usr/bin/env python3
"""Weights-resident dataflow LSTM backward (rnn_persist_bwd.hip) against the wavefront schedule at the per-rank shape of configs[2]
(b = 128, 4 x LSTM(1024), bf16): same saved forward state and output gradient, comparison of every dG buffer, status record, time per pass.
   python tests/tuning/persist/ab_persist_bwd.py [T] [reps] [B]"""
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
dev = torch.device("cuda", 0)
dt = torch.bfloat16
G4 = 4 * H
ldw, ldh, ldwT, ldg = H + PAD, H + PAD, G4 + PAD, G4 + PAD
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.03)
Wih = [None] + [rnd(G4, ldw).to(dt) for _ in range(NL - 1)]
Whh = [rnd(G4, ldw).to(dt) for _ in range(NL)]
bias = [None] + [rnd(G4) * 10 for _ in range(NL - 1)]
gx0 = rnd(B, G4) * 30


def tr(w):      # [4H, ldw] -> [H, ldwT] (K-contiguous for the backward contraction)
    o = torch.zeros(H, ldwT, device=dev, dtype=dt)
    o[:, :G4] = w[:, :H].t()
    return o


WihT = [None] + [tr(w) for w in Wih[1:]]
WhhT = [tr(w) for w in Whh]
hs = [torch.zeros(T, B, ldh, device=dev, dtype=dt) for _ in range(NL)]
cs = [torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)]
gates = [torch.zeros(T, B, G4, device=dev, dtype=dt) for _ in range(NL)]
cstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
ops.rnn_fwd(L.CELL_LSTM, dt, T, B, H, gx0, 0, Wih, [ldw] * NL, Whh, [ldw] * NL, bias, hs, ldh, cs, gates, cstate, persist=False)
dy = rnd(T, B, H) * 3


def bufs():
    return dict(dG=[torch.zeros(T, B, ldg, device=dev, dtype=dt) for _ in range(NL)], dstate=[torch.zeros(2, B, H, device=dev) for _ in range(NL)])


def bwd(b, persist):
    ops.rnn_bwd(L.CELL_LSTM, dt, T, B, H, WhhT, [ldwT] * NL, WihT, [ldwT] * NL, dy.view(T * B, H), H, hs, ldh, cs, gates, b["dG"], b["dstate"], ldg=ldg,
                persist=persist)


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


a, p = bufs(), bufs()
bwd(a, False); torch.cuda.synchronize()
bwd(p, True); torch.cuda.synchronize()
ops.persist_check(sync=True)
print("persistent launch status: ok")
worst = 0.0
for l in range(NL):
    x, y = a["dG"][l].float()[:, :, :G4], p["dG"][l].float()[:, :, :G4]
    d = (x - y).abs().max().item()
    rel = d / x.abs().max().item()
    worst = max(worst, rel)
    bad = ((x - y).abs() > 0.02 * x.abs().max()).nonzero()
    print(f"dG[{l}] max |wavefront - dataflow| = {d:.3e}  (max |value| {x.abs().max().item():.3e}, rel {rel:.2e}, equal bits: {bool(torch.equal(x, y))}, finite: {bool(torch.isfinite(y).all())})")
    if len(bad) or not torch.isfinite(y).all():
        worst = float("inf")
        bad = (~((x - y).abs() <= 0.02 * x.abs().max())).nonzero()
        print("   first mismatches (t, row, col):", bad[:6].tolist(), " count", len(bad))
        if os.environ.get("PB_DIAG"):
            tt, rr, cc = bad[:, 0], bad[:, 1], bad[:, 2]
            print("   by t:", torch.bincount(tt, minlength=T).tolist())
            print("   by row // 8:", torch.bincount(rr // 8, minlength=B // 8).tolist())
            print("   by gate:", torch.bincount(cc // H, minlength=4).tolist())
            print("   by unit % 8:", torch.bincount(cc % 8, minlength=8).tolist())
            print("   by (unit // 8) % 8:", torch.bincount((cc % H // 8) % 8, minlength=8).tolist())
            print("   by unit block of 64:", torch.bincount(cc % H // 64, minlength=16).tolist())
            print("   nan count:", int(torch.isnan(y).sum()))
nl = T + NL - 1
for label, persist in (("wavefront (2 x (T+3) launches)", False), ("dataflow  (one launch)        ", True)):
    ms = timeit(lambda: bwd(a if not persist else p, persist), REPS)
    print(f"{label}: {ms:.3f} ms per pass  {1e3 * ms / nl:.2f} us per diagonal  ({2 * B * G4 * (7 * H) * T / ms / 1e9:.0f} TFLOP/s)")
ops.persist_check(sync=True)
print("RESULT", "OK" if worst <= 1e-2 else "MISMATCH", f"(largest difference {worst:.2e} of the buffer's largest value)")
if os.environ.get("MVAE_LIB"):
    from molecular_vae_amd.ops import Scratch
    ws = Scratch.get(1, dev, tag="rnn_persist_bwd")
    head = (64 + NL * T * 64 * 4 + 4095) // 4096 * 4096
    off = head + 16 * 1024 * 1024
    st = ws[off:off + NL * T * 64 * 64].view(torch.int64).view(NL, T, 64, 8).cpu().double() * 0.01      # us
    names = ["start", "hflag poll", "hflag ok", "half0 slots done", "half1 slots done", "half1 partial sent", "partials received", "dG stored"]
    for l in range(NL):
        d = st[l].flip(0)          # [step in execution order][workgroup][stamp]
        per = (d[3:, :, 0] - d[2:-1, :, 0]).mean().item()
        print(f"layer {l}: step period {per:.2f} us; offsets from the layer's EARLIEST step start, mean over steps >= 2: mean over workgroups / min / max")
        base = d[2:, :, 0].min(dim=1, keepdim=True).values
        for k in range(0, 8):
            if l == NL - 1 and k == 1:
                continue
            x = (d[2:, :, k] - base).mean(dim=0)        # per workgroup
            print(f"    {names[k]:22s} {x.mean().item():7.2f} {x.min().item():7.2f} {x.max().item():7.2f}   slowest wg j={int(x.argmax())} fastest j={int(x.argmin())}")
        busy = ((d[2:, :, 4] - d[2:, :, 2]).mean(dim=0))
        print("    streaming (hflag ok -> half1 slots done) per workgroup: " + " ".join(f"{v:.1f}" for v in busy.tolist()))
if os.environ.get("PB_DUMP"):
    l = NL - 1
    x, y = a["dG"][l].float()[T - 1, :, :G4], p["dG"][l].float()[T - 1, :, :G4]
    okm = ((x - y).abs() <= 0.02 * x.abs().max())
    print("top layer, t = T-1: rows x (gate 0, units 0..23) ok-map (1 = ok)")
    for r in range(0, 40):
        print(f"  row {r:3d}: " + "".join("1" if okm[r, u] else "." for u in range(24)) + "   gate1: " + "".join("1" if okm[r, H + u] else "." for u in range(24)))
    r, u = 1, 0
    print("expected", x[1, :8].tolist()); print("got     ", y[1, :8].tolist())
    # does the value sit somewhere else?
    v = x[1, 0].item()
    hit = ((y - v).abs() < 1e-7 * max(1.0, abs(v))).nonzero()
    print("expected value of (row 1, unit 0) found at", hit[:8].tolist())
if os.environ.get("PB_DUMP"):
    bad = (~okm).nonzero()
    print("top layer wrong entries:", len(bad), " by gate", torch.bincount(bad[:, 1] // H, minlength=4).tolist(), " nan", int(torch.isnan(y).sum()))
    print("  by row % 32:", torch.bincount(bad[:, 0] % 32, minlength=32).tolist())
    print("  by unit % 64:", torch.bincount(bad[:, 1] % 64, minlength=64).tolist())
    for k in range(0, min(len(bad), 400), 40):
        r_, c_ = int(bad[k, 0]), int(bad[k, 1])
        print(f"   (row {r_}, gate {c_ // H}, unit {c_ % H}): expected {x[r_, c_].item():.6e} got {y[r_, c_].item():.6e}")
