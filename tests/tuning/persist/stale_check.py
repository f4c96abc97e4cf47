#!/usr/bin/env python3
"""Back-to-back persistent passes into the SAME buffers with DIFFERENT inputs: a hand-off that read a stale cache line of the previous
launch would reproduce the previous pass's values.  Prints the worst difference to the wavefront schedule for every launch."""
import os
os.environ.setdefault("MVAE_TUNING", "1")
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from molecular_vae_amd import ops, _lib as L   # noqa: E402
T = int(sys.argv[1]) if len(sys.argv) > 1 else 120
SAVE = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
B, H, NL, PAD = 128, 1024, 4, 64
dev = torch.device("cuda", 0); dt = torch.bfloat16
G4, ldw, ldh = 4 * H, H + PAD, H + PAD
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.03)
Wih = [None] + [rnd(G4, ldw).to(dt) for _ in range(NL - 1)]
Whh = [rnd(G4, ldw).to(dt) for _ in range(NL)]
bias = [None] + [rnd(G4) * 10 for _ in range(NL - 1)]
mk = lambda: dict(hs=[torch.zeros(T, B, ldh, device=dev, dtype=dt) for _ in range(NL)], cs=[torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)],
                  gates=[torch.zeros(T, B, G4, device=dev, dtype=dt) for _ in range(NL)], cstate=[torch.zeros(2, B, H, device=dev) for _ in range(NL)])
a, p = mk(), mk()
def fwd(b, gx0, persist):
    ops.rnn_fwd(L.CELL_LSTM, dt, T, B, H, gx0, 0, Wih, [ldw] * NL, Whh, [ldw] * NL, bias, b["hs"], ldh, b["cs"] if SAVE else None,
                b["gates"] if SAVE else None, b["cstate"], persist=persist)
worst = 0.0
inputs = [rnd(B, G4) * 30 for _ in range(6)]
for k, gx0 in enumerate(inputs):          # launches queued back to back, no synchronisation in between
    fwd(p, gx0, True)
outs = None
torch.cuda.synchronize(); ops.persist_check(sync=True)
fwd(a, inputs[-1], False); torch.cuda.synchronize()
for l in range(NL):
    d = (a["hs"][l].float() - p["hs"][l].float()).abs()
    print(f"after 6 back-to-back launches: hs[{l}] max diff {d.max().item():.4e} at t={int(d.amax(dim=(1, 2)).argmax())}, mean {d.mean().item():.3e}")
    worst = max(worst, d.max().item())
# and one at a time
for k, gx0 in enumerate(inputs[:3]):
    fwd(p, gx0, True); fwd(a, gx0, False); torch.cuda.synchronize()
    w = max((a["hs"][l].float() - p["hs"][l].float()).abs().max().item() for l in range(NL))
    print(f"launch {k} alone: max diff {w:.4e}")
    worst = max(worst, w)
print("RESULT", "OK" if worst < 0.02 else "STALE/MISMATCH", worst)
