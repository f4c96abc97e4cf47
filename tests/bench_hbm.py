#!/usr/bin/env python3
"""Stand-alone timing of the head's HBM-bound kernels (softmax fwd / bwd, BCE+KL fwd / bwd) against the 8 TB/s roof:
   python tests/bench_hbm.py [B L C] ...     (default: the headline shape and BASELINE configs[4])"""
import os
os.environ.setdefault("MVAE_TUNING", "1")   # schedule knobs are honoured only under this switch
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molecular_vae_amd as mv   # noqa: E402,F401
from molecular_vae_amd import ops  # noqa: E402

dev = torch.device("cuda")
shapes = [(1024, 120, 35), (2048, 256, 64)]
if len(sys.argv) > 3:
    a = list(map(int, sys.argv[1:]))
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B, L, C in shapes:
    n, m = B * L * C, B * 292
    logits = torch.randn(L * B, C, device=dev)
    recon = torch.empty(B, L, C, device=dev)
    ohe = torch.nn.functional.one_hot(torch.randint(0, C, (B, L), device=dev), C).float()
    mu, lv = torch.randn(B, 292, device=dev), torch.randn(B, 292, device=dev)
    out3 = torch.empty(3, device=dev); drecon = torch.empty_like(recon); dmu = torch.empty_like(mu); dlv = torch.empty_like(lv)
    dl = torch.zeros(L * B + 8, 128, device=dev, dtype=torch.bfloat16)
    for tiled in ("1", "0"):
        os.environ["MVAE_SOFTMAX_TILED"] = tiled
        r = {}
        r["softmax_fwd"] = (4 * 2 * n, timeit(lambda: ops.softmax_tb_fwd(logits, C, recon, B, L, C)))
        r["bce_kl_fwd"] = (4 * (2 * n + 2 * m), timeit(lambda: ops.bce_kl_loss_fwd(recon, ohe, mu, lv, float(L), out3)))
        r["bce_kl_bwd"] = (4 * (3 * n + 4 * m), timeit(lambda: ops.bce_kl_loss_bwd(recon, ohe, mu, lv, float(L), None, drecon, dmu, dlv)))
        r["softmax_bwd"] = (4 * 2 * n + 2 * n, timeit(lambda: ops.softmax_tb_bwd(recon, drecon, dl[:L * B], None, B, L, C)))
        print(f"B={B} L={L} C={C} tiled={tiled}: " + "  ".join(f"{k} {us:.1f} us {b / us / 1e6:.2f} TB/s ({b / us / 8e6:.2f})" for k, (b, us) in r.items()), flush=True)
