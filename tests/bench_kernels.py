#!/usr/bin/env python3
"""Kernel micro-benchmarks at the decoder's shapes (B=512, H=1024, 4 layers, T steps): wavefront fwd / bwd steps and the
weight-gradient GEMM.  Used for tuning and for the PMC profiles under profiles/.   python tests/bench_kernels.py [T] [B]"""
import os
os.environ.setdefault("MVAE_TUNING", "1")   # schedule knobs are honoured only under this switch
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molecular_vae_amd as mv       # noqa: E402
from molecular_vae_amd import ops, _lib as L   # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
which = sys.argv[3] if len(sys.argv) > 3 else "fwd,bwd,gemm"
H, NL, PAD = int(os.environ.get("BK_H", 1024)), int(os.environ.get("BK_NL", 4)), int(os.environ.get("LDPAD", 64))
CELL = L.CELL_GRU if os.environ.get("BK_CELL", "lstm") == "gru" else L.CELL_LSTM
dev = torch.device("cuda")
dt = torch.bfloat16
G4 = 4 * H
ldw, ldwT, ldh, ldg = H + PAD, G4 + PAD, H + PAD, G4 + PAD
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05)

Wih = [None] + [rnd(G4, ldw).to(dt) for _ in range(NL - 1)]
Whh = [rnd(G4, ldw).to(dt) for _ in range(NL)]
WihT = [None] + [rnd(H, ldwT).to(dt) for _ in range(NL - 1)]
WhhT = [rnd(H, ldwT).to(dt) for _ in range(NL)]
bias = [None] + [rnd(G4) for _ in range(NL - 1)]
gx0 = rnd(B, G4)
hs = [rnd(T, B, ldh).to(dt) * 10 for _ in range(NL)]      # random, so that MVAE_DBG=1 (epilogue skipped) still multiplies real operands
cs = [torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)]
cstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
gates = [torch.sigmoid(rnd(T, B, G4) * 20).to(dt) for _ in range(NL)]
dG = [rnd(T, B, ldg).to(dt) for _ in range(NL)]
dstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
dy = rnd(T * B, H)


def fwd():
    ops.rnn_fwd(CELL, dt, T, B, H, gx0, 0, Wih, [ldw] * NL, Whh, [ldw] * NL, bias, hs, ldh, None if CELL == L.CELL_GRU else cs, gates, cstate)


def bwd():
    ops.rnn_bwd(CELL, dt, T, B, H, WhhT, [ldwT] * NL, WihT, [ldwT] * NL, dy, H, hs, ldh, None if CELL == L.CELL_GRU else cs, gates, dG, dstate, ldg=ldg)


dW = torch.zeros(G4, H, device=dev)


def gemm():
    ops.gemm_tn(dG[1].view(T * B, ldg), hs[0].view(T * B, ldh), dW, G4, H, T * B, lda=ldg, ldb=ldh)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def host_us(fn, launches):
    """Host time to ENQUEUE one pass on an idle GPU (no wait inside), us per launch."""
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / launches


nl = T + NL - 1
per_t = 2 * B * G4 * (H + 3 * 2 * H)
if "hostn" in which:
    import time
    for n in (1, 2, 4, 8, 16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            bwd()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"bwd x{n}: enqueue {1e6 * (t1 - t0) / (n * nl):.2f} us/launch, GPU {1e6 * (t2 - t0) / (n * nl):.2f} us/launch ({n * nl} launches queued back to back)")
if "host" in which.split(","):
    print(f"host enqueue: fwd {host_us(fwd, nl):.2f} us/launch, bwd {host_us(bwd, nl):.2f} us/launch (bwd split forms: two launches per diagonal)")
if "fwd" in which:
    ms = timeit(fwd)
    print(f"fwd : {ms:.3f} ms  {1e3 * ms / nl:.1f} us/launch  {per_t * T / ms / 1e9:.1f} TFLOP/s")
if "bwd" in which:
    ms = timeit(bwd)
    print(f"bwd : {ms:.3f} ms  {1e3 * ms / nl:.1f} us/launch  {per_t * T / ms / 1e9:.1f} TFLOP/s")
if "gemmcs" in which:
    db = torch.zeros(G4, device=dev)
    ms = timeit(lambda: ops.gemm_tn_colsum(dG[1].view(T * B, ldg), hs[0].view(T * B, ldh), dW, db, G4, H, T * B, lda=ldg, ldb=ldh))
    print(f"gemm+colsum: {ms:.3f} ms  M={G4} N={H} K={T * B}  {2 * G4 * H * T * B / ms / 1e9:.1f} TFLOP/s")
    ms = timeit(lambda: ops.colsum_t(dG[1].view(T * B, ldg), T * B, G4, db, ldx=ldg))
    print(f"colsum_t alone: {ms:.3f} ms")
if "gemmg" in which.split(","):
    # the step's own form: all dW_ih / dW_hh of the upper two layers as ONE grouped full-K launch (4 problems x 64 tiles of 256 x 256)
    db = [torch.zeros(G4, device=dev) for _ in range(2)]
    dWs = [torch.zeros(G4, H, device=dev) for _ in range(4)]
    TB = T * B

    def gemmg():
        pr = []
        for i, l in enumerate((NL - 1, NL - 2)):
            a = dG[l].view(TB, ldg)
            pr.append(dict(A=a, B=hs[l - 1].view(TB, ldh), out=dWs[2 * i], M=G4, N=H, K=TB, lda=ldg, ldb=ldh, colsum_out=db[i]))
            pr.append(dict(A=a[B:], B=hs[l].view(TB, ldh), out=dWs[2 * i + 1], M=G4, N=H, K=TB - B, lda=ldg, ldb=ldh))
        ops.gemm_tn_grouped(pr)
    ms = timeit(gemmg)
    fl = 2 * G4 * H * (2 * TB + 2 * (TB - B))
    print(f"gemm grouped (4 problems, one launch): {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
if "gemm" in which.split(","):
    ms = timeit(gemm)
    print(f"gemm: {ms:.3f} ms  M={G4} N={H} K={T * B}  {2 * G4 * H * T * B / ms / 1e9:.1f} TFLOP/s")
