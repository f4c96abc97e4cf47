#!/usr/bin/env python3
"""What would a LIBRARY GEMM (hipBLASLt / rocBLAS through torch.mm) do on the decoder's weight-gradient contraction dW = dG^T . X
(M = 4096, N = 1024, K = T * B, bf16 operands, strided rows) against mvae_gemm_tn_grouped?  Timing probe only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import molecular_vae_amd as mv
from molecular_vae_amd import ops
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T, H = 120, 1024
K, G4 = T * B, 4 * H
dG = [(torch.randn(K, G4 + 64, device=dev) * 0.05).to(torch.bfloat16) for _ in range(4)]
X = [(torch.randn(K, H + 64, device=dev) * 0.05).to(torch.bfloat16) for _ in range(4)]
out = [torch.zeros(G4, H, device=dev) for _ in range(4)]


def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def ours():
    ops.gemm_tn_grouped([dict(A=dG[i], B=X[i], out=out[i], M=G4, N=H, K=K, lda=G4 + 64, ldb=H + 64) for i in range(4)])


def lib_bf16():
    for i in range(4):
        torch.mm(dG[i][:, :G4].t(), X[i][:, :H])


def lib_f32out():
    for i in range(4):
        torch.mm(dG[i][:, :G4].t(), X[i][:, :H], out_dtype=torch.float32)


fl = 4 * 2.0 * G4 * H * K
for name, fn in (("mvae_gemm_tn_grouped (4 problems)", ours), ("torch.mm bf16 out (4 calls)", lib_bf16), ("torch.mm fp32 out (4 calls)", lib_f32out)):
    try:
        ms = ev(fn)
        print(f"B={B} {name:40s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    except Exception as ex:
        print(f"B={B} {name}: {type(ex).__name__}: {str(ex)[:200]}", flush=True)
