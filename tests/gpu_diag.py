#!/usr/bin/env python3
"""Bring-up diagnostic: runs every HIP op against the oracle and PRINTS errors per stage without stopping at the first
mismatch.  Usage on the GPU box:  python tests/gpu_diag.py > gpurun_out/diag.log 2>&1"""
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_helpers import *   # noqa: F401,F403
from molecular_vae_amd import ops, _lib as L

dev = torch.device("cuda")


def section(name):
    print(f"\n=== {name}", flush=True)


def t_gemm():
    section("gemm_nt")
    rs = np.random.RandomState(0)
    for dt, tol in ((torch.float32, 2e-6), (torch.bfloat16, 2e-2)):
        for (M, N, K) in ((64, 64, 64), (128, 128, 256), (300, 200, 96), (35, 1024, 2048), (17, 9, 40), (4096, 1024, 1024),
                          (288, 72, 2880), (1, 5, 8), (300, 200, 512), (130, 70, 4096), (4096, 1024, 61440 // 8), (512, 4096, 292 if dt == torch.float32 else 296)):
            A = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).to(dev).to(dt)
            B = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32)).to(dev).to(dt)
            bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(dev)
            out = torch.full((M, N), float("nan"), device=dev)
            ops.gemm_nt(A, B, out, M, N, K, bias=bias, act=L.ACT_SELU)
            ref = A.double().cpu().numpy() @ B.double().cpu().numpy().T + bias.double().cpu().numpy()
            ref = O.selu(ref)
            e = rel(out.cpu().numpy(), ref)
            print(f"  {str(dt):16s} M={M} N={N} K={K}: rel={e:.3e} {'OK' if e < tol else 'FAIL'}", flush=True)


def t_cast_transpose():
    section("cast_transpose / permute")
    rs = np.random.RandomState(1)
    for (R, C, ldd, ldt) in ((37, 53, 56, 40), (128, 64, 64, 128), (5, 3, 8, 8)):
        src = torch.from_numpy(rs.standard_normal((R, C)).astype(np.float32)).to(dev)
        for dt in (torch.float32, torch.bfloat16):
            dst = torch.full((R, ldd), 7.0, device=dev, dtype=dt); dstT = torch.full((C, ldt), 7.0, device=dev, dtype=dt)
            ops.cast_transpose(src, R, C, dst=dst, dstT=dstT)
            ok = torch.equal(dst[:, :C], src.to(dt)) and torch.equal(dstT[:, :R], src.to(dt).t()) \
                and float(dst[:, C:].abs().sum()) == 0 and float(dstT[:, R:].abs().sum()) == 0
            print(f"  R={R} C={C} {dt}: {'OK' if ok else 'FAIL'}")
    x = torch.from_numpy(rs.standard_normal((3, 5, 7)).astype(np.float32)).to(dev)
    out = torch.empty(3, 7, 5, device=dev)
    ops.permute021(x, out, 3, 5, 7)
    print("  permute021:", "OK" if torch.equal(out, x.permute(0, 2, 1).contiguous()) else "FAIL")


def t_lstm(dt, tol, T=7, B=5, H=32, NL=2, In=16):
    section(f"lstm fwd/bwd {dt} T={T} B={B} H={H} NL={NL}")
    import test_gpu_parity as tp
    errs = tp._lstm_case(dt, T, B, H, NL, In)
    for k, v in errs.items():
        print(f"  {k}: rel={v:.3e} {'OK' if v < tol else 'FAIL'}")


def t_small_ops():
    section("loss / lambda / softmax / adam")
    rs = np.random.RandomState(3)
    B, Lq, C, o = 5, 11, 12, 16
    logits = rs.standard_normal((Lq * B, C)) * 2
    tl = torch.from_numpy(logits.astype(np.float32)).to(dev)
    recon = torch.empty(B, Lq, C, device=dev)
    ops.softmax_tb_fwd(tl, C, recon, B, Lq, C)
    e = np.exp(logits - logits.max(1, keepdims=True)); pr = (e / e.sum(1, keepdims=True)).reshape(Lq, B, C).transpose(1, 0, 2)
    print(f"  softmax fwd rel={rel(recon.cpu().numpy(), pr):.3e}")
    idx = rs.randint(0, C, (B, Lq))
    ohe = O.one_hot(idx, C)
    mu = rs.standard_normal((B, o)); lv = rs.standard_normal((B, o))
    out = torch.empty(3, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.float32)).to(dev)
    ops.bce_kl_loss_fwd(recon, t(ohe), t(mu), t(lv), Lq, out)
    ref = O.bce_kl_loss(pr, ohe, mu, lv, Lq)
    print(f"  loss fwd hip={out.cpu().numpy()} ref={ref}")
    drecon = torch.empty_like(recon); dmu = torch.empty(B, o, device=dev); dlv = torch.empty(B, o, device=dev)
    ops.bce_kl_loss_bwd(recon, t(ohe), t(mu), t(lv), Lq, None, drecon, dmu, dlv)
    r = O.bce_kl_loss_bwd(pr, ohe, mu, lv, Lq)
    print(f"  loss bwd rel drecon={rel(drecon.cpu().numpy(), r[0]):.3e} dmu={rel(dmu.cpu().numpy(), r[1]):.3e} dlv={rel(dlv.cpu().numpy(), r[2]):.3e}")
    # softmax bwd
    Cp = 16; ldT = (Lq * B + 7) // 8 * 8 + 8
    for dt in (torch.float32, torch.bfloat16):
        dl = torch.full((Lq * B, Cp), 3.0, device=dev, dtype=dt); dlT = torch.zeros(C, ldT, device=dev, dtype=dt)
        ops.softmax_tb_bwd(recon, drecon, dl, dlT, B, Lq, C)
        dp = r[0].transpose(1, 0, 2).reshape(Lq * B, C); p2 = pr.transpose(1, 0, 2).reshape(Lq * B, C)
        ref_dl = p2 * (dp - (dp * p2).sum(1, keepdims=True))
        print(f"  softmax bwd {dt}: rel={rel(dl[:, :C].float().cpu().numpy(), ref_dl):.3e} pad0={float(dl[:, C:].abs().sum())} "
              f"T rel={rel(dlT[:, :Lq * B].float().cpu().numpy(), ref_dl.T):.3e}")
    X = rs.standard_normal((1000, 150)).astype(np.float32)
    out = torch.zeros(150, device=dev)
    ops.colsum(t(X), 1000, 150, out)
    print(f"  colsum rel={rel(out.cpu().numpy(), X.astype(np.float64).sum(0)):.3e}")
    for dt in (torch.float32, torch.bfloat16):
        Xt = t(X).to(dt)
        out = torch.zeros(1000, device=dev)
        ops.rowsum(Xt[:, :144], 1000, 144, out, ldx=150)
        print(f"  rowsum {dt} rel={rel(out.cpu().numpy(), Xt[:, :144].double().cpu().numpy().sum(1)):.3e}")
    # adam
    n = 200000
    p0 = rs.standard_normal(n); g0 = rs.standard_normal(n) * 0.05
    P = {"w": p0.copy()}; st = {}
    tp, tg, tm, tv = t(p0), t(g0), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    partial = torch.zeros((n + 65535) // 65536, device=dev); norm = torch.zeros(1, device=dev)
    for step in range(1, 4):
        gc, total = O.clip_grad_norm({"w": g0}, 3.0)
        P = O.adam_step(P, gc, st, lr=8e-4)
        ops.sumsq(tg, partial)
        ops.clip_adam(tp, tg, tm, tv, partial, 1.0, 3.0, 8e-4, 0.9, 0.999, 1e-8, step, norm)
    print(f"  adam 3 steps rel={rel(tp.cpu().numpy(), P['w']):.3e} norm hip={float(norm):.6f} ref={total:.6f}")


def t_model(name, dims, params, idx, eps, dtype, tol):
    section(f"model {name} dtype={dtype}")
    p64 = {k: np.asarray(v, np.float64) for k, v in params.items()}
    t0 = time.time()
    ref = O.molvae_loss_and_grads(p64, idx, np.asarray(eps, np.float64), max_len=dims["i"], num_lstm=dims["n_enc"], num_gru=dims["n_dec"])
    print(f"  oracle {time.time() - t0:.1f}s loss={ref['loss']:.9f}")
    enc, dec = build_modules(dims, params, dtype)
    out = run_hip(enc, dec, idx, eps, dims["i"])
    print(f"  hip loss={out['loss']:.9f} rel={abs(out['loss'] - ref['loss']) / abs(ref['loss']):.3e}")
    for k in ("mu", "logvar", "z", "recon"):
        print(f"  {k}: rel={rel(out[k], ref[k]):.3e}")
    rep = grad_report(out["grads"], ref["grads"])
    worst = max(rep.values())
    for k, v in rep.items():
        print(f"    grad {k}: rel={v:.3e} {'' if v < tol else '<-- FAIL'}")
    print(f"  worst grad rel={worst:.3e} {'OK' if worst < tol else 'FAIL'}", flush=True)


def main():
    print(torch.cuda.get_device_name(0), flush=True)
    steps = [t_gemm, t_cast_transpose, lambda: t_lstm(torch.float32, 2e-5), lambda: t_lstm(torch.bfloat16, 3e-2),
             lambda: t_lstm(torch.float32, 2e-5, T=5, B=70, H=72, NL=3, In=8),
             lambda: t_lstm(torch.bfloat16, 3e-2, T=6, B=130, H=128, NL=3, In=8),      # LDS-direct pipelined path, ragged rows
             lambda: t_lstm(torch.float32, 2e-5, T=4, B=200, H=64, NL=2, In=8),        # pipelined path, f32
             lambda: t_lstm(torch.bfloat16, 3e-2, T=3, B=256, H=192, NL=4, In=8),
             t_small_ops]
    for s in steps:
        try:
            s()
        except Exception:
            traceback.print_exc()
    try:
        g = np.load(os.path.join(ROOT, "tests/golden/g1_small.npz"))
        t_model("g1", G1, g1_dims_params(np.float32), g["idx"], g["eps"], torch.float32, 2e-4)
        t_model("g1", G1, g1_dims_params(np.float32), g["idx"], g["eps"], torch.bfloat16, 5e-2)
    except Exception:
        traceback.print_exc()
    try:
        g = np.load(os.path.join(ROOT, "tests/golden/g2_full.npz"))
        params = ip.init_params(ip.molvae_shapes(), 202, 1.5, np.float32)
        t_model("g2-full", FULL, params, g["idx"], g["eps"], torch.float32, 5e-4)
        t_model("g2-full", FULL, params, g["idx"], g["eps"], torch.bfloat16, 5e-2)
    except Exception:
        traceback.print_exc()


if __name__ == "__main__":
    main()
