"""GPU parity tests (pytest -m gpu): the HIP path through the C ABI vs the CPU oracle on the same seeded inputs, the golden
fixtures generated from the reference, and size-independent properties at BASELINE.json's full sizes.

Tolerances (north_star: 1e-4 relative on the ELBO and the encoder mu / log sigma^2):
  f32 path   loss, mu, logvar, recon <= 1e-5 rel (well inside 1e-4); every parameter gradient <= 2e-4 rel (max-norm)
  bf16 path  loss <= 1e-4 rel; mu / logvar identical to the f32 path (the encoder is always f32); recon <= 5e-3;
             parameter gradients <= 5e-2 rel (bf16 storage of decoder activations, fp32 accumulate)
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gpu_helpers as gh            # noqa: E402
from gpu_helpers import O, ip, mv, rel   # noqa: E402
from molecular_vae_amd import ops, _lib as L   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda")


def t(a, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.float32)).to(dev).to(dt)


def test_native_library_is_the_one_running():
    lib = L.load()
    assert lib.mvae_abi_version() == L.ABI_VERSION >= 10
    assert os.path.samefile(L.LIB_PATH, os.path.join(ROOT, "molecular-vae_amd", "libmvae_hip.so"))
    assert any("libmvae_hip.so" in line for line in open("/proc/self/maps"))


@pytest.mark.parametrize("dt,tol", [(torch.float32, 5e-6), (torch.bfloat16, 1e-6)])
@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 128, 256), (300, 200, 96), (35, 1024, 2048), (17, 9, 40), (1, 5, 8),
                                   (4096, 1024, 1024), (288, 72, 2880), (300, 200, 512), (130, 70, 4096), (512, 4096, 296)])
def test_gemm_nt(dt, tol, shape):
    """All tile variants: 64/128 tiles, split-K, ragged edges, register-staged and LDS-direct pipelined main loops.
    Inputs are pre-rounded to the storage type, so the check isolates the kernel (fp32 accumulate)."""
    M, N, K = shape
    rs = np.random.RandomState(M * 7 + N * 3 + K)
    A, B = t(rs.standard_normal((M, K)), dt), t(rs.standard_normal((N, K)), dt)
    bias = t(rs.standard_normal(N))
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_nt(A, B, out, M, N, K, bias=bias, act=L.ACT_SELU)
    ref = O.selu(A.double().cpu().numpy() @ B.double().cpu().numpy().T + bias.double().cpu().numpy())
    assert rel(out.cpu().numpy(), ref) < tol


@pytest.mark.parametrize("shape", [(256, 128, 256), (4096, 120, 2304), (300, 64, 1152)])
def test_gemm_nt_f32_operands_as_three_bf16_products(shape):
    """MVAE_F32X3: fp32 operands multiplied as hi.hi + hi.lo + lo.hi on the bf16 MFMA (the conv input-gradient GEMMs of the bf16 training
    mode): about 16 mantissa bits per product -- three orders of magnitude tighter than plain bf16, looser than the exact-fp32 kernel."""
    M, N, K = shape
    rs = np.random.RandomState(M + N)
    A, B = t(rs.standard_normal((M, K))), t(rs.standard_normal((N, K)))
    ref = A.double().cpu().numpy() @ B.double().cpu().numpy().T
    out3 = torch.full((M, N), float("nan"), device=dev); ops.gemm_nt(A, B, out3, M, N, K, x3=True)
    out1 = torch.full((M, N), float("nan"), device=dev); ops.gemm_nt(A, B, out1, M, N, K)
    e3, e1 = rel(out3.cpu().numpy(), ref), rel(out1.cpu().numpy(), ref)
    assert e1 < 5e-6 and e1 < e3 < 3e-5, (e1, e3)


def _lstm_case(dt, T, B, H, NL, In, seed=2, persist=None):
    """persist=True: the shape / layout the weights-resident dataflow passes serve (rnn_persist*.hip: ldh = H + 64, ldg = 4H + 64, a
    time-invariant layer-0 input, weights scaled so that a 1024-wide contraction does not saturate the gates) and both passes REQUIRED to
    take that schedule -- B distinct random rows against the numpy oracle."""
    rs = np.random.RandomState(seed)
    G4 = 4 * H
    p = {}
    a = 0.4 if not persist else 0.4 * np.sqrt(32.0 / H)
    for l in range(NL):
        inp = In if l == 0 else H
        p[f"g.weight_ih_l{l}"] = rs.uniform(-0.4 if l == 0 else -a, 0.4 if l == 0 else a, (G4, inp)); p[f"g.weight_hh_l{l}"] = rs.uniform(-a, a, (G4, H))
        p[f"g.bias_ih_l{l}"] = rs.uniform(-0.2, 0.2, G4); p[f"g.bias_hh_l{l}"] = rs.uniform(-0.2, 0.2, G4)
    if dt == torch.bfloat16:
        for k in p:
            if "weight" in k:
                p[k] = torch.from_numpy(p[k]).bfloat16().double().numpy()
    x = rs.standard_normal((T, B, In))
    if persist:
        x[:] = x[0]                          # models.py:163: the decoder's input is the latent repeated over time
    y, caches = O.lstm_fwd(x, p, "g", NL)
    dy = rs.standard_normal((T, B, H))
    grads = {}
    dx = O.lstm_bwd(dy, caches, grads, "g")
    gx0 = t((x.reshape(T * B, In) @ p["g.weight_ih_l0"].T + p["g.bias_ih_l0"] + p["g.bias_hh_l0"]).reshape(T, B, G4))
    ldh, ldg = (H + 8, G4 + 8) if not persist else (H + 64, G4 + 64)
    before = dict(ops.PERSIST_STATS)
    hs = [torch.zeros(T, B, ldh, device=dev, dtype=dt) for _ in range(NL)]
    cs = [torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)]
    cstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
    gates = [torch.zeros(T, B, G4, device=dev, dtype=dt) for _ in range(NL)]
    w_ih = [None] + [t(p[f"g.weight_ih_l{l}"], dt) for l in range(1, NL)]
    w_hh = [t(p[f"g.weight_hh_l{l}"], dt) for l in range(NL)]
    bias = [None] + [t(p[f"g.bias_ih_l{l}"] + p[f"g.bias_hh_l{l}"]) for l in range(1, NL)]
    ops.rnn_fwd(L.CELL_LSTM, dt, T, B, H, gx0[0] if persist else gx0, 0 if persist else B * G4, w_ih, [H] * NL, w_hh, [H] * NL, bias, hs, ldh, cs, gates,
                cstate, persist=persist)
    w_hhT = [t(p[f"g.weight_hh_l{l}"].T, dt) for l in range(NL)]
    w_ihT = [None] + [t(p[f"g.weight_ih_l{l}"].T, dt) for l in range(1, NL)]
    dG = [torch.zeros(T, B, ldg, device=dev, dtype=dt) for _ in range(NL)]
    dstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
    ops.rnn_bwd(L.CELL_LSTM, dt, T, B, H, w_hhT, [G4] * NL, w_ihT, [G4] * NL, t(dy), H, hs, ldh, cs, gates, dG, dstate, ldg=ldg, persist=persist)
    torch.cuda.synchronize()
    if persist:
        assert ops.PERSIST_STATS["launches"] == before["launches"] + 1 and ops.PERSIST_STATS["bwd_launches"] == before["bwd_launches"] + 1
        assert ops.PERSIST_STATS["failures"] == before["failures"]
    errs = {}
    for l in range(NL):
        h_np = hs[l][:, :, :H].double().cpu().numpy()
        errs[f"h{l}"] = rel(h_np, caches[l][1])
        errs[f"c{l}"] = rel(cs[l].double().cpu().numpy(), caches[l][2])
        errs[f"gates{l}"] = rel(gates[l].double().cpu().numpy(), caches[l][3])
        assert float(hs[l][:, :, H:].abs().sum()) == 0.0 and float(dG[l][:, :, G4:].abs().sum()) == 0.0   # pads stay zero
        dp = dG[l][:, :, :G4].double().cpu().numpy().reshape(T * B, G4)
        hprev = np.concatenate([np.zeros((1, B, H)), h_np[:-1]], 0).reshape(T * B, H)
        errs[f"dWhh{l}"] = rel(dp.T @ hprev, grads[f"g.weight_hh_l{l}"]) if T > 1 else 0.0
        errs[f"db{l}"] = rel(dp.sum(0), grads[f"g.bias_ih_l{l}"])
        if l > 0:
            xin = hs[l - 1][:, :, :H].double().cpu().numpy().reshape(T * B, H)
            errs[f"dWih{l}"] = rel(dp.T @ xin, grads[f"g.weight_ih_l{l}"])
        if dt == torch.bfloat16 and T > 1:        # the product path: hardware-transposed TN GEMM + vectorised column sum
            dwhh = torch.zeros(G4, H, device=dev)
            ops.gemm_tn(dG[l].view(T * B, ldg)[B:], hs[l].view(T * B, ldh), dwhh, G4, H, T * B - B, lda=ldg, ldb=ldh)
            errs[f"tn_dWhh{l}"] = rel(dwhh.cpu().numpy(), dp.T @ hprev) * 10     # vs the same bf16 inputs: fp32 accumulate only
            db = torch.zeros(G4, device=dev)
            ops.colsum_t(dG[l].view(T * B, ldg), T * B, G4, db, ldx=ldg)
            errs[f"tn_db{l}"] = rel(db.cpu().numpy(), dp.sum(0)) * 10
    dx_h = dG[0][:, :, :G4].float().cpu().numpy().reshape(T * B, G4) @ p["g.weight_ih_l0"]
    errs["dx"] = rel(dx_h.reshape(T, B, In), dx)
    return errs


@pytest.mark.parametrize("split", ["1", "0", "2", "1284", "644", "1281", "641", "2562", "2561"])
def test_lstm_bwd_output_gradient_as_a_product(split, monkeypatch):
    """mvae_rnn_bwd with dy given as dy_a . dy_w^T (contracted by the top layer's cell as its second K-segment) against the same call with the
    materialised fp32 dy = dy_a . dy_w^T, in every backward schedule (fused, 2- and 4-way split, unsplit 128 x 128, 256 x 128)."""
    monkeypatch.setenv("MVAE_BWD_SPLIT", split)
    torch.manual_seed(11)
    dt, T, B, H, NL, C = torch.bfloat16, 4, 256, 128, 3, 35
    G4, ldh, ldg, KP = 4 * H, H + 8, 4 * H + 8, 128
    rnd = lambda *sh: torch.randn(*sh, device=dev)
    hs = [(0.5 * rnd(T, B, ldh)).to(dt) for _ in range(NL)]
    cs = [(0.5 * rnd(T, B, H)).to(dt) for _ in range(NL)]
    gates = [torch.sigmoid(rnd(T, B, G4)).to(dt) for _ in range(NL)]
    w_hhT = [(0.1 * rnd(H, G4)).to(dt) for _ in range(NL)]
    w_ihT = [None] + [(0.1 * rnd(H, G4)).to(dt) for _ in range(1, NL)]
    dy_a = torch.zeros(T * B + 8, KP, device=dev, dtype=dt); dy_a[:T * B, :C] = (0.3 * rnd(T * B, C)).to(dt)
    dy_w = torch.zeros(H, KP, device=dev, dtype=dt); dy_w[:, :C] = (0.3 * rnd(H, C)).to(dt)
    dy = dy_a[:T * B].float() @ dy_w.float().t()                      # what models.py used to materialise (fp32 accumulate of bf16 operands)
    out = []
    for mode in ("tensor", "product"):
        dG = [torch.zeros(T, B, ldg, device=dev, dtype=dt) for _ in range(NL)]
        dstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
        if mode == "tensor":
            ops.rnn_bwd(L.CELL_LSTM, dt, T, B, H, w_hhT, [G4] * NL, w_ihT, [G4] * NL, dy, H, hs, ldh, cs, gates, dG, dstate, ldg=ldg)
        else:
            ops.rnn_bwd(L.CELL_LSTM, dt, T, B, H, w_hhT, [G4] * NL, w_ihT, [G4] * NL, None, 0, hs, ldh, cs, gates, dG, dstate, ldg=ldg,
                        dy_a=dy_a[:T * B], dy_w=dy_w, dy_k=KP)
        torch.cuda.synchronize()
        out.append([g.float().cpu().numpy() for g in dG] + [d.cpu().numpy() for d in dstate])
    for a, b in zip(*out):
        assert np.abs(a).max() > 0 and rel(b, a) < 6e-3, rel(b, a)      # dG is stored in bf16: the two summation orders differ by an ulp of it


@pytest.mark.parametrize("shape", [(128, 128, 64), (4096, 1024, 1024), (35, 1024, 4000), (300, 200, 513), (288, 72, 7), (130, 64, 61440 // 4)])
def test_gemm_tn_bf16(shape):
    """C = A^T . B from K-major bf16 operands (ds_read_b64_tr_b16 fragments): ragged M/N/K, split-K, over-read rows."""
    M, N, K = shape
    rs = np.random.RandomState(M + N + K)
    lda, ldb = (M + 7) // 8 * 8 + 8, (N + 7) // 8 * 8 + 16
    A = torch.zeros(K, lda, device=dev, dtype=torch.bfloat16); B = torch.zeros(K, ldb, device=dev, dtype=torch.bfloat16)
    A[:, :M] = t(rs.standard_normal((K, M)), torch.bfloat16); B[:, :N] = t(rs.standard_normal((K, N)), torch.bfloat16)
    A[:, M:] = 3.0; B[:, N:] = -2.0                      # pads must not leak into the result
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_tn(A, B, out, M, N, K, lda=lda, ldb=ldb)
    ref = A[:, :M].double().cpu().numpy().T @ B[:, :N].double().cpu().numpy()
    assert rel(out.cpu().numpy(), ref) < 2e-6


@pytest.mark.parametrize("shape", [(4096, 1024, 5000), (2048, 1024, 4096 + 64)])
def test_gemm_tn_colsum_fused(shape):
    """Weight gradient and bias gradient from one pass over dG: C = A^T . B and colsum[m] = sum_k A[k][m] (ones-fragment MFMA)."""
    M, N, K = shape
    rs = np.random.RandomState(K)
    lda, ldb = M + 64, N + 64
    A = torch.zeros(K + 8, lda, device=dev, dtype=torch.bfloat16); B = torch.zeros(K + 8, ldb, device=dev, dtype=torch.bfloat16)
    A[:K, :M] = t(rs.standard_normal((K, M)), torch.bfloat16); B[:K, :N] = t(rs.standard_normal((K, N)), torch.bfloat16)
    A[K:] = 5.0                                              # rows beyond K must not be summed
    assert ops.gemm_tn_colsum_supported(A, M, N, K)
    out = torch.full((M, N), float("nan"), device=dev); cs = torch.full((M,), float("nan"), device=dev)
    assert ops.gemm_tn_colsum(A, B, out, cs, M, N, K, lda=lda, ldb=ldb)
    Ad = A[:K, :M].double().cpu().numpy()
    ref = Ad.T @ B[:K, :N].double().cpu().numpy()
    assert rel(out.cpu().numpy(), ref) < 2e-6 and rel(cs.cpu().numpy(), Ad.sum(0)) < 2e-6
    assert ops.gemm_tn_colsum(A, B, out, cs, M, N, K, lda=lda, ldb=ldb, colsum_accumulate=True)
    assert rel(cs.cpu().numpy(), 2 * Ad.sum(0)) < 2e-6
    assert not ops.gemm_tn_colsum_supported(A, 35, 1024, K)  # small shapes: the caller falls back to gemm_tn + colsum_t


def test_gemm_tn_grouped_one_launch_no_split_k():
    """The grouped weight-gradient launch: several C = A^T . B problems of different K (and operand offsets) in one grid, each 256 x 256 tile
    accumulated over its full K, column sums of A riding along (fresh and accumulated), accumulate onto C -- against f64 on the same bf16 inputs."""
    rs = np.random.RandomState(3)
    M, N, lda, ldb = 4096, 1024, 4096 + 64, 1024 + 64
    Ks = [700, 520, 1000]
    A = [t(rs.standard_normal((K, lda)) * 0.1, torch.bfloat16) for K in Ks]
    Bm = [t(rs.standard_normal((K, ldb)) * 0.1, torch.bfloat16) for K in Ks]
    outs = [torch.full((M, N), float("nan"), device=dev), torch.full((M, N), float("nan"), device=dev), torch.ones(M, N, device=dev)]
    cs0 = torch.full((M,), float("nan"), device=dev); cs2 = torch.full((M,), 2.0, device=dev)
    probs = [dict(A=A[0], B=Bm[0], out=outs[0], M=M, N=N, K=Ks[0], lda=lda, ldb=ldb, colsum_out=cs0),
             dict(A=A[1][8:], B=Bm[1], out=outs[1], M=M, N=N, K=Ks[1] - 8, lda=lda, ldb=ldb),
             dict(A=A[2], B=Bm[2], out=outs[2], M=M, N=N, K=Ks[2], lda=lda, ldb=ldb, accumulate=True, colsum_out=cs2, colsum_accumulate=True)]
    assert all(ops.gemm_tn_grouped_supported(q["A"], M, N, q["K"], lda, ldb) for q in probs)
    ops.gemm_tn_grouped(probs)
    torch.cuda.synchronize()
    a64 = [x.double().cpu().numpy() for x in A]; b64 = [x.double().cpu().numpy() for x in Bm]
    assert rel(outs[0].cpu().numpy(), a64[0][:, :M].T @ b64[0][:, :N]) < 2e-6
    assert rel(outs[1].cpu().numpy(), a64[1][8:, :M].T @ b64[1][:Ks[1] - 8, :N]) < 2e-6
    assert rel(outs[2].cpu().numpy(), 1.0 + a64[2][:, :M].T @ b64[2][:, :N]) < 2e-6
    assert rel(cs0.cpu().numpy(), a64[0][:, :M].sum(0)) < 2e-6 and rel(cs2.cpu().numpy(), 2.0 + a64[2][:, :M].sum(0)) < 2e-6
    out2 = [torch.empty_like(o) for o in outs[:2]]
    ops.gemm_tn_grouped([dict(probs[0], out=out2[0], colsum_out=None), dict(probs[1], out=out2[1])])
    assert torch.equal(out2[0], outs[0]) and torch.equal(out2[1], outs[1])          # deterministic, independent of the grouping
    # capped grid (mvae_gemm_tn_grouped_capped): 5 / 64 / 1000 workgroups loop over the 128 tiles -- same tiles, same bits
    for cap in (5, 64, 1000):
        out3 = [torch.full((M, N), float("nan"), device=dev), torch.full((M, N), float("nan"), device=dev)]
        cs3 = torch.full((M,), float("nan"), device=dev)
        ops.gemm_tn_grouped([dict(probs[0], out=out3[0], colsum_out=cs3), dict(probs[1], out=out3[1])], max_workgroups=cap)
        assert torch.equal(out3[0], outs[0]) and torch.equal(out3[1], outs[1]) and torch.equal(cs3, cs0), cap


@pytest.mark.parametrize("shape", [(64, 64, 32), (300, 200, 513), (288, 72, 4000), (35, 1024, 1000), (120, 2304, 700), (5, 3, 7)])
def test_gemm_tn_f32_exact(shape):
    """C = A^T . B from K-major f32 operands on the exact-f32 MFMA kernel (conv / encoder weight gradients): ragged M/N/K, split over K."""
    M, N, K = shape
    rs = np.random.RandomState(M * 3 + N + K)
    lda, ldb = (M + 3) // 4 * 4 + 4, (N + 3) // 4 * 4 + 8
    A = torch.zeros(K, lda, device=dev); B = torch.zeros(K, ldb, device=dev)
    A[:, :M] = t(rs.standard_normal((K, M))); B[:, :N] = t(rs.standard_normal((K, N)))
    A[:, M:] = 3.0; B[:, N:] = -2.0                      # pads must not leak into the result
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_tn(A, B, out, M, N, K, lda=lda, ldb=ldb)
    ref = A[:, :M].double().cpu().numpy().T @ B[:, :N].double().cpu().numpy()
    assert rel(out.cpu().numpy(), ref) < 2e-6
    acc = out.clone()
    ops.gemm_tn(A, B, acc, M, N, K, lda=lda, ldb=ldb, accumulate=True)
    assert rel(acc.cpu().numpy(), 2 * ref) < 2e-6


@pytest.mark.parametrize("bm", ["128", "256", "512"])
@pytest.mark.parametrize("shape", [(300, 200, 513), (4096 + 35, 1024, 200), (256, 128, 64)])
def test_gemm_tn_bf16_tile_variants(shape, bm, monkeypatch):
    """All TN tiles (128 x 128, 256 x 128, 256 x 256 = tag 512) on ragged shapes, forced through MVAE_TN_BM."""
    monkeypatch.setenv("MVAE_TN_BM", bm)
    test_gemm_tn_bf16(shape)


@pytest.mark.parametrize("case", [
    (torch.float32, 7, 5, 32, 2, 16, 2e-5),       # tiny, generic register-staged path
    (torch.float32, 4, 6, 20, 2, 8, 2e-5),        # H not a multiple of 8: element-wise (non-vector) epilogues, partial 8-unit groups
    (torch.bfloat16, 7, 5, 32, 2, 16, 3e-2),
    (torch.float32, 5, 70, 72, 3, 8, 2e-5),       # encoder-like H=72, 3 layers: the row-resident schedule (4 rows per workgroup, ragged last one)
    (torch.float32, 1, 6, 72, 1, 8, 2e-5),        # row-resident edge cases: T = 1, single layer (no input gradient)
    (torch.float32, 3, 9, 72, 2, 8, 2e-5),
    (torch.bfloat16, 6, 130, 128, 3, 8, 3e-2),    # LDS-direct pipelined path, ragged rows
    (torch.float32, 4, 200, 64, 2, 8, 2e-5),      # pipelined path, f32 MFMA
    (torch.bfloat16, 3, 256, 192, 4, 8, 3e-2),    # 4 layers, 128-row tiles
    (torch.bfloat16, 1, 9, 64, 2, 8, 3e-2),       # T = 1 edge case
    (torch.bfloat16, 3, 512, 512, 3, 8, 3e-2),    # short contraction with a workgroup per CU: the fused backward is chosen over the split pair
])
def test_lstm_wavefront_fwd_bwd_vs_oracle(case):
    dt, T, B, H, NL, In, tol = case
    errs = _lstm_case(dt, T, B, H, NL, In)
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad


@pytest.mark.parametrize("hoist", ["0", "1"])
def test_row_resident_lstm_with_and_without_hoisted_input_projection(hoist, monkeypatch):
    """The encoder's row-resident schedule (H = 72, f32) in both of its forms: upper layers contracting [x_t | h_{t-1}] per step (what
    large batches take) and with x . W_ih^T for all t hoisted into one GEMM written in place into the saved-gates buffer (small T * B)."""
    monkeypatch.setenv("MVAE_ROWRES_HOIST", hoist)
    errs = _lstm_case(torch.float32, 6, 37, 72, 3, 8)
    bad = {k: v for k, v in errs.items() if v > 2e-5}
    assert not bad, bad


@pytest.mark.parametrize("env", [{"MVAE_BJ": "64"}, {"MVAE_BM": "128", "MVAE_NBUF_FWD": "4", "MVAE_NBUF_BWD": "5"},
                                 {"MVAE_BM": "64", "MVAE_BJ": "32", "MVAE_NBUF_FWD": "5", "MVAE_NBUF_BWD": "3"},
                                 {"MVAE_BM": "64", "MVAE_BJ": "32", "MVAE_NBUF_FWD": "2", "MVAE_FWD_GM": "0"}])
def test_lstm_tile_variants_vs_oracle(env, monkeypatch):
    """The tile / ring-depth variants the heuristics pick at production sizes, forced here at an oracle-checkable size."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    errs = _lstm_case(torch.bfloat16, 4, 200, 192, 3, 8)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, bad


@pytest.mark.parametrize("split", ["0", "2", "2562", "1284", "644", "1281", "641", "2561"])
def test_lstm_bwd_split_segment_schedule(split, monkeypatch):
    """Backward with the contraction split across workgroups -- by K-segment (128x128 or 256x128 partial tiles) or by half segment
    (4 partial tiles per output: the small-batch schedules) + element-wise second launch -- vs the fused single-launch form."""
    monkeypatch.setenv("MVAE_BWD_SPLIT", split)
    errs = _lstm_case(torch.bfloat16, 5, 256, 128, 3, 8)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, bad


@pytest.mark.parametrize("gm", ["256256", "256128", "128128", "128064"])
@pytest.mark.parametrize("shape", [(4, 200, 192, 3), (3, 256, 128, 4), (1, 70, 64, 2)])
def test_lstm_fwd_gate_major_tiles(gm, shape, monkeypatch):
    """The gate-major forward tile (weights as the MFMA A operand, permuted LDS rows, cell update in registers) in each of its four
    (weight rows, batch rows) shapes, forced at oracle-checkable sizes incl. ragged batch tiles and T = 1."""
    monkeypatch.setenv("MVAE_FWD_GM", gm)
    T, B, H, NL = shape
    errs = _lstm_case(torch.bfloat16, T, B, H, NL, 8)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, bad


@pytest.mark.parametrize("case", [
    (3, 7, 23, 5, 4, 8, 8, 29),        # tiny, nothing aligned to a K-step: generic loops, ragged tiles
    (9, 24, 40, 20, 18, 32, 32, 44),   # LDS-direct path (k * ldx % 32 == 0), padded channels, batch stride larger than W * ldx
    (130, 120, 72, 120, 18, 128, 128, 96),   # conv_1 of the model at batch 130
    (64, 64, 38, 64, 18, 64, 64, 38),  # conv_3
])
def test_conv1d_selu_sliding_window_vs_oracle(case):
    """Channels-last sliding-window Conv1d + SELU (forward, dX, dW, db) against the oracle's im2col formulation."""
    B, Cin, W, Cout, k, ldx, ldo, Wbuf = case
    rs = np.random.RandomState(B + Cin + W)
    x = rs.standard_normal((B, Cin, W)) * 0.5
    w = rs.standard_normal((Cout, Cin, k)) * (1.0 / np.sqrt(Cin * k)); b = rs.standard_normal(Cout) * 0.1
    y_ref, cache = O.conv_selu_fwd(x, w, b)                       # [B, Cout, Wout]
    Wout = W - k + 1
    dy = rs.standard_normal(y_ref.shape)
    dx_ref, dw_ref, db_ref = O.conv_selu_bwd(dy, cache)
    xd = torch.zeros(B, Wbuf, ldx, device=dev); xd[:, :W, :Cin] = t(x.transpose(0, 2, 1))
    wd, bd = t(w), t(b)
    wp = torch.full((Cout, k * ldx), 7.0, device=dev); wq = torch.full((Cin, k * ldo), 7.0, device=dev)
    ops.conv1d_pack_weights(wd, Cin, Cout, k, ldx, wp, ldo, wq)
    y = torch.zeros(B * Wout, ldo, device=dev)
    ops.conv1d_selu_fwd(xd, B, W, ldx, Wbuf * ldx, Cout, k, wp, bd, y, ldo)
    assert rel(y.view(B, Wout, ldo)[:, :, :Cout].cpu().numpy().transpose(0, 2, 1), y_ref) < 1e-5
    assert float(y[:, Cout:].abs().sum()) == 0
    dyd = torch.zeros(B * Wout, ldo, device=dev); dyd[:, :Cout] = t(dy.transpose(0, 2, 1).reshape(B * Wout, Cout))
    dzp = torch.full((B * (Wout + 2 * k - 2), ldo), 3.0, device=dev)
    dw = torch.empty(Cout, Cin, k, device=dev); db = torch.empty(Cout, device=dev); dx = torch.zeros(B * W, ldx, device=dev)
    ops.conv1d_selu_bwd(B, W, Cin, ldx, Wbuf * ldx, Cout, ldo, k, dyd, y, xd, wq, dzp, dw, db, dx, ldx)
    assert rel(dw.cpu().numpy(), dw_ref) < 1e-5 and rel(db.cpu().numpy(), db_ref) < 1e-5      # f32 sums of up to 3e4 terms vs f64
    assert rel(dx.view(B, W, ldx)[:, :, :Cin].cpu().numpy().transpose(0, 2, 1), dx_ref) < 1e-5
    # the bf16 training mode's form of the two gradient GEMMs (fp32 operands as 3 x bf16 products, MVAE_CONV_BWD_X3): ~16 mantissa bits
    dw3 = torch.empty_like(dw); db3 = torch.empty_like(db); dx3 = torch.zeros_like(dx)
    ops.conv1d_selu_bwd(B, W, Cin, ldx, Wbuf * ldx, Cout, ldo, k, dyd, y, xd, wq, dzp, dw3, db3, dx3, ldx, x3=True)
    assert rel(dw3.cpu().numpy(), dw_ref) < 5e-5 and rel(db3.cpu().numpy(), db_ref) < 1e-5
    assert rel(dx3.view(B, W, ldx)[:, :, :Cin].cpu().numpy().transpose(0, 2, 1), dx_ref) < 5e-5


def test_small_ops_vs_oracle():
    rs = np.random.RandomState(3)
    B, Lq, C, o = 5, 11, 12, 16
    logits = rs.standard_normal((Lq * B, C)) * 2
    recon = torch.empty(B, Lq, C, device=dev)
    ops.softmax_tb_fwd(t(logits), C, recon, B, Lq, C)
    e = np.exp(logits - logits.max(1, keepdims=True)); pr = (e / e.sum(1, keepdims=True)).reshape(Lq, B, C).transpose(1, 0, 2)
    assert rel(recon.cpu().numpy(), pr) < 1e-6
    idx = rs.randint(0, C, (B, Lq)); ohe = O.one_hot(idx, C)
    mu = rs.standard_normal((B, o)); lv = rs.standard_normal((B, o))
    out = torch.empty(3, device=dev)
    ops.bce_kl_loss_fwd(recon, t(ohe), t(mu), t(lv), Lq, out)
    ref = O.bce_kl_loss(pr, ohe, mu, lv, Lq)
    assert rel(out.cpu().numpy(), np.array(ref)) < 1e-6
    drecon = torch.empty_like(recon); dmu = torch.empty(B, o, device=dev); dlv = torch.empty(B, o, device=dev)
    ops.bce_kl_loss_bwd(recon, t(ohe), t(mu), t(lv), Lq, None, drecon, dmu, dlv)
    r = O.bce_kl_loss_bwd(pr, ohe, mu, lv, Lq)
    assert rel(drecon.cpu().numpy(), r[0]) < 1e-5 and rel(dmu.cpu().numpy(), r[1]) < 1e-6 and rel(dlv.cpu().numpy(), r[2]) < 1e-6
    # BCELoss edge cases: p exactly 0 / 1 hit the -100 log clamp and the 1e-12 denominator clamp
    pe = torch.tensor([0.0, 1.0, 1.0, 0.0, 0.5], device=dev); te = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0], device=dev)
    ops.bce_kl_loss_fwd(pe, te, t(mu), t(lv), 1.0, out)
    assert abs(float(out[1]) - (100 + 100 + 0 + 0 + np.log(2)) / 5) < 1e-4
    # adam + clip
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
    assert rel(tp.cpu().numpy(), P["w"]) < 1e-6 and abs(float(norm) - total) < 1e-4 * total


def _model_check(dims, params, idx, eps, dtype, tol_loss, tol_lat, tol_recon, tol_grad, golden_loss=None):
    p64 = {k: np.asarray(v, np.float64) for k, v in params.items()}
    ref = O.molvae_loss_and_grads(p64, idx, np.asarray(eps, np.float64), max_len=dims["i"], num_lstm=dims["n_enc"], num_gru=dims["n_dec"])
    enc, dec = gh.build_modules(dims, params, dtype)
    out = gh.run_hip(enc, dec, idx, eps, dims["i"])
    assert abs(out["loss"] - ref["loss"]) < tol_loss * abs(ref["loss"])
    if golden_loss is not None:          # fixture generated from the reference itself
        assert abs(out["loss"] - golden_loss) < tol_loss * abs(golden_loss)
    assert rel(out["mu"], ref["mu"]) < tol_lat and rel(out["logvar"], ref["logvar"]) < tol_lat
    assert rel(out["recon"], ref["recon"]) < tol_recon
    rep = gh.grad_report(out["grads"], ref["grads"])
    bad = {k: v for k, v in rep.items() if v > tol_grad}
    assert not bad, bad
    return out, ref


def test_g1_small_model_f32_and_bf16(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_small.npz"))
    params = gh.g1_dims_params(np.float32)
    out, _ = _model_check(gh.G1, params, g["idx"], g["eps"], torch.float32, 1e-5, 1e-5, 1e-5, 2e-4, float(g["loss"]))
    assert rel(out["mu"], g["mu"]) < 1e-5 and rel(out["logvar"], g["logvar"]) < 1e-5 and rel(out["recon"], g["recon"]) < 1e-5
    for k in [f for f in g.files if f.startswith("grad.")]:
        assert rel(out["grads"][k[5:]], g[k]) < 2e-4, k
    _model_check(gh.G1, params, g["idx"], g["eps"], torch.bfloat16, 1e-4, 1e-5, 5e-3, 5e-2, float(g["loss"]))


def test_g2_full_dims_f32_and_bf16(golden_dir):
    """Full model (L=120, C=35, o=292, 4x LSTM-1024), B=4, against the oracle AND the reference-generated fixture."""
    g = np.load(os.path.join(golden_dir, "g2_full.npz"))
    params = ip.init_params(ip.molvae_shapes(), 202, 1.5, np.float32)
    out, _ = _model_check(gh.FULL, params, g["idx"], g["eps"], torch.float32, 1e-5, 1e-5, 1e-5, 5e-4, float(g["loss"]))
    assert rel(out["mu"], g["mu"]) < 1e-5 and rel(out["logvar"], g["logvar"]) < 1e-5
    assert rel(out["recon"][:, ::17, :], g["recon_rows"]) < 1e-5
    for k, gr in out["grads"].items():
        assert abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - float(g["gnorm." + k])) < 5e-4 * float(g["gnorm." + k]) + 1e-12, k
    _model_check(gh.FULL, params, g["idx"], g["eps"], torch.bfloat16, 1e-4, 1e-5, 5e-3, 5e-2, float(g["loss"]))


def test_g4_clip_adam_trajectory(golden_dir):
    """5 steps of train.py:95-104 (clip 3.0 + Adam 8e-4) with FusedAdam reproduce the reference's loss trajectory."""
    g = np.load(os.path.join(golden_dir, "g4_traj.npz")); g1 = np.load(os.path.join(golden_dir, "g1_small.npz"))
    enc, dec = gh.build_modules(gh.G1, gh.g1_dims_params(np.float32), torch.float32)
    opt = mv.FusedAdam(list(enc.parameters()) + list(dec.parameters()), lr=0.0008, max_grad_norm=3.0)
    idx = torch.from_numpy(g1["idx"]).to(dev)
    ohe = torch.nn.functional.one_hot(idx, gh.G1["c"]).float()
    for step in range(5):
        opt.zero_grad(set_to_none=True)
        z, mu, logvar = enc(idx, t(g["eps"][step]))
        loss = mv.bce_kl_loss(dec(z), ohe, mu, logvar, gh.G1["i"])
        loss.backward()
        opt.step()
        assert abs(float(loss) - g["losses"][step]) < 2e-4 * g["losses"][step], (step, float(loss), g["losses"][step])
        assert abs(float(opt.last_grad_norm) - g["gnorms"][step]) < 1e-3 * g["gnorms"][step]
    for pre, m in (("encoder.", enc), ("decoder.", dec)):
        for k, v in m.named_parameters():
            a = v.detach().cpu().numpy().astype(np.float64)
            assert abs(np.sqrt((a ** 2).sum()) - float(g["fnorm." + pre + k])) < 1e-4 * (1 + float(g["fnorm." + pre + k])), k


@pytest.mark.parametrize("B", [512, 128, 1024])
def test_full_size_properties_batch512(B):
    """BASELINE.json full sizes -- configs[1] (B=512), the per-rank shape of configs[2] (b=128 = 1024 / 8) and the metric's own batch
    (B=1024, one GPU) at L=120, C=35: properties that need no oracle run.
       * recon rows are probability vectors; * the fused loss equals a plain torch evaluation of train.py:31-38 on the same
       recon; * run-to-run bitwise determinism of loss and gradients; * batch-row independence: the first 64 molecules give the
       same mu as a B=64 run."""
    torch.manual_seed(42)
    model = mv.MolecularVAE().to(dev)
    gen = torch.Generator().manual_seed(1234)
    idx = torch.randint(0, 35, (B, 120), generator=gen).to(dev)
    eps = (1e-2 * torch.randn(B, 292, generator=gen)).to(dev)
    ohe = torch.nn.functional.one_hot(idx, 35).float()

    def run(ix, ep, oh):
        model.zero_grad(set_to_none=True)
        recon, mu, logvar = model(ix, ep)
        loss = mv.bce_kl_loss(recon, oh, mu, logvar, 120)
        loss.backward()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters()))
        return recon.detach(), mu.detach(), logvar.detach(), loss.detach(), gn

    recon, mu, logvar, loss, gn = run(idx, eps, ohe)
    assert torch.allclose(recon.sum(-1), torch.ones(B, 120, device=dev), atol=1e-5) and float(recon.min()) >= 0
    bce = torch.nn.functional.binary_cross_entropy(recon.reshape(-1), ohe.reshape(-1))
    ref_loss = 120 * bce - 0.5 * torch.mean(1. + mu - logvar ** 2 - torch.exp(mu))
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    recon2, mu2, logvar2, loss2, gn2 = run(idx, eps, ohe)
    assert torch.equal(recon, recon2) and torch.equal(mu, mu2) and float(loss) == float(loss2) and float(gn) == float(gn2)
    _, mu64, _, _, _ = run(idx[:64], eps[:64], ohe[:64])
    assert torch.allclose(mu64, mu[:64], rtol=1e-5, atol=1e-6)
    assert np.isfinite(float(gn)) and float(gn) > 0


def test_training_trajectory_vs_stock_torch_cpu_port():
    """Five train.py-style steps (clip 3.0 + Adam 8e-4) of MolecularVAE with FusedAdam (gradients written into the optimiser's flat
    buffer, weight-gradient GEMMs forked / parked on the side stream) against stock torch.nn modules + torch.optim.Adam on the CPU with
    the same weights, batches and noise: losses and pre-clip gradient norms must track step by step -- to 1e-6 in f32 mode, and within
    the bf16 decoder's rounding (2e-4 on the loss) in the default mode."""
    from oracle import torch_ref
    Lq, V, B = 120, 26, 16
    torch.manual_seed(42)
    m32 = mv.MolecularVAE(i=Lq, c=V, dtype=torch.float32)
    ref = torch_ref.CpuPort(i=Lq, o=292, c=V)
    ref.load_state_dict(m32.state_dict())
    mbf = mv.MolecularVAE(i=Lq, c=V)
    mbf.load_state_dict(m32.state_dict())
    m32, mbf = m32.to(dev), mbf.to(dev)
    o32 = mv.FusedAdam(m32.parameters(), lr=8e-4, max_grad_norm=3.0)
    obf = mv.FusedAdam(mbf.parameters(), lr=8e-4, max_grad_norm=3.0)
    ropt = torch.optim.Adam(ref.parameters(), lr=8e-4)
    loss_fn = mv.make_loss_function(Lq)
    g = torch.Generator().manual_seed(3)
    for s in range(5):
        idx = torch.randint(0, V, (B, Lq), generator=g)
        eps = 1e-2 * torch.randn(B, 292, generator=g)
        ohe = torch.nn.functional.one_hot(idx, V).float()
        l32 = mv.train_step(m32, o32, loss_fn, idx.to(dev), ohe.to(dev), eps=eps.to(dev))
        lbf = mv.train_step(mbf, obf, loss_fn, idx.to(dev), ohe.to(dev), eps=eps.to(dev))
        ropt.zero_grad(set_to_none=True)
        rr, rmu, rlv = ref(idx, eps)
        rloss = torch_ref.elbo(rr, ohe, rmu, rlv, Lq)
        rloss.backward()
        gn = float(torch.nn.utils.clip_grad_norm_(ref.parameters(), 3.0))
        ropt.step()
        rl = float(rloss.detach())
        assert abs(float(l32) - rl) < 1e-6 * abs(rl) and abs(float(o32.last_grad_norm) - gn) < 1e-4 * gn, (s, float(l32), rl)
        assert abs(float(lbf) - rl) < 2e-4 * abs(rl) and abs(float(obf.last_grad_norm) - gn) < 5e-2 * gn, (s, float(lbf), rl)


def test_long_sequence_config_seq256_vocab64():
    """BASELINE.json configs[4] shape (seq_len 256, vocab 64) at a modest batch: f32 path against stock torch.nn modules carrying the same
    weights (oracle/torch_ref.CpuPort), bf16 path against the f32 path, determinism."""
    from oracle import torch_ref
    Lq, V, B = 256, 64, 8
    torch.manual_seed(7)
    m32 = mv.MolecularVAE(i=Lq, c=V, dtype=torch.float32)
    ref = torch_ref.CpuPort(i=Lq, o=292, c=V)
    ref.load_state_dict(m32.state_dict())
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, V, (B, Lq), generator=gen)
    eps = 1e-2 * torch.randn(B, 292, generator=gen)
    ohe = torch.nn.functional.one_hot(idx, V).float()
    r_recon, r_mu, r_lv = ref(idx, eps)
    r_loss = torch_ref.elbo(r_recon, ohe, r_mu, r_lv, Lq)
    r_loss.backward()
    m32 = m32.to(dev)
    recon, mu, lv = m32(idx.to(dev), eps.to(dev))
    loss = mv.bce_kl_loss(recon, ohe.to(dev), mu, lv, Lq)
    loss.backward()
    assert abs(float(loss.detach()) - float(r_loss.detach())) < 1e-5 * abs(float(r_loss.detach()))
    assert rel(mu.detach().cpu().numpy(), r_mu.detach().numpy()) < 1e-4 and rel(lv.detach().cpu().numpy(), r_lv.detach().numpy()) < 1e-4
    gr = dict(ref.named_parameters())
    worst = max(rel(p.grad.cpu().numpy(), gr[n].grad.numpy()) for n, p in m32.named_parameters())
    assert worst < 2e-4, worst
    mb = mv.MolecularVAE(i=Lq, c=V).to(dev)                 # bf16 decoder storage
    mb.load_state_dict(m32.state_dict())
    out = []
    for _ in range(2):
        mb.zero_grad(set_to_none=True)
        rb, mub, lvb = mb(idx.to(dev), eps.to(dev))
        lb = mv.bce_kl_loss(rb, ohe.to(dev), mub, lvb, Lq)
        lb.backward()
        out.append((float(lb.detach()), float(sum(p.grad.double().abs().sum() for p in mb.parameters()))))
    assert out[0] == out[1]
    assert abs(out[0][0] - float(loss.detach())) < 1e-4 * abs(float(loss.detach())) and torch.equal(mub, mu)


def test_long_sequence_config_full_size_batch2048():
    """BASELINE.json configs[4] at its FULL size -- seq_len 256, vocab 64, batch 2048 (K = T*B = 524288 rows in the weight-gradient
    contractions: operands beyond one 2 GiB buffer descriptor, chunked inside the launcher): one complete optimiser step, twice from the
    same state.  Size-independent properties: probability rows, the fused loss against a plain evaluation of train.py:31-38 on the same
    recon, bitwise run-to-run determinism of loss / gradient norm / updated parameters, finite non-zero gradients for every parameter,
    and the weight gradient of the top decoder layer against a direct f32 contraction of the saved K-major operands on a row slice."""
    Lq, V, B = 256, 64, 2048
    gen = torch.Generator().manual_seed(77)
    idx = torch.randint(0, V, (B, Lq), generator=gen).to(dev)
    eps = (1e-2 * torch.randn(B, 292, generator=gen)).to(dev)
    ohe = torch.nn.functional.one_hot(idx, V).float()
    loss_fn = mv.make_loss_function(Lq)
    res = []
    for rep in range(2):
        torch.manual_seed(42)
        model = mv.MolecularVAE(i=Lq, c=V).to(dev)
        opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
        opt.zero_grad(set_to_none=True)
        recon, mu, lv = model(idx, eps)
        loss = loss_fn(recon, ohe, mu, lv)
        loss.backward()
        if rep == 0:
            assert torch.allclose(recon.sum(-1), torch.ones(B, Lq, device=dev), atol=1e-5) and float(recon.min()) >= 0
            bce = torch.nn.functional.binary_cross_entropy(recon.reshape(-1), ohe.reshape(-1))
            ref_loss = Lq * bce - 0.5 * torch.mean(1. + mu - lv ** 2 - torch.exp(mu))
            assert abs(float(loss) - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
            for n, p_ in model.named_parameters():
                g = p_.grad
                assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0, n
            # dW_hh of the top layer = dG[3][1:]^T . hs[3][:-1]: check 64 output rows against a direct contraction (f32, chunked over time)
            ws = model.decoder._ws
            dG = [b for k, b in ws.bufs.items() if k[0] == "dG3"][0]; hs = [b for k, b in ws.bufs.items() if k[0] == "hs3"][0]
            ref = torch.zeros(64, 1024, device=dev)
            for t in range(1, Lq):
                ref += dG[t, :, 4096 - 64:4096].float().t() @ hs[t - 1, :, :1024].float()
            got = model.decoder.gru.weight_hh_l3.grad[4096 - 64:4096]
            assert float((got - ref).abs().max()) < 2e-3 * float(ref.abs().max())
        opt.step()
        torch.cuda.synchronize()
        res.append((float(loss.detach()), float(opt.last_grad_norm), float(sum(p_.detach().double().abs().sum() for p_ in model.parameters()))))
        del model, opt, recon, mu, lv, loss
        ops.release_caches()
        torch.cuda.empty_cache()
    assert res[0] == res[1] and np.isfinite(res[0][0]) and res[0][1] > 0


# ---------------------------------------------------------------------------------------------- MOSES GRU path (mosesvae.py)
def _moses_setup(golden_dir, dtype):
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    g = np.load(os.path.join(golden_dir, "g3_moses.npz"))
    chars = [chr(ord("a") + i) for i in range(26)]
    v = VC.OneHotVocab(chars)
    V = len(v)
    assert V == int(g["V"]) and v.pad == int(g["pad"]) and v.bos == int(g["bos"])
    model = MV.VAE(v, dtype=dtype)
    params = ip.init_params(ip.moses_shapes(V), 303, 1.5, np.float32)
    sd = {k: torch.from_numpy(params[k]) for k in params}
    model.load_state_dict({k: sd[_moses_base(k)] for k in model.state_dict()})
    return g, model.to(dev).eval(), params        # g3 was recorded with model.eval() (dropout off); tests that want train() say so


def _moses_base(k):
    for a, b in (("vae.0.", "x_emb."), ("vae.1.0.", "x_emb."), ("encoder.0.", "x_emb."), ("encoder.1.", "encoder_rnn."),
                 ("encoder.2.", "q_mu."), ("encoder.3.", "q_logvar."), ("decoder.0.", "decoder_rnn."), ("decoder.1.", "decoder_lat."),
                 ("decoder.2.", "decoder_fc."), ("vae.1.1.", "encoder_rnn."), ("vae.1.2.", "q_mu."), ("vae.1.3.", "q_logvar."),
                 ("vae.2.0.", "decoder_rnn."), ("vae.2.1.", "decoder_lat."), ("vae.2.2.", "decoder_fc.")):
        if k.startswith(a):
            return b + k[len(a):]
    return k


@pytest.mark.parametrize("dtype,tl,tg", [(torch.float32, 2e-5, 5e-4), (torch.bfloat16, 5e-3, 6e-2)])
def test_g3_moses_gru_vae(golden_dir, dtype, tl, tg):
    """mosesvae.VAE.forward on the reference-generated fixture (30 symbols, B=6 ragged lengths, dropout off) and vs the oracle:
    kl, recon, z, logvar, logits y, and every parameter gradient of  kl_w * kl + recon."""
    g, model, params = _moses_setup(golden_dir, dtype)
    seqs = [torch.from_numpy(g[f"seq{b}"]) for b in range(6)]
    eps = torch.from_numpy(g["eps"].astype(np.float32)).to(dev)
    kl, recon, z, logvar, x, y = model(seqs, eps)
    assert (x.cpu().numpy() == g["x"]).all()
    assert abs(float(kl) - float(g["kl"])) < tl * abs(float(g["kl"])) and abs(float(recon) - float(g["recon"])) < tl * abs(float(g["recon"]))
    assert rel(z.detach().cpu().numpy(), g["z"]) < tl and rel(logvar.detach().cpu().numpy(), g["logvar"]) < tl
    assert rel(y.detach().cpu().numpy(), g["y"]) < max(tl, 1e-5)
    kl_w = float(g["kl_w"])
    model.zero_grad(set_to_none=True)
    (kl_w * kl + recon).backward()
    torch.cuda.synchronize()
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref = O.moses_forward(p64, [g[f"seq{b}"] for b in range(6)], g["eps"], int(g["pad"]))
    rg = ref["grads_for"](kl_w)
    bad = {}
    for k, p_ in model.named_parameters():
        e = rel(p_.grad.cpu().numpy(), rg[k])
        if e > tg:
            bad[k] = e
        gr = p_.grad.double().cpu().numpy()
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["gnorm." + k])) < 2 * tg * float(g["gnorm." + k]) + 1e-12, k   # reference fixture
    assert not bad, bad


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gru_token_table_addend_equals_gathered_sequence(dtype):
    """mvae_rnn_fwd with add_table / add_index (+ a per-sequence addend) gives the same bits as the gathered [T, B, 4H] sequence it
    replaces (mvae_gather_rows_tb -> add0), ragged lengths included; ids outside the table are clamped the same way."""
    from molecular_vae_amd import _lib as LL
    torch.manual_seed(3)
    T, B, H, V = 9, 70, 128, 30
    ld = H + 64
    w = (0.1 * torch.randn(4 * H, ld, device=dev)).to(dtype); w[:, H:] = 0
    bias = 0.1 * torch.randn(4 * H, device=dev)
    tbl = torch.randn(V, 4 * H, device=dev); base = torch.randn(B, 4 * H, device=dev)
    idx = torch.randint(0, V, (B, T + 3), device=dev)
    idx[0, 0], idx[1, 2] = -4, V + 9                                   # clamped
    lengths = torch.sort(torch.randint(1, T + 1, (B,)), descending=True).values.to(torch.int32).to(dev)
    out = []
    for mode in ("gathered", "table"):
        hs = torch.zeros(T, B, ld, dtype=dtype, device=dev); gates = torch.zeros(T, B, 4 * H, dtype=dtype, device=dev)
        hstate = torch.zeros(2, B, H, device=dev)
        if mode == "gathered":
            add = torch.empty(T, B, 4 * H, device=dev)
            ops.gather_rows_tb(idx[:, :T].contiguous(), tbl, add, B, T, V, 4 * H, base=base)
            ops.rnn_fwd(LL.CELL_GRU, dtype, T, B, H, add, B * 4 * H, [w], [ld], [w], [ld], [bias], [hs], ld, None, [gates], [hstate], lengths=lengths)
        else:
            ops.rnn_fwd(LL.CELL_GRU, dtype, T, B, H, base, 0, [w], [ld], [w], [ld], [bias], [hs], ld, None, [gates], [hstate], lengths=lengths,
                        add_table=tbl, add_index=idx)
        torch.cuda.synchronize()
        out.append((hs.float().cpu(), gates.float().cpu(), hstate.cpu()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert out[0][0].abs().sum() > 0


def test_token_scatter_as_onehot_contraction():
    """dtable = onehot(x)^T . d (mvae_onehot_tb + the TN GEMM) against the deterministic scatter kernel it replaces on the bf16 path."""
    torch.manual_seed(4)
    B, T, V, W, ldd = 96, 21, 30, 768, 768 + 64
    idx = torch.randint(0, V, (B, T), device=dev)
    d = torch.randn(T * B + 8, ldd, device=dev).to(torch.bfloat16)
    ref = torch.empty(V, W, device=dev)
    ops.scatter_rows_tb(idx, d[:T * B].view(T, B, ldd), ref, B, T, V, W, ldd=ldd)
    oh = torch.full((T * B + 8, 32), 7.0, device=dev, dtype=torch.bfloat16)
    ops.onehot_tb(idx, oh[:T * B], B, T, V)
    want = torch.nn.functional.one_hot(idx.t().reshape(-1), 32).to(torch.bfloat16)
    assert torch.equal(oh[:T * B], want) and (oh[T * B:] == 7.0).all()
    oh[T * B:] = 0
    got = torch.empty(V, W, device=dev)
    ops.gemm_tn(oh[:T * B], d[:T * B], got, V, W, T * B, lda=32, ldb=ldd)
    torch.cuda.synchronize()
    exact = torch.zeros(V, W, dtype=torch.float64, device=dev).index_add_(0, idx.t().reshape(-1), d[:T * B, :W].double())
    assert rel(got.cpu().numpy(), exact.cpu().numpy()) < 1e-6 and rel(ref.cpu().numpy(), exact.cpu().numpy()) < 1e-6


def test_moses_larger_batch_vs_oracle():
    """B=40, lengths 12..60 (MOSES-like), V=30: GRU kernels on the LDS-direct path (H=256/512 are whole K-steps), bf16."""
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    chars = [chr(ord("a") + i) for i in range(26)]
    v = VC.OneHotVocab(chars)
    V = len(v)
    params = ip.init_params(ip.moses_shapes(V), 11, 1.0, np.float32)
    model = MV.VAE(v, dtype=torch.bfloat16)
    model.load_state_dict({k: torch.from_numpy(params[_moses_base(k)]) for k in model.state_dict()})
    model = model.to(dev).eval()
    rs = np.random.RandomState(5)
    lens = sorted(rs.randint(10, 58, size=40).tolist(), reverse=True)
    seqs = [np.concatenate([[v.bos], rs.randint(0, 26, size=n), [v.eos]]).astype(np.int64) for n in lens]
    eps = rs.standard_normal((40, 160)).astype(np.float32)
    kl, recon, z, logvar, x, y = model([torch.from_numpy(s) for s in seqs], torch.from_numpy(eps).to(dev))
    (0.5 * kl + recon).backward()
    ref = O.moses_forward({k: a.astype(np.float64) for k, a in params.items()}, seqs, eps.astype(np.float64), v.pad)
    assert abs(float(kl) - ref["kl"]) < 5e-3 * abs(ref["kl"]) and abs(float(recon) - ref["recon"]) < 5e-3 * abs(ref["recon"])
    rg = ref["grads_for"](0.5)
    bad = {k: rel(p_.grad.cpu().numpy(), rg[k]) for k, p_ in model.named_parameters() if rel(p_.grad.cpu().numpy(), rg[k]) > 8e-2}
    assert not bad, bad


@pytest.mark.parametrize("dtype,tl,tg", [(torch.float32, 2e-5, 5e-4), (torch.bfloat16, 5e-3, 6e-2)])
def test_g6_moses_train_mode_dropout_vs_reference_fixture(golden_dir, dtype, tl, tg):
    """mosesvae.VAE in train() mode: the decoder GRU's inter-layer dropout (mosesvae.py:73-79).  The fixture holds the keep masks the
    REFERENCE drew (reconstructed from its generator, tests/golden/make_golden.py make_g6) with its kl / recon / y / gradients; the HIP
    path gets the same masks injected and must land on the same numbers; eval() on the same model must give g3's (different) numbers."""
    g3, model, params = _moses_setup(golden_dir, dtype)
    g = np.load(os.path.join(golden_dir, "g6_moses_train.npz"))
    seqs = [torch.from_numpy(g[f"seq{b}"]) for b in range(6)]
    eps = torch.from_numpy(g["eps"].astype(np.float32)).to(dev)
    model.train()
    kl, recon, z, logvar, x, y = model(seqs, eps, drop_mask=g["masks"])
    assert abs(float(kl) - float(g["kl"])) < tl * abs(float(g["kl"])) and abs(float(recon) - float(g["recon"])) < max(tl, 1e-5) * abs(float(g["recon"]))
    assert rel(y.detach().cpu().numpy(), g["y"]) < 10 * tl
    assert abs(float(recon) - float(g3["recon"])) > 1e-3 * abs(float(g3["recon"]))          # dropout really changed the network
    kl_w = float(g["kl_w"])
    model.zero_grad(set_to_none=True)
    (kl_w * kl + recon).backward()
    torch.cuda.synchronize()
    for k, p_ in model.named_parameters():
        gr = p_.grad.double().cpu().numpy()
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["gnorm." + k])) < 2 * tg * float(g["gnorm." + k]) + 1e-12, k
        sl = gr.reshape(-1)[:: max(1, gr.size // 64)][:64]
        assert rel(sl, g["gslice." + k]) < 4 * tg or np.abs(g["gslice." + k]).max() < 1e-12, k
    model.eval()                                            # same sequences, g3's noise: the deterministic network, g3's numbers
    kl_e, recon_e, *_ = model(seqs, torch.from_numpy(g3["eps"].astype(np.float32)).to(dev))
    assert abs(float(recon_e) - float(g3["recon"])) < max(tl, 1e-5) * abs(float(g3["recon"]))


@pytest.mark.parametrize("split", ["1", "644"])
def test_moses_device_generated_dropout_mask_vs_oracle(split, monkeypatch):
    """Train mode with the mask generated ON DEVICE from an explicit seed (counter-based hash, include/mvae.h mvae_dropout_keep): the
    oracle, fed with the host restatement of the same hash, must agree -- on the fused backward tiles and on the split-K schedule whose
    element-wise kernel applies the mask (forced at B = 128); the same seed reproduces the loss bit for bit, another seed does not."""
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    monkeypatch.setenv("MVAE_BWD_SPLIT", split)
    chars = [chr(ord("a") + i) for i in range(26)]
    v = VC.OneHotVocab(chars)
    V = len(v)
    params = ip.init_params(ip.moses_shapes(V), 12, 1.0, np.float32)
    model = MV.VAE(v, dtype=torch.bfloat16)
    model.load_state_dict({k: torch.from_numpy(params[_moses_base(k)]) for k in model.state_dict()})
    model = model.to(dev).train()
    rs = np.random.RandomState(6)
    B = 128
    lens = sorted(rs.randint(8, 30, size=B).tolist(), reverse=True)
    seqs = [np.concatenate([[v.bos], rs.randint(0, 26, size=n), [v.eos]]).astype(np.int64) for n in lens]
    eps = rs.standard_normal((B, 160)).astype(np.float32)
    tseqs = [torch.from_numpy(s) for s in seqs]
    kl, recon, *_ = model(tseqs, torch.from_numpy(eps).to(dev), drop_seed=20241)
    assert model.last_drop_seed == 20241
    (0.5 * kl + recon).backward()
    T = max(len(s) for s in seqs)
    masks = ops.dropout_keep_mask(20241, (2, T, B, 512), 0.2)
    ref = O.moses_forward({k: a.astype(np.float64) for k, a in params.items()}, seqs, eps.astype(np.float64), v.pad, drop_masks=masks, drop_p=0.2)
    assert abs(float(kl) - ref["kl"]) < 5e-3 * abs(ref["kl"]) and abs(float(recon) - ref["recon"]) < 5e-3 * abs(ref["recon"])
    rg = ref["grads_for"](0.5)
    bad = {k: rel(p_.grad.cpu().numpy(), rg[k]) for k, p_ in model.named_parameters() if rel(p_.grad.cpu().numpy(), rg[k]) > 8e-2}
    assert not bad, bad
    with torch.no_grad():
        r_same = float(model(tseqs, torch.from_numpy(eps).to(dev), drop_seed=20241)[1])
        r_other = float(model(tseqs, torch.from_numpy(eps).to(dev), drop_seed=7)[1])
        torch.manual_seed(3); r_a = float(model(tseqs, torch.from_numpy(eps).to(dev))[1]); s_a = model.last_drop_seed
        torch.manual_seed(3); r_b = float(model(tseqs, torch.from_numpy(eps).to(dev))[1])
    assert r_same == float(recon) and r_other != r_same and r_a == r_b and s_a == model.last_drop_seed


def test_moses_sample_greedy_matches_teacher_forced_logits(golden_dir):
    """VAE.sample() drives the GRU kernels one token at a time (T=1 launches, state carried through h0): at near-zero temperature every
    generated token must be the arg-max of the oracle's teacher-forced logits for the same latent and prefix."""
    g, model, params = _moses_setup(golden_dir, torch.float32)
    z = torch.from_numpy(g["z"].astype(np.float32)).to(dev)
    torch.manual_seed(0)
    toks, z_out = model.sample(6, max_len=12, z=z, temp=1e-4, return_tokens=True)
    assert len(toks) == 6 and torch.equal(z_out.cpu(), z.cpu())
    torch.manual_seed(0)
    strings, _ = model.sample(6, max_len=12, z=z, temp=1e-4)
    assert all(isinstance(s, str) for s in strings)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    checked = 0
    for b, tk in enumerate(toks):
        ids = tk.numpy().astype(np.int64)
        assert ids[0] == model.bos
        if len(ids) < 2:
            continue
        ref = O.moses_forward(p64, [ids], np.zeros((1, 160)), int(g["pad"]), want_grads=False, z_override=g["z"][b:b + 1])
        logits = ref["y"][0]                                  # [T, V]: position i predicts token i+1
        top2 = np.sort(logits, -1)[:, -2:]
        for i in range(len(ids) - 1):
            if top2[i, 1] - top2[i, 0] > 1e-3:                # skip numerical ties
                assert int(logits[i].argmax()) == int(ids[i + 1]), (b, i)
                checked += 1
    assert checked >= 20


# ---------------------------------------------------------------------------------------------- models2d.VAE (models2d.py)
@pytest.mark.parametrize("dtype,tl,tg", [(torch.float32, 1e-5, 5e-4), (torch.bfloat16, 2e-3, 6e-2)])
def test_g7_models2d_conv_relu_gru_vae(golden_dir, dtype, tl, tg):
    """The conv(ReLU) + GRU(2 -> 501, 3 layers)-over-the-one-hot-block variant against the fixture recorded from the imported reference
    models2d.py (train mode with its randn draw injected; eval mode z = mu) and against the oracle: loss, mu / logvar, recon, every gradient.
    The hidden size runs padded 501 -> 512 inside the kernels; gradients come back in the reference's [1503, 501] layout."""
    from molecular_vae_amd import models2d as M2
    g = np.load(os.path.join(golden_dir, "g7_models2d.npz"))
    params = ip.init_params(ip.models2d_shapes(), 404, 2.0, np.float32)
    model = M2.VAE(dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(dev).train()
    x = torch.nn.functional.one_hot(torch.from_numpy(g["idx"]), 35).float().to(dev)
    eps = torch.from_numpy(g["eps"].astype(np.float32)).to(dev)
    recon, mu, logvar = model(x, eps)
    loss = mv.bce_kl_loss(recon, x, mu, logvar, 120)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(g["loss"])) < tl * abs(float(g["loss"]))
    assert rel(mu.detach().cpu().numpy(), g["mu"]) < 1e-5 and rel(logvar.detach().cpu().numpy(), g["logvar"]) < 1e-5      # the encoder is f32 in both modes
    assert rel(recon.detach().cpu().numpy(), g["recon"]) < 20 * tl
    assert torch.allclose(recon.sum(-1), torch.ones(3, 120, device=dev), atol=1e-5)
    ref = O.models2d_loss_and_grads({k: v.astype(np.float64) for k, v in params.items()}, O.one_hot(g["idx"], 35), g["eps"], 120)
    bad = {}
    for k, p_ in model.named_parameters():
        gr = p_.grad.double().cpu().numpy()
        e = rel(gr, ref["grads"][k])
        if e > tg:
            bad[k] = e
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["gnorm." + k])) < 2 * tg * float(g["gnorm." + k]) + 1e-12, k     # reference fixture
    assert not bad, bad
    model.eval()
    with torch.no_grad():
        r2, mu2, lv2 = model(x)
        l2 = mv.bce_kl_loss(r2, x, mu2, lv2, 120)
        rz = model.decode(mu2)
    assert abs(float(l2) - float(g["eval_loss"])) < tl * abs(float(g["eval_loss"]))
    assert rel(r2.cpu().numpy()[:, ::17, :], g["eval_recon_rows"]) < 20 * tl and torch.equal(rz, r2)
    m_e, lv_e = model.encode(x)
    assert torch.equal(m_e, mu2) and torch.equal(lv_e, lv2)


def test_models2d_full_batch_training_steps():
    """B = 1024 (BASELINE's batch) through FusedAdam: finite, deterministic run to run, loss decreases over a few steps."""
    from molecular_vae_amd import models2d as M2
    out = []
    for rep in range(2):
        torch.manual_seed(42)
        model = M2.VAE().to(dev)
        opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
        loss_fn = mv.make_loss_function(120)
        gen = torch.Generator().manual_seed(1)
        idx = torch.randint(0, 35, (1024, 120), generator=gen).to(dev)
        x = torch.nn.functional.one_hot(idx, 35).float()
        eps = torch.randn(1024, 2, generator=gen).to(dev)
        ls = []
        for _ in range(4):
            opt.zero_grad(set_to_none=True)
            recon, mu, lv = model(x, eps)
            loss = loss_fn(recon, x, mu, lv)
            loss.backward()
            opt.step()
            ls.append(float(loss.detach()))
        out.append(ls)
    assert out[0] == out[1] and all(np.isfinite(out[0])) and out[0][-1] < out[0][0]
