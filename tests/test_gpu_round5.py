"""GPU parity tests added in round 5 (-m gpu).

* the reparameterisation draw of models.py:92 / mosesvae.py:159 / models2d.py:34 as a LIBRARY op (mvae_lambda_fwd / mvae_moses_latent_fwd with
  eps == NULL, mvae_normal_fill): equal to the host restatement of the counter hash, reproducible from (seed, counter), statistically
  N(0, scale^2), and -- the parity statement -- a step that draws on the device equals the same step with the restated noise INJECTED;
* the "cpu" source keeps the reference's CPU generator stream through a pinned ring.
Checker = oracle/ (numpy) and the host restatements in molecular_vae_amd.ops."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import gpu_helpers as gh
    from gpu_helpers import O, ip, mv, rel
    from molecular_vae_amd import ops, _lib as LL
    dev = torch.device("cuda", 0)


# ---------------------------------------------------------------------------------------------- device-side reparameterisation noise
def test_normal_fill_equals_the_host_restatement_and_is_normal():
    """mvae_normal_fill: every element equals scale * n(seed, offset + i) of the numpy restatement (hash words bit-exact on the host test; the
    float transform to 2e-6 absolute in units of sigma), across the 2^31 counter fold; moments and a Kolmogorov-Smirnov test against
    N(0, 1e-4) (= models.py:92's 1e-2 * randn) at the headline size B x o = 1024 x 292."""
    from scipy import stats
    for seed, off, n in ((1234, 0, 1024 * 292), (7, (1 << 31) - 1000, 4096), (0xFFFFFFFF, (1 << 45) + 3, 4096)):
        out = ops.normal_fill(torch.empty(n, device=dev), 1e-2, seed, off).cpu().numpy().astype(np.float64)
        want, _, _ = ops.normal_draw(seed, off, n, scale=1e-2)
        assert np.abs(out - want).max() < 2e-6 * 1e-2 * 6, (seed, off, np.abs(out - want).max())
    x = ops.normal_fill(torch.empty(1024 * 292, device=dev), 1e-2, 1234, 0).cpu().numpy().astype(np.float64)
    n = x.size
    assert abs(x.mean()) < 4 * 1e-2 / np.sqrt(n) and abs(x.std() / 1e-2 - 1) < 6e-3
    assert stats.kstest(x / 1e-2, "norm").pvalue > 1e-3
    a = ops.normal_fill(torch.empty(1000, device=dev), 1.0, 5, 100)
    b = ops.normal_fill(torch.empty(2000, device=dev), 1.0, 5, 0)
    assert torch.equal(a, b[100:1100])                     # a draw depends on (seed, counter) only, not on the launch shape


def test_lambda_draws_its_noise_inside_the_launch():
    """Lambda(noise="device") (the default): z = mu + exp(log_v / 2) * scale * n(seed, counter) with the restated normals; the same seed
    reproduces z bit for bit, the counter advances by B * o per forward, gradients use the drawn block (formulas in float64)."""
    torch.manual_seed(3)
    lam = mv.Lambda(i=435, o=292, scale=1e-2).to(dev)
    assert lam.noise == "device"
    rs = np.random.RandomState(0)
    x = rs.standard_normal((9, 435)); gz = rs.standard_normal((9, 292))
    tx = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    lam.seed_noise(77)
    z, mu, lv = lam(tx)
    assert lam.noise_stream.state() == dict(seed=77, counter=9 * 292)
    eps, _, _ = ops.normal_draw(77, 0, 9 * 292, scale=1e-2)
    eps = eps.reshape(9, 292)
    Wm, bm = lam.z_mean.weight.detach().double().cpu().numpy(), lam.z_mean.bias.detach().double().cpu().numpy()
    Wv, bv = lam.z_log_var.weight.detach().double().cpu().numpy(), lam.z_log_var.bias.detach().double().cpu().numpy()
    rmu, rlv = x @ Wm.T + bm, x @ Wv.T + bv
    assert rel(z.detach().cpu().numpy(), rmu + np.exp(rlv / 2) * eps) < 1e-5
    assert np.abs((z - mu).detach().cpu().numpy() / np.exp(rlv / 2) - eps).max() < 2e-6      # the noise itself, not hidden under mu
    (z * torch.tensor(gz, device=dev).float()).sum().backward()
    dlv = gz * eps * 0.5 * np.exp(rlv / 2)
    assert rel(lam.z_log_var.weight.grad.cpu().numpy(), dlv.T @ x) < 1e-5 and rel(lam.z_mean.weight.grad.cpu().numpy(), gz.T @ x) < 1e-5
    z2, _, _ = lam(tx.detach())                                    # next block of the stream: different noise
    assert not torch.equal(z2, z.detach())
    eps2, _, _ = ops.normal_draw(77, 9 * 292, 9 * 292, scale=1e-2)
    assert rel(z2.detach().cpu().numpy(), rmu + np.exp(rlv / 2) * eps2.reshape(9, 292)) < 1e-5
    lam.seed_noise(77)
    z3, _, _ = lam(tx.detach())
    assert torch.equal(z3, z.detach())                             # bit-reproducible from (seed, counter)
    # the "cpu" source: scale * randn on the CPU default generator (models.py:92), through the pinned ring -- more draws than ring slots
    lam.noise = "cpu"
    torch.manual_seed(123)
    zs = [lam(tx.detach())[0].detach().cpu() for _ in range(mv.Lambda.EPS_RING + 2)]
    torch.manual_seed(123)
    for zc in zs:
        want = 1e-2 * torch.randn(9, 292)
        assert torch.allclose(zc, torch.tensor(rmu + np.exp(rlv / 2) * want.double().numpy()).float(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_molecular_vae_step_with_device_noise_equals_the_step_with_that_noise_injected(dtype):
    """MolecularVAE (product default, noise drawn inside mvae_lambda_fwd) against the SAME model fed the host restatement of that draw as
    `eps`: loss, mu, logvar, recon and every gradient agree to fp32 rounding of eps (the injected path is the one the fixtures pin), for two
    consecutive steps (the second uses the next counter block)."""
    torch.manual_seed(11)
    model = mv.MolecularVAE(dtype=dtype).to(dev)
    assert model.encoder.lmbd.noise == "device"
    loss_fn = mv.make_loss_function(120)
    B = 16
    idx = torch.randint(0, 35, (B, 120), generator=torch.Generator().manual_seed(5)).to(dev)
    ohe = torch.nn.functional.one_hot(idx, 35).float()
    model.encoder.lmbd.seed_noise(2024)
    for step in range(2):
        outs = []
        for inject in (False, True):
            model.zero_grad(set_to_none=True)
            if inject:
                e, _, _ = ops.normal_draw(2024, step * B * 292, B * 292, scale=1e-2)
                recon, mu, lv = model(idx, eps=torch.tensor(e.reshape(B, 292), dtype=torch.float32, device=dev))
            else:
                recon, mu, lv = model(idx)
            loss = loss_fn(recon, ohe, mu, lv)
            loss.backward()
            torch.cuda.synchronize()
            outs.append((float(loss), mu.detach().cpu().numpy(), recon.detach().cpu().numpy(),
                         {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}))
        (la, mua, ra, ga), (lb, mub, rb, gb) = outs
        tol = 1e-5 if dtype == torch.float32 else 2e-3       # bf16: an eps difference of 1 fp32 ulp can flip a bf16 rounding downstream
        assert abs(la - lb) < 1e-6 * abs(lb) + (0 if dtype == torch.float32 else 1e-5 * abs(lb))
        assert np.array_equal(mua, mub) and rel(ra, rb) < tol
        bad = {k: rel(ga[k], gb[k]) for k in ga if rel(ga[k], gb[k]) > 10 * tol}
        assert not bad, bad
    assert model.encoder.lmbd.noise_stream.counter == 2 * B * 292     # injected forwards consume nothing


def test_moses_and_models2d_draw_their_noise_in_the_library(golden_dir):
    """mosesvae.VAE (mosesvae.py:159 randn_like) and models2d.VAE (models2d.py:34): the default forward draws inside the latent launch; it
    equals the forward with the restated normals injected, and is reproducible from the seed."""
    from test_gpu_parity import _moses_setup
    g, model, params = _moses_setup(golden_dir, torch.float32)
    assert model.noise == "device"
    seqs = [torch.from_numpy(g[f"seq{b}"]) for b in range(6)]
    model.seed_noise(31)
    kl, recon, z, lv, _, y = model(seqs)
    model.zero_grad(set_to_none=True); (kl + recon).backward(); torch.cuda.synchronize()
    ga = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    e, _, _ = ops.normal_draw(31, 0, 6 * 160)
    kl2, recon2, z2, lv2, _, y2 = model(seqs, eps=torch.tensor(e.reshape(6, 160), dtype=torch.float32, device=dev))
    model.zero_grad(set_to_none=True); (kl2 + recon2).backward(); torch.cuda.synchronize()
    assert abs(float(kl) - float(kl2)) < 1e-6 * abs(float(kl2)) and abs(float(recon) - float(recon2)) < 1e-5 * abs(float(recon2))
    assert rel(z.detach().cpu().numpy(), z2.detach().cpu().numpy()) < 1e-5
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert rel(ga[k], p.grad.cpu().numpy()) < 1e-4, k
    model.seed_noise(31)
    z3 = model.forward_encoder(seqs)[0]
    assert torch.equal(z3.detach(), z.detach())
    assert float(model.sample_z_prior(64).abs().max()) == 0.0          # the reference as written (mosesvae.py:211)
    zp = model.sample_z_prior(64, normal=True)
    want, _, _ = ops.normal_draw(31, 6 * 160, 64 * 160)
    assert np.abs(zp.cpu().numpy().reshape(-1) - want).max() < 1e-5
    model.noise = "torch"
    torch.manual_seed(9); za = model.forward_encoder(seqs)[0]
    torch.manual_seed(9); zb = model.forward_encoder(seqs)[0]
    assert torch.equal(za, zb) and not torch.equal(za.detach(), z.detach())
    # models2d
    from molecular_vae_amd import models2d
    torch.manual_seed(4)
    m2 = models2d.VAE().to(dev).train()
    x = torch.nn.functional.one_hot(torch.randint(0, models2d.VOCAB, (8, models2d.SEQ), generator=torch.Generator().manual_seed(1)), models2d.VOCAB).float().to(dev)
    m2.noise_stream.reseed(17)
    ra, mua, lva = m2(x)
    e2, _, _ = ops.normal_draw(17, 0, 16)
    rb, mub, lvb = m2(x, eps=torch.tensor(e2.reshape(8, 2), dtype=torch.float32, device=dev))
    assert torch.equal(mua, mub) and rel(ra.detach().cpu().numpy(), rb.detach().cpu().numpy()) < 2e-3


# ---------------------------------------------------------------------------------------------- persistent kernels against the oracle
@pytest.mark.parametrize("B", [128, 256])
def test_persistent_dataflow_passes_vs_the_numpy_oracle(B):
    """rnn_persist.hip / rnn_persist_bwd.hip at the shape they serve (LSTM 4 x 1024, bf16, b = 128; 256 = two passes) with B DISTINCT random
    rows, T = 6, against oracle/np_oracle.lstm_fwd / lstm_bwd (models.py:156,164): every layer's h, c, saved gates, dG through the weight /
    bias gradients it yields, and the input gradient -- the tolerance of the other bf16 LSTM cases (3e-2).  Both passes are REQUIRED to take
    the dataflow schedule (the helper asserts the launch counters)."""
    from test_gpu_parity import _lstm_case
    errs = _lstm_case(torch.bfloat16, 6, B, 1024, 4, 16, seed=7, persist=True)
    bad = {k: v for k, v in errs.items() if v > 3e-2}
    assert not bad, bad


# ---------------------------------------------------------------------------------------------- a launch that gives up is survivable
def _b128_step_setup(dtype=torch.bfloat16, B=128):
    torch.manual_seed(21)
    model = mv.MolecularVAE(dtype=dtype).to(dev)
    opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
    lf = mv.make_loss_function(120)
    idx = torch.randint(0, 35, (B, 120), generator=torch.Generator().manual_seed(3)).to(dev)
    ohe = torch.nn.functional.one_hot(idx, 35).float()
    eps = (1e-2 * torch.randn(B, 292, generator=torch.Generator().manual_seed(4))).to(dev)
    return model, opt, lf, idx, ohe, eps


@pytest.mark.parametrize("knob", ["MVAE_PERSIST_SPIN", "MVAE_ROWRES_SPIN"])
def test_training_step_whose_persistent_launch_gave_up_is_skipped_on_the_device(knob, monkeypatch):
    """b = 128 (the per-rank shape of configs[2]): a decoder dataflow pass (MVAE_PERSIST_SPIN=1) or a layer-concurrent encoder pass
    (MVAE_ROWRES_SPIN=1) that gives up stores a NaN into the optimiser's poison slot; mvae_clip_adam then leaves parameters and moments
    untouched (bit for bit) and counts the skip -- no exception, no host wait inside the step; the poison slot is clean again, the next step
    trains normally, and after three failures the spinning schedules are off and the step runs on the wavefront / layer-by-layer forms."""
    model, opt, lf, idx, ohe, eps = _b128_step_setup()
    mv.train_step(model, opt, lf, idx, ohe, eps=eps)
    torch.cuda.synchronize()
    f = opt._flat[0]
    assert float(opt.skipped_steps) == 0 and float(f["poison"]) == 0.0
    n0 = dict(ops.PERSIST_STATS)
    assert n0["launches"] >= 1 and n0["bwd_launches"] >= 1 and n0["rowres_pipe"] >= 2         # the schedules under test are the ones running
    p0, m0 = f["p"].clone(), f["m"].clone()
    monkeypatch.setenv(knob, "1")
    with pytest.warns(RuntimeWarning, match="gave up"):
        mv.train_step(model, opt, lf, idx, ohe, eps=eps)
        torch.cuda.synchronize()
        ops.persist_check(sync=True)
    assert torch.equal(f["p"], p0) and torch.equal(f["m"], m0)                                 # the update did not happen
    assert float(opt.skipped_steps) == 1 and float(f["poison"]) == 0.0 and not np.isfinite(float(opt.last_grad_norm))
    assert ops.PERSIST_STATS["failures"] >= 1 and ops.PERSIST_STATS["reruns"] == 0             # nothing waited, nothing was re-run
    monkeypatch.delenv(knob)
    if not ops.PERSIST_STATS["disabled"]:
        mv.train_step(model, opt, lf, idx, ohe, eps=eps)
        torch.cuda.synchronize()
        assert float(opt.skipped_steps) == 1 and not torch.equal(f["p"], p0) and np.isfinite(float(opt.last_grad_norm))
    # enough failures: the schedules switch themselves off and training goes on
    monkeypatch.setenv(knob, "1")
    with pytest.warns(RuntimeWarning):
        for _ in range(3):
            mv.train_step(model, opt, lf, idx, ohe, eps=eps)
            torch.cuda.synchronize()
            ops.persist_check(sync=True)
    assert ops.PERSIST_STATS["disabled"]
    before = dict(ops.PERSIST_STATS)
    p1 = f["p"].clone()
    mv.train_step(model, opt, lf, idx, ohe, eps=eps)
    torch.cuda.synchronize()
    assert ops.persist_check(sync=True) == 0 and not torch.equal(f["p"], p1)
    for k in ("launches", "bwd_launches", "rowres_pipe"):
        assert ops.PERSIST_STATS[k] == before[k], k


def test_evaluation_forward_re_runs_a_pass_that_gave_up(monkeypatch):
    """No optimiser in sight (no_grad forward): the call waits for the status record of its persistent launch and, when it gave up, runs the
    pass again without spins -- the caller gets the wavefront schedule's outputs and a warning."""
    model, opt, lf, idx, ohe, eps = _b128_step_setup()
    model.eval()
    with torch.no_grad():
        ops.PERSIST_DEFAULT = "0"
        try:
            want, mu0, _ = model(idx, eps=eps)
        finally:
            ops.PERSIST_DEFAULT = "1"
        n0 = ops.PERSIST_STATS["launches"]
        got, mu1, _ = model(idx, eps=eps)
        assert ops.PERSIST_STATS["launches"] == n0 + 1
        assert rel(got.cpu().numpy(), want.cpu().numpy()) < 2e-2 and rel(mu1.cpu().numpy(), mu0.cpu().numpy()) < 1e-5
        monkeypatch.setenv("MVAE_PERSIST_SPIN", "1")
        monkeypatch.setenv("MVAE_ROWRES_SPIN", "1")
        with pytest.warns(RuntimeWarning, match="gave up"):
            again, mu2, _ = model(idx, eps=eps)
        assert ops.PERSIST_STATS["reruns"] >= 1
        assert torch.equal(again, want) and torch.equal(mu2, mu0)


def test_no_status_is_read_where_no_spinning_kernel_ran():
    """the host used to re-derive which schedule the library took and read a status word the library might never have written
    (sequence length > 256, > 64 rows per table block, dropout, a dh_last ...).  The library now reports the status location itself
    (status_out): an encoder with i = 260 -- served by the wavefront / layer-by-layer forms -- runs forward and backward without any
    spurious failure report and equals the oracle."""
    dims = dict(i=260, o=16, c=12, emb=30, h_enc=72, n_enc=3, h_dec=32, n_dec=2)
    shapes = ip.molvae_shapes(dims["i"], dims["o"], dims["c"], dims["emb"], dims["h_enc"], dims["n_enc"], dims["h_dec"], dims["n_dec"])
    params = ip.init_params(shapes, 77, 2.0, np.float32)
    enc, dec = gh.build_modules(dims, params, torch.float32)
    B = 8
    idx = ip.seeded_indices(5, B, dims["i"], dims["c"]); eps = ip.seeded_eps(5, B, dims["o"], dtype=np.float64)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        out = gh.run_hip(enc, dec, idx, eps, dims["i"])
        assert ops.persist_check(sync=True) == 0
    ref = O.molvae_loss_and_grads({k: v.astype(np.float64) for k, v in params.items()}, idx, eps, max_len=dims["i"], num_lstm=3, num_gru=2)
    assert abs(out["loss"] - ref["loss"]) < 1e-5 * abs(ref["loss"])
    worst = max(gh.grad_report(out["grads"], ref["grads"]).values())
    assert worst < 1e-3, worst
    assert ops.PERSIST_STATS["failures"] == 0


# ---------------------------------------------------------------------------------------------- the encoder's parameter-gradient GEMMs in one launch
def test_tn_f32_multi_launch_vs_float64_and_the_single_launches():
    """mvae_gemm_tn_f32_multi: problems of different shapes (the encoder's heads / dense / LSTM(72) weight gradients at b = 128, a tiny one, one
    that accumulates, one in the 3 x bf16 form) in ONE launch against numpy float64 -- exact-f32 products: 2e-6 relative; x3: 5e-5 -- and
    against the same contractions launched one by one (bit-equal column sums are not required: the split over K differs)."""
    rs = np.random.RandomState(0)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    shapes = [(292, 512, 128, True, False, False), (512, 1344, 128, True, False, False), (288, 72, 15360 - 128, False, False, False),
              (288, 72, 15360, True, False, False), (288, 4, 15360, True, False, False), (35, 288, 700, False, True, False),
              (64, 2304, 2688, True, False, True), (5, 3, 9, False, False, False)]
    batch, want, outs = ops.TnF32Batch(dev), [], []
    for (M, N, K, cs, acc, x3) in shapes:
        lda, ldb = (M + 3) // 4 * 4 + 4, (N + 3) // 4 * 4 + 8
        A = np.zeros((K, lda)); A[:, :M] = rs.standard_normal((K, M))
        Bm = np.zeros((K, ldb)); Bm[:, :N] = rs.standard_normal((K, N))
        c0 = rs.standard_normal((M, N)) if acc else np.zeros((M, N))
        out, csum = t(c0), (torch.full((M,), 7.0, device=dev) if cs else None)
        batch.add(t(A), t(Bm), out, M, N, K, lda=lda, ldb=ldb, colsum_out=csum, accumulate=acc, x3=x3)
        want.append((c0 + A[:, :M].T @ Bm[:, :N], A[:, :M].sum(0), 5e-5 if x3 else 2e-6))
        outs.append((out, csum))
    batch.run()
    torch.cuda.synchronize()
    for (out, csum), (w, wcs, tol), sh in zip(outs, want, shapes):
        assert rel(out.cpu().numpy(), w) < tol, sh
        if csum is not None:
            assert rel(csum.cpu().numpy(), wcs) < tol, sh
    # more problems than one launch takes: split into several launches
    many = ops.TnF32Batch(dev)
    res = []
    for i in range(ops.TnF32Batch.MAX + 3):
        A, Bm = rs.standard_normal((40 + i, 8)), rs.standard_normal((40 + i, 12))
        o_ = torch.zeros(8, 12, device=dev)
        many.add(t(A), t(Bm), o_, 8, 12, 40 + i); res.append((o_, A.T @ Bm))
    many.run()
    for o_, w in res:
        assert rel(o_.cpu().numpy(), w) < 2e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-4)])
def test_encoder_parameter_gradients_batched_equal_the_one_by_one_launches(dtype, tol, monkeypatch):
    """MolEncoder backward with every parameter-gradient GEMM in ONE mvae_gemm_tn_f32_multi launch (the default) against MVAE_ENC_DW_BATCH=0 (one
    launch each, where autograd would put it): every encoder gradient agrees to fp32 summation order (bf16 mode: the conv gradients use 3 x bf16
    products in both forms); b = 128 and a ragged batch."""
    for B in (128, 37):
        torch.manual_seed(5)
        model = mv.MolecularVAE(dtype=dtype).to(dev)
        lf = mv.make_loss_function(120)
        idx = torch.randint(0, 35, (B, 120), generator=torch.Generator().manual_seed(B)).to(dev)
        ohe = torch.nn.functional.one_hot(idx, 35).float()
        eps = (1e-2 * torch.randn(B, 292, generator=torch.Generator().manual_seed(1))).to(dev)
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MVAE_ENC_DW_BATCH", mode)
            model.zero_grad(set_to_none=True)
            recon, mu, lv = model(idx, eps=eps)
            lf(recon, ohe, mu, lv).backward()
            torch.cuda.synchronize()
            got[mode] = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.encoder.named_parameters()}
        bad = {k: rel(got["1"][k], got["0"][k]) for k in got["1"] if rel(got["1"][k], got["0"][k]) > tol}
        assert not bad, (B, bad)


def test_f32_model_samples_with_a_vocabulary_above_32_tokens():
    """mvae_moses_sample_step keeps the decoder_fc head in LDS; in fp32 with H = 512 a vocabulary above 32 tokens needs more than the
    default 64 KB of dynamic LDS -- the launcher now opts in to the CU's 160 KB (V <= 64 by the kernel's own limit).  A 45-symbol fp32 model
    samples, reproducibly from its seed, and greedy decoding (temp -> 0) equals the arg-max of a teacher-forced forward."""
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    chars = [chr(ord("A") + i) for i in range(26)] + [chr(ord("a") + i) for i in range(15)]
    v = VC.OneHotVocab(chars)
    assert 32 < len(v) <= 64
    torch.manual_seed(8)
    model = MV.VAE(v, dtype=torch.float32).to(dev).eval()
    a, za = model.sample(16, max_len=12, seed=5, return_tokens=True)
    b, zb = model.sample(16, max_len=12, seed=5, return_tokens=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and len(a) == 16
    c, _ = model.sample(16, max_len=12, seed=6, return_tokens=True)
    assert any(not torch.equal(x, y) for x, y in zip(a, c))


def test_two_ranks_skip_together_when_one_ranks_persistent_launch_gives_up(tmp_path):
    """ADVICE r04 #1 / VERDICT r04 missing #3 in the data-parallel setting: two ranks (fresh child processes sharing this GPU, gloo), b = 128 each
    in bf16 -- the shape of the persistent decoder passes.  At step 1 ONLY rank 1's persistent launches give up (poll budget 1): its poison slot
    becomes NaN, the slot travels with the last gradient bucket, BOTH ranks' optimiser kernels skip the update (no rank raises, no rank is left
    waiting in a collective), the parameters stay identical on both ranks through every step, step 1 leaves them untouched, step 2 trains."""
    import json
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = os.path.join(ROOT, "tests", "dp_equiv.py")
    out = os.path.join(str(tmp_path), "poison.json")
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MVAE_PERSIST_SPIN"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script, "--out", out, "--b", "128", "--steps", "3", "--dtype", "bf16", "--poison-rank", "1",
                        "--poison-step", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.load(open(out))
    r0, r1 = res["ranks"]
    assert r0["persist"] > 0 and r1["persist"] > 0                       # both ranks ran the persistent schedules
    assert r1["failures"] >= 1                                           # rank 1's launches gave up (forced) ...
    assert r0["skipped"] == r1["skipped"] >= 1.0                         # ... and BOTH ranks skipped the same updates
    assert r0["psum"] == r1["psum"]                                      # the replicas never diverge
    assert r0["psum"][1] == r0["psum"][0]                                # step 1 left the parameters untouched on both
    if r0["failures"] == 0 and r0["skipped"] == 1.0:                     # (two processes share this GPU: an unforced give-up on rank 0 would be
        assert r0["psum"][2] != r0["psum"][1]                            #  survived the same way and only void this last check) step 2 trains

