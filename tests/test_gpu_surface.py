"""GPU tests (pytest -m gpu) of the surface around the training step: forward-only evaluation (train.py:120-153), accuracy
(train.py:109-113), optimiser hand-over to / from torch.optim.Adam (train.py:81,173), the stand-alone Lambda head (models.py:80-94)
and the two-rank data-parallel equivalence of the HIP path (train_distributed.py:72 replaced by per-process DP)."""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gpu_helpers as gh            # noqa: E402,F401
from gpu_helpers import mv          # noqa: E402
from molecular_vae_amd import _lib as L   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_only_pass_equals_training_forward(dtype):
    """Under no_grad the decoder hands the step kernels no save buffers (gates / cell states are not written); outputs must be
    bit-identical to the training forward, and backward through such a pass must be refused, not silently wrong."""
    torch.manual_seed(42)
    model = mv.MolecularVAE(dtype=dtype).to(dev)
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, 35, (48, 120), generator=g).to(dev)
    eps = (1e-2 * torch.randn(48, 292, generator=g)).to(dev)
    recon_t, mu_t, lv_t = model(idx, eps)
    ws = model.decoder._ws
    gates = [b for k, b in ws.bufs.items() if k[0] == "gates0"][0]
    gates.fill_(7.0)
    with torch.no_grad():
        recon_e, mu_e, lv_e = model(idx, eps)
    assert torch.equal(recon_t, recon_e) and torch.equal(mu_t, mu_e) and torch.equal(lv_t, lv_e)
    assert float(gates.float().min()) == 7.0 and float(gates.float().max()) == 7.0      # the forward-only pass did not touch them
    # decoding from a latent (train_sample.py:32 model.decoder(sampler)) is the same forward-only path
    with torch.no_grad():
        z = torch.randn(5, 292, device=dev)
        p = model.decoder(z)
    assert p.shape == (5, 120, 35) and torch.allclose(p.sum(-1), torch.ones(5, 120, device=dev), atol=1e-5)


def test_evaluate_and_exact_match_accuracy_vs_cpu():
    """train.py:106-113 / 120-153: exact-match accuracy (row-wise torch.equal of the arg-max) and the forward-only evaluation loop."""
    g = torch.Generator().manual_seed(11)
    data = torch.randint(0, 35, (64, 120), generator=g)
    recon = torch.rand(64, 120, 35, generator=g) * 0.5
    right_rows = [0, 3, 17, 40, 63]
    for b in range(64):
        recon[b].scatter_(1, data[b].view(-1, 1), 1.0)           # arg-max == data everywhere ...
        if b not in right_rows:
            t = int(torch.randint(0, 120, (1,), generator=g))
            recon[b, t, (int(data[b, t]) + 1) % 35] = 2.0        # ... except one position of the wrong rows
    want = sum(int(torch.equal(recon[b].argmax(1), data[b])) for b in range(64)) / 64.0     # the reference's Python loop
    assert want == len(right_rows) / 64.0
    got = float(mv.exact_match_accuracy(recon.to(dev), data.to(dev)))
    assert got == want
    # evaluate(): mean of per-batch losses, accuracy over all sequences, model left in its previous mode
    torch.manual_seed(42)
    model = mv.MolecularVAE().to(dev)
    model.encoder.lmbd.draw_eps = lambda b, o, d: torch.zeros(b, o, device=d)
    loss_fn = mv.make_loss_function(120)
    batches = []
    for _ in range(3):
        idx = torch.randint(0, 35, (32, 120), generator=g).to(dev)
        batches.append((idx, torch.nn.functional.one_hot(idx, 35).float()))
    model.train()
    val, acc = mv.evaluate(model, loss_fn, batches)
    assert model.training
    per, hits = [], 0
    for idx, ohe in batches:
        recon, mu, lv = model(idx)
        per.append(float(loss_fn(recon, ohe, mu, lv)))
        r, d = recon.detach().cpu(), idx.cpu()
        hits += sum(int(torch.equal(r[b].argmax(1), d[b])) for b in range(32))
    assert abs(val - sum(per) / 3) < 1e-6 * abs(val) and acc == hits / 96.0
    # the optimiser keeps stepping after a forward-only pass sat between forward and backward of different batches
    opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
    l0 = float(mv.train_step(model, opt, loss_fn, *batches[0]))
    mv.evaluate(model, loss_fn, batches[:1])
    l1 = float(mv.train_step(model, opt, loss_fn, *batches[0]))
    assert np.isfinite(l0) and np.isfinite(l1) and l1 != l0


def test_fused_adam_hands_over_to_torch_adam_and_back():
    """Two steps with FusedAdam, state_dict() into torch.optim.Adam (the reference's optimiser, train.py:81), one more step each on the
    same gradients: parameters must agree; then torch's state back into a fresh FusedAdam, one more step: same again."""
    torch.manual_seed(7)
    m1 = mv.MolecularVAE(i=24, o=16, c=12, dtype=torch.float32).to(dev)
    torch.manual_seed(7)
    m2 = mv.MolecularVAE(i=24, o=16, c=12, dtype=torch.float32).to(dev)
    fa = mv.FusedAdam(m1.parameters(), lr=8e-4, max_grad_norm=3.0)
    loss_fn = mv.make_loss_function(24)
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 12, (6, 24), generator=g).to(dev)
    ohe = torch.nn.functional.one_hot(idx, 12).float()
    eps = (1e-2 * torch.randn(6, 16, generator=g)).to(dev)
    for _ in range(2):
        mv.train_step(m1, fa, loss_fn, idx, ohe, eps=eps)
    m2.load_state_dict(m1.state_dict())
    ta = torch.optim.Adam(m2.parameters(), lr=1.0)
    ta.load_state_dict(copy.deepcopy(fa.state_dict()))     # as a checkpoint file would: torch's loader keeps references to what it is given

    def grads(m):
        m.zero_grad(set_to_none=True)
        recon, mu, lv = m(idx, eps)
        loss_fn(recon, ohe, mu, lv).backward()
        torch.cuda.synchronize()

    grads(m2)
    torch.nn.utils.clip_grad_norm_(m2.parameters(), 3.0)
    ta.step()
    mv.train_step(m1, fa, loss_fn, idx, ohe, eps=eps)
    for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-7), k
    torch.manual_seed(7)
    m3 = mv.MolecularVAE(i=24, o=16, c=12, dtype=torch.float32).to(dev)
    fb = mv.FusedAdam(m3.parameters(), lr=1.0, max_grad_norm=3.0)
    mv.load_checkpoint({"model_state_dict": {"module." + k: v for k, v in m2.state_dict().items()},
                        "optimizer_state_dict": copy.deepcopy(ta.state_dict())}, m3, fb)
    grads(m2)
    torch.nn.utils.clip_grad_norm_(m2.parameters(), 3.0)
    ta.step()
    mv.train_step(m3, fb, loss_fn, idx, ohe, eps=eps)
    for (k, a), (_, b) in zip(m3.named_parameters(), m2.named_parameters()):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-7), k


def test_standalone_lambda_head_vs_numpy():
    """models.py:80-94 with its default i=435 (not a multiple of 4: exercises the padded GEMM path), injected eps; forward and every
    gradient against the formulas in float64."""
    torch.manual_seed(3)
    lam = mv.Lambda(i=435, o=292, scale=1e-2).to(dev)
    rs = np.random.RandomState(0)
    x = rs.standard_normal((9, 435)); eps = 1e-2 * rs.standard_normal((9, 292))
    gz, gmu, glv = rs.standard_normal((9, 292)), rs.standard_normal((9, 292)), rs.standard_normal((9, 292))
    tx = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
    z, mu, lv = lam(tx, torch.tensor(eps, dtype=torch.float32, device=dev))
    assert lam.mu is mu and lam.log_v is lv
    (z * torch.tensor(gz, device=dev).float() + mu * torch.tensor(gmu, device=dev).float() + lv * torch.tensor(glv, device=dev).float()).sum().backward()
    Wm, bm = lam.z_mean.weight.detach().double().cpu().numpy(), lam.z_mean.bias.detach().double().cpu().numpy()
    Wv, bv = lam.z_log_var.weight.detach().double().cpu().numpy(), lam.z_log_var.bias.detach().double().cpu().numpy()
    rmu, rlv = x @ Wm.T + bm, x @ Wv.T + bv
    rz = rmu + np.exp(rlv / 2) * eps
    dmu = gz + gmu
    dlv = gz * eps * 0.5 * np.exp(rlv / 2) + glv
    rel = gh.rel
    assert rel(mu.detach().cpu().numpy(), rmu) < 1e-5 and rel(lv.detach().cpu().numpy(), rlv) < 1e-5 and rel(z.detach().cpu().numpy(), rz) < 1e-5
    assert rel(lam.z_mean.weight.grad.cpu().numpy(), dmu.T @ x) < 1e-5 and rel(lam.z_mean.bias.grad.cpu().numpy(), dmu.sum(0)) < 1e-5
    assert rel(lam.z_log_var.weight.grad.cpu().numpy(), dlv.T @ x) < 1e-5 and rel(lam.z_log_var.bias.grad.cpu().numpy(), dlv.sum(0)) < 1e-5
    assert rel(tx.grad.cpu().numpy(), dmu @ Wm + dlv @ Wv) < 1e-5
    # noise="cpu": without eps the draw is scale * randn on the CPU default generator (models.py:92): reproducible under manual_seed
    # (the default source, "device", is covered by tests/test_gpu_round5.py)
    lam.noise = "cpu"
    torch.manual_seed(123); z1, _, _ = lam(tx.detach())
    torch.manual_seed(123); want = 1e-2 * torch.randn(9, 292)
    assert torch.allclose(z1.cpu(), torch.tensor(rmu + np.exp(rlv / 2) * want.double().numpy()).float(), rtol=1e-4, atol=1e-6)


def _run(cmd, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}"
    return r


def test_two_rank_moses_data_parallel_uses_the_global_token_mean(tmp_path):
    """mosesvae.VAE under DP: the reconstruction loss is a mean over non-pad targets (mosesvae.py:193-197); ranks hold different token
    counts, so each rank rescales its local mean by cnt_r * world / sum cnt (one 4-byte all-reduce).  Two ranks on interleaved shards must
    then follow the single-process run on the global batch (loss = KL + CE, parameters, clipped-gradient norm) through 3 Adam steps."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = os.path.join(ROOT, "tests", "dp_equiv.py")
    a, b = os.path.join(str(tmp_path), "m1.json"), os.path.join(str(tmp_path), "m2.json")
    _run([sys.executable, script, "--model", "moses", "--out", a, "--b", "24", "--steps", "3"])
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
          "--master-port", str(port), script, "--model", "moses", "--out", b, "--b", "24", "--steps", "3"])
    ra, rb = json.load(open(a)), json.load(open(b))
    assert rb["world"] == 2
    for k in ("loss", "psum", "gnorm"):
        for x, y in zip(ra[k], rb[k]):
            assert abs(x - y) <= 2e-5 * abs(x), (k, ra[k], rb[k])


@pytest.mark.parametrize("dtype,tol", [("f32", dict(loss=2e-6, psum=2e-6, gnorm=2e-6)), ("bf16", dict(loss=5e-4, psum=1e-5, gnorm=3e-2))])
def test_two_rank_data_parallel_equals_single_process(tmp_path, dtype, tol):
    """The PRODUCT path under data parallelism: 4 optimiser steps of the full-size model, (a) one process, global batch 2b;
    (b) two ranks (fresh child processes sharing this GPU, gloo), b each, GradSync with the early all-reduce started from backward.
    Losses, parameter sums and pre-clip gradient norms must agree to fp32 re-association noise (<= 2e-6 relative) in the exact-f32 mode;
    in the bf16 mode (the bench configuration) the two layouts pick different tiles, whose fp32 summation orders round to different bf16
    values now and then: loss within 5e-4, gradient norm within 3 %."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = os.path.join(ROOT, "tests", "dp_equiv.py")
    a, b = os.path.join(str(tmp_path), "dp1.json"), os.path.join(str(tmp_path), "dp2.json")
    _run([sys.executable, script, "--out", a, "--b", "32", "--steps", "4", "--dtype", dtype])
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
          "--master-port", str(port), script, "--out", b, "--b", "32", "--steps", "4", "--dtype", dtype])
    ra, rb = json.load(open(a)), json.load(open(b))
    # the two-rank run really took the early-all-reduce path: one range per decoder LSTM layer and step, in reverse layer order (the head
    # travels with layer 3)
    assert rb["world"] == 2 and rb["early_ranges"] == 4 * 4
    for k in ("loss", "psum", "gnorm"):
        for x, y in zip(ra[k], rb[k]):
            assert abs(x - y) <= tol[k] * abs(x), (k, ra[k], rb[k])


def test_two_rank_sharded_optimizer_equals_the_all_reduce_form_bit_for_bit(tmp_path):
    """FusedAdam(shard_optimizer=True): reduce-scatter of the flat gradient, clip + Adam on this rank's 1/world slice, all-gather of the
    parameters -- against the all-reduce form on the same two ranks (gloo, fresh child processes sharing this GPU): losses, gradient norms,
    parameter sums AND the gathered Adam moments must be IDENTICAL (world 2: the same two addends per element; the norm is formed from the
    same 64K-element partial sums in the same order)."""
    import socket
    script = os.path.join(ROOT, "tests", "dp_equiv.py")
    res = []
    for extra in ([], ["--shard"]):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = os.path.join(str(tmp_path), "sh%d.json" % len(res))
        _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), script, "--out", out, "--b", "16", "--steps", "3", "--dtype", "bf16"] + extra)
        res.append(json.load(open(out)))
    a, b = res
    assert a["world"] == b["world"] == 2 and a["early_ranges"] == 4 * 3 and b["early_ranges"] == 0
    for k in ("loss", "psum", "gnorm", "pcheck"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["mcheck"] == b["mcheck"]


def test_two_rank_bf16_gradient_all_reduce_tracks_the_fp32_one(tmp_path):
    """GradSync(compress="bf16"): the gradient crosses the links as bfloat16.  Against fp32 on the wire (same ranks, same data, bf16 model):
    the loss trajectory stays within 1e-3, the pre-clip norm within 1 % -- the option's measured cost."""
    import socket
    script = os.path.join(ROOT, "tests", "dp_equiv.py")
    res = []
    for extra in ([], ["--compress", "bf16"]):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = os.path.join(str(tmp_path), "cp%d.json" % len(res))
        _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), script, "--out", out, "--b", "16", "--steps", "4", "--dtype", "bf16"] + extra)
        res.append(json.load(open(out)))
    a, b = res
    for x, y in zip(a["loss"], b["loss"]):
        assert abs(x - y) < 1e-3 * abs(x), (a["loss"], b["loss"])
    for x, y in zip(a["gnorm"], b["gnorm"]):
        assert abs(x - y) < 1e-2 * abs(x), (a["gnorm"], b["gnorm"])
    assert a["loss"] != b["loss"] or a["gnorm"] != b["gnorm"]           # it really took the rounded path
