"""GPU parity tests added in round 4 (-m gpu): the remaining pieces of the reference's module surface -- mosesvae.VAE.forward_encoder as an
encoder-only pass, forward_decoder(x, z) on a caller-supplied latent, the encoder / decoder optimiser split and the ``model.apply(init_weights)``
hook of moses_train_distrib_logp.py:48-51,262-268.  Checker = oracle/ (numpy) and the reference-generated fixtures in tests/golden/."""
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
    from test_gpu_parity import _moses_base, _moses_setup
    dev = torch.device("cuda", 0)

ENC = ("x_emb.", "encoder_rnn.", "q_mu.", "q_logvar.")
DEC = ("x_emb.", "decoder_rnn.", "decoder_lat.", "decoder_fc.")


def _g3(golden_dir, dtype):
    g, model, params = _moses_setup(golden_dir, dtype)
    seqs = [torch.from_numpy(g[f"seq{b}"]) for b in range(6)]
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    return g, model, p64, seqs


# ---------------------------------------------------------------------------------------------- forward_decoder(x, z)
@pytest.mark.parametrize("dtype,tl,tg", [(torch.float32, 2e-5, 5e-4), (torch.bfloat16, 5e-3, 6e-2)])
def test_forward_decoder_on_the_fixture_latent(golden_dir, dtype, tl, tg):
    """mosesvae.py:166-199 with the z the REFERENCE recorded in g3: (recon_loss, x_padded, y) equal the fixture's, the decoder-half
    parameter gradients and d recon / d z equal the oracle's, and no encoder parameter receives a gradient."""
    g, model, p64, seqs = _g3(golden_dir, dtype)
    z = torch.from_numpy(g["z"].astype(np.float32)).to(dev).requires_grad_(True)
    recon, x, y = model.forward_decoder(seqs, z)
    assert (x.cpu().numpy() == g["x"]).all()
    assert abs(float(recon) - float(g["recon"])) < tl * abs(float(g["recon"]))
    assert rel(y.detach().cpu().numpy(), g["y"]) < max(tl, 1e-5)
    model.zero_grad(set_to_none=True)
    recon.backward()
    torch.cuda.synchronize()
    ref = O.moses_forward(p64, [g[f"seq{b}"] for b in range(6)], g["eps"], int(g["pad"]))
    rg = ref["grads_for"](0.0)
    assert rel(z.grad.cpu().numpy(), rg["_dz"]) < tg
    bad = {}
    for k, p_ in model.named_parameters():
        if k.startswith(DEC):
            want = rg["_x_emb_dec"] if k == "x_emb.weight" else rg[k]
            e = rel(p_.grad.cpu().numpy(), want)
            if e > tg:
                bad[k] = e
        else:
            assert p_.grad is None, k
    assert not bad, bad


def test_forward_decoder_on_an_arbitrary_latent_vs_oracle():
    """A latent the encoder never produced (interpolation / the aggressive loop of moses_train_distrib_logp.py:289-318): B = 40 ragged
    sequences, bf16 kernels on the LDS-direct path, train-mode dropout with an injected mask; logits, loss, decoder gradients and dz vs the
    oracle run with the same z."""
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    v = VC.OneHotVocab([chr(ord("a") + i) for i in range(26)])
    V = len(v)
    params = ip.init_params(ip.moses_shapes(V), 21, 1.0, np.float32)
    model = MV.VAE(v, dtype=torch.bfloat16)
    model.load_state_dict({k: torch.from_numpy(params[_moses_base(k)]) for k in model.state_dict()})
    model = model.to(dev).train()
    rs = np.random.RandomState(9)
    B = 40
    lens = sorted(rs.randint(10, 58, size=B).tolist(), reverse=True)
    seqs = [np.concatenate([[v.bos], rs.randint(0, 26, size=n), [v.eos]]).astype(np.int64) for n in lens]
    T = max(len(s) for s in seqs)
    zin = rs.standard_normal((B, 160)).astype(np.float32) * 0.7
    mask = (rs.uniform(size=(2, T, B, 512)) >= 0.2).astype(np.uint8)
    z = torch.from_numpy(zin).to(dev).requires_grad_(True)
    recon, x, y = model.forward_decoder([torch.from_numpy(s) for s in seqs], z, drop_mask=torch.from_numpy(mask))
    recon.backward()
    ref = O.moses_forward({k: a.astype(np.float64) for k, a in params.items()}, seqs, np.zeros((B, 160)), v.pad, z_override=zin.astype(np.float64),
                          drop_masks=[mask[0], mask[1]], drop_p=0.2)
    assert abs(float(recon) - ref["recon"]) < 5e-3 * abs(ref["recon"])
    assert rel(y.detach().cpu().numpy(), ref["y"]) < 2e-2
    rg = ref["grads_for"](0.0)
    assert rel(z.grad.cpu().numpy(), rg["_dz"]) < 8e-2
    bad = {}
    for k, p_ in model.named_parameters():
        if k.startswith(DEC):
            want = rg["_x_emb_dec"] if k == "x_emb.weight" else rg[k]
            e = rel(p_.grad.cpu().numpy(), want)
            if e > 8e-2:
                bad[k] = e
    assert not bad, bad
    with pytest.raises(ValueError):
        model.forward_decoder([torch.from_numpy(s) for s in seqs], z[:, :100])


# ---------------------------------------------------------------------------------------------- forward_encoder alone
@pytest.mark.parametrize("dtype,tl,tg", [(torch.float32, 2e-5, 5e-4), (torch.bfloat16, 5e-3, 6e-2)])
def test_forward_encoder_is_encoder_only(golden_dir, dtype, tl, tg):
    """mosesvae.py:142-164: (z, kl, logvar) equal g3's; NO decoder kernel is launched (launch groups are tagged: ops.PROFILE records
    `moses_enc_fwd` but not `moses_dec_fwd`, and the decoder half's saved-state generation does not move); the gradient of kl equals the
    oracle's kl-only gradient, decoder parameters receive none."""
    g, model, p64, seqs = _g3(golden_dir, dtype)
    eps = torch.from_numpy(g["eps"].astype(np.float32)).to(dev)
    ops.PROFILE = {}
    try:
        z, kl, logvar = model.forward_encoder(seqs, eps)
        tags_enc = set(ops.PROFILE)
        ops.PROFILE = {}
        model(seqs, eps)
        tags_full = set(ops.PROFILE)
    finally:
        ops.PROFILE = None
    assert "moses_enc_fwd" in tags_enc and "moses_dec_fwd" not in tags_enc, tags_enc
    assert {"moses_enc_fwd", "moses_dec_fwd"} <= tags_full, tags_full
    assert abs(float(kl) - float(g["kl"])) < tl * abs(float(g["kl"]))
    assert rel(z.detach().cpu().numpy(), g["z"]) < tl and rel(logvar.detach().cpu().numpy(), g["logvar"]) < tl
    z, kl, logvar = model.forward_encoder(seqs, eps)           # (the full forward above took the encoder half's saved-state slot)
    model.zero_grad(set_to_none=True)
    kl.backward()
    torch.cuda.synchronize()
    ref = O.moses_forward(p64, [g[f"seq{b}"] for b in range(6)], g["eps"], int(g["pad"]))
    g1, g0 = ref["grads_for"](1.0), ref["grads_for"](0.0)       # linear in kl_w: the difference is the gradient of kl alone
    bad = {}
    for k, p_ in model.named_parameters():
        if k.startswith(ENC):
            key = "_x_emb_enc" if k == "x_emb.weight" else k
            e = rel(p_.grad.cpu().numpy(), g1[key] - g0[key])
            if e > tg:
                bad[k] = e
        else:
            assert p_.grad is None, k
    assert not bad, bad


@pytest.mark.parametrize("dtype,tg", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_encoder_then_decoder_equals_the_fused_forward(golden_dir, dtype, tg):
    """The reference's own composition (mosesvae.py:135-140: forward = forward_encoder, then forward_decoder on its z) through the two
    separate autograd nodes gives the fused node's losses, logits and EVERY parameter gradient (x_emb receives both halves' shares)."""
    g, model, p64, seqs = _g3(golden_dir, dtype)
    eps = torch.from_numpy(g["eps"].astype(np.float32)).to(dev)
    kl_w = float(g["kl_w"])
    kl, recon, z, logvar, x, y = model(seqs, eps)
    model.zero_grad(set_to_none=True)
    (kl_w * kl + recon).backward()
    fused = {k: p_.grad.clone() for k, p_ in model.named_parameters()}
    z2, kl2, lv2 = model.forward_encoder(seqs, eps)
    recon2, x2, y2 = model.forward_decoder(seqs, z2)
    assert torch.equal(z2, z) and torch.equal(kl2, kl) and torch.equal(lv2, logvar) and torch.equal(x2, x)
    assert torch.equal(y2, y) and torch.equal(recon2, recon)
    model.zero_grad(set_to_none=True)
    (kl_w * kl2 + recon2).backward()
    torch.cuda.synchronize()
    bad = {k: rel(p_.grad.cpu().numpy(), fused[k].cpu().numpy()) for k, p_ in model.named_parameters()
           if rel(p_.grad.cpu().numpy(), fused[k].cpu().numpy()) > tg}
    assert not bad, bad


def test_halves_keep_separate_saved_state(golden_dir):
    """forward_encoder, forward_decoder, then both backward passes in either order; a second forward_decoder invalidates only the first
    forward_decoder's state (saved_state_depth = 1), not the encoder's."""
    g, model, p64, seqs = _g3(golden_dir, torch.float32)
    eps = torch.from_numpy(g["eps"].astype(np.float32)).to(dev)
    z, kl, lv = model.forward_encoder(seqs, eps)
    r1, _, _ = model.forward_decoder(seqs, z.detach())
    r2, _, _ = model.forward_decoder(seqs, z.detach() * 0.5)
    kl.backward()                                               # encoder half untouched by the two decoder passes
    r2.backward()
    with pytest.raises(LL.MvaeError):
        r1.backward()
    model.saved_state_depth = 2
    z, kl, lv = model.forward_encoder(seqs, eps)
    r1, _, _ = model.forward_decoder(seqs, z.detach())
    r2, _, _ = model.forward_decoder(seqs, z.detach() * 0.5)
    r1.backward(); r2.backward(); kl.backward()


# ---------------------------------------------------------------------------------------------- optimiser split + init hook
def _moses_batches(v, n, B, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        lens = sorted(rs.randint(8, 40, size=B).tolist(), reverse=True)
        out.append([torch.from_numpy(np.concatenate([[v.bos], rs.randint(0, 26, size=k), [v.eos]]).astype(np.int64)) for k in lens])
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_optimizers_over_encoder_and_decoder_parameters(dtype):
    """moses_train_distrib_logp.py:267-268: Adam(model.encoder.parameters()) + Adam(model.decoder.parameters()) on ONE mosesvae.VAE (the
    aliasing ModuleLists; x_emb belongs to the encoder list) against one optimiser over model.parameters(): same learning rate, the
    reference's sequence clip_grad_norm_(all parameters) -> encoder step -> decoder step; 4 steps, identical losses and parameters."""
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    v = VC.OneHotVocab([chr(ord("a") + i) for i in range(26)])

    def build():
        torch.manual_seed(7)
        return MV.VAE(v, dtype=dtype).to(dev).train()

    batches = _moses_batches(v, 4, 32, 3)
    eps = [torch.from_numpy(np.random.RandomState(50 + i).standard_normal((32, 160)).astype(np.float32)).to(dev) for i in range(4)]
    m1, m2 = build(), build()
    n_enc, n_dec = len(list(m2.encoder.parameters())), len(list(m2.decoder.parameters()))
    assert n_enc + n_dec == len(list(m2.parameters())) == 29
    one = mv.FusedAdam(m1.parameters(), lr=5e-4)
    enc = mv.FusedAdam(m2.encoder.parameters(), lr=5e-4)
    dec = mv.FusedAdam(m2.decoder.parameters(), lr=5e-4)
    for i in range(4):
        losses = []
        for m, opts in ((m1, (one,)), (m2, (enc, dec))):
            for o in opts:
                o.zero_grad(set_to_none=True)
            kl, recon, _, _, _, _ = m(batches[i], eps[i], drop_seed=100 + i)
            loss = 0.3 * kl + recon
            loss.backward()
            torch.nn.utils.clip_grad_norm_((p for p in m.parameters() if p.requires_grad), 50)
            for o in opts:
                o.step()
            losses.append(float(loss))
        assert losses[0] == losses[1], (i, losses)
    for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), k


def test_model_apply_init_weights_reaches_the_linear_layers():
    """moses_train_distrib_logp.py:48-51,262: ``model.apply(init_weights)`` tests ``type(m) == nn.Linear``.  The parameter holders are real
    nn.Linear objects, so the hook re-initialises q_mu / q_logvar / decoder_lat / decoder_fc (6 layers) exactly as on the reference model,
    and the packed weight shadows follow (the hook writes the bias through ``.data``, which torch's version counters do not see)."""
    import torch.nn as nn
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    v = VC.OneHotVocab([chr(ord("a") + i) for i in range(26)])
    torch.manual_seed(3)
    model = MV.VAE(v, dtype=torch.float32).to(dev).eval()
    seqs = _moses_batches(v, 1, 8, 1)[0]
    eps = torch.zeros(8, 160, device=dev)
    before = model(seqs, eps)
    hit = []

    def init_weights(m):
        if type(m) == nn.Linear:
            torch.nn.init.xavier_uniform_(m.weight)
            m.bias.data.fill_(0.01)
            hit.append(m)

    torch.manual_seed(4)
    model.apply(init_weights)
    assert len(set(map(id, hit))) == 6
    assert all(float((m.bias - 0.01).abs().max()) == 0 for m in hit)
    after = model(seqs, eps)
    assert float((after[5] - before[5]).abs().max()) > 1e-3                      # the logits moved: shadows were refreshed
    params = {k: p_.detach().double().cpu().numpy() for k, p_ in model.named_parameters()}
    ref = O.moses_forward(params, [s.numpy() for s in seqs], np.zeros((8, 160)), v.pad, want_grads=False)
    assert rel(after[5].detach().cpu().numpy(), ref["y"]) < 2e-5 and abs(float(after[0]) - ref["kl"]) < 2e-5 * abs(ref["kl"])
    # the same on the main model: every nn.Linear of MolecularVAE (dense_1, the two lambda heads, latent_input, decoded_mean)
    m2 = mv.MolecularVAE(dtype=torch.float32)
    hit.clear()
    m2.apply(init_weights)
    assert len(hit) == 5
    with pytest.raises(RuntimeError):
        hit[0](torch.zeros(1, hit[0].in_features))                               # holders never compute on torch


# ---------------------------------------------------------------------------------------------- RCCL on the one GPU this box has
def _child(cmd, timeout=900):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}"
    return r


@pytest.mark.parametrize("extra", [[], ["--shard"], ["--model", "moses"], ["--persistent-shape"]],
                         ids=["allreduce_early_ranges", "reduce_scatter_sharded_adam", "moses_token_mean", "b128_bf16_persistent_passes"])
def test_rccl_one_rank_forced_collectives_equal_the_plain_run_bit_for_bit(tmp_path, extra):
    """train_distributed.py:72 replaced by per-process DP over RCCL: the collectives themselves, run on the ONE GPU of this box.  A fresh
    child process initialises torch.distributed with backend "nccl" (== RCCL), world_size 1, and GradSync(force=True) issues every collective
    of the N-rank step anyway -- the four early per-layer all-reduces started from the SIDE stream inside backward, the bucketed rest in
    step(), reduce-scatter + all-gather of the sharded form, the MOSES token-count all-reduce.  This is the part gloo cannot rehearse:
    ProcessGroupNCCL runs collectives on its own stream and hands over through events at issue and at wait().  Sums over one rank are the
    identity, so 4 optimiser steps (exact-f32 mode) must equal the non-distributed run BIT FOR BIT; a missing stream dependency shows up as
    a difference."""
    import json
    script = os.path.join(ROOT, "tests", "dp_equiv.py")
    a, b = os.path.join(str(tmp_path), "plain.json"), os.path.join(str(tmp_path), "rccl.json")
    base = [sys.executable, script, "--b", "32", "--steps", "4", "--dtype", "f32"] + [x for x in extra if x not in ("--shard", "--persistent-shape")]
    if "--persistent-shape" in extra:
        # the per-rank shape of the 8-GPU job (B = 128, bf16): the decoder passes are the ONE-launch dataflow kernels, which need every CU --
        # an RCCL kernel still resident when one of them starts would make it give up (MvaeError in the child); same kernels in both runs,
        # so bit-for-bit holds in bf16 as well
        base = [sys.executable, script, "--b", "64", "--steps", "4", "--dtype", "bf16"]
    _child(base + ["--out", a])
    _child(base + ["--out", b, "--backend", "nccl", "--force-comm"] + (["--shard"] if "--shard" in extra else []))
    ra, rb = json.load(open(a)), json.load(open(b))
    assert rb["buckets"] > 0, rb
    if not extra or "--persistent-shape" in extra:
        assert rb["early_ranges"] == 4 * 4                       # one range per decoder LSTM layer and step, issued from the side stream
    for k in ("loss", "psum", "gnorm", "pcheck"):
        assert ra[k] == rb[k], (k, ra[k], rb[k])
    if "--shard" in extra:
        assert rb["stale_state_dict_refused"] is True
        assert rb["mcheck"] == ra["mcheck"] == rb["mcheck_reloaded"] and rb["vcheck"] == rb["vcheck_reloaded"]


def test_two_rank_sharded_optimizer_checkpoint_needs_gather_state(tmp_path):
    """with FusedAdam(shard_optimizer=True) a rank holds current Adam moments for its own slice only.  state_dict()
    right after step() must refuse; after gather_state() on every rank it loads into an unsharded optimiser with every moment in place
    (two gloo ranks sharing this GPU)."""
    import json, socket
    script = os.path.join(ROOT, "tests", "dp_equiv.py")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = os.path.join(str(tmp_path), "sh.json")
    _child([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script, "--out", out, "--b", "16", "--steps", "3", "--dtype", "bf16", "--shard"])
    r = json.load(open(out))
    assert r["world"] == 2 and r["stale_state_dict_refused"] is True
    assert r["mcheck"] == r["mcheck_reloaded"] > 0 and r["vcheck"] == r["vcheck_reloaded"] > 0


# ---------------------------------------------------------------------------------------------- weights-resident dataflow LSTM forward (b = 128)
def _persist_case(T, seed=0, B=128):
    from molecular_vae_amd import _lib as L
    H, NL, PAD = 1024, 4, 64
    G4, ldw, ldh = 4 * H, H + PAD, H + PAD
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.03)
    dt = torch.bfloat16
    W = dict(Wih=[None] + [rnd(G4, ldw).to(dt) for _ in range(NL - 1)], Whh=[rnd(G4, ldw).to(dt) for _ in range(NL)],
             bias=[None] + [rnd(G4) * 10 for _ in range(NL - 1)], gx0=rnd(B, G4) * 30)

    def run(persist, save=True):
        b = dict(hs=[torch.zeros(T, B, ldh, device=dev, dtype=dt) for _ in range(NL)], cs=[torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)],
                 gates=[torch.zeros(T, B, G4, device=dev, dtype=dt) for _ in range(NL)], cstate=[torch.zeros(2, B, H, device=dev) for _ in range(NL)])
        ops.rnn_fwd(L.CELL_LSTM, dt, T, B, H, W["gx0"], 0, W["Wih"], [ldw] * NL, W["Whh"], [ldw] * NL, W["bias"], b["hs"], ldh,
                    b["cs"] if save else None, b["gates"] if save else None, b["cstate"], persist=persist)
        torch.cuda.synchronize()
        return b
    return run, (T, B, H, NL)


def _close_bf16(x, y, ulps=2.0):
    """|x - y| <= `ulps` bf16 steps at the magnitude of the buffer's largest value (the two schedules add the same fp32 products in
    different orders, so a value may round to the neighbouring bf16)."""
    x, y = x.float(), y.float()
    step = 2.0 ** (math.floor(math.log2(max(float(x.abs().max()), 1e-30))) - 7)
    return float((x - y).abs().max()) <= ulps * step


import math


def test_persistent_dataflow_forward_serves_256_rows_as_two_passes():
    """B = 256 (the per-rank batch of a 4-GPU job): two passes over 128 independent rows each, same buffers, same results as the wavefront."""
    run, (T, B, H, NL) = _persist_case(9, seed=2, B=256)
    n0 = ops.PERSIST_STATS["launches"]
    a, p = run(False), run(True)
    ops.persist_check(sync=True)
    assert ops.PERSIST_STATS["launches"] == n0 + 1
    for l in range(NL):
        assert _close_bf16(a["hs"][l][:, :, :H], p["hs"][l][:, :, :H]) and _close_bf16(a["gates"][l], p["gates"][l]) and _close_bf16(a["cs"][l], p["cs"][l]), l
        assert _close_bf16(a["cstate"][l][(T - 1) & 1], p["cstate"][l][(T - 1) & 1], ulps=4.0), l


@pytest.mark.parametrize("T", [1, 2, 7, 120])
def test_persistent_dataflow_lstm_forward_equals_the_wavefront_schedule(T):
    """rnn_persist.hip (ONE launch, weights resident in registers, h_t handed between workgroups through write-through stores + flag words)
    against the wavefront schedule (T + 3 launches) on the same random operands at the shape it serves (LSTM 4 x 1024, b = 128, bf16):
    every layer output, saved cell state, saved gate and the final fp32 cell state; the padding columns of hs stay untouched; the
    forward-only form (no saved state) gives the same outputs; the launch reports no failed hand-off."""
    run, (T, B, H, NL) = _persist_case(T)
    n0 = ops.PERSIST_STATS["launches"]
    a, p = run(False), run(True)
    ops.persist_check(sync=True)
    assert ops.PERSIST_STATS["launches"] == n0 + 1
    for l in range(NL):
        assert _close_bf16(a["hs"][l][:, :, :H], p["hs"][l][:, :, :H]), ("hs", l)
        assert float(p["hs"][l][:, :, H:].float().abs().max()) == 0.0, "padding columns written"
        assert _close_bf16(a["cs"][l], p["cs"][l]) and _close_bf16(a["gates"][l], p["gates"][l]), l
        assert _close_bf16(a["cstate"][l][(T - 1) & 1], p["cstate"][l][(T - 1) & 1], ulps=4.0), l
    q = run(True, save=False)
    ops.persist_check(sync=True)
    for l in range(NL):
        assert torch.equal(q["hs"][l], p["hs"][l]), l


def test_persistent_dataflow_counted_waits_equal_the_drained_form(monkeypatch):
    """The operand ring's counted s_waitcnt vmcnt(N) assume that loads and stores retire in issue order (what hipcc's own wait-count insertion
    assumes on gfx9-family targets, which have ONE counter for both).  MVAE_PERSIST_SAFE=1 drains every group of stores at once, so that no
    counted wait ever has a store in flight: the outputs must be the same bits."""
    run, (T, B, H, NL) = _persist_case(24, seed=3)
    p = run(True)
    monkeypatch.setenv("MVAE_PERSIST_SAFE", "1")
    q = run(True)
    ops.persist_check(sync=True)
    for k in ("hs", "cs", "gates", "cstate"):
        for l in range(NL):
            assert torch.equal(p[k][l], q[k][l]), (k, l)


def test_persistent_dataflow_failed_hand_off_is_reported_not_hung(monkeypatch):
    """Bounded spins: with the poll budget cut to ONE round (MVAE_PERSIST_SPIN=1) a consumer gives up at the first flag that is not there yet,
    every workgroup drains, the launch ENDS -- never a hang -- and the call (no poison slot given: it verifies its own launch) warns, runs the
    pass again on the wavefront schedule and returns CORRECT results; the next launch with the normal budget is fine."""
    run, (T, B, H, NL) = _persist_case(16, seed=5)
    a = run(False)
    monkeypatch.setenv("MVAE_PERSIST_SPIN", "1")
    with pytest.warns(RuntimeWarning, match="gave up"):
        q = run(True)
    assert ops.PERSIST_STATS["failures"] == 1 and ops.PERSIST_STATS["reruns"] == 1 and not ops.PERSIST_STATS["disabled"]
    for l in range(NL):
        assert torch.equal(a["hs"][l], q["hs"][l]) and torch.equal(a["gates"][l], q["gates"][l]), l       # the re-run IS the wavefront schedule
    monkeypatch.delenv("MVAE_PERSIST_SPIN")
    p = run(True)
    assert ops.PERSIST_STATS["failures"] == 1
    assert _close_bf16(a["hs"][3][:, :, :H], p["hs"][3][:, :, :H])


def test_persistent_schedule_inside_the_training_step_b128(golden_dir):
    """The product path at the per-rank batch of the 8-GPU configuration takes the persistent decoder forward (and the layer-concurrent encoder
    passes) by default; against the same model with the schedules switched off: loss within 1e-4, mu within 2e-6, every parameter gradient
    within the bf16 tolerance of the fixture tests."""
    g = np.load(os.path.join(golden_dir, "g2_full.npz"))
    params = ip.init_params(ip.molvae_shapes(), 202, 1.5, np.float32)
    rep = 32
    idx = torch.from_numpy(np.tile(g["idx"], (rep, 1))).to(dev)
    eps = torch.from_numpy(np.tile(g["eps"].astype(np.float32), (rep, 1))).to(dev)
    ohe = torch.nn.functional.one_hot(idx, 35).float()
    out = {}
    for mode in ("1", "0"):
        ops.PERSIST_DEFAULT = mode
        try:
            model = mv.MolecularVAE(dtype=torch.bfloat16)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
            model = model.to(dev)
            n0, nb0 = ops.PERSIST_STATS["launches"], ops.PERSIST_STATS["bwd_launches"]
            recon, mu, logvar = model(idx, eps)
            loss = mv.bce_kl_loss(recon, ohe, mu, logvar, 120)
            loss.backward()
            torch.cuda.synchronize()
            ops.persist_check(sync=True)
            # ... and the forward-only form right behind a training step (same workspace: the head GEMM and the backward pass have just
            # read hs with plain loads on every XCD; a hand-off that could hit a stale copy would show up here)
            model.eval()
            with torch.no_grad():
                recon_eval, _, _ = model(idx.flip(0), eps)
            torch.cuda.synchronize()
            ops.persist_check(sync=True)
            out[mode] = (float(loss), mu.detach().clone(), {k: p_.grad.clone() for k, p_ in model.named_parameters()}, ops.PERSIST_STATS["launches"] - n0,
                         recon_eval.clone(), ops.PERSIST_STATS["bwd_launches"] - nb0)
        finally:
            ops.PERSIST_DEFAULT = "1"
    assert out["1"][3] == 2 and out["0"][3] == 0
    assert out["1"][5] == 1 and out["0"][5] == 0            # ... and the persistent decoder backward
    assert float((out["1"][4] - out["0"][4]).abs().max()) < 2e-3          # evaluation (forward-only form) reconstructions
    assert abs(out["1"][0] - out["0"][0]) < 1e-4 * abs(out["0"][0]) and abs(out["1"][0] - float(g["loss"])) < 1e-4 * float(g["loss"])
    # (the switch also selects the encoder's layer-concurrent row-resident form, whose layers >= 1 contract [x | h] in one chain instead of
    # adding a hoisted input projection: exact-f32 re-association, ~1e-7)
    assert rel(out["1"][1].cpu().numpy(), out["0"][1].cpu().numpy()) < 2e-6
    bad = {k: rel(out["1"][2][k].cpu().numpy(), out["0"][2][k].cpu().numpy()) for k in out["1"][2]
           if rel(out["1"][2][k].cpu().numpy(), out["0"][2][k].cpu().numpy()) > 5e-2}
    assert not bad, bad


# ---------------------------------------------------------------------------------------------- weights-resident dataflow LSTM backward (b = 128)
def _persist_bwd_case(T, seed=0, B=128):
    """Saved forward state from the wavefront forward, a random output gradient, and a runner for either backward schedule."""
    from molecular_vae_amd import _lib as L
    H, NL, PAD = 1024, 4, 64
    G4, ldw, ldh, ldwT, ldg = 4 * H, H + PAD, H + PAD, 4 * H + PAD, 4 * H + PAD
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.03)
    dt = torch.bfloat16
    Wih = [None] + [rnd(G4, ldw).to(dt) for _ in range(NL - 1)]
    Whh = [rnd(G4, ldw).to(dt) for _ in range(NL)]
    bias = [None] + [rnd(G4) * 10 for _ in range(NL - 1)]
    gx0 = rnd(B, G4) * 30

    def tr(w):
        o = torch.zeros(H, ldwT, device=dev, dtype=dt)
        o[:, :G4] = w[:, :H].t()
        return o
    WihT, WhhT = [None] + [tr(w) for w in Wih[1:]], [tr(w) for w in Whh]
    hs = [torch.zeros(T, B, ldh, device=dev, dtype=dt) for _ in range(NL)]
    cs = [torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)]
    gates = [torch.zeros(T, B, G4, device=dev, dtype=dt) for _ in range(NL)]
    cstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
    ops.rnn_fwd(L.CELL_LSTM, dt, T, B, H, gx0, 0, Wih, [ldw] * NL, Whh, [ldw] * NL, bias, hs, ldh, cs, gates, cstate, persist=False)
    dy = rnd(T, B, H) * 3

    def run(persist):
        dG = [torch.full((T, B, ldg), 7.0, device=dev, dtype=dt) for _ in range(NL)]
        dstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
        ops.rnn_bwd(L.CELL_LSTM, dt, T, B, H, WhhT, [ldwT] * NL, WihT, [ldwT] * NL, dy.view(T * B, H), H, hs, ldh, cs, gates, dG, dstate, ldg=ldg,
                    persist=persist)
        torch.cuda.synchronize()
        return dG
    return run, (T, B, H, NL)


def _close_rel(x, y, tol):
    x, y = x.float(), y.float()
    return bool(torch.isfinite(y).all()) and float((x - y).abs().max()) <= tol * float(x.abs().max())


@pytest.mark.parametrize("T", [1, 2, 7, 120])
def test_persistent_dataflow_lstm_backward_equals_the_wavefront_schedule(T):
    """rnn_persist_bwd.hip (ONE launch: weights resident in registers, dG tiles and K-quarter partial sums handed between workgroups through
    write-through stores + flag words / phase-tagged payloads) against the wavefront backward (2 x (T + 3) launches) on the same saved state
    and output gradient at the shape it serves: every layer's pre-activation gradient within 2 bf16 steps of the buffer's largest value
    (the two schedules add the same fp32 products in different orders); the padding columns of dG stay untouched; no failed hand-off."""
    run, (T, B, H, NL) = _persist_bwd_case(T)
    n0 = ops.PERSIST_STATS["bwd_launches"]
    a, p = run(False), run(True)
    ops.persist_check(sync=True)
    assert ops.PERSIST_STATS["bwd_launches"] == n0 + 1
    for l in range(NL):
        assert _close_rel(a[l][:, :, :4 * H], p[l][:, :, :4 * H], 2.0 ** -6), l
        assert float((p[l][:, :, 4 * H:].float() - 7.0).abs().max()) == 0.0, "padding columns written"


def test_persistent_dataflow_backward_serves_256_rows_as_two_passes_and_is_row_position_independent():
    """B = 256: two passes over 128 independent rows each.  The batch holds every sequence twice (rows r and 128 + r, i.e. the same row of the
    two passes) and rows 5 and 77 of a pass are copies of each other as well: all copies must get the same BITS (the partial sums are added
    in an order that does not depend on where a row sits)."""
    from molecular_vae_amd import _lib as L
    T, B, H, NL, PAD = 6, 256, 1024, 4, 64
    G4, ldw, ldh, ldwT, ldg = 4 * H, H + PAD, H + PAD, 4 * H + PAD, 4 * H + PAD
    g = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.03)
    dt = torch.bfloat16
    Wih = [None] + [rnd(G4, ldw).to(dt) for _ in range(NL - 1)]
    Whh = [rnd(G4, ldw).to(dt) for _ in range(NL)]
    tr = lambda w: torch.cat([w[:, :H].t().contiguous(), torch.zeros(H, PAD, device=dev, dtype=dt)], dim=1).contiguous()
    WihT, WhhT = [None] + [tr(w) for w in Wih[1:]], [tr(w) for w in Whh]
    half = rnd(128, G4) * 30
    half[77] = half[5]
    gx0 = torch.cat([half, half], 0).contiguous()
    hs = [torch.zeros(T, B, ldh, device=dev, dtype=dt) for _ in range(NL)]
    cs = [torch.zeros(T, B, H, device=dev, dtype=dt) for _ in range(NL)]
    gates = [torch.zeros(T, B, G4, device=dev, dtype=dt) for _ in range(NL)]
    cstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
    ops.rnn_fwd(L.CELL_LSTM, dt, T, B, H, gx0, 0, Wih, [ldw] * NL, Whh, [ldw] * NL, [None] * NL, hs, ldh, cs, gates, cstate, persist=True)
    dyh = rnd(T, 128, H) * 3
    dyh[:, 77] = dyh[:, 5]
    dy = torch.cat([dyh, dyh], 1).contiguous()
    out = {}
    for persist in (False, True):
        dG = [torch.zeros(T, B, ldg, device=dev, dtype=dt) for _ in range(NL)]
        dstate = [torch.zeros(2, B, H, device=dev) for _ in range(NL)]
        ops.rnn_bwd(L.CELL_LSTM, dt, T, B, H, WhhT, [ldwT] * NL, WihT, [ldwT] * NL, dy.view(T * B, H), H, hs, ldh, cs, gates, dG, dstate, ldg=ldg,
                    persist=persist)
        torch.cuda.synchronize()
        out[persist] = dG
    ops.persist_check(sync=True)
    for l in range(NL):
        assert _close_rel(out[False][l][:, :, :G4], out[True][l][:, :, :G4], 2.0 ** -6), l
        assert torch.equal(out[True][l][:, :128], out[True][l][:, 128:]), ("passes differ", l)
        assert torch.equal(out[True][l][:, 5], out[True][l][:, 77]), ("row position", l)


def test_persistent_dataflow_backward_counted_waits_equal_the_drained_form(monkeypatch):
    """MVAE_PERSIST_SAFE=1 drains every group of stores at once (no counted wait ever has a store in flight): same bits."""
    run, (T, B, H, NL) = _persist_bwd_case(24, seed=3)
    p = run(True)
    monkeypatch.setenv("MVAE_PERSIST_SAFE", "1")
    q = run(True)
    ops.persist_check(sync=True)
    for l in range(NL):
        assert torch.equal(p[l], q[l]), l


def test_persistent_dataflow_backward_failed_hand_off_is_reported_not_hung(monkeypatch):
    """Poll budget of ONE round: somebody gives up at the first flag / partial that is not there yet, every workgroup drains, the launch ends;
    the call warns, re-runs the pass on the wavefront schedule (correct dG) and counts the failure; three failures switch the spinning
    schedules off for the process; the next launch with the normal budget is fine."""
    run, (T, B, H, NL) = _persist_bwd_case(16, seed=5)
    a = run(False)
    monkeypatch.setenv("MVAE_PERSIST_SPIN", "1")
    with pytest.warns(RuntimeWarning, match="gave up"):
        q = run(True)
    for l in range(NL):
        assert torch.equal(a[l], q[l]), l
    assert ops.PERSIST_STATS["failures"] == 1 and ops.PERSIST_STATS["reruns"] == 1
    monkeypatch.delenv("MVAE_PERSIST_SPIN")
    p = run(True)
    assert _close_rel(a[0][:, :, :4 * H], p[0][:, :, :4 * H], 2.0 ** -6) and ops.PERSIST_STATS["failures"] == 1
    monkeypatch.setenv("MVAE_PERSIST_SPIN", "1")
    with pytest.warns(RuntimeWarning):
        run(True); run(True)
    assert ops.PERSIST_STATS["failures"] == 3 and ops.PERSIST_STATS["disabled"]
    monkeypatch.delenv("MVAE_PERSIST_SPIN")
    n0 = ops.PERSIST_STATS["bwd_launches"]
    r = run(None)                                         # default choice: the schedule is off now -> wavefront
    assert ops.PERSIST_STATS["bwd_launches"] == n0 and all(torch.equal(a[l], r[l]) for l in range(NL))


def test_persistent_backward_is_refused_off_its_shape_and_switched_by_its_knob(monkeypatch):
    """persist=True on a shape the dataflow kernel does not serve raises (never a silent fallback when the caller asked for it); the default
    (persist=None) follows ops.PERSIST_DEFAULT / MVAE_PERSIST_BWD (honoured under MVAE_TUNING=1, which the test session sets)."""
    run64, _ = _persist_bwd_case(3, seed=9, B=64)
    with pytest.raises(LL.MvaeError):
        run64(True)
    run, (T, B, H, NL) = _persist_bwd_case(3, seed=9)
    n0 = ops.PERSIST_STATS["bwd_launches"]
    a = run(None)
    assert ops.PERSIST_STATS["bwd_launches"] == n0 + 1
    monkeypatch.setenv("MVAE_PERSIST_BWD", "0")
    b = run(None)
    assert ops.PERSIST_STATS["bwd_launches"] == n0 + 1          # the wavefront schedule this time
    ops.persist_check(sync=True)
    for l in range(NL):
        assert _close_rel(b[l][:, :, :4 * H], a[l][:, :, :4 * H], 2.0 ** -6), l


def test_persistent_backward_back_to_back_launches_into_the_same_buffers():
    """Six launches queued without synchronisation, different output gradients, the SAME dG / exchange / flag buffers: the last one's result
    must be the last gradient's (a hand-off that hit a stale line or a stale phase bit of an earlier launch would reproduce old values)."""
    from molecular_vae_amd import _lib as L
    run, (T, B, H, NL) = _persist_bwd_case(12, seed=8)
    a = run(False)
    for _ in range(5):
        run(True)
    p = run(True)
    ops.persist_check(sync=True)
    for l in range(NL):
        assert _close_rel(a[l][:, :, :4 * H], p[l][:, :, :4 * H], 2.0 ** -6), l


# ---------------------------------------------------------------------------------------------- sampling step (mosesvae.py:236-253)
def test_sampling_step_distribution_and_bookkeeping():
    """mvae_moses_sample_step on a fixed logit row: (1) the draw is the inverse CDF of softmax(y / temp) at the documented uniform
    hash(seed, step * B + b) / 2^32 -- recomputed on the host, >= 99.9 % of 8192 rows agree exactly (the rest sit on a cumulative-sum
    boundary in fp32) -- and the frequencies match the probabilities within 4 sigma (against torch.multinomial's own frequencies too);
    (2) the reference's bookkeeping: x[b, i] is written unless the row has ended, a first <eos> sets end_pads = i + 1 and the mask;
    (3) the next step's input rows are table[w] + base."""
    B, V, H, W = 8192, 30, 64, 128
    rs = np.random.RandomState(0)
    logits = rs.standard_normal(V).astype(np.float32) * 2.0
    temp, seed, step, eos = 0.7, 12345, 3, 5
    h = torch.zeros(B, H, device=dev); h[:, 0] = 1.0
    wfc = torch.zeros(V, H, device=dev); wfc[:, 0] = torch.from_numpy(logits).to(dev)
    table = torch.from_numpy(rs.standard_normal((V, W)).astype(np.float32)).to(dev)
    base = torch.from_numpy(rs.standard_normal((B, W)).astype(np.float32)).to(dev)
    add = torch.empty(B, W, device=dev)
    x = torch.full((B, 8), 99, dtype=torch.long, device=dev)
    end_pads = torch.full((B,), 8, dtype=torch.long, device=dev)
    eos_mask = torch.zeros(B, dtype=torch.uint8, device=dev); eos_mask[::7] = 1       # some rows have already ended
    w = torch.empty(B, dtype=torch.long, device=dev)
    ops.moses_sample_step(h, H, wfc, None, temp, seed, step, eos, table, base, add, x, end_pads, eos_mask, w, B, V, H)
    torch.cuda.synchronize()
    wn = w.cpu().numpy()
    p = np.exp((logits - logits.max()) / temp).astype(np.float64); p /= p.sum()
    u = ops.sample_uniform(seed, step, B)
    want = np.minimum((np.cumsum(p)[None, :] <= u[:, None]).sum(1), V - 1)
    assert (wn == want).mean() >= 0.999, (wn != want).sum()
    freq = np.bincount(wn, minlength=V) / B
    assert (np.abs(freq - p) < 4 * np.sqrt(p * (1 - p) / B) + 1e-4).all(), np.abs(freq - p).max()
    tm = torch.multinomial(torch.from_numpy(p).float().expand(B, V), 1)[:, 0].numpy()
    assert (np.abs(np.bincount(tm, minlength=V) / B - freq) < 6 * np.sqrt(p * (1 - p) / B) + 1e-4).all()
    ended = np.zeros(B, bool); ended[::7] = True
    xs, ep, em = x.cpu().numpy(), end_pads.cpu().numpy(), eos_mask.cpu().numpy()
    assert (xs[~ended, step] == wn[~ended]).all() and (xs[ended, step] == 99).all() and (np.delete(xs, step, 1) == 99).all()
    new_eos = ~ended & (wn == eos)
    assert (ep[new_eos] == step + 1).all() and (ep[~new_eos] == 8).all() and (em.astype(bool) == (ended | new_eos)).all()
    assert torch.equal(add, table[w] + base)
    # a different step / seed gives a different, again reproducible draw
    w2 = torch.empty_like(w); ops.moses_sample_step(h, H, wfc, None, temp, seed, step + 1, eos, table, base, add, x, end_pads, eos_mask, w2, B, V, H)
    w3 = torch.empty_like(w); ops.moses_sample_step(h, H, wfc, None, temp, seed, step + 1, eos, table, base, add, x, end_pads, eos_mask, w3, B, V, H)
    assert torch.equal(w2, w3) and not torch.equal(w2, w)


def test_moses_sample_is_reproducible_from_its_seed_and_launches_four_kernels_per_token(golden_dir):
    """VAE.sample: explicit seed -> identical strings; per generated token the loop issues ONE GRU wavefront pass (3 launches) + ONE sampling
    launch through the C ABI and no torch arithmetic (counted with a torch profiler around the loop)."""
    g, model, params = _moses_setup(golden_dir, torch.bfloat16)
    a, za = model.sample(64, max_len=20, temp=1.0, seed=7, return_tokens=True)
    b, zb = model.sample(64, max_len=20, temp=1.0, seed=7, z=za, return_tokens=True)
    c, _ = model.sample(64, max_len=20, temp=1.0, seed=8, z=za, return_tokens=True)
    assert all(torch.equal(x_, y_) for x_, y_ in zip(a, b)) and not all(torch.equal(x_, y_) for x_, y_ in zip(a, c))
    assert all(int(t[0]) == model.bos for t in a) and all(len(t) <= 20 for t in a)
    from torch.profiler import profile, ProfilerActivity
    model.sample(64, max_len=6, seed=1)                                   # warm
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        model.sample(64, max_len=26, seed=1)
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    per_token = sum(1 for n_ in names if "moses_sample_step" in n_)
    assert per_token == 25
    steps = sum(1 for n_ in names if "gru_step" in n_ or "lstm_step" in n_ or "step_fwd" in n_)
    assert steps == 3 * 25, (steps, sorted(set(names)))


@pytest.mark.parametrize("B", [8, 128, 200])
def test_layer_concurrent_row_resident_encoder_passes_equal_the_layer_by_layer_form(B, monkeypatch):
    """rnn_rowres.hip, round 4: the encoder LSTM(72) x 3 forward and backward as ONE launch each with the layers running concurrently (a
    pipeline over per-workgroup progress words, agent-scope stores / loads of the hand-off rows, bounded polls) against the layer-after-layer
    launches (MVAE_ROWRES_PIPE=0, input projection not hoisted so that both add the same products in the same order): outputs, saved state
    and every pre-activation gradient agree to exact-f32 rounding (the two forms are separate instantiations: the compiler's fma contraction
    may differ by an ulp, 4e-8 observed), for a batch that is not a multiple of the 4 rows a workgroup owns too."""
    from molecular_vae_amd import _lib as L
    T, H, NL, Hp, G4 = 120, 72, 3, 96, 288
    g = torch.Generator(device="cuda").manual_seed(B)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.2
    Wih = [None] + [torch.zeros(G4, Hp, device=dev) for _ in range(NL - 1)]
    Whh = [torch.zeros(G4, Hp, device=dev) for _ in range(NL)]
    for w in Wih[1:] + Whh:
        w[:, :H] = rnd(G4, H)
    WihT = [None] + [w[:, :H].t().contiguous() for w in Wih[1:]]
    WhhT = [w[:, :H].t().contiguous() for w in Whh]
    bias = [None] + [rnd(G4) for _ in range(NL - 1)]
    tbl = rnd(35, G4)
    idx = torch.randint(0, 35, (B, T), device=dev, generator=g)
    dy = rnd(T, B, H)

    def run(pipe):
        monkeypatch.setenv("MVAE_ROWRES_PIPE", pipe)
        monkeypatch.setenv("MVAE_ROWRES_HOIST", "0")
        b = dict(hs=[torch.zeros(T, B, Hp, device=dev) for _ in range(NL)], cs=[torch.zeros(T, B, H, device=dev) for _ in range(NL)],
                 gates=[torch.zeros(T, B, G4, device=dev) for _ in range(NL)], cstate=[torch.zeros(2, B, H, device=dev) for _ in range(NL)],
                 dG=[torch.zeros(T, B, G4, device=dev) for _ in range(NL)], dstate=[torch.zeros(2, B, H, device=dev) for _ in range(NL)])
        ops.rnn_fwd(L.CELL_LSTM, torch.float32, T, B, H, None, 0, Wih, [Hp] * NL, Whh, [Hp] * NL, bias, b["hs"], Hp, b["cs"], b["gates"], b["cstate"],
                    zero_padded_k=True, add_table=tbl, add_index=idx, persist=(pipe == "1"))
        ops.rnn_bwd(L.CELL_LSTM, torch.float32, T, B, H, WhhT, [G4] * NL, WihT, [G4] * NL, dy, H, b["hs"], Hp, b["cs"], b["gates"], b["dG"], b["dstate"])
        torch.cuda.synchronize()
        ops.persist_check(sync=True)
        return b
    n0 = ops.PERSIST_STATS["rowres_pipe"]
    a, p = run("0"), run("1")
    assert ops.PERSIST_STATS["rowres_pipe"] == n0 + 2        # the forward AND the backward pass of run("1") reported a status record (= took the pipelined form)
    for k, tol in (("hs", 2e-6), ("cs", 2e-6), ("gates", 2e-6), ("dG", 2e-5)):
        for l in range(NL):
            assert rel(a[k][l].cpu().numpy(), p[k][l].cpu().numpy()) < tol, (k, l, float((a[k][l] - p[k][l]).abs().max()))


def test_gru_rowres_counted_waits_equal_the_drained_form_at_the_bench_shape(monkeypatch):
    """gru_rowres_fwd / _bwd (the MOSES encoder GRU(256), one launch per direction) retire their prefetched operands
    with COUNTED s_waitcnt vmcnt(N), which is exact only if loads and stores retire in issue order -- what hipcc's own wait-count insertion
    assumes on gfx9-family targets (one counter for both kinds; tests/test_host_logic.py checks the per-step instruction counts the N's are
    built from).  MVAE_GRU_ROWRES_SAFE=1 turns every counted wait into vmcnt(0): at the bench shape (B = 1024, lengths ~N(38, 8)) the whole
    training step must give the same bits either way."""
    import bench_extra
    outs = []
    for safe in ("0", "1"):
        monkeypatch.setenv("MVAE_GRU_ROWRES_SAFE", safe)
        torch.manual_seed(5)
        wl = bench_extra.MosesWorkload(1024, "bf16", dev, 0, None)
        wl.model.eval()                                    # (dropout off: the comparison is about the encoder kernels)
        kl, recon, z, logvar, x, y = wl.model(wl.batch, eps=torch.zeros(1024, 160, device=dev))
        (0.5 * kl + recon).backward()
        torch.cuda.synchronize()
        outs.append((kl.detach().clone(), z.detach().clone(), {k: p_.grad.clone() for k, p_ in wl.model.named_parameters() if k.startswith(("encoder_rnn", "x_emb"))}))
        del wl
        ops.release_caches()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k
