"""CPU tests of the host-side surface around the hot path: the MOSES vocabulary / collate against the reference-generated fixture
(g5), the mosesvae state-dict aliases against g3, the schedule helpers of moses_train_distrib.py:47-89, the optimiser / checkpoint
dictionaries of train.py:170-177, and bench.py's launch contract (no GPU is touched anywhere here)."""
import io
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import molecular_vae_amd as mv
from molecular_vae_amd import _lib as L
from molecular_vae_amd import vocab as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ vocab.py:10-87 vs g5
def test_charvocab_matches_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_vocab.npz"))
    order = [str(s) for s in g["order"]]                       # the reference's collate ordering of the fixture's SMILES list
    smiles = sorted(order, key=lambda s: order.index(s))       # any order: from_data takes the union of characters
    v = V.CharVocab.from_data(smiles)
    assert [v.i2c[i] for i in range(len(v))] == [str(s) for s in g["symbols"]]
    assert (v.bos, v.eos, v.pad, v.unk) == (int(g["bos"]), int(g["eos"]), int(g["pad"]), int(g["unk"]))
    assert v.char2id("?") == int(g["unk_probe"]) == v.unk      # unknown character -> <unk>
    # the fixture stores string2ids(add_bos, add_eos) of the ORIGINAL list order; recover that order from the lengths + contents
    ids_fixture = [g[f"ids{n}"].tolist() for n in range(len(order))]
    by_ids = {tuple(v.string2ids(s, add_bos=True, add_eos=True)): s for s in smiles}
    assert sorted(by_ids) == sorted(tuple(i) for i in ids_fixture)
    for ids in ids_fixture:
        s = by_ids[tuple(ids)]
        assert v.ids2string(ids) == s                           # round trip strips <bos>/<eos>
        assert v.ids2string(ids, rem_bos=False, rem_eos=False) == "<bos>" + s + "<eos>"
        assert torch.equal(V.string2tensor(v, s), torch.tensor(ids))
    # collate: stable sort by length, longest first (moses_train_distrib.py:127-135), then <bos> ids <eos> as int64
    original = [by_ids[tuple(i)] for i in ids_fixture]
    out = V.get_collate_fn(v)(list(original))
    assert [v.ids2string(t.tolist()) for t in out] == order
    assert all(t.dtype == torch.long for t in out)
    with pytest.raises(ValueError):
        V.CharVocab(set("C") | {"<pad>"})                      # vocab.py:22 ValueError('SS in chars')
    oh = V.OneHotVocab.from_data(smiles)
    assert torch.equal(oh.vectors, torch.eye(len(v)))


def test_mosesvae_state_dict_aliases_match_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_moses.npz"))
    chars = [chr(ord("a") + i) for i in range(26)]
    vocab = V.OneHotVocab(set(chars))
    assert len(vocab) == int(g["V"]) == 30
    model = mv.mosesvae.VAE(vocab)
    assert sorted(model.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]     # 88 keys over 29 tensors
    assert len(list(model.parameters())) == 29 == len(g["grad_names"])
    assert sorted(n for n, _ in model.named_parameters()) == [str(k) for k in g["grad_names"]]
    assert (model.bos, model.eos, model.pad, model.unk) == (int(g["bos"]), int(g["eos"]), int(g["pad"]), int(g["unk"]))
    # reference-layout checkpoints load, DataParallel-prefixed ones after the strip (mosesanalyize.py:171-173)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(mv.strip_module_prefix({"module." + k: v for k, v in sd.items()}))
    with pytest.raises(ValueError):
        mv.mosesvae.VAE(type("Bad", (), dict(bos=0, eos=1, pad=2, unk=3, vectors=torch.eye(30)[:, :7], __len__=lambda s: 30))())


def test_models2d_module_surface_matches_reference_layout():
    from oracle import initparams as ip
    m = mv.models2d.VAE()
    shapes = ip.models2d_shapes()                                # models2d.py:12-21
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) or set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    # same RNG consumption as stock torch.nn modules constructed in the reference's order
    torch.manual_seed(9); a = mv.models2d.VAE().state_dict()
    torch.manual_seed(9)
    c1 = torch.nn.Conv1d(120, 9, 9); torch.nn.Conv1d(9, 9, 9); torch.nn.Conv1d(9, 10, 11); torch.nn.Linear(90, 435)
    torch.nn.Linear(435, 2); torch.nn.Linear(435, 2); torch.nn.Linear(2, 2); gru = torch.nn.GRU(2, 501, 3, batch_first=True); f3 = torch.nn.Linear(501, 35)
    assert torch.equal(a["conv1d1.weight"], c1.weight) and torch.equal(a["gru.weight_hh_l2"], gru.weight_hh_l2) and torch.equal(a["fc3.bias"], f3.bias)
    with pytest.raises(L.MvaeError):
        m(torch.zeros(2, 120, 35))                               # no CPU fallback
    with pytest.raises(ValueError):
        m(torch.zeros(2, 35, 120))


# ------------------------------------------------------------------------------------------------ schedules
def test_kl_annealer_and_cosine_restart_follow_the_reference_formulas():
    k = mv.KLAnnealer(100)                                      # moses_train_distrib.py:47-58
    assert k(0) == 0 and abs(k(37) - 0.37) < 1e-12 and abs(k(100) - 1.0) < 1e-12
    p = torch.nn.Parameter(torch.zeros(4))
    opt = mv.FusedAdam([p], lr=3e-4, max_grad_norm=50.0)        # moses_train_distrib.py:188,227
    sched = mv.CosineAnnealingLRWithRestart(opt)                # :61-89; the constructor takes the first step
    seen = [opt.param_groups[0]["lr"]]
    for _ in range(24):
        sched.step()
        seen.append(opt.param_groups[0]["lr"])
    for i, lr in enumerate(seen):
        kk = i % 10 + 1                                         # current_epoch runs 1..10, lr_end is reached at 10, then restarts
        want = 1e-4 + (3e-4 - 1e-4) * (1 + math.cos(math.pi * kk / 10)) / 2
        assert abs(lr - want) < 1e-15, (i, lr, want)
    assert abs(seen[9] - 1e-4) < 1e-15 and seen[10] > seen[9]
    assert mv.cosine_lr_with_restart(3e-4, 0) == 3e-4
    # against torch's own _LRScheduler machinery driving the same step() logic
    sd = sched.state_dict()
    s2 = mv.CosineAnnealingLRWithRestart(mv.FusedAdam([torch.nn.Parameter(torch.zeros(1))], lr=3e-4))
    s2.load_state_dict(sd)
    assert s2.current_epoch == sched.current_epoch and s2.get_lr() == sched.get_lr()


# ------------------------------------------------------------------------------------------------ optimiser / checkpoints
def _tiny():
    torch.manual_seed(3)
    return mv.MolecularVAE(i=24, o=16, c=12)


def test_fused_adam_state_dict_round_trips_with_torch_adam():
    m = _tiny()
    fa = mv.FusedAdam(m.parameters(), lr=8e-4, max_grad_norm=3.0)
    ps = list(m.parameters())
    # every parameter is a view of the flat buffer and keeps its value
    m2 = _tiny()
    for a, b in zip(ps, m2.parameters()):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(0)
    for p in ps:                                                 # pretend three steps happened
        fa.state[p]["exp_avg"].copy_(torch.randn(p.shape, generator=g)); fa.state[p]["exp_avg_sq"].copy_(torch.rand(p.shape, generator=g))
        fa.state[p]["step"] = torch.tensor(3.0)
    for f in fa._flat:
        f["step"] = 3
    sd = fa.state_dict()
    ta = torch.optim.Adam(m2.parameters(), lr=1.0)
    ta.load_state_dict(sd)                                       # FusedAdam -> torch.optim.Adam
    assert ta.param_groups[0]["lr"] == 8e-4 and ta.param_groups[0]["amsgrad"] is False
    for p in m2.parameters():
        p.grad = torch.ones_like(p) * 1e-3
    ta.step()                                                    # and torch's step runs on the loaded dict (step count 3 -> 4)
    assert all(float(ta.state[p]["step"]) == 4.0 for p in m2.parameters())
    m3 = _tiny()
    fb = mv.FusedAdam(m3.parameters(), lr=1.0, max_grad_norm=3.0)
    epoch0 = L.PARAM_EPOCH[0]
    fb.load_state_dict(ta.state_dict())                          # torch.optim.Adam -> FusedAdam
    assert fb.param_groups[0]["lr"] == 8e-4 and fb.param_groups[0]["max_grad_norm"] == 3.0
    for p2, p3 in zip(m2.parameters(), m3.parameters()):
        assert torch.equal(ta.state[p2]["exp_avg"], fb.state[p3]["exp_avg"]) and float(fb.state[p3]["step"]) == 4.0
    assert fb._flat[0]["step"] == 4 and L.PARAM_EPOCH[0] >= epoch0
    # the moments are still views of the flat buffers after loading
    assert fb.state[next(iter(m3.parameters()))]["exp_avg"].data_ptr() == fb._flat[0]["m"].data_ptr()
    with pytest.raises(ValueError):
        fb.load_state_dict({"state": {}, "param_groups": [{"params": [0]}]})
    with pytest.raises(L.MvaeError):
        fb.step()                                                # no CPU fallback


def test_checkpoint_dictionary_of_train_py(tmp_path):
    m = _tiny()
    opt = mv.FusedAdam(m.parameters(), lr=8e-4, max_grad_norm=3.0)
    charset = {0: " ", 1: "C"}
    path = os.path.join(str(tmp_path), "save_128_adam_16.pt")
    d = mv.save_checkpoint(path, m, opt, epoch=7, charset=charset, max_len=24, latent_size=16)
    assert set(d) == {"model_state_dict", "optimizer_state_dict", "epoch", "charset", "max_len", "lr", "latent_size"}   # train.py:170-177
    torch.manual_seed(99)
    m2 = mv.MolecularVAE(i=24, o=16, c=12)
    opt2 = mv.FusedAdam(m2.parameters(), lr=1.0, max_grad_norm=3.0)
    ck = mv.load_checkpoint(path, m2, opt2)
    assert ck["epoch"] == 7 and ck["charset"] == charset and ck["lr"] == 8e-4 and opt2.param_groups[0]["lr"] == 8e-4
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert m2.encoder.embedding.weight.data_ptr() == opt2._flat[0]["p"].data_ptr()        # still a view of the optimiser's flat buffer
    # train_distributed.py:145-151: DataParallel keys, no latent_size
    buf = io.BytesIO()
    torch.save({"model_state_dict": {"module." + k: v for k, v in m.state_dict().items()}, "optimizer_state_dict": opt.state_dict(),
                "epoch": 1, "charset": charset, "max_len": 24, "lr": 8e-4}, buf)
    buf.seek(0)
    m3 = mv.MolecularVAE(i=24, o=16, c=12)
    mv.load_checkpoint(buf, m3)
    assert torch.equal(m3.decoder.gru.weight_hh_l1, m.decoder.gru.weight_hh_l1)
    # and into stock torch.nn modules under the reference's key names (what a reference user would load it into)
    from oracle import torch_ref
    ref = torch_ref.CpuPort(i=24, o=16, c=12)
    ref.load_state_dict(m.state_dict())
    assert torch.equal(ref.decoder.gru.weight_ih_l0, m.decoder.gru.weight_ih_l0)


def test_grad_sinks_die_with_their_optimizer():
    m = _tiny()
    ps = list(m.decoder.parameters())
    o1 = mv.FusedAdam(m.parameters(), lr=1e-3)
    r = L.grad_sink_range(ps)
    assert r is not None and r[0] is o1 and r[3] - r[2] == sum(p.numel() for p in ps)
    del o1, r
    import gc; gc.collect()
    assert L.grad_sink_range(ps) is None and not hasattr(ps[0], "_mvae_sink")             # dead optimiser: entry dropped on sight
    o2 = mv.FusedAdam(m.parameters(), lr=1e-3)
    assert L.grad_sink_range(ps)[0] is o2
    assert L.grad_sink_range(ps[::-1]) is None                                             # not one contiguous in-order range


def test_gather_grads_refuses_a_cloned_gradient_inside_an_early_range():
    m = _tiny()
    sync = mv.GradSync()
    opt = mv.FusedAdam(m.parameters(), lr=1e-3, grad_sync=sync)
    ps = list(m.parameters())
    flat = opt._flat[0]["g"]
    off = sum(p.numel() for p in ps[:5])
    ps[5].grad = torch.ones_like(ps[5])                          # NOT the sink view: what AccumulateGrad leaves behind when it clones
    sync.early.append((flat, off, off + ps[5].numel()))          # as if backward had started this range's all-reduce
    with pytest.raises(L.MvaeError):
        opt.gather_grads()
    sync.early.clear()
    opt.gather_grads()                                           # without an early range the copy is fine
    assert float(flat[off:off + ps[5].numel()].sum()) == ps[5].numel()


# ------------------------------------------------------------------------------------------------ bench.py launch contract
def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in [k for k in e if k.startswith("MVAE_")] + ["WORLD_SIZE", "RANK", "LOCAL_RANK"]:
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=300)


def test_bench_refuses_more_gpus_than_the_box_has():
    r = _bench("--gpus", "64", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "GPU" in r.stderr and not r.stdout.strip()      # errors instead of printing n_gpus: 1


def test_bench_refuses_world_size_mismatch_and_tuning_variables():
    r = _bench("--gpus", "2", "--steps", "1", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()
    r = _bench("--gpus", "1", "--steps", "1", env={"MVAE_DBG": "1"})
    assert r.returncode != 0 and "MVAE_DBG" in r.stderr and not r.stdout.strip()


def test_graft_entry_build_hook_runs():
    """The driver's "does it build" hook: compiles every HIP source for gfx950 and loads the library (no GPU needed)."""
    import __graft_entry__ as ge
    ge.build()


# ------------------------------------------------------------------------------------------------ round 4: surface rows a13 / a14, init hooks, knobs
def test_mosesvae_exposes_the_reference_method_surface():
    """mosesvae.py:126-262: forward / forward_encoder / forward_decoder / sample / sample_z_prior / string2tensor / tensor2string."""
    vocab = V.OneHotVocab(set(chr(ord("a") + i) for i in range(26)))
    m = mv.mosesvae.VAE(vocab)
    for name in ("forward", "forward_encoder", "forward_decoder", "sample", "sample_z_prior", "string2tensor", "tensor2string"):
        assert callable(getattr(m, name)), name
    # the optimiser split of moses_train_distrib_logp.py:267-268 covers every parameter exactly once
    enc, dec = list(m.encoder.parameters()), list(m.decoder.parameters())
    assert len(enc) + len(dec) == len(list(m.parameters())) and not (set(map(id, enc)) & set(map(id, dec)))
    assert [n for n, _ in zip(*m._half_params("enc"))][0] == "x_emb.weight" and set(map(id, m._half_params("enc")[1])) == set(map(id, enc))
    assert set(map(id, m._half_params("dec")[1])) == set(map(id, dec)) | {id(m.x_emb.weight)}      # the embedding feeds the decoder GRU too
    seqs = [torch.tensor([vocab.bos, 1, 2, vocab.eos])]
    with pytest.raises(L.MvaeError):
        m.forward_encoder(seqs)                                   # no CPU fallback
    with pytest.raises(L.MvaeError):
        m.forward_decoder(seqs, torch.zeros(1, 160))
    with pytest.raises(ValueError):
        m.forward_decoder(seqs, torch.zeros(1, 10))


def test_linear_holders_are_real_nn_linear_for_type_keyed_hooks():
    """moses_train_distrib_logp.py:48-51: ``if type(m) == nn.Linear`` must find the layers; ``apply`` invalidates the packed shadows."""
    vocab = V.OneHotVocab(set(chr(ord("a") + i) for i in range(26)))
    m = mv.mosesvae.VAE(vocab)
    lin = [x for x in m.modules() if type(x) == torch.nn.Linear]
    assert len(lin) == 6
    with pytest.raises(RuntimeError):
        lin[0](torch.zeros(1, lin[0].in_features))
    torch.manual_seed(5); a = mv.mosesvae.VAE(vocab).state_dict()
    import copy
    c = copy.deepcopy(m)
    assert [type(x) for x in c.modules()] == [type(x) for x in m.modules()]
    with pytest.raises(RuntimeError):
        [x for x in c.modules() if type(x) == torch.nn.Linear][0](torch.zeros(1, 256))
    e0 = L.PARAM_EPOCH[0]
    m.apply(lambda mod: None)
    assert L.PARAM_EPOCH[0] > e0
    big = mv.MolecularVAE(i=24, o=16, c=12)
    assert len([x for x in big.modules() if type(x) == torch.nn.Linear]) == 5
    e0 = L.PARAM_EPOCH[0]
    big.apply(lambda mod: None)
    assert L.PARAM_EPOCH[0] > e0


def test_schedule_knobs_are_ignored_without_the_tuning_switch(monkeypatch):
    lib = L.load()
    monkeypatch.setenv("MVAE_BM", "64")
    monkeypatch.delenv("MVAE_TUNING", raising=False)
    assert lib.mvae_knob_int(b"MVAE_BM", 7) == 7 and L.knob("MVAE_MOSES_FORK", "1") == "1"
    monkeypatch.setenv("MVAE_MOSES_FORK", "0")
    assert L.knob("MVAE_MOSES_FORK", "1") == "1"
    monkeypatch.setenv("MVAE_TUNING", "1")
    assert lib.mvae_knob_int(b"MVAE_BM", 7) == 64 and L.knob("MVAE_MOSES_FORK", "1") == "0"
    assert lib.mvae_knob_int(b"MVAE_NOT_SET", 3) == 3


def test_sharded_fused_adam_refuses_a_stale_state_dict():
    p = torch.nn.Parameter(torch.zeros(8))
    opt = mv.FusedAdam([p], lr=1e-3)
    opt.state_dict()                                               # unsharded: always allowed
    opt.shard, opt._moments_stale = True, True                     # what step() leaves behind in the sharded form
    with pytest.raises(L.MvaeError):
        opt.state_dict()
    with pytest.raises(L.MvaeError):
        mv.save_checkpoint(io.BytesIO(), torch.nn.Linear(2, 2), opt, 0, ["a"], 4)
    opt._moments_stale = False                                      # gather_state() clears it
    opt.state_dict()
