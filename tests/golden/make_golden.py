#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference on disk).  It imports the
reference's ``models.py`` / ``mosesvae.py`` / ``vocab.py`` as they lie there, loads
weights from the build-owned seeded initialiser (oracle/initparams.py), feeds
seeded inputs and records inputs + expected outputs.  Only DATA is written here
(.npz); no reference source text is copied.

    python tests/golden/make_golden.py            # writes g1..g5 *.npz next to this file

Fixtures:
  g1_small.npz   reduced dims, B=3: every stage, recon/mu/logvar, loss, all parameter grads
  g2_full.npz    full dims (L=120,C=35,o=292), B=4: seed + loss, mu, logvar, recon rows, grad norms/slices
  g3_moses.npz   mosesvae.VAE, 30-symbol OneHotVocab, B=6 varlen, dropout off: kl, recon, mu, logvar, y, grads
  g4_traj.npz    5 steps of clip(3.0)+Adam(8e-4) on the g1 model: loss per step, final checksum
  g5_vocab.npz   CharVocab round trips + collate ordering on a fixed SMILES list
  g7_models2d.npz     models2d.VAE (conv+ReLU encoder, GRU(2->501,3) decoder over the one-hot block), B=3, train + eval mode
  g6_moses_train.npz  g3's model in train() mode: the inter-layer dropout masks torch drew (reconstructed), kl, recon, y, grads
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import models as ref_models          # noqa: E402  (reference, read-only)
import mosesvae as ref_moses         # noqa: E402
import vocab as ref_vocab            # noqa: E402
from oracle import initparams as ip  # noqa: E402

torch.set_num_threads(8)


def ref_loss_function(recon_x, x, mu, logvar, max_len):
    # train.py:31-38 (train.py itself cannot be imported: comet_ml + hard-coded paths at import)
    recon_x = recon_x.contiguous().view(-1)
    x = x.contiguous().view(-1)
    bce = torch.nn.BCELoss(reduction="mean")
    xent_loss = max_len * bce(recon_x, x)
    kl_loss = -0.5 * torch.mean(1. + mu - logvar ** 2. - torch.exp(mu))
    return xent_loss + kl_loss


def load(module, params, prefix=""):
    sd = {k[len(prefix):]: torch.from_numpy(v.astype(np.float64)) for k, v in params.items()
          if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=True)
    return missing


def run_molvae(enc, dec, params, idx, seed_eps, max_len, dtype=torch.float64):
    enc = enc.to(dtype); dec = dec.to(dtype)
    load(enc, params, "encoder."); load(dec, params, "decoder.")
    B = idx.shape[0]
    o = enc.lmbd.z_mean.out_features
    torch.manual_seed(seed_eps)
    eps = (1e-2 * torch.randn(B, o).to(dtype)).numpy().astype(np.float64)  # what Lambda draws: scale * randn.type_as(log_v), models.py:92-93
    torch.manual_seed(seed_eps)
    tidx = torch.from_numpy(idx)
    C = enc.embedding.num_embeddings
    stages = {}
    # stage taps (forward hooks on the reference's own submodules)
    hooks = []
    def tap(name):
        def fn(mod, inp, out):
            out = out[0] if isinstance(out, tuple) else out
            stages[name] = out.detach().numpy().copy()
        return fn
    hooks.append(enc.embedding.register_forward_hook(tap("enc_emb")))
    hooks.append(enc.gru.register_forward_hook(tap("enc_lstm_out")))
    hooks.append(enc.conv_1.register_forward_hook(tap("enc_conv1")))
    hooks.append(enc.conv_2.register_forward_hook(tap("enc_conv2")))
    hooks.append(enc.conv_3.register_forward_hook(tap("enc_conv3")))
    hooks.append(enc.dense_1.register_forward_hook(tap("enc_dense")))
    hooks.append(dec.latent_input.register_forward_hook(tap("dec_latent")))
    hooks.append(dec.gru.register_forward_hook(tap("dec_lstm_out")))
    z, mu, logvar = enc(tidx)
    recon = dec(z)
    for h in hooks:
        h.remove()
    ohe = torch.nn.functional.one_hot(tidx, C).to(dtype)
    loss = ref_loss_function(recon, ohe, mu, logvar, max_len)
    for m in (enc, dec):
        m.zero_grad()
    loss.backward()
    grads = {"encoder." + k: v.grad.numpy().copy() for k, v in enc.named_parameters()}
    grads.update({"decoder." + k: v.grad.numpy().copy() for k, v in dec.named_parameters()})
    return dict(eps=eps, z=z.detach().numpy(), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
                recon=recon.detach().numpy(), loss=float(loss), grads=grads, stages=stages)


# ------------------------------------------------------------------ G1
G1 = dict(i=24, o=16, c=12, emb=30, h_enc=56, n_enc=2, h_dec=32, n_dec=2, B=3, seed=101, gain=2.0)


def make_g1():
    g = G1
    shapes = ip.molvae_shapes(g["i"], g["o"], g["c"], g["emb"], g["h_enc"], g["n_enc"], g["h_dec"], g["n_dec"])
    params = ip.init_params(shapes, g["seed"], g["gain"], np.float64)
    enc = ref_models.MolEncoder(i=g["i"], o=g["o"], c=g["c"], h_size=g["h_enc"], num_lstm=g["n_enc"])
    dec = ref_models.MolDecoder(i=g["o"], o=g["i"], c=g["c"], num_gru=g["n_dec"], h_size=g["h_dec"])
    idx = ip.seeded_indices(g["seed"], g["B"], g["i"], g["c"])
    r = run_molvae(enc, dec, params, idx, 555, max_len=g["i"])
    out = dict(idx=idx, eps=r["eps"], z=r["z"], mu=r["mu"], logvar=r["logvar"], recon=r["recon"],
               loss=np.float64(r["loss"]))
    out.update({"stage." + k: v for k, v in r["stages"].items()})
    out.update({"grad." + k: v.astype(np.float32) for k, v in r["grads"].items()})   # f32 storage: <1e-7 rel
    np.savez_compressed(os.path.join(HERE, "g1_small.npz"), **out)
    print("g1 loss", r["loss"])
    return params, enc, dec, idx


# ------------------------------------------------------------------ G2
G2 = dict(B=4, seed=202, gain=1.5)


def make_g2():
    shapes = ip.molvae_shapes()
    params = ip.init_params(shapes, G2["seed"], G2["gain"], np.float32)
    m = ref_models.MolecularVAE()          # i=120,o=292,c=35   models.py:98
    idx = ip.seeded_indices(G2["seed"], G2["B"], 120, 35)
    r = run_molvae(m.encoder, m.decoder, params, idx, 777, max_len=120)
    out = dict(idx=idx, eps=r["eps"], mu=r["mu"], logvar=r["logvar"], z=r["z"], loss=np.float64(r["loss"]),
               recon_rows=r["recon"][:, ::17, :], recon_sum=np.float64(r["recon"].sum()),
               recon_sq=np.float64((r["recon"] ** 2).sum()))
    for k, gr in r["grads"].items():
        out["gnorm." + k] = np.float64(np.sqrt((gr ** 2).sum()))
        out["gslice." + k] = gr.reshape(-1)[:: max(1, gr.size // 64)][:64].copy()
    np.savez_compressed(os.path.join(HERE, "g2_full.npz"), **out)
    print("g2 loss", r["loss"])


# ------------------------------------------------------------------ G3
def make_g3():
    chars = [chr(ord('a') + i) for i in range(26)]
    v = ref_vocab.OneHotVocab(chars)                    # 26 + 4 specials = 30 symbols
    V = len(v)
    model = ref_moses.VAE(v).double()
    params = ip.init_params(ip.moses_shapes(V), 303, 1.5, np.float64)
    sd = model.state_dict()
    for k in sd:                                         # 88 aliased keys over 29 tensors
        base = k
        for pre in ("vae.0.", "vae.1.0.", "encoder.0."):
            if k.startswith(pre):
                base = "x_emb." + k[len(pre):]
        for a, b in (("encoder.1.", "encoder_rnn."), ("encoder.2.", "q_mu."), ("encoder.3.", "q_logvar."),
                     ("decoder.0.", "decoder_rnn."), ("decoder.1.", "decoder_lat."), ("decoder.2.", "decoder_fc."),
                     ("vae.1.1.", "encoder_rnn."), ("vae.1.2.", "q_mu."), ("vae.1.3.", "q_logvar."),
                     ("vae.2.0.", "decoder_rnn."), ("vae.2.1.", "decoder_lat."), ("vae.2.2.", "decoder_fc.")):
            if k.startswith(a):
                base = b + k[len(a):]
        sd[k] = torch.from_numpy(params[base])
    model.load_state_dict(sd)
    model.eval()                                         # dropout off (decoder_rnn dropout=0.2 in train mode)
    rs = np.random.RandomState(31)
    lens = sorted(rs.randint(4, 14, size=6).tolist(), reverse=True)
    strings = ["".join(chars[j] for j in rs.randint(0, 26, size=n)) for n in lens]
    seqs = [model.string2tensor(s, device="cpu") for s in strings]
    out = dict(lens=np.array([len(s) for s in seqs]), pad=np.int64(v.pad), bos=np.int64(v.bos), eos=np.int64(v.eos),
               unk=np.int64(v.unk), V=np.int64(V))
    for b, s in enumerate(seqs):
        out[f"seq{b}"] = s.numpy()
    out["state_dict_keys"] = np.array(sorted(model.state_dict().keys()))
    torch.manual_seed(999)
    eps = torch.randn(6, 160, dtype=torch.float64).numpy()
    torch.manual_seed(999)
    kl, recon, z, logvar, x, y = model(seqs)
    out.update(eps=eps, kl=np.float64(kl.item()), recon=np.float64(recon.item()), z=z.detach().numpy(),
               logvar=logvar.detach().numpy(), x=x.numpy(), y=y.detach().numpy().astype(np.float32))
    kl_w = 0.37
    model.zero_grad()
    (kl_w * kl + recon).backward()
    seen = {}
    for k, p_ in model.named_parameters():              # named_parameters de-duplicates aliases
        seen[k] = p_.grad.numpy().copy()
    out["kl_w"] = np.float64(kl_w)
    out["grad_names"] = np.array(sorted(seen.keys()))
    for k, gval in seen.items():                         # norms + strided slices (full grads would be 36 MB)
        out["gnorm." + k] = np.float64(np.sqrt((gval ** 2).sum()))
        out["gslice." + k] = gval.reshape(-1)[:: max(1, gval.size // 64)][:64].copy()
    np.savez_compressed(os.path.join(HERE, "g3_moses.npz"), **out)
    print("g3 kl", kl.item(), "recon", recon.item(), "params", sorted(seen.keys()))


# ------------------------------------------------------------------ G6: mosesvae.VAE in TRAIN mode (decoder_rnn dropout 0.2 active)
def make_g6():
    """Same model / weights / sequences as g3, model.train().  torch draws the inter-layer dropout noise inside _VF.gru from the CPU
    default generator; the same draws are repeated here (manual_seed -> randn_like(mu) -> bernoulli_(0.8) on the packed layer outputs
    of decoder layers 0 and 1) and unpacked to [layer][T][B][H] keep masks.  The numpy oracle fed with these masks must reproduce the
    reference's train-mode kl / recon / gradients -- that check runs here (a wrong reconstruction cannot produce a fixture)."""
    from oracle import np_oracle as O
    chars = [chr(ord('a') + i) for i in range(26)]
    v = ref_vocab.OneHotVocab(chars)
    V = len(v)
    model = ref_moses.VAE(v).double()
    params = ip.init_params(ip.moses_shapes(V), 303, 1.5, np.float64)
    sd = model.state_dict()
    for k in sd:
        base = k
        for pre in ("vae.0.", "vae.1.0.", "encoder.0."):
            if k.startswith(pre):
                base = "x_emb." + k[len(pre):]
        for a, b in (("encoder.1.", "encoder_rnn."), ("encoder.2.", "q_mu."), ("encoder.3.", "q_logvar."),
                     ("decoder.0.", "decoder_rnn."), ("decoder.1.", "decoder_lat."), ("decoder.2.", "decoder_fc."),
                     ("vae.1.1.", "encoder_rnn."), ("vae.1.2.", "q_mu."), ("vae.1.3.", "q_logvar."),
                     ("vae.2.0.", "decoder_rnn."), ("vae.2.1.", "decoder_lat."), ("vae.2.2.", "decoder_fc.")):
            if k.startswith(a):
                base = b + k[len(a):]
        sd[k] = torch.from_numpy(params[base])
    model.load_state_dict(sd)
    model.train()                                        # decoder_rnn: dropout=0.2 between its 3 layers (mosesvae.py:73-79)
    rs = np.random.RandomState(31)
    lens = sorted(rs.randint(4, 14, size=6).tolist(), reverse=True)
    strings = ["".join(chars[j] for j in rs.randint(0, 26, size=n)) for n in lens]
    seqs = [model.string2tensor(s, device="cpu") for s in strings]
    B, Hd, p_drop = len(seqs), 512, 0.2
    SEED = 4242
    torch.manual_seed(SEED)
    kl, recon, z, logvar, x, y = model(seqs)
    kl_w = 0.37
    model.zero_grad()
    (kl_w * kl + recon).backward()
    # ---- repeat the generator draws
    torch.manual_seed(SEED)
    eps = torch.randn(B, 160, dtype=torch.float64)
    slens = [len(s) for s in seqs]
    T = max(slens)
    n_packed = sum(slens)
    masks = np.zeros((2, T, B, Hd), np.uint8)
    for l in range(2):
        noise = torch.empty(n_packed, Hd, dtype=torch.float64).bernoulli_(1 - p_drop).numpy()
        r = 0
        for t in range(T):
            for b in range(B):
                if slens[b] > t:
                    masks[l, t, b] = noise[r]; r += 1
        assert r == n_packed
    # ---- the oracle with these masks must land on the reference's numbers
    res = O.moses_forward(params, [s.numpy() for s in seqs], eps.numpy(), v.pad, drop_masks=masks, drop_p=p_drop)
    assert abs(res["kl"] - kl.item()) < 1e-10 * abs(kl.item()), (res["kl"], kl.item())
    assert abs(res["recon"] - recon.item()) < 1e-10 * abs(recon.item()), (res["recon"], recon.item())
    og = res["grads_for"](kl_w)
    out = dict(lens=np.array(slens), pad=np.int64(v.pad), V=np.int64(V), eps=eps.numpy(), masks=masks, drop_p=np.float64(p_drop),
               kl=np.float64(kl.item()), recon=np.float64(recon.item()), y=y.detach().numpy().astype(np.float32), kl_w=np.float64(kl_w))
    for b, s_ in enumerate(seqs):
        out[f"seq{b}"] = s_.numpy()
    worst = 0.0
    for k, p_ in model.named_parameters():
        gval = p_.grad.numpy()
        worst = max(worst, float(np.abs(og[k] - gval).max() / (np.abs(gval).max() + 1e-300)))
        out["gnorm." + k] = np.float64(np.sqrt((gval ** 2).sum()))
        out["gslice." + k] = gval.reshape(-1)[:: max(1, gval.size // 64)][:64].copy()
    assert worst < 1e-9, worst
    np.savez_compressed(os.path.join(HERE, "g6_moses_train.npz"), **out)
    print("g6 (train mode) kl", kl.item(), "recon", recon.item(), "keep fraction", masks[:, :, :, :].mean(), "oracle grad err", worst)


# ------------------------------------------------------------------ G7: models2d.VAE (conv + ReLU encoder, GRU over the one-hot block)
def make_g7():
    """models2d.py:8-52 imported as it lies (it has no loss of its own: train.py:31-38's loss_function on its outputs).  B=3, train
    mode with the recorded randn_like draw, and eval mode (z = mu)."""
    import models2d as ref_m2d                          # noqa: E402  (reference, read-only)
    from oracle import np_oracle as O
    params = ip.init_params(ip.models2d_shapes(), 404, 2.0, np.float64)
    model = ref_m2d.VAE().double()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    idx = ip.seeded_indices(404, 3, 120, 35)
    x = torch.nn.functional.one_hot(torch.from_numpy(idx), 35).double()
    model.train()
    torch.manual_seed(31337)
    eps = torch.randn(3, 2, dtype=torch.float64).numpy()       # models2d.py:34 randn_like(std): the first draw of the forward
    torch.manual_seed(31337)
    recon, mu, logvar = model(x)
    loss = ref_loss_function(recon, x, mu, logvar, 120)
    model.zero_grad(); loss.backward()
    out = dict(idx=idx, eps=eps, mu=mu.detach().numpy(), logvar=logvar.detach().numpy(), recon=recon.detach().numpy().astype(np.float32),
               loss=np.float64(loss.item()))
    ref = O.models2d_loss_and_grads(params, x.numpy(), eps, 120, train=True)
    assert abs(ref["loss"] - loss.item()) < 1e-10 * abs(loss.item()), (ref["loss"], loss.item())
    worst = 0.0
    for k, p_ in model.named_parameters():
        gval = p_.grad.numpy()
        worst = max(worst, float(np.abs(ref["grads"][k] - gval).max() / (np.abs(gval).max() + 1e-300)))
        out["gnorm." + k] = np.float64(np.sqrt((gval ** 2).sum()))
        out["gslice." + k] = gval.reshape(-1)[:: max(1, gval.size // 64)][:64].copy()
    assert worst < 1e-8, worst
    model.eval()
    r2, mu2, lv2 = model(x)
    out["eval_loss"] = np.float64(ref_loss_function(r2, x, mu2, lv2, 120).item())
    out["eval_recon_rows"] = r2.detach().numpy()[:, ::17, :].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g7_models2d.npz"), **out)
    print("g7 loss", loss.item(), "eval loss", float(out["eval_loss"]), "oracle grad err", worst)


# ------------------------------------------------------------------ G4
def make_g4(params, enc, dec, idx):
    g = G1
    load(enc, params, "encoder."); load(dec, params, "decoder.")
    plist = list(enc.parameters()) + list(dec.parameters())
    opt = torch.optim.Adam(plist, lr=0.0008)             # train.py:81
    tidx = torch.from_numpy(idx)
    ohe = torch.nn.functional.one_hot(tidx, g["c"]).double()
    losses, norms, eps_all = [], [], []
    for step in range(5):
        torch.manual_seed(1000 + step)
        eps_all.append((1e-2 * torch.randn(g["B"], g["o"]).double()).numpy())
        torch.manual_seed(1000 + step)
        opt.zero_grad()
        z, mu, logvar = enc(tidx)
        recon = dec(z)
        loss = ref_loss_function(recon, ohe, mu, logvar, g["i"])
        loss.backward()
        tn = torch.nn.utils.clip_grad_norm_(plist, 3.0)  # train.py:102
        opt.step()                                       # train.py:104
        losses.append(loss.item()); norms.append(float(tn))
    fin = {}
    for pre, mod in (("encoder.", enc), ("decoder.", dec)):   # norms + strided slices of the final parameters
        for k, v in mod.named_parameters():
            a = v.detach().numpy()
            fin["fnorm." + pre + k] = np.float64(np.sqrt((a ** 2).sum()))
            fin["fslice." + pre + k] = a.reshape(-1)[:: max(1, a.size // 64)][:64].copy()
    np.savez_compressed(os.path.join(HERE, "g4_traj.npz"), losses=np.array(losses), gnorms=np.array(norms),
                        eps=np.stack(eps_all), **fin)
    print("g4 losses", losses, "norms", norms)


# ------------------------------------------------------------------ G5
SMILES = ["CCO", "c1ccccc1", "CC(=O)Oc1ccccc1C(=O)O", "CN1CCC[C@H]1c2cccnc2", "O=C(O)c1ccccc1",
          "C1CC1", "N#Cc1ccc(Br)cc1", "CC(C)Cc1ccc(cc1)[C@@H](C)C(=O)O"]


def make_g5():
    v = ref_vocab.CharVocab.from_data(SMILES)
    syms = [v.i2c[i] for i in range(len(v))]
    ids = [np.array(v.string2ids(s, add_bos=True, add_eos=True)) for s in SMILES]
    back = [v.ids2string(list(i)) for i in ids]
    assert back == SMILES
    # collate ordering, moses_train_distrib.py:127-135: data.sort(key=len, reverse=True) (stable)
    data = list(SMILES)
    data.sort(key=len, reverse=True)
    out = dict(symbols=np.array(syms), order=np.array(data), unk_probe=np.int64(v.char2id("?")),
               bos=np.int64(v.bos), eos=np.int64(v.eos), pad=np.int64(v.pad), unk=np.int64(v.unk))
    for n, i in enumerate(ids):
        out[f"ids{n}"] = i
    np.savez_compressed(os.path.join(HERE, "g5_vocab.npz"), **out)
    print("g5 symbols", syms)


if __name__ == "__main__":
    if len(sys.argv) > 1:                                # e.g. `make_golden.py g6`: only the named fixtures
        for name in sys.argv[1:]:
            globals()["make_" + name]()
        sys.exit(0)
    params, enc, dec, idx = make_g1()
    make_g4(params, enc, dec, idx)
    make_g2()
    make_g3()
    make_g5()
    make_g6()
    make_g7()
