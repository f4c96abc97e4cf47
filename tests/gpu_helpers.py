"""Shared helpers for the GPU parity tests: run the HIP path and the numpy oracle on the same seeded inputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import np_oracle as O          # noqa: E402  (checker only)
from oracle import initparams as ip        # noqa: E402
import molecular_vae_amd as mv             # noqa: E402

G1 = dict(i=24, o=16, c=12, emb=30, h_enc=56, n_enc=2, h_dec=32, n_dec=2, B=3, seed=101, gain=2.0)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def build_modules(dims, params, dtype, dev="cuda"):
    enc = mv.MolEncoder(i=dims["i"], o=dims["o"], c=dims["c"], word_embedding_size=dims["emb"], h_size=dims["h_enc"],
                        num_lstm=dims["n_enc"])
    dec = mv.MolDecoder(i=dims["o"], o=dims["i"], c=dims["c"], num_gru=dims["n_dec"], h_size=dims["h_dec"], dtype=dtype)
    enc.load_state_dict({k[len("encoder."):]: torch.from_numpy(np.asarray(v, np.float32)) for k, v in params.items()
                         if k.startswith("encoder.")})
    dec.load_state_dict({k[len("decoder."):]: torch.from_numpy(np.asarray(v, np.float32)) for k, v in params.items()
                         if k.startswith("decoder.")})
    import weakref
    dec.__dict__["_peer"] = weakref.ref(enc)      # pair them as MolecularVAE does: exercises the side-stream weight-gradient fork
    return enc.to(dev), dec.to(dev)


def run_hip(enc, dec, idx, eps, max_len):
    dev = next(enc.parameters()).device
    tidx = torch.from_numpy(idx).to(dev)
    teps = torch.from_numpy(np.asarray(eps, np.float32)).to(dev)
    C = enc.embedding.num_embeddings
    for m in (enc, dec):
        m.zero_grad(set_to_none=True)
    z, mu, logvar = enc(tidx, teps)
    dec.__dict__["_z_from_peer"] = True           # as MolecularVAE.forward does: z is the paired encoder's output, so the fork is allowed
    recon = dec(z)
    dec.__dict__["_z_from_peer"] = False
    ohe = torch.nn.functional.one_hot(tidx, C).float()
    loss = mv.bce_kl_loss(recon, ohe, mu, logvar, max_len)
    loss.backward()
    torch.cuda.synchronize()
    grads = {"encoder." + k: p.grad.detach().cpu().numpy() for k, p in enc.named_parameters()}
    grads.update({"decoder." + k: p.grad.detach().cpu().numpy() for k, p in dec.named_parameters()})
    return dict(loss=float(loss.detach()), z=z.detach().cpu().numpy(), mu=mu.detach().cpu().numpy(),
                logvar=logvar.detach().cpu().numpy(), recon=recon.detach().cpu().numpy(), grads=grads)


def g1_dims_params(dt=np.float64):
    shapes = ip.molvae_shapes(G1["i"], G1["o"], G1["c"], G1["emb"], G1["h_enc"], G1["n_enc"], G1["h_dec"], G1["n_dec"])
    return ip.init_params(shapes, G1["seed"], G1["gain"], dt)


FULL = dict(i=120, o=292, c=35, emb=30, h_enc=72, n_enc=3, h_dec=1024, n_dec=4)


def grad_report(hip_grads, ref_grads):
    """per-parameter relative error (max-norm) -> dict"""
    return {k: rel(hip_grads[k], ref_grads[k]) for k in ref_grads}
