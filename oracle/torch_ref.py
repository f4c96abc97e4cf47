"""CPU baseline: the reference ARCHITECTURE executing on host cores through stock torch.nn primitives.

TEST / BENCH INFRASTRUCTURE ONLY (never imported by the product package).  This is a restatement, written from
SURVEY.md section 3.4 / 8a, of what the reference runs when `.cuda()` is dropped from train.py:77 -- the same
torch.nn.Embedding / LSTM / Conv1d / Linear / BCELoss / clip_grad_norm_ / Adam calls, arranged as
models.py:109-165 arranges them and stepped as train.py:94-104 steps them.  It is pinned to the same golden fixtures
as the numpy oracle (tests/test_oracle_golden.py::test_torch_port_matches_golden).  `bench.py` times it as
``cpu_baseline`` (kind "port"); the reference's own files never travel to the GPU box.
"""
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

ALPHA = 1.6732632423543772848170429916717
SCALE = 1.0507009873554804934193349852946


def _selu(x):
    return SCALE * F.elu(x, ALPHA)


class CpuPort(nn.Module):
    """Parameter names are the reference's state_dict keys, so fixtures / seeded weights load directly."""

    def __init__(self, i=120, o=292, c=35, emb=30, h_enc=72, n_enc=3, h_dec=1024, n_dec=4):
        super().__init__()
        self.i = i
        enc = nn.Module()
        enc.embedding = nn.Embedding(c, emb)
        enc.gru = nn.LSTM(emb, h_enc, n_enc, batch_first=True)
        enc.conv_1 = nn.Sequential(nn.Conv1d(i, 120, 18))
        enc.conv_2 = nn.Sequential(nn.Conv1d(120, 64, 18))
        enc.conv_3 = nn.Sequential(nn.Conv1d(64, 64, 18))
        enc.dense_1 = nn.Sequential(nn.Linear((h_enc - 51) * 64, 512))
        enc.lmbd = nn.Module()
        enc.lmbd.z_mean = nn.Linear(512, o)
        enc.lmbd.z_log_var = nn.Linear(512, o)
        dec = nn.Module()
        dec.latent_input = nn.Sequential(nn.Linear(o, o))
        dec.gru = nn.LSTM(o, h_dec, n_dec, batch_first=True)
        dec.decoded_mean = nn.Module()
        dec.decoded_mean.module = nn.Sequential(nn.Linear(h_dec, c))
        self.encoder, self.decoder = enc, dec

    def forward(self, idx, eps):
        e, d = self.encoder, self.decoder
        h, _ = e.gru(e.embedding(idx))                       # [B, L, h]; L plays the conv-channel role below
        h = _selu(e.conv_3(_selu(e.conv_2(_selu(e.conv_1(h))))))
        h = _selu(e.dense_1(h.flatten(1)))
        mu, logvar = e.lmbd.z_mean(h), e.lmbd.z_log_var(h)
        z = mu + torch.exp(logvar / 2.) * eps
        y = _selu(d.latent_input(z)).unsqueeze(1).expand(-1, self.i, -1)
        y, _ = d.gru(y)
        B, L, _ = y.shape
        p = torch.softmax(d.decoded_mean.module(y.reshape(B * L, -1)), dim=1).view(B, L, -1)
        return p, mu, logvar


def elbo(recon, onehot, mu, logvar, max_len):
    xent = max_len * F.binary_cross_entropy(recon.reshape(-1), onehot.reshape(-1), reduction="mean")
    kl = -0.5 * torch.mean(1. + mu - logvar ** 2. - torch.exp(mu))     # as train.py:36-37 computes it
    return xent + kl


def time_cpu_training(batch=32, steps=6, warmup=2, threads=None, seed=42, L=120, C=35, log=None, budget_s=60.0):
    """Times `steps` train.py-style steps (fwd + loss + bwd + clip 3.0 + Adam 8e-4) on host cores."""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    model = CpuPort(i=L, c=C)
    opt = torch.optim.Adam(model.parameters(), lr=0.0008)
    g = torch.Generator().manual_seed(1234)
    idx = torch.randint(0, C, (batch, L), generator=g)
    onehot = F.one_hot(idx, C).float()
    losses, t0 = [], None
    for s in range(warmup + steps):
        if s == warmup:
            t0 = time.perf_counter()
        eps = 1e-2 * torch.randn(batch, 292)
        opt.zero_grad()
        recon, mu, logvar = model(idx, eps)
        loss = elbo(recon, onehot, mu, logvar, L)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0)
        opt.step()
        losses.append(loss.item())
        if log:
            log(f"cpu step {s}: loss {losses[-1]:.4f}")
        if t0 is not None and time.perf_counter() - t0 > budget_s and s + 1 < warmup + steps:
            steps = s + 1 - warmup          # bounded sample: stop early on a slow host
            break
    dt = time.perf_counter() - t0
    return dict(molecules_per_s=batch * steps / dt, s_per_step=dt / steps, threads=torch.get_num_threads(),
                batch=batch, steps=steps, losses=losses)
