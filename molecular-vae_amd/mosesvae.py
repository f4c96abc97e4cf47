"""Host-side mirror of ``mosesvae.VAE`` (mosesvae.py:27-199): GRU encoder / 3-layer GRU decoder character VAE.

Same constructor (``VAE(vocab)``), hard-coded hyper-parameters (mosesvae.py:31-40), attribute names, aliasing ``ModuleList``s
(88 ``state_dict`` keys over 29 tensors) and ``forward`` 6-tuple ``(kl_loss, recon_loss, z, logvar, x, y)`` as the reference.
The whole forward is one ``torch.autograd.Function`` over the C ABI: embedding folded into the layer-0 GRU input projection
(a table gather), wavefront GRU kernels with per-sequence length masking (== pack_sequence / pad_packed_sequence semantics:
a finished sequence keeps its state and emits zeros), fused heads + reparameterisation + KL, fused token cross-entropy.

Train mode (``model.train()``): the decoder GRU's inter-layer dropout (``d_dropout = 0.2``, mosesvae.py:73-79) runs inside the step
kernels -- layer l+1 reads ``h^l * keep / (1 - p)`` -- with the keep mask generated on device by a counter-based hash from an explicit
seed (``forward(..., drop_seed=)``; by default drawn from torch's CPU generator, so ``torch.manual_seed`` makes a run reproducible) or
injected (``forward(..., drop_mask=)``: uint8 ``[layers-1, T, B, H]``; the parity tests inject the mask the reference drew).  ``eval()``
computes the deterministic network.  Under data parallelism (an initialised ``torch.distributed`` group) a TRAINING forward (train mode,
gradients enabled) takes the token mean of the reconstruction loss (mosesvae.py:193-197) over the GLOBAL number of non-pad targets
(``dp_global_token_mean``, one 4-byte all-reduce on ``dp_group``), so that the all-reduced gradient equals the single-process gradient of
the global batch; eval / no_grad forwards never communicate.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import ops
from .models import (LinearWeights, EmbeddingWeights, RNNWeights, _Workspace, _SavedState, _params_key, _pad, _require_cuda, _LDPAD, _dyk,
                     _apply_and_mark)
from .vocab import PaddedBatch, pad_batch


class ReLU(nn.Module):
    """Marker (fused into the producing GEMM's epilogue)."""


def _slots4(w3, H, order):
    """[3H, K] gate rows (r, z, n) -> [4H, K] slot rows; order 'x' = (r, z, n, 0) for W_ih, 'h' = (r, z, 0, n) for W_hh."""
    out = torch.zeros(4 * H, *w3.shape[1:], dtype=w3.dtype, device=w3.device)
    out[:2 * H] = w3[:2 * H]
    if order == "x":
        out[2 * H:3 * H] = w3[2 * H:]
    else:
        out[3 * H:] = w3[2 * H:]
    return out


class VAE(nn.Module, _SavedState):
    def __init__(self, vocab, dtype=torch.bfloat16):
        super().__init__()
        q_d_h, q_n_layers, d_n_layers, d_dropout, d_z, d_d_h = 256, 1, 3, 0.2, 160, 512     # mosesvae.py:31-40
        self.vocabulary = vocab
        for ss in ("bos", "eos", "unk", "pad"):
            setattr(self, ss, getattr(vocab, ss))
        n_vocab, d_emb = len(vocab), vocab.vectors.size(1)
        if d_emb != n_vocab or vocab.vectors.size(0) != n_vocab:
            # the kernels fold the (identity-initialised, trainable) embedding into the layer-0 projections as a [V, V] table
            raise ValueError(f"mosesvae.VAE needs a one-hot vocabulary (vocab.vectors [{n_vocab}, {n_vocab}], mosesvae.py:48-50); "
                             f"got {tuple(vocab.vectors.shape)}")
        self.x_emb = EmbeddingWeights(n_vocab, d_emb)
        self.x_emb.padding_idx = self.pad
        self.x_emb.weight.data.copy_(vocab.vectors)                 # mosesvae.py:50 (overwrites the zeroed pad row too)
        self.encoder_rnn = RNNWeights("GRU", d_emb, q_d_h, q_n_layers)
        self.q_mu = nn.Sequential(LinearWeights(q_d_h, 256), ReLU(), LinearWeights(256, d_z))
        self.q_logvar = nn.Sequential(LinearWeights(q_d_h, 256), ReLU(), LinearWeights(256, d_z))
        self.decoder_rnn = RNNWeights("GRU", d_emb + d_z, d_d_h, d_n_layers)
        self.decoder_lat = LinearWeights(d_z, d_d_h)
        self.decoder_fc = LinearWeights(d_d_h, n_vocab)
        self.encoder = nn.ModuleList([self.x_emb, self.encoder_rnn, self.q_mu, self.q_logvar])
        self.decoder = nn.ModuleList([self.decoder_rnn, self.decoder_lat, self.decoder_fc])
        self.vae = nn.ModuleList([self.x_emb, self.encoder, self.decoder])
        self.d_z, self.d_dropout = d_z, d_dropout
        self.dp_global_token_mean = True     # DP: normalise the CE by the global non-pad token count (see module docstring)
        self.dp_force = False                # GradSync(force=True): make the token-count all-reduce even on a one-rank group (RCCL rehearsal)
        self.dp_group = None                 # process group of that reduction (None: the default group); moses_train_step sets it from the optimiser's GradSync
        self.last_drop_seed = None           # seed of the most recent train-mode forward (None: eval / injected mask)
        self.compute_dtype = dtype
        self.prior = "zeros"                 # sample_z_prior: "zeros" = the reference as written (mosesvae.py:211), "normal" = N(0, I) as its docstring says
        self.noise = "device"                # reparameterisation noise: "device" = drawn by the library inside the latent launch; "torch" = torch.randn on the device generator
        self.__dict__["noise_stream"] = ops.NoiseStream()
        self._init_saved_state()
        self._pack_key, self._packed = None, {}
        self.__dict__["_side"] = None

    apply = _apply_and_mark

    def _side_stream(self, dev):
        return ops.side_stream(dev)

    @property
    def device(self):
        return next(self.parameters()).device

    def string2tensor(self, string, device="model"):
        ids = self.vocabulary.string2ids(string, add_bos=True, add_eos=True)
        return torch.tensor(ids, dtype=torch.long, device=self.device if device == "model" else device)

    def tensor2string(self, tensor):
        return self.vocabulary.ids2string(tensor.tolist(), rem_bos=True, rem_eos=True)

    # -- unique parameters in a fixed order (parameters() already de-duplicates the aliases)
    def _plist(self):
        return list(self.parameters())

    def _batch(self, x):
        """The reference's list of per-sequence LongTensors (sorted by length descending, as collate() yields them) or an already collated
        PaddedBatch -> (x_pad [B, T] int64, lengths [B] int32) on the model's device, one transfer."""
        dev = self.device
        if isinstance(x, PaddedBatch):                                # already collated into the kernels' layout (vocab.get_padded_collate_fn)
            return x.x_pad.to(dev, non_blocking=True), x.lengths.to(dev, non_blocking=True).to(torch.int32)
        b = pad_batch(list(x), self.pad)                              # padded where the tensors live (device tensors stay on the device)
        return b.x_pad.to(dev, non_blocking=True), b.lengths.to(dev, non_blocking=True)

    def _eps(self, B, eps):
        """The reparameterisation noise handed to the encoder node: the caller's tensor; None under noise == "device" (the library draws it
        inside the latent launch from self.noise_stream: an explicit seed + element counter, ops.NoiseStream); torch's device generator under
        noise == "torch" (mosesvae.py:159 as written: randn_like on the CUDA generator)."""
        if eps is not None:
            return eps.contiguous().float()
        if self.noise == "device":
            return None
        return torch.randn(B, self.d_z, device=self.device)

    def seed_noise(self, seed, counter=0):
        self.noise_stream.reseed(seed, counter)

    def _draw_drop(self, T, B, drop_mask, drop_seed):
        """Train mode: the decoder GRU's inter-layer dropout draw (nn.GRU(dropout=d_dropout if d_n_layers > 1 else 0)) as (p, seed, mask | None)."""
        NL = self.decoder_rnn.num_layers
        if not (self.training and self.d_dropout > 0 and NL > 1):
            return None
        if drop_mask is not None:
            m = torch.as_tensor(drop_mask).to(torch.uint8)
            if tuple(m.shape) != (NL - 1, T, B, self.decoder_rnn.hidden_size):
                raise ValueError(f"drop_mask must be [layers-1, T, B, H] = {(NL - 1, T, B, self.decoder_rnn.hidden_size)}, got {tuple(m.shape)}")
            self.last_drop_seed = None
            return (float(self.d_dropout), 0, m.contiguous().to(self.device))
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,))) if drop_seed is None else int(drop_seed)
        self.last_drop_seed = seed
        return (float(self.d_dropout), seed, None)

    def _dp_token_mean(self, recon, ntok):
        # Data-parallel TRAINING steps only (train mode, gradients enabled): every rank of `dp_group` makes this call once per step.  Evaluation,
        # no_grad encoding and rank-0-only validation stay collective-free (a collective only some ranks reach would deadlock).
        if (self.dp_global_token_mean and self.training and torch.is_grad_enabled() and dist.is_available() and dist.is_initialized()
                and (dist.get_world_size(self.dp_group) > 1 or self.dp_force)):
            # local mean = num_r / cnt_r; the gradient all-reduce averages over ranks, so scale to  num_r * world / sum_r cnt_r
            tot = ntok.detach().clone()
            dist.all_reduce(tot, group=self.dp_group)
            recon = recon * (ntok.detach() * dist.get_world_size(self.dp_group) / tot)
        return recon

    def _half_params(self, half):
        """(names, parameters) one half's autograd node owns: forward_encoder -> x_emb, encoder_rnn, q_mu, q_logvar; forward_decoder -> x_emb,
        decoder_rnn, decoder_lat, decoder_fc (the embedding feeds both GRUs)."""
        pre = ("x_emb.", "encoder_rnn.", "q_mu.", "q_logvar.") if half == "enc" else ("x_emb.", "decoder_rnn.", "decoder_lat.", "decoder_fc.")
        sel = [(n, p) for n, p in self.named_parameters() if n.startswith(pre)]
        return [n for n, _ in sel], [p for _, p in sel]

    def forward(self, x, eps=None, drop_mask=None, drop_seed=None):
        """x: list of LongTensors (one per sequence, sorted by length descending, as collate() yields them).
        Returns (kl_loss, recon_loss, z, logvar, x_padded, y) -- mosesvae.py:126-140 -- from ONE fused autograd node (the encoder and decoder
        halves of forward_encoder / forward_decoder, with the decoder's parameter gradients on a side stream beside the encoder's backward).
        eps / drop_mask / drop_seed inject the reparameterisation noise and the train-mode inter-layer dropout draw (parity tests)."""
        x_pad, len_t = self._batch(x)
        B, T = x_pad.shape
        eps = self._eps(B, eps)                                        # None: mosesvae.py:159's randn_like(mu) is drawn inside mvae_moses_latent_fwd
        drop = self._draw_drop(T, B, drop_mask, drop_seed)
        if x_pad.is_cuda and torch.is_grad_enabled():
            self._side_stream(x_pad.device)                            # first use probes for a concurrent stream: here, not inside the backward
        kl, recon, z, logvar, y, ntok = _MosesFn.apply(self, x_pad, len_t, eps, drop, *self._plist())
        return kl, self._dp_token_mean(recon, ntok), z, logvar, x_pad, y

    def forward_encoder(self, x, eps=None):
        """mosesvae.py:142-164: x -> (z, kl_loss, logvar).  Runs the encoder half only (GRU(256), heads, reparameterisation + KL); differentiable
        w.r.t. x_emb / encoder_rnn / q_mu / q_logvar."""
        x_pad, len_t = self._batch(x)
        _, params = self._half_params("enc")
        return _MosesEncFn.apply(self, x_pad, len_t, self._eps(x_pad.shape[0], eps), *params)

    def forward_decoder(self, x, z, drop_mask=None, drop_seed=None):
        """mosesvae.py:166-199: teacher-forced decoder on the caller's latent z [B, d_z] -> (recon_loss, x_padded, y); differentiable w.r.t. z
        and x_emb / decoder_rnn / decoder_lat / decoder_fc.  Train mode applies the inter-layer dropout as `forward` does."""
        x_pad, len_t = self._batch(x)
        B, T = x_pad.shape
        if tuple(z.shape) != (B, self.d_z):
            raise ValueError(f"forward_decoder: z must be [{B}, {self.d_z}], got {tuple(z.shape)}")
        drop = self._draw_drop(T, B, drop_mask, drop_seed)
        _, params = self._half_params("dec")
        recon, y, ntok = _MosesDecFn.apply(self, x_pad, len_t, z.to(self.device), drop, *params)
        return self._dp_token_mean(recon, ntok), x_pad, y

    def sample_z_prior(self, n_batch, normal=None):
        """mosesvae.py:200-211.  The reference AS WRITTEN returns zeros (its randn line is commented out, :207-210) although its docstring says
        z ~ N(0, I); a drop-in keeps that: `self.prior == "zeros"` (the default).  `normal=True` / `self.prior = "normal"` gives the documented
        N(0, I) draw -- from the library's counter hash (noise == "device") or torch's device generator."""
        if not (self.prior == "normal" if normal is None else normal):
            return torch.zeros((n_batch, self.d_z), device=self.device)
        if self.noise == "device" and self.device.type == "cuda":
            seed, off = self.noise_stream.take(n_batch * self.d_z)
            return ops.normal_fill(torch.empty(n_batch, self.d_z, device=self.device), 1.0, seed, off)
        return torch.randn(n_batch, self.d_z, device=self.device)

    @torch.no_grad()
    def sample(self, n_batch, max_len=100, z=None, temp=1.0, return_tokens=False, seed=None):
        """mosesvae.py:214-262 (autoregressive decoding, multinomial sampling at temperature `temp`) on the GRU step kernels: per generated token
        one wavefront pass of the 3-layer stack (T = 1) and ONE sampling launch (head GEMV + softmax + multinomial + eos / end-pad bookkeeping +
        the next token's input rows: mvae_moses_sample_step) -- 4 launches per token, no torch arithmetic in the loop.  Randomness is explicit:
        `seed` (default: drawn from torch's CPU generator, so torch.manual_seed makes a run reproducible) feeds a counter hash of (step, row).
        Upstream bugs fixed: boolean masks, a real d_z.  Returns (list of strings, z)."""
        dev = self.device
        _require_cuda(dev, "mosesvae.VAE.sample")
        P = self._pack(dev)
        ws, dt, f32 = self._ws, self.compute_dtype, torch.float32
        if z is None:
            z = self.sample_z_prior(n_batch)
        z = z.to(dev).float().contiguous()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)))
        B, V, dz = n_batch, self.x_emb.num_embeddings, self.d_z
        pd = P["dec"]; Hd = pd["H"]; NL = self.decoder_rnn.num_layers
        Vp, ldh = _pad(V, 4), Hd + _LDPAD
        W = lambda name, shape, d=f32: ws.get("smp_" + name, shape, d, dev)
        h0 = W("h0", (B, Hd)); ops.gemm_nt(z, self.decoder_lat.weight, h0, B, Hd, dz, bias=self.decoder_lat.bias)
        tbl3 = W("tbl3", (V, 3 * Hd)); ops.gemm_nt(P["E_p"], pd["Wx_p"], tbl3, V, 3 * Hd, Vp)
        tbl4 = W("tbl4", (V, 4 * Hd)); tbl4[:, :3 * Hd].copy_(tbl3)
        zp3 = W("zp3", (B, 3 * Hd)); ops.gemm_nt(z, P["Wz"], zp3, B, 3 * Hd, dz)
        zp4 = W("zp4", (B, 4 * Hd)); zp4[:, :3 * Hd].copy_(zp3)
        hbuf = [W(f"h{l}", (2, B, ldh), dt) for l in range(NL)]        # [0] = previous state, [1] = new state (swapped every token)
        for l in range(NL):
            ops.cast_transpose(h0, B, Hd, dst=hbuf[l][0])
        gates = [W(f"gates{l}", (1, B, 4 * Hd), dt) for l in range(NL)]
        hstate = [W(f"hstate{l}", (2, B, Hd)) for l in range(NL)]
        add = W("add", (1, B, 4 * Hd))
        w = torch.full((B,), self.bos, dtype=torch.long, device=dev)
        x = torch.full((B, max_len), self.pad, dtype=torch.long, device=dev)
        x[:, 0] = self.bos
        end_pads = torch.full((B,), max_len, dtype=torch.long, device=dev)
        eos_mask = torch.zeros(B, dtype=torch.uint8, device=dev)
        ops.gather_rows_tb(w.view(B, 1), tbl4, add, B, 1, V, 4 * Hd, base=zp4)      # the <bos> input rows; later ones come out of the sampling launch
        cur = 0
        for i in range(1, max_len):
            ops.rnn_fwd(L.CELL_GRU, dt, 1, B, Hd, add, 0, pd["Wih"], [pd["ldw"]] * NL, pd["Whh"], [pd["ldw"]] * NL, pd["bias"],
                        [h[1 - cur:2 - cur] for h in hbuf], ldh, None, gates, hstate, h0=[h[cur] for h in hbuf], ldh0=ldh, persist=False)
            ops.moses_sample_step(hbuf[-1][1 - cur], ldh, P["Wfc"], self.decoder_fc.bias, temp, seed, i, self.eos, tbl4, zp4, add, x, end_pads,
                                  eos_mask, w, B, V, Hd)
            cur = 1 - cur
        xs, ends = x.cpu(), end_pads.cpu()
        if return_tokens:                                                # raw id tensors (specials included), for tests / downstream scoring
            return [xs[b, :ends[b]] for b in range(B)], z
        return [self.tensor2string(xs[b, :ends[b]]) for b in range(B)], z

    # -- packed shadows: ONE multi-tensor pack launch (ops.PackList) instead of ~60 few-microsecond launches per optimiser step
    def _pack(self, dev):
        params = self._plist()
        key = _params_key(params) + (self.compute_dtype,)
        if key == self._pack_key:
            return self._packed
        ptrs = (dev, self.compute_dtype) + tuple((id(p), p.data_ptr()) for p in params)     # id: a deepcopy must rebuild its own job table
        if self.__dict__.get("_pack_ptrs") != ptrs:
            with torch.no_grad():                      # the job table keeps plain (non-autograd) views of the parameters
                self._build_pack(dev)
            self.__dict__["_pack_ptrs"] = ptrs
        with torch.no_grad():
            self.__dict__["_pack_list"].run()
        self._pack_key = key
        return self._packed

    def _build_pack(self, dev):
        """Allocate the shadows (zeroed: the empty gate slot of every 4-slot matrix and all padding are never written) and record the jobs
        that fill them, each straight from a parameter (jobs of one launch must not depend on each other)."""
        ws, dt, f32 = self._ws, self.compute_dtype, torch.float32
        V = self.x_emb.num_embeddings
        Vp = _pad(V, 4)
        P, pl = {}, ops.PackList()
        E = self.x_emb.weight
        P["E_p"] = ws.get("E_p", (V, Vp), f32, dev); P["ET_p"] = ws.get("ET_p", (V, Vp), f32, dev)
        pl.cast_transpose(E, V, V, dst=P["E_p"], dstT=P["ET_p"])
        for name, rnn, nl in (("enc", self.encoder_rnn, 1), ("dec", self.decoder_rnn, self.decoder_rnn.num_layers)):
            H = rnn.hidden_size
            ldw, ldwT = H + _LDPAD, 4 * H + _LDPAD
            P[name] = dict(H=H, ldw=ldw, ldwT=ldwT, Whh=[], WhhT=[], Wih=[None], WihT=[None], bias=[])
            for l in range(nl):
                # gate rows (r, z, n) -> slots: W_hh (r, z, 0, n), W_ih (r, z, n, 0)
                whh = getattr(rnn, f"weight_hh_l{l}")
                w = ws.get(f"{name}_Whh{l}", (4 * H, ldw), dt, dev); wT = ws.get(f"{name}_WhhT{l}", (H, ldwT), dt, dev)
                pl.cast_transpose(whh[:2 * H], 2 * H, H, dst=w[:2 * H], dstT=wT[:, :2 * H])
                pl.cast_transpose(whh[2 * H:], H, H, dst=w[3 * H:], dstT=wT[:, 3 * H:4 * H])
                P[name]["Whh"].append(w); P[name]["WhhT"].append(wT)
                if l > 0:
                    wih = getattr(rnn, f"weight_ih_l{l}")
                    w = ws.get(f"{name}_Wih{l}", (4 * H, ldw), dt, dev); wT = ws.get(f"{name}_WihT{l}", (H, ldwT), dt, dev)
                    pl.cast_transpose(wih, 3 * H, H, dst=w[:3 * H], dstT=wT[:, :3 * H])
                    P[name]["Wih"].append(w); P[name]["WihT"].append(wT)
                bi, bh = getattr(rnn, f"bias_ih_l{l}"), getattr(rnn, f"bias_hh_l{l}")
                b4 = ws.get(f"{name}_bias{l}", (4 * H,), f32, dev)
                pl.add(bi[:2 * H], bh[:2 * H], b4[:2 * H]); pl.copy(bi[2 * H:], b4[2 * H:3 * H]); pl.copy(bh[2 * H:], b4[3 * H:])
                P[name]["bias"].append(b4)
            # layer-0 input weights: the embedding part becomes a table, the z part (decoder) a dense projection
            w0 = getattr(rnn, "weight_ih_l0")
            wx = ws.get(f"{name}_Wx_p", (3 * H, Vp), f32, dev); wxT = ws.get(f"{name}_WxT", (V, 3 * H), f32, dev)
            pl.cast_transpose(w0[:, :V], 3 * H, V, dst=wx, dstT=wxT)
            P[name]["Wx_p"], P[name]["WxT"] = wx, wxT
        dz = self.d_z
        Hd = P["dec"]["H"]
        w0 = self.decoder_rnn.weight_ih_l0
        P["Wz"] = ws.get("dec_Wz", (3 * Hd, dz), f32, dev); P["WzT"] = ws.get("dec_WzT", (dz, 3 * Hd), f32, dev)
        pl.cast_transpose(w0[:, V:], 3 * Hd, dz, dst=P["Wz"], dstT=P["WzT"])
        for n, seq in (("mu", self.q_mu), ("lv", self.q_logvar)):
            P[n + "_W0T"] = ws.get(n + "_W0T", (seq[0].in_features, seq[0].out_features), f32, dev)
            pl.cast_transpose(seq[0].weight, seq[0].out_features, seq[0].in_features, dstT=P[n + "_W0T"])
            P[n + "_W2T"] = ws.get(n + "_W2T", (seq[2].in_features, seq[2].out_features), f32, dev)
            pl.cast_transpose(seq[2].weight, seq[2].out_features, seq[2].in_features, dstT=P[n + "_W2T"])
        P["WlatT"] = ws.get("WlatT", (dz, Hd), f32, dev); pl.cast_transpose(self.decoder_lat.weight, Hd, dz, dstT=P["WlatT"])
        Vp8 = _pad(V, 8)
        P["Wfc"] = ws.get("Wfc", (V, Hd + _LDPAD), dt, dev); P["WfcT"] = ws.get("WfcT", (Hd, _dyk(V) if dt == torch.bfloat16 else Vp8), dt, dev)
        pl.cast_transpose(self.decoder_fc.weight, V, Hd, dst=P["Wfc"], dstT=P["WfcT"])
        self._packed = P
        self.__dict__["_pack_list"] = pl


def _kmajor_gemm(ws, tag, A, lda, M, Bm, ldb, N, K, out, dev):
    """out[M,N] = A[:K,:M]^T . Bm[:K,:N] for K-major operands: TN kernel for bf16, transposes + NT for f32."""
    if A.dtype == torch.bfloat16:
        ops.gemm_tn(A, Bm, out, M, N, K, lda=lda, ldb=ldb)
        return
    ldT = _pad(K, 8) + 8
    AT = ws.get(tag + "_AT", (M, ldT), torch.float32, dev); BT = ws.get(tag + "_BT", (N, ldT), torch.float32, dev)
    ops.cast_transpose(A, K, M, dstT=AT, lds=lda); ops.cast_transpose(Bm, K, N, dstT=BT, lds=ldb)
    ops.gemm_nt(AT, BT, out, M, N, ldT, lda=ldT, ldb=ldT)


# ------------------------------------------------------------------------------------------------------------------------------------
# The two halves of the step (mosesvae.py:142-164 encoder, :166-199 decoder) as helpers over a workspace.  Three autograd Functions are built
# from them: _MosesFn (the fused `forward`, mosesvae.py:126-140: both halves in one node, decoder parameter gradients on a side stream beside
# the encoder's backward), _MosesEncFn (`forward_encoder`: the encoder half alone) and _MosesDecFn (`forward_decoder(x, z)`: the decoder half
# on a latent the caller supplies, with a gradient w.r.t. that latent).  The halves use disjoint buffer names and keep a generation count
# each (`_SavedState._next_saved_ws(half)`), so `forward_encoder` followed by `forward_decoder` (the reference's composition) overwrites nothing.
def _enc_forward(mod, ws, P, x_pad, lengths, eps):
    dev, dt, f32 = x_pad.device, mod.compute_dtype, torch.float32
    B, T = x_pad.shape
    V, dz = mod.x_emb.num_embeddings, mod.d_z
    Vp = _pad(V, 4)
    W = lambda name, shape, d=f32: ws.get(name, shape, d, dev)
    # ---------------- encoder GRU (mosesvae.py:150-156): embedding folded into a [V, 4H] table
    pe = P["enc"]; Hq = pe["H"]
    tbl4 = W("enc_tbl4", (V, 4 * Hq)); ops.gemm_nt(P["E_p"], pe["Wx_p"], tbl4, V, 3 * Hq, Vp)      # slot 3 stays zero (ldc = 4H)
    # the table rows are added in the step epilogue (add_table / add_index): no gathered [T, B, 4H] copy
    ldh_e = Hq + _LDPAD
    hsx_e = [W("enc_hsx0", (T + 1, B, ldh_e), dt)]                 # slot 0 = initial state (zeros), slots 1.. = outputs
    gates_e = [W("enc_gates0", (T, B, 4 * Hq), dt)]
    hstate_e = [W("enc_hstate0", (2, B, Hq))]
    ops.rnn_fwd(L.CELL_GRU, dt, T, B, Hq, None, 0, pe["Wih"], [pe["ldw"]], pe["Whh"], [pe["ldw"]], pe["bias"],
                [hsx_e[0][1:]], ldh_e, None, gates_e, hstate_e, lengths=lengths, add_table=tbl4, add_index=x_pad, tag="moses_enc_fwd")
    h_last = hstate_e[0][(T - 1) & 1]                               # fp32 [B,Hq]: last valid state of every sequence
    # ---------------- heads + reparameterisation + KL (mosesvae.py:158-162)
    m1 = W("m1", (B, 256)); l1 = W("l1", (B, 256)); mu = W("mu", (B, dz)); lv = W("lv", (B, dz))
    ops.gemm_nt(h_last, mod.q_mu[0].weight, m1, B, 256, Hq, bias=mod.q_mu[0].bias, act=L.ACT_RELU)
    ops.gemm_nt(m1, mod.q_mu[2].weight, mu, B, dz, 256, bias=mod.q_mu[2].bias)
    ops.gemm_nt(h_last, mod.q_logvar[0].weight, l1, B, 256, Hq, bias=mod.q_logvar[0].bias, act=L.ACT_RELU)
    ops.gemm_nt(l1, mod.q_logvar[2].weight, lv, B, dz, 256, bias=mod.q_logvar[2].bias)
    z = torch.empty(B, dz, device=dev); kl = torch.empty(1, device=dev)
    if eps is None:
        seed, off = mod.noise_stream.take(B * dz)
        eps = W("eps", (B, dz))
        ops.moses_latent_fwd(mu, lv, None, z, kl, B, dz, seed=seed, offset=off, eps_out=eps)
    else:
        ops.moses_latent_fwd(mu, lv, eps, z, kl, B, dz)
    return z, kl, lv, eps


def _dec_forward(mod, ws, P, x_pad, lengths, z, drop):
    dev, dt, f32 = x_pad.device, mod.compute_dtype, torch.float32
    B, T = x_pad.shape
    V, dz = mod.x_emb.num_embeddings, mod.d_z
    Vp = _pad(V, 4)
    W = lambda name, shape, d=f32: ws.get(name, shape, d, dev)
    # ---------------- decoder (mosesvae.py:172-197)
    pd = P["dec"]; Hd = pd["H"]; NL = mod.decoder_rnn.num_layers
    W("z_saved", (B, dz)).copy_(z)                                  # the backward's operand (dW_z, decoder_lat): z is an output / input tensor we do not own
    h0 = W("h0", (B, Hd)); ops.gemm_nt(z, mod.decoder_lat.weight, h0, B, Hd, dz, bias=mod.decoder_lat.bias)
    tbl4d = W("dec_tbl4", (V, 4 * Hd)); ops.gemm_nt(P["E_p"], pd["Wx_p"], tbl4d, V, 3 * Hd, Vp)
    zp4 = W("zp4", (B, 4 * Hd)); ops.gemm_nt(z, P["Wz"], zp4, B, 3 * Hd, dz)
    ldh_d = Hd + _LDPAD
    hsx_d = [W(f"dec_hsx{l}", (T + 1, B, ldh_d), dt) for l in range(NL)]
    for l in range(NL):
        ops.cast_transpose(h0, B, Hd, dst=hsx_d[l][0])              # h_0 = decoder_lat(z) for every layer (mosesvae.py:185-186)
    gates_d = [W(f"dec_gates{l}", (T, B, 4 * Hd), dt) for l in range(NL)]
    hstate_d = [W(f"dec_hstate{l}", (2, B, Hd)) for l in range(NL)]
    hd = None
    if drop is not None:                                            # train mode: dropped copies of the outputs of layers 0 .. NL-2
        hd = [W(f"dec_hd{l}", (T, B, ldh_d), dt) for l in range(NL - 1)] + [None]
    # layer-0 input [emb(x_t), z] (mosesvae.py:176-188): z part = time-invariant addend zp4, token part = table row x[b, t]
    ops.rnn_fwd(L.CELL_GRU, dt, T, B, Hd, zp4, 0, pd["Wih"], [pd["ldw"]] * NL, pd["Whh"], [pd["ldw"]] * NL, pd["bias"],
                [h[1:] for h in hsx_d], ldh_d, None, gates_d, hstate_d, h0=[h[0] for h in hsx_d], ldh0=ldh_d, lengths=lengths,
                add_table=tbl4d, add_index=x_pad,
                hdrop=hd, drop_mask=(None if drop is None or drop[2] is None else [drop[2][l] for l in range(NL - 1)]),
                drop_p=(drop[0] if drop else 0.0), drop_seed=(drop[1] if drop else 0), tag="moses_dec_fwd")
    TB = T * B
    y_tb = W("y_tb", (TB, V))
    ops.gemm_nt(hsx_d[-1][1:].reshape(TB, ldh_d), P["Wfc"], y_tb, TB, V, Hd, bias=mod.decoder_fc.bias)
    y = torch.empty(B, T, V, device=dev); ops.permute102(y_tb, y, T, B, V)
    loss2 = W("loss2", (2,)); ops.ce_loss_fwd(y_tb, V, x_pad, mod.pad, loss2, B, T, V)
    return loss2[0].clone(), y, loss2[1].clone()


def _lin_bwd(ws, grads, dev, B, tag, dy, x, WT, wname, bname, M_in, N_out, need_dx=True):
    """y = x W^T + b:  dW = dy^T x (exact-f32 TN kernel straight from the batch-major operands) with db = colsum(dy) as its virtual
    ones column, dx = dy W (via the packed transpose WT [in, out])."""
    ops.gemm_tn_f32_colsum(dy, x, grads[wname], grads[bname], N_out, M_in, B, lda=dy.stride(0), ldb=x.stride(0))
    if not need_dx:
        return None
    dx = ws.get(tag + "_dx", (B, M_in), torch.float32, dev)
    ops.gemm_nt(dy, WT, dx, B, M_in, N_out)
    return dx


def _onehot(mod, ws, x_pad):
    """bf16: the token scatter as a contraction, dtbl = onehot(x)^T . dG0 over the T*B rows (both halves' table gradients use the matrix)."""
    dt = mod.compute_dtype
    if dt != torch.bfloat16:
        return None
    B, T = x_pad.shape
    V = mod.x_emb.num_embeddings
    oh = ws.get("onehot_tb", (T * B + 8, _pad(V, 8)), dt, x_pad.device)[:T * B]
    ops.onehot_tb(x_pad, oh, B, T, V)
    return oh


def _dec_backward(mod, ws, P, grads, x_pad, lengths, drop, drecon, dy_ext, dz_ext, fork, onehot):
    """Gradients of the decoder half: fills grads[decoder_*] and the decoder's share dE of the embedding gradient; returns (dz, finish) --
    dz [B, d_z] = gradient w.r.t. the latent (incl. dz_ext), finish() = the bias column sums + the join of the side stream, to be called
    once the caller has enqueued whatever it wants to run beside the side stream's weight-gradient GEMMs."""
    dev, dt, f32 = x_pad.device, mod.compute_dtype, torch.float32
    B, T = x_pad.shape
    V, dz = mod.x_emb.num_embeddings, mod.d_z
    Vp, Vp8, TB, Bp = _pad(V, 4), _pad(V, 8), T * B, _pad(B, 4)
    W = lambda name, shape, d=f32: ws.get(name, shape, d, dev)
    c = lambda t: t.contiguous().float() if t is not None else None
    g1 = lambda t: c(t).reshape(1) if t is not None else None
    pd = P["dec"]
    Hd, NL = pd["H"], mod.decoder_rnn.num_layers
    ldh_d = Hd + _LDPAD
    # ---------------- decoder_fc + cross-entropy
    y_tb, loss2 = W("y_tb", (TB, V)), W("loss2", (2,))
    fuse_dy = dt == torch.bfloat16 and (4 * Hd) % 64 == 0   # the top GRU cell contracts dl . W_fc itself (pad / finished positions: zero rows in dl)
    ldl = _dyk(V) if fuse_dy else Vp8
    dl = W("dl", (TB + 8, ldl), dt)[:TB]
    if drecon is None:
        drecon = torch.zeros((), device=dev)
    ops.ce_loss_bwd(y_tb, V, x_pad, mod.pad, loss2, g1(drecon), c(dy_ext), dl, B, T, V)
    hsx_d = [W(f"dec_hsx{l}", (T + 1, B, ldh_d), dt) for l in range(NL)]
    out_seq = hsx_d[-1][1:].reshape(TB, ldh_d)
    # Everything below that only produces PARAMETER gradients of the decoder (decoder_fc, the GRU stack's dW / db, the token table) is
    # throughput-bound and independent of dz, while what follows on the path to the encoder -- heads, the encoder GRU's reverse
    # pass: latency-bound, a fraction of the chip -- is not: the former runs on a side stream beside the latter (joined by finish()).
    side = mod._side_stream(dev) if fork else None

    def fc_grads():
        _kmajor_gemm(ws, "fc", dl, ldl, V, out_seq, ldh_d, Hd, TB, grads["decoder_fc.weight"], dev)
        dbp = W("dbfc_p", (Vp8,)); ops.colsum_t(dl, TB, Vp8, dbp, ldx=ldl); grads["decoder_fc.bias"].copy_(dbp[:V])

    if not fork:
        fc_grads()
    if fuse_dy and dy_ext is not None:
        # an external gradient w.r.t. the returned logits sits in dl at finished positions too (it counts for decoder_fc.bias, above); the
        # padded output carries no gradient back to h there (pad_packed_sequence, mosesvae.py:189), and the top cell contracts dl AS IS:
        # clear those rows first.  (The CE's own gradient is zero there already: the target is pad.)
        ops.mask_rows_tb(dl, T, B, lengths)
    dyd = None
    if not fuse_dy:
        dyd = W("dy_dec", (TB, Hd)); ops.gemm_nt(dl, P["WfcT"], dyd, TB, Hd, Vp8)
    # ---------------- decoder GRU stack, reverse wavefront (+ gradient w.r.t. h_0 of every layer)
    ldg_d = 4 * Hd + _LDPAD
    gates_d = [W(f"dec_gates{l}", (T, B, 4 * Hd), dt) for l in range(NL)]
    dG_d = [W(f"dec_dG{l}", (T, B, ldg_d), dt) for l in range(NL)]
    dstate_d = [W(f"dec_dstate{l}", (2, B, Hd)) for l in range(NL)]
    dh0 = [W(f"dec_dh0_{l}", (B, Hd)) for l in range(NL)]
    ops.rnn_bwd(L.CELL_GRU, dt, T, B, Hd, pd["WhhT"], [pd["ldwT"]] * NL, pd["WihT"], [pd["ldwT"]] * NL, dyd, Hd,
                [h[1:] for h in hsx_d], ldh_d, None, gates_d, dG_d, dstate_d, ldg=ldg_d, h0=[h[0] for h in hsx_d], ldh0=ldh_d,
                lengths=lengths, dh0=dh0,
                drop_mask=(None if drop is None or drop[2] is None else [drop[2][l] for l in range(NL - 1)]),
                drop_p=(drop[0] if drop else 0.0), drop_seed=(drop[1] if drop else 0), tag="moses_dec_bwd",
                dy_a=(dl if fuse_dy else None), dy_w=(P["WfcT"] if fuse_dy else None), dy_k=(_dyk(V) if fuse_dy else 0))
    hd = [W(f"dec_hd{l}", (T, B, ldh_d), dt) for l in range(NL - 1)] if drop is not None else None
    dE = grads["_dE_dec"]

    def dec_bias_grads():
        # column sums of dG (bandwidth-bound passes over [T*B, 4H]): on the MAIN stream behind the encoder's backward, beside the side
        # stream's compute-bound weight-gradient GEMMs -- the side stream is the longer of the two chains since the encoder's backward is one launch
        s4 = W("dec_s4", (4 * Hd,))
        for l in range(NL):
            ops.colsum_t(dG_d[l].view(TB, ldg_d), TB, 4 * Hd, s4, ldx=ldg_d)
            grads[f"decoder_rnn.bias_ih_l{l}"].copy_(s4[:3 * Hd])
            grads[f"decoder_rnn.bias_hh_l{l}"][:2 * Hd].copy_(s4[:2 * Hd]); grads[f"decoder_rnn.bias_hh_l{l}"][2 * Hd:].copy_(s4[3 * Hd:])

    def dec_weight_grads():
        # the 3 x (dW_hh[r, z], dW_hh[n], dW_ih) contractions of the stack, one split-K TN GEMM each.  MVAE_MOSES_DW_GROUPED=1: ONE grouped launch of
        # full-K 256 x 256 tiles instead (8 problems, 66 tiles) -- measured in round 5 and NOT kept: 5.46 vs 5.18 ms per step at B = 1024 (66
        # workgroups with K = 62 k each run longer than the whole tail they sit beside; the chip-wide split-K launches finish sooner)
        probs = []
        for l in range(NL):
            a = dG_d[l].view(TB, ldg_d)
            hprev = hsx_d[l][:T].reshape(TB, ldh_d)                  # h_{t-1} for every t (slot 0 = h_0)
            gw = grads[f"decoder_rnn.weight_hh_l{l}"]
            probs.append(dict(A=a, B=hprev, out=gw[:2 * Hd], M=2 * Hd, N=Hd, K=TB, lda=ldg_d, ldb=ldh_d, tag="dwhh_rz"))
            probs.append(dict(A=a[:, 3 * Hd:], B=hprev, out=gw[2 * Hd:], M=Hd, N=Hd, K=TB, lda=ldg_d, ldb=ldh_d, tag="dwhh_n"))
            if l > 0:                                                 # the layer's input: the output of layer l-1 (after its dropout in train mode)
                xin = hd[l - 1].view(TB, ldh_d) if hd is not None else hsx_d[l - 1][1:].reshape(TB, ldh_d)
                probs.append(dict(A=a, B=xin, out=grads[f"decoder_rnn.weight_ih_l{l}"], M=3 * Hd, N=Hd, K=TB, lda=ldg_d, ldb=ldh_d, tag="dwih"))
        if (L.knob("MVAE_MOSES_DW_GROUPED", "0") != "0" and dt == torch.bfloat16
                and all(ops.gemm_tn_grouped_supported(q["A"], q["M"], q["N"], q["K"], q["lda"], q["ldb"]) for q in probs)):
            ops.gemm_tn_grouped(probs)
        else:
            for q in probs:
                _kmajor_gemm(ws, q["tag"], q["A"], q["lda"], q["M"], q["B"], q["ldb"], q["N"], q["K"], q["out"], dev)
        # layer-0 input = [emb(x_t), z]: table gradient for the embedding part (the z part is on the path to the encoder: main stream)
        dtbl3 = W("dec_dtbl3", (V, 3 * Hd))
        if onehot is not None:
            ops.gemm_tn(onehot, dG_d[0].view(TB, ldg_d), dtbl3, V, 3 * Hd, TB, lda=Vp8, ldb=ldg_d)
        else:
            dtbl4 = W("dec_dtbl4", (V, 4 * Hd)); ops.scatter_rows_tb(x_pad, dG_d[0], dtbl4, B, T, V, 4 * Hd, ldd=ldg_d)
            dtbl3.copy_(dtbl4[:, :3 * Hd])
        ops.gemm_nt(dtbl3, pd["WxT"], dE, V, V, 3 * Hd)
        dtblT = W("dec_dtblT", (3 * Hd, Vp)); ops.cast_transpose(dtbl3, V, 3 * Hd, dstT=dtblT)
        dwx = W("dec_dwx", (3 * Hd, V)); ops.gemm_nt(dtblT, P["ET_p"], dwx, 3 * Hd, V, Vp)
        grads["decoder_rnn.weight_ih_l0"][:, :V].copy_(dwx)

    side_done = None
    if fork:
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        for t in grads["_flat"]:
            t.record_stream(side)
        with torch.cuda.stream(side):
            fc_grads()
            dec_weight_grads()
            side_done = torch.cuda.Event(); side_done.record()
    else:
        dec_weight_grads()
    dzp4 = W("dzp4", (B, ldg_d)); ops.timesum(dG_d[0], T, B, ldg_d, dzp4)
    dzp3 = W("dzp3", (B, 3 * Hd)); dzp3.copy_(dzp4[:, :3 * Hd])
    z = W("z_saved", (B, dz))
    dzp3T, zT = W("dzp3T", (3 * Hd, Bp)), W("zT", (dz, Bp))
    ops.cast_transpose(dzp3, B, 3 * Hd, dstT=dzp3T); ops.cast_transpose(z, B, dz, dstT=zT)
    dwz = W("dec_dwz", (3 * Hd, dz)); ops.gemm_nt(dzp3T, zT, dwz, 3 * Hd, dz, Bp)
    grads["decoder_rnn.weight_ih_l0"][:, V:].copy_(dwz)          # (the token columns [:, :V] are written by dec_weight_grads)
    dz_tot = W("dz_tot", (B, dz)); ops.gemm_nt(dzp3, P["WzT"], dz_tot, B, dz, 3 * Hd)
    # h_0 = decoder_lat(z), shared by the layers
    dh0s = dh0[0]
    for l in range(1, NL):
        dh0s.add_(dh0[l])
    dzl = _lin_bwd(ws, grads, dev, B, "lat", dh0s, z, P["WlatT"], "decoder_lat.weight", "decoder_lat.bias", dz, Hd)
    dz_tot.add_(dzl)
    if dz_ext is not None:
        dz_tot.add_(c(dz_ext))

    def finish():
        dec_bias_grads()
        if side_done is not None:
            torch.cuda.current_stream().wait_event(side_done)        # the decoder's parameter gradients (and dE) from the side stream

    return dz_tot, finish


def _enc_backward(mod, ws, P, grads, x_pad, lengths, eps, dz_tot, dkl, dlv_ext, onehot):
    """Gradients of the encoder half from dz_tot [B, d_z] (None: no gradient through z), the KL weight and an external gradient w.r.t. logvar:
    fills grads[encoder_rnn.*, q_mu.*, q_logvar.*] and the encoder's share dE2 of the embedding gradient."""
    dev, dt, f32 = x_pad.device, mod.compute_dtype, torch.float32
    B, T = x_pad.shape
    V, dz = mod.x_emb.num_embeddings, mod.d_z
    Vp, Vp8, TB = _pad(V, 4), _pad(V, 8), T * B
    W = lambda name, shape, d=f32: ws.get(name, shape, d, dev)
    c = lambda t: t.contiguous().float() if t is not None else None
    g1 = lambda t: c(t).reshape(1) if t is not None else None
    pe = P["enc"]
    Hq = pe["H"]
    ldh_e = Hq + _LDPAD
    lin_bwd = lambda *a, **k: _lin_bwd(ws, grads, dev, B, *a, **k)
    # ---------------- reparameterisation + KL, heads
    mu, lv = W("mu", (B, dz)), W("lv", (B, dz))
    dmu, dlv = W("dmu", (B, dz)), W("dlv", (B, dz))
    if dz_tot is None:
        dz_tot = W("dz_zero", (B, dz))                                # allocated zero, never written
    ops.moses_latent_bwd(mu, lv, eps, dz_tot, g1(dkl), c(dlv_ext), dmu, dlv, B, dz)
    m1, l1 = W("m1", (B, 256)), W("l1", (B, 256))
    h_last = W("enc_hstate0", (2, B, Hq))[(T - 1) & 1]
    dm1 = lin_bwd("mu2", dmu, m1, P["mu_W2T"], "q_mu.2.weight", "q_mu.2.bias", 256, dz); ops.relu_bwd(dm1, m1)
    dhq = lin_bwd("mu0", dm1, h_last, P["mu_W0T"], "q_mu.0.weight", "q_mu.0.bias", Hq, 256)
    dhq_tot = W("dhq_tot", (B, Hq)); dhq_tot.copy_(dhq)
    dl1 = lin_bwd("lv2", dlv, l1, P["lv_W2T"], "q_logvar.2.weight", "q_logvar.2.bias", 256, dz); ops.relu_bwd(dl1, l1)
    dhq2 = lin_bwd("lv0", dl1, h_last, P["lv_W0T"], "q_logvar.0.weight", "q_logvar.0.bias", Hq, 256)
    dhq_tot.add_(dhq2)
    # ---------------- encoder GRU: only the final state receives gradient; masked steps pass it back to each sequence's last step
    ldg_e = 4 * Hq + _LDPAD
    hsx_e = [W("enc_hsx0", (T + 1, B, ldh_e), dt)]
    gates_e = [W("enc_gates0", (T, B, 4 * Hq), dt)]
    dG_e = [W("enc_dG0", (T, B, ldg_e), dt)]
    dstate_e = [W("enc_dstate0", (2, B, Hq))]
    ops.rnn_bwd(L.CELL_GRU, dt, T, B, Hq, pe["WhhT"], [pe["ldwT"]], pe["WihT"], [pe["ldwT"]], None, 0,
                [hsx_e[0][1:]], ldh_e, None, gates_e, dG_e, dstate_e, ldg=ldg_e, h0=[hsx_e[0][0]], ldh0=ldh_e,
                lengths=lengths, dh_last=[dhq_tot], tag="moses_enc_bwd")
    a = dG_e[0].view(TB, ldg_e)
    hprev = hsx_e[0][:T].reshape(TB, ldh_e)
    gw = grads["encoder_rnn.weight_hh_l0"]
    _kmajor_gemm(ws, "e_dwhh_rz", a, ldg_e, 2 * Hq, hprev, ldh_e, Hq, TB, gw[:2 * Hq], dev)
    _kmajor_gemm(ws, "e_dwhh_n", a[:, 3 * Hq:], ldg_e, Hq, hprev, ldh_e, Hq, TB, gw[2 * Hq:], dev)
    s4e = W("enc_s4", (4 * Hq,)); ops.colsum_t(a, TB, 4 * Hq, s4e, ldx=ldg_e)
    grads["encoder_rnn.bias_ih_l0"].copy_(s4e[:3 * Hq])
    grads["encoder_rnn.bias_hh_l0"][:2 * Hq].copy_(s4e[:2 * Hq]); grads["encoder_rnn.bias_hh_l0"][2 * Hq:].copy_(s4e[3 * Hq:])
    etbl3 = W("enc_dtbl3", (V, 3 * Hq))
    if onehot is not None:
        ops.gemm_tn(onehot, a, etbl3, V, 3 * Hq, TB, lda=Vp8, ldb=ldg_e)
    else:
        etbl4 = W("enc_dtbl4", (V, 4 * Hq)); ops.scatter_rows_tb(x_pad, dG_e[0], etbl4, B, T, V, 4 * Hq, ldd=ldg_e)
        etbl3.copy_(etbl4[:, :3 * Hq])
    ops.gemm_nt(etbl3, pe["WxT"], grads["_dE_enc"], V, V, 3 * Hq)
    etblT = W("enc_dtblT", (3 * Hq, Vp)); ops.cast_transpose(etbl3, V, 3 * Hq, dstT=etblT)
    ops.gemm_nt(etblT, P["ET_p"], grads["encoder_rnn.weight_ih_l0"], 3 * Hq, V, Vp)


def _grad_views(mod, ws, names, params, dev):
    """Zeroed flat fp32 buffer + per-parameter views for `names`, plus the two halves' scratch shares of the embedding gradient."""
    gflat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    grads, off = {"_flat": [gflat]}, 0
    for n, p in zip(names, params):
        grads[n] = gflat[off:off + p.numel()].view(p.shape); off += p.numel()
    V = mod.x_emb.num_embeddings
    grads["_dE_dec"] = ws.get("dE", (V, V), torch.float32, dev)
    grads["_dE_enc"] = ws.get("dE2", (V, V), torch.float32, dev)
    return grads


def _fork_allowed(dy_ext):
    return L.knob("MVAE_MOSES_FORK", "1") != "0" and dy_ext is None      # MVAE_MOSES_FORK=0: one stream (A/B knob)


class _MosesFn(torch.autograd.Function):
    """mosesvae.VAE.forward (mosesvae.py:126-140): both halves in ONE autograd node."""

    @staticmethod
    @ops.traced("moses_step_fwd")
    def forward(ctx, mod, x_pad, lengths, eps, drop, *params):
        dev = x_pad.device
        _require_cuda(dev, "mosesvae.VAE")
        P = mod._pack(dev)
        slot, ws, gen = mod._next_saved_ws("enc")
        dslot, dws, dgen = mod._next_saved_ws("dec")
        z, kl, lv, eps = _enc_forward(mod, ws, P, x_pad, lengths, eps)
        recon, y, ntok = _dec_forward(mod, dws, P, x_pad, lengths, z, drop)
        ctx.mod, ctx.x_pad, ctx.lengths, ctx.eps, ctx.drop = mod, x_pad, lengths, eps, drop
        ctx.slot, ctx.gen, ctx.dslot, ctx.dgen = slot, gen, dslot, dgen
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(ntok)
        return kl[0].clone(), recon, z, lv.clone(), y, ntok

    @staticmethod
    @ops.traced("moses_step_bwd")
    def backward(ctx, dkl, drecon, dz_ext, dlv_ext, dy_ext, _dntok=None):
        mod, x_pad, lengths, eps, drop = ctx.mod, ctx.x_pad, ctx.lengths, ctx.eps, ctx.drop
        ws = mod._saved_ws(ctx.slot, ctx.gen, "mosesvae.VAE", "enc")
        dws = mod._saved_ws(ctx.dslot, ctx.dgen, "mosesvae.VAE", "dec")
        dev = x_pad.device
        P = mod._packed
        params = mod._plist()
        names = [n for n, _ in mod.named_parameters()]
        grads = _grad_views(mod, ws, names, params, dev)
        onehot = _onehot(mod, ws, x_pad)
        dz_tot, finish = _dec_backward(mod, dws, P, grads, x_pad, lengths, drop, drecon, dy_ext, dz_ext, _fork_allowed(dy_ext), onehot)
        _enc_backward(mod, ws, P, grads, x_pad, lengths, eps, dz_tot, dkl, dlv_ext, onehot)
        finish()
        ge = grads["x_emb.weight"]; torch.add(grads["_dE_dec"], grads["_dE_enc"], out=ge)
        ge[mod.pad].zero_()                                           # nn.Embedding(padding_idx=pad): no gradient to the pad row
        return (None, None, None, None, None) + tuple(grads[n] for n in names)


class _MosesEncFn(torch.autograd.Function):
    """mosesvae.VAE.forward_encoder (mosesvae.py:142-164): the encoder half ALONE -- no decoder kernel is launched."""

    @staticmethod
    @ops.traced("moses_encoder_fwd")
    def forward(ctx, mod, x_pad, lengths, eps, *params):
        dev = x_pad.device
        _require_cuda(dev, "mosesvae.VAE.forward_encoder")
        P = mod._pack(dev)
        slot, ws, gen = mod._next_saved_ws("enc")
        z, kl, lv, eps = _enc_forward(mod, ws, P, x_pad, lengths, eps)
        ctx.mod, ctx.slot, ctx.gen, ctx.x_pad, ctx.lengths, ctx.eps = mod, slot, gen, x_pad, lengths, eps
        ctx.set_materialize_grads(False)
        return z, kl[0].clone(), lv.clone()

    @staticmethod
    @ops.traced("moses_encoder_bwd")
    def backward(ctx, dz_ext, dkl, dlv_ext):
        mod, x_pad, lengths, eps = ctx.mod, ctx.x_pad, ctx.lengths, ctx.eps
        ws = mod._saved_ws(ctx.slot, ctx.gen, "mosesvae.VAE.forward_encoder", "enc")
        dev = x_pad.device
        names, params = mod._half_params("enc")
        grads = _grad_views(mod, ws, names, params, dev)
        onehot = _onehot(mod, ws, x_pad)
        dz = dz_ext.contiguous().float() if dz_ext is not None else None
        _enc_backward(mod, ws, mod._packed, grads, x_pad, lengths, eps, dz, dkl, dlv_ext, onehot)
        ge = grads["x_emb.weight"]; ge.copy_(grads["_dE_enc"])
        ge[mod.pad].zero_()
        return (None, None, None, None) + tuple(grads[n] for n in names)


class _MosesDecFn(torch.autograd.Function):
    """mosesvae.VAE.forward_decoder (mosesvae.py:166-199): teacher-forced decoder on a latent the caller supplies; differentiable in z."""

    @staticmethod
    @ops.traced("moses_decoder_fwd")
    def forward(ctx, mod, x_pad, lengths, z, drop, *params):
        dev = x_pad.device
        _require_cuda(dev, "mosesvae.VAE.forward_decoder")
        P = mod._pack(dev)
        dslot, dws, dgen = mod._next_saved_ws("dec")
        recon, y, ntok = _dec_forward(mod, dws, P, x_pad, lengths, z.contiguous().float(), drop)
        ctx.mod, ctx.x_pad, ctx.lengths, ctx.drop = mod, x_pad, lengths, drop
        ctx.dslot, ctx.dgen = dslot, dgen
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(ntok)
        return recon, y, ntok

    @staticmethod
    @ops.traced("moses_decoder_bwd")
    def backward(ctx, drecon, dy_ext, _dntok=None):
        mod, x_pad, lengths, drop = ctx.mod, ctx.x_pad, ctx.lengths, ctx.drop
        ws = mod._saved_ws(ctx.dslot, ctx.dgen, "mosesvae.VAE.forward_decoder", "dec")
        dev = x_pad.device
        names, params = mod._half_params("dec")
        grads = _grad_views(mod, ws, names, params, dev)
        onehot = _onehot(mod, ws, x_pad)
        dz_tot, finish = _dec_backward(mod, ws, mod._packed, grads, x_pad, lengths, drop, drecon, dy_ext, None, _fork_allowed(dy_ext), onehot)
        finish()
        ge = grads["x_emb.weight"]; ge.copy_(grads["_dE_dec"])
        ge[mod.pad].zero_()
        dz = dz_tot.clone() if ctx.needs_input_grad[3] else None
        return (None, None, None, dz, None) + tuple(grads[n] for n in names)
