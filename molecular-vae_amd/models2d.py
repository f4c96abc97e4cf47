"""Host-side mirror of ``models2d.VAE`` (models2d.py:8-52), the conv(ReLU) + GRU-over-the-one-hot-block variant.

Same constructor (no arguments), attribute names (``conv1d1..3``, ``fc0``, ``fc11``, ``fc12``, ``fc2``, ``gru``, ``fc3``), ``state_dict`` keys /
shapes / default-init RNG consumption, ``encode`` / ``reparametrize`` / ``decode`` / ``forward`` signatures as the reference:
``forward(x [B,120,35] one-hot float) -> (recon [B,120,35] probabilities, mu [B,2], logvar [B,2])``; in ``train()`` mode
``z = mu + exp(logvar / 2) * randn_like``, in ``eval()`` mode ``z = mu`` (models2d.py:31-38).  The reference wires this model to no trainer
and defines no loss for it; it pairs with ``train.py``'s ``loss_function`` (``make_loss_function(120)``) and ``FusedAdam`` like ``MolecularVAE``.

Underneath, one ``torch.autograd.Function`` over the C ABI (no torch.nn compute, no CPU fallback):
  * encoder: the one-hot block is read ONCE, channels-last ``[b][v][t]`` (sequence position = conv channel, as the reference's
    ``Conv1d(120, 9, 9)`` on ``[B,120,35]`` has it); three sliding-window Conv1d + bias + ReLU GEMMs (exact f32), dense + SELU, stacked
    mu | logvar heads, reparameterisation -- mvae_conv1d_act_{fwd,bwd}, mvae_gemm_nt, mvae_lambda_{fwd,bwd};
  * decoder: ``fc2`` + SELU; the GRU input is the latent repeated 120 times, so its layer-0 projection is computed once (time-invariant
    addend); ``nn.GRU(2, 501, 3)`` runs on the layer-wavefront GRU step kernels with the hidden size padded 501 -> 512 by zero weight
    rows / columns (a padded unit stays exactly 0: r = z = 1/2, n = tanh(0) = 0, h' = (1 - z) n + z h = 0), which puts every launch on
    the LDS-direct / vectorised paths; ``fc3`` + softmax over the class axis -- mvae_rnn_{fwd,bwd}, mvae_softmax_tb_{fwd,bwd}.
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .models import LinearWeights, Conv1dWeights, RNNWeights, _Workspace, _SavedState, _params_key, _pad, _require_cuda, _LDPAD, _dyk
from .mosesvae import _kmajor_gemm

SEQ, VOCAB, HID, HP, NLAY = 120, 35, 501, 512, 3


def _slots4p(w3, H, Hp, Kp, order):
    """[3H, K] GRU gate rows (r, z, n) -> [4Hp, Kp] slot rows, zero padded; order 'x' = (r, z, n, 0) for W_ih, 'h' = (r, z, 0, n) for W_hh."""
    K = w3.shape[1]
    out = torch.zeros(4 * Hp, Kp, dtype=w3.dtype, device=w3.device)
    out[0:H, :K] = w3[0:H]
    out[Hp:Hp + H, :K] = w3[H:2 * H]
    s = 2 if order == "x" else 3
    out[s * Hp:s * Hp + H, :K] = w3[2 * H:3 * H]
    return out


class VAE(nn.Module, _SavedState):
    def __init__(self, dtype=torch.bfloat16):
        super().__init__()
        self.conv1d1 = Conv1dWeights(SEQ, 9, 9)          # models2d.py:12-21, same construction order (same RNG stream under manual_seed)
        self.conv1d2 = Conv1dWeights(9, 9, 9)
        self.conv1d3 = Conv1dWeights(9, 10, 11)
        self.fc0 = LinearWeights(90, 435)
        self.fc11 = LinearWeights(435, 2)
        self.fc12 = LinearWeights(435, 2)
        self.fc2 = LinearWeights(2, 2)
        self.gru = RNNWeights("GRU", 2, HID, NLAY)
        self.fc3 = LinearWeights(HID, VOCAB)
        self.compute_dtype = dtype
        self.__dict__["noise_stream"] = ops.NoiseStream()           # (seed, counter) of the reparameterisation draws made inside mvae_lambda_fwd
        self._init_saved_state()
        self._pack_key, self._packed = None, {}

    # -- the reference's method surface
    def forward(self, x, eps=None):
        dev = x.device
        B = x.shape[0]
        if tuple(x.shape[1:]) != (SEQ, VOCAB):
            raise ValueError(f"models2d.VAE expects the one-hot block [B, {SEQ}, {VOCAB}], got {tuple(x.shape)}")
        if not self.training:
            eps = torch.zeros(B, 2, device=dev)                         # models2d.py:37-38: z = mu
        # training, eps None: models2d.py:34's randn_like(std) is drawn inside mvae_lambda_fwd (counter hash of self.noise_stream)
        infer = not torch.is_grad_enabled()
        return _Models2dFn.apply(self, x.contiguous().float(), eps.contiguous().float() if eps is not None else None, infer, *list(self.parameters()))

    def encode(self, x):
        with torch.no_grad():
            was = self.training
            self.eval()
            _, mu, logvar = self.forward(x)
            self.train(was)
        return mu, logvar

    def reparametrize(self, mu, logvar):
        if self.training:
            return torch.randn_like(logvar) * torch.exp(0.5 * logvar) + mu      # [B,2] plumbing-sized; the fused path does this in lambda_fwd
        return mu

    def decode(self, z):
        """Decode given latents [B, 2] (forward-only)."""
        with torch.no_grad():
            return _decode_only(self, z.contiguous().float())

    # -- packed shadows
    def _pack(self, dev):
        params = list(self.parameters())
        key = _params_key(params) + (self.compute_dtype,)
        if key == self._pack_key:
            return self._packed
        ws, dt, f32 = self._ws, self.compute_dtype, torch.float32
        P = {}
        with torch.no_grad():
            for n, conv, ldx, ldo, want_q in (("c1", self.conv1d1, SEQ, 12, False), ("c2", self.conv1d2, 12, 12, True), ("c3", self.conv1d3, 12, 12, True)):
                Ci, Co, k = conv.in_channels, conv.out_channels, conv.kernel_size
                P[n + "_wp"] = ws.get(n + "_wp", (Co, k * ldx), f32, dev)
                P[n + "_wq"] = ws.get(n + "_wq", (Ci, k * ldo), f32, dev) if want_q else None
                ops.conv1d_pack_weights(conv.weight, Ci, Co, k, ldx, P[n + "_wp"], ldo if want_q else 0, P[n + "_wq"])
            P["W0p"] = ws.get("W0p", (435, 92), f32, dev); P["W0p"][:, :90].copy_(self.fc0.weight)
            P["W0T"] = ws.get("W0T", (92, 436), f32, dev); ops.cast_transpose(P["W0p"], 435, 92, dstT=P["W0T"])
            P["Wml"] = ws.get("Wml", (4, 436), f32, dev); P["Wml"][:2, :435].copy_(self.fc11.weight); P["Wml"][2:, :435].copy_(self.fc12.weight)
            P["bml"] = ws.get("bml", (4,), f32, dev); P["bml"][:2].copy_(self.fc11.bias); P["bml"][2:].copy_(self.fc12.bias)
            P["WmlT"] = ws.get("WmlT", (436, 4), f32, dev); ops.cast_transpose(P["Wml"], 4, 436, dstT=P["WmlT"])
            P["W2p"] = ws.get("W2p", (2, 4), f32, dev); P["W2p"][:, :2].copy_(self.fc2.weight)
            P["W2T"] = ws.get("W2T", (4, 4), f32, dev); P["W2T"][:2, :2].copy_(self.fc2.weight.t())
            g = self.gru
            ldw, ldwT = HP + _LDPAD, 4 * HP + _LDPAD
            P.update(ldw=ldw, ldwT=ldwT, Whh=[], WhhT=[], Wih=[None], WihT=[None], bias=[])
            for l in range(NLAY):
                whh4 = _slots4p(getattr(g, f"weight_hh_l{l}"), HID, HP, HP, "h")
                w = ws.get(f"Whh{l}", (4 * HP, ldw), dt, dev); wT = ws.get(f"WhhT{l}", (HP, ldwT), dt, dev)
                ops.cast_transpose(whh4, 4 * HP, HP, dst=w, dstT=wT)
                P["Whh"].append(w); P["WhhT"].append(wT)
                if l > 0:
                    wih4 = _slots4p(getattr(g, f"weight_ih_l{l}"), HID, HP, HP, "x")
                    w = ws.get(f"Wih{l}", (4 * HP, ldw), dt, dev); wT = ws.get(f"WihT{l}", (HP, ldwT), dt, dev)
                    ops.cast_transpose(wih4, 4 * HP, HP, dst=w, dstT=wT)
                    P["Wih"].append(w); P["WihT"].append(wT)
                bi, bh = getattr(g, f"bias_ih_l{l}"), getattr(g, f"bias_hh_l{l}")
                b4 = ws.get(f"bias{l}", (4 * HP,), f32, dev)
                b4.zero_()
                b4[0:HID] = bi[0:HID] + bh[0:HID]; b4[HP:HP + HID] = bi[HID:2 * HID] + bh[HID:2 * HID]
                b4[2 * HP:2 * HP + HID] = bi[2 * HID:]; b4[3 * HP:3 * HP + HID] = bh[2 * HID:]
                P["bias"].append(b4)
            P["Wx0"] = ws.get("Wx0", (4 * HP, 4), f32, dev); P["Wx0"].copy_(_slots4p(g.weight_ih_l0, HID, HP, 4, "x"))
            P["Wx0T"] = ws.get("Wx0T", (4, 4 * HP), f32, dev); ops.cast_transpose(P["Wx0"], 4 * HP, 4, dstT=P["Wx0T"])
            Cp = _pad(VOCAB, 8)
            w3p = torch.zeros(VOCAB, HP, device=dev); w3p[:, :HID].copy_(self.fc3.weight)
            P["Wfc"] = ws.get("Wfc", (VOCAB, ldw), dt, dev); P["WfcT"] = ws.get("WfcT", (HP, _dyk(VOCAB) if dt == torch.bfloat16 else Cp), dt, dev)
            ops.cast_transpose(w3p, VOCAB, HP, dst=P["Wfc"], dstT=P["WfcT"])
        self._pack_key, self._packed = key, P
        return P


def _decoder_forward(mod, P, W, zp, B, dev, infer):
    """zp [B,4] (z in columns 0,1) -> recon [B,120,35]; fills the workspace buffers backward reads."""
    dt, f32 = mod.compute_dtype, torch.float32
    d2p = W("d2p", (B, 4))
    ops.gemm_nt(zp, P["W2p"], d2p, B, 2, 4, bias=mod.fc2.bias, act=L.ACT_SELU)                # models2d.py:41
    gx0 = W("gx0", (B, 4 * HP))
    ops.gemm_nt(d2p, P["Wx0"], gx0, B, 4 * HP, 4)                                              # layer-0 input projection, once (repeat(1,120,1), :42)
    ldh = HP + _LDPAD
    hsx = [W(f"hsx{l}", (SEQ + 1, B, ldh), dt) for l in range(NLAY)]                           # slot 0 = h_0 = 0
    gates = [None if infer else W(f"gates{l}", (SEQ, B, 4 * HP), dt) for l in range(NLAY)]
    hstate = [W(f"hstate{l}", (2, B, HP)) for l in range(NLAY)]
    ops.rnn_fwd(L.CELL_GRU, dt, SEQ, B, HP, gx0, 0, P["Wih"], [P["ldw"]] * NLAY, P["Whh"], [P["ldw"]] * NLAY, P["bias"],
                [h[1:] for h in hsx], ldh, None, gates, hstate, tag="m2d_gru_fwd")
    TB = SEQ * B
    logits = W("logits", (TB, VOCAB))
    ops.gemm_nt(hsx[-1][1:].reshape(TB, ldh), P["Wfc"], logits, TB, VOCAB, HP, bias=mod.fc3.bias)     # :44
    recon = torch.empty(B, SEQ, VOCAB, dtype=f32, device=dev)
    ops.softmax_tb_fwd(logits, VOCAB, recon, B, SEQ, VOCAB)                                            # :45 softmax over the class axis
    return recon


def _decode_only(mod, z):
    dev = z.device
    _require_cuda(dev, "models2d.VAE.decode")
    P = mod._pack(dev)
    _, ws = mod._next_saved_ws()
    B = z.shape[0]
    W = lambda name, shape, d=torch.float32: ws.get(name, shape, d, dev)
    zp = W("zp", (B, 4)); zp[:, :2].copy_(z)
    return _decoder_forward(mod, P, W, zp, B, dev, True)


class _Models2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, eps, infer, *params):
        dev = x.device
        _require_cuda(dev, "models2d.VAE")
        P = mod._pack(dev)
        f32 = torch.float32
        slot, ws = mod._next_saved_ws()
        B = x.shape[0]
        W = lambda name, shape, d=f32: ws.get(name, shape, d, dev)
        # ---- encoder (models2d.py:23-29): the one-hot block, channels-last [b][v][t]
        xcl = W("xcl", (B * VOCAB, SEQ)); ops.permute021(x, xcl, B, SEQ, VOCAB)
        c1, c2, c3 = mod.conv1d1, mod.conv1d2, mod.conv1d3
        W1, W2, W3 = VOCAB - 8, VOCAB - 16, VOCAB - 26                      # 27, 19, 9
        y1 = W("y1", (B * W1, 12)); ops.conv1d_selu_fwd(xcl, B, VOCAB, SEQ, VOCAB * SEQ, 9, 9, P["c1_wp"], c1.bias, y1, 12, act=L.ACT_RELU)
        y2 = W("y2", (B * W2, 12)); ops.conv1d_selu_fwd(y1, B, W1, 12, W1 * 12, 9, 9, P["c2_wp"], c2.bias, y2, 12, act=L.ACT_RELU)
        y3 = W("y3", (B * W3, 12)); ops.conv1d_selu_fwd(y2, B, W2, 12, W2 * 12, 10, 11, P["c3_wp"], c3.bias, y3, 12, act=L.ACT_RELU)
        flat = W("flat", (B, 12 * W3)); ops.permute021(y3, flat, B, W3, 12)   # h.view(B, -1): index = channel * 9 + position (channels 10, 11 are zero pads)
        f0 = W("f0", (B, 436)); ops.gemm_nt(flat, P["W0p"], f0, B, 435, 92, lda=12 * W3, bias=mod.fc0.bias, act=L.ACT_SELU)
        mulv = W("mulv", (B, 4)); ops.gemm_nt(f0, P["Wml"], mulv, B, 4, 436, bias=P["bml"])
        z = torch.empty(B, 2, device=dev); mu = torch.empty_like(z); logvar = torch.empty_like(z)
        if eps is None:
            seed, off = mod.noise_stream.take(B * 2)
            eps = W("eps", (B, 2))
            ops.lambda_fwd(mulv, None, z, mu, logvar, B, 2, scale=1.0, seed=seed, offset=off, eps_out=eps)
        else:
            ops.lambda_fwd(mulv, eps, z, mu, logvar, B, 2)                  # :31-38 (eval mode: eps = 0 -> z = mu)
        zp = W("zp", (B, 4)); zp[:, :2].copy_(z)
        # ---- decoder (:40-47)
        recon = _decoder_forward(mod, P, W, zp, B, dev, infer)
        ctx.mod, ctx.slot, ctx.gen, ctx.eps, ctx.B = mod, slot, (-1 if infer else ws.generation), eps, B
        ctx.save_for_backward(recon)
        return recon, mu, logvar

    @staticmethod
    def backward(ctx, drecon, dmu_ext, dlv_ext):
        mod, eps, B = ctx.mod, ctx.eps, ctx.B
        (recon,) = ctx.saved_tensors
        ws = mod._saved_ws(ctx.slot, ctx.gen, "models2d.VAE")
        dev = recon.device
        P, dt, f32 = mod._packed, mod.compute_dtype, torch.float32
        W = lambda name, shape, d=f32: ws.get(name, shape, d, dev)
        params = list(mod.parameters())
        names = [n for n, _ in mod.named_parameters()]
        gflat = torch.zeros(sum(p.numel() for p in params), dtype=f32, device=dev)
        grads, off = {}, 0
        for n, p in zip(names, params):
            grads[n] = gflat[off:off + p.numel()].view(p.shape); off += p.numel()
        Bp, TB, Cp = _pad(B, 4), SEQ * B, _pad(VOCAB, 8)
        ldh, ldg = HP + _LDPAD, 4 * HP + _LDPAD

        def lin_grads(tag, dy, ldy, n_out, x, ldx, n_in, gw, gb):
            """y = x W^T + b:  dW [n_out, n_in] = dy^T x (contraction over the batch), db = column sums of dy."""
            dyT, xT = W(tag + "_dyT", (n_out, Bp)), W(tag + "_xT", (n_in, Bp))
            ops.cast_transpose(dy, B, n_out, dstT=dyT, lds=ldy); ops.cast_transpose(x, B, n_in, dstT=xT, lds=ldx)
            tmp = W(tag + "_dw", (n_out, n_in))
            ops.gemm_nt(dyT, xT, tmp, n_out, n_in, Bp)
            gw.copy_(tmp)
            ops.colsum(dy, B, n_out, gb, ldx=ldy)

        # ---- fc3 + softmax
        drecon = drecon.contiguous().float()
        fuse_dy = dt == torch.bfloat16 and (4 * HP) % 64 == 0   # the top GRU cell contracts dl . W_fc3 itself
        ldl = _dyk(VOCAB) if fuse_dy else Cp
        dl = W("dl", (TB + 8, ldl), dt)[:TB]
        ops.softmax_tb_bwd(recon, drecon, dl, None, B, SEQ, VOCAB)
        hsx = [W(f"hsx{l}", (SEQ + 1, B, ldh), dt) for l in range(NLAY)]
        out_seq = hsx[-1][1:].reshape(TB, ldh)
        dw3 = W("dw3", (VOCAB, HP))
        _kmajor_gemm(ws, "fc3", dl, ldl, VOCAB, out_seq, ldh, HP, TB, dw3, dev)
        grads["fc3.weight"].copy_(dw3[:, :HID])
        dbp = W("db3p", (Cp,)); ops.colsum_t(dl, TB, Cp, dbp, ldx=ldl); grads["fc3.bias"].copy_(dbp[:VOCAB])
        dy = None
        if not fuse_dy:
            dy = W("dy", (TB, HP)); ops.gemm_nt(dl, P["WfcT"], dy, TB, HP, Cp)
        # ---- GRU stack, reverse wavefront
        gates = [W(f"gates{l}", (SEQ, B, 4 * HP), dt) for l in range(NLAY)]
        dG = [W(f"dG{l}", (SEQ, B, ldg), dt) for l in range(NLAY)]
        dstate = [W(f"dstate{l}", (2, B, HP)) for l in range(NLAY)]
        ops.rnn_bwd(L.CELL_GRU, dt, SEQ, B, HP, P["WhhT"], [P["ldwT"]] * NLAY, P["WihT"], [P["ldwT"]] * NLAY, dy, HP,
                    [h[1:] for h in hsx], ldh, None, gates, dG, dstate, ldg=ldg, h0=[h[0] for h in hsx], ldh0=ldh, tag="m2d_gru_bwd",
                    dy_a=(dl if fuse_dy else None), dy_w=(P["WfcT"] if fuse_dy else None), dy_k=(_dyk(VOCAB) if fuse_dy else 0))
        s4 = W("s4", (4 * HP,))
        tmp = W("dw_gru", (4 * HP, HP))
        for l in range(NLAY):
            a = dG[l].view(TB, ldg)
            hprev = hsx[l][:SEQ].reshape(TB, ldh)                        # h_{t-1} for every t (slot 0 = 0)
            _kmajor_gemm(ws, "dwhh_rz", a, ldg, 2 * HP, hprev, ldh, HP, TB, tmp[:2 * HP], dev)          # slots r, z
            _kmajor_gemm(ws, "dwhh_n", a[:, 3 * HP:], ldg, HP, hprev, ldh, HP, TB, tmp[3 * HP:], dev)    # slot n_h (slot n_x has no W_hh rows)
            gw = grads[f"gru.weight_hh_l{l}"]
            gw[0:HID].copy_(tmp[0:HID, :HID]); gw[HID:2 * HID].copy_(tmp[HP:HP + HID, :HID]); gw[2 * HID:].copy_(tmp[3 * HP:3 * HP + HID, :HID])
            if l > 0:
                _kmajor_gemm(ws, "dwih", a, ldg, 3 * HP, hsx[l - 1][1:].reshape(TB, ldh), ldh, HP, TB, tmp[:3 * HP], dev)   # slots r, z, n_x
                gw = grads[f"gru.weight_ih_l{l}"]
                gw[0:HID].copy_(tmp[0:HID, :HID]); gw[HID:2 * HID].copy_(tmp[HP:HP + HID, :HID]); gw[2 * HID:].copy_(tmp[2 * HP:2 * HP + HID, :HID])
            ops.colsum_t(a, TB, 4 * HP, s4, ldx=ldg)
            gb = grads[f"gru.bias_ih_l{l}"]
            gb[0:HID].copy_(s4[0:HID]); gb[HID:2 * HID].copy_(s4[HP:HP + HID]); gb[2 * HID:].copy_(s4[2 * HP:2 * HP + HID])
            gb = grads[f"gru.bias_hh_l{l}"]
            gb[0:HID].copy_(s4[0:HID]); gb[HID:2 * HID].copy_(s4[HP:HP + HID]); gb[2 * HID:].copy_(s4[3 * HP:3 * HP + HID])
        # layer 0: the input is time-invariant -> its gradient is the time sum of dG[0]
        dgx0 = W("dgx0", (B, ldg)); ops.timesum(dG[0], SEQ, B, ldg, dgx0)
        d2p, zp = W("d2p", (B, 4)), W("zp", (B, 4))
        dgT, d2T = W("dgx0T", (4 * HP, Bp)), W("d2T", (4, Bp))
        ops.cast_transpose(dgx0, B, 4 * HP, dstT=dgT, lds=ldg); ops.cast_transpose(d2p, B, 4, dstT=d2T)
        dwx = W("dwx0", (4 * HP, 4)); ops.gemm_nt(dgT, d2T, dwx, 4 * HP, 4, Bp)
        gw = grads["gru.weight_ih_l0"]
        gw[0:HID].copy_(dwx[0:HID, :2]); gw[HID:2 * HID].copy_(dwx[HP:HP + HID, :2]); gw[2 * HID:].copy_(dwx[2 * HP:2 * HP + HID, :2])
        dd2 = W("dd2", (B, 4)); ops.gemm_nt(dgx0, P["Wx0T"], dd2, B, 4, 4 * HP, lda=ldg)
        ops.selu_bwd(dd2, d2p)
        lin_grads("fc2", dd2, 4, 2, zp, 4, 2, grads["fc2.weight"], grads["fc2.bias"])
        dz = W("dz", (B, 2)); ops.gemm_nt(dd2, P["W2T"], dz, B, 2, 4)
        # ---- reparameterisation + heads
        mulv, f0 = W("mulv", (B, 4)), W("f0", (B, 436))
        dmulv = W("dmulv", (B, 4))
        c = lambda t: t.contiguous().float() if t is not None else None
        ops.lambda_bwd(mulv, eps, dz, c(dmu_ext), c(dlv_ext), dmulv, B, 2)
        dwml, dbml = W("dwml", (4, 435)), W("dbml", (4,))
        lin_grads("ml", dmulv, 4, 4, f0, 436, 435, dwml, dbml)
        grads["fc11.weight"].copy_(dwml[:2]); grads["fc12.weight"].copy_(dwml[2:])
        grads["fc11.bias"].copy_(dbml[:2]); grads["fc12.bias"].copy_(dbml[2:])
        df0 = W("df0", (B, 436)); ops.gemm_nt(dmulv, P["WmlT"], df0, B, 436, 4)
        ops.selu_bwd(df0, f0)
        flat = W("flat", (B, 108))
        dw0 = W("dw0", (435, 92))
        lin_grads("fc0", df0, 436, 435, flat, 108, 92, dw0, grads["fc0.bias"])
        grads["fc0.weight"].copy_(dw0[:, :90])
        dflat = W("dflat", (B, 108)); ops.gemm_nt(df0, P["W0T"], dflat, B, 92, 436, lda=436, ldc=108)     # columns 92.. (pad channels) stay zero
        # ---- conv stack (ReLU), channels-last
        W1, W2, W3 = VOCAB - 8, VOCAB - 16, VOCAB - 26
        dy3 = W("dy3", (B * W3, 12)); ops.permute021(dflat, dy3, B, 12, W3)
        xcl, y1, y2, y3 = W("xcl", (B * VOCAB, SEQ)), W("y1", (B * W1, 12)), W("y2", (B * W2, 12)), W("y3", (B * W3, 12))
        dy2, dy1 = W("dy2", (B * W2, 12)), W("dy1", (B * W1, 12))
        dzp3, dzp2, dzp1 = W("dzp3", (B * (W3 + 20), 12)), W("dzp2", (B * (W2 + 16), 12)), W("dzp1", (B * (W1 + 16), 12))
        c1, c2, c3 = mod.conv1d1, mod.conv1d2, mod.conv1d3
        ops.conv1d_selu_bwd(B, W2, 9, 12, W2 * 12, 10, 12, 11, dy3, y3, y2, P["c3_wq"], dzp3, grads["conv1d3.weight"], grads["conv1d3.bias"], dy2, 12,
                            act=L.ACT_RELU)
        ops.conv1d_selu_bwd(B, W1, 9, 12, W1 * 12, 9, 12, 9, dy2, y2, y1, P["c2_wq"], dzp2, grads["conv1d2.weight"], grads["conv1d2.bias"], dy1, 12,
                            act=L.ACT_RELU)
        ops.conv1d_selu_bwd(B, VOCAB, SEQ, SEQ, VOCAB * SEQ, 9, 12, 9, dy1, y1, xcl, None, dzp1, grads["conv1d1.weight"], grads["conv1d1.bias"],
                            None, 0, act=L.ACT_RELU)
        return (None, None, None, None) + tuple(grads[n] for n in names)
