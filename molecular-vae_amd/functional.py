"""Loss and small-op autograd glue over the C ABI: the ELBO of train.py:31-38 and the stand-alone Lambda head."""
import torch

from . import _lib as L
from . import ops


class _BceKlLossFn(torch.autograd.Function):
    """max_len * BCELoss(mean)(recon, x) + (-0.5) * mean(1 + mu - logvar^2 - exp(mu))  -- train.py:31-38 verbatim
    (binary CE on softmax outputs, log clamp at -100, mu/logvar swapped in the KL term)."""

    @staticmethod
    def forward(ctx, recon, target, mu, logvar, max_len):
        if recon.device.type != "cuda":
            raise L.MvaeError("loss_function runs on the MI355X only (no CPU fallback)")
        recon_c, target_c = recon.contiguous().float(), target.contiguous().float()
        mu_c, logvar_c = mu.contiguous().float(), logvar.contiguous().float()
        out = torch.empty(3, dtype=torch.float32, device=recon.device)
        with ops._Timed("hbm_bce_kl_loss_fwd"):
            ops.bce_kl_loss_fwd(recon_c, target_c, mu_c, logvar_c, max_len, out)
        ctx.save_for_backward(recon_c, target_c, mu_c, logvar_c)
        ctx.max_len = max_len
        ctx.parts = out
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        recon, target, mu, logvar = ctx.saved_tensors
        drecon = torch.empty_like(recon); dmu = torch.empty_like(mu); dlogvar = torch.empty_like(logvar)
        g = grad_out.contiguous().float().reshape(1)
        with ops._Timed("hbm_bce_kl_loss_bwd"):
            ops.bce_kl_loss_bwd(recon, target, mu, logvar, ctx.max_len, g, drecon, dmu, dlogvar)
        return drecon, None, dmu, dlogvar, None


def bce_kl_loss(recon_x, x, mu, logvar, max_len):
    """Fused HIP ELBO (train.py:31-38).  x is the float one-hot target [B, L, C] the DataLoader yields."""
    return _BceKlLossFn.apply(recon_x, x, mu, logvar, float(max_len))


def make_loss_function(max_len):
    """``loss_function(recon_x, x, mu, logvar)`` with the reference's signature; ``max_len`` is the module-level
    global of train.py:43 there."""
    def loss_function(recon_x, x, mu, logvar):
        return bce_kl_loss(recon_x, x, mu, logvar, max_len)
    return loss_function


class _LambdaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, eps, wm, bm, wv, bv):
        if x.device.type != "cuda":
            raise L.MvaeError("Lambda runs on the MI355X only (no CPU fallback)")
        x = x.contiguous().float()
        B, K = x.shape
        o = wm.shape[0]
        dev = x.device
        W = torch.cat([wm, wv], 0).contiguous(); b = torch.cat([bm, bv], 0).contiguous()
        Kp = (K + 3) // 4 * 4
        if Kp != K:
            xp = torch.zeros(B, Kp, device=dev); xp[:, :K] = x
            Wp = torch.zeros(2 * o, Kp, device=dev); Wp[:, :K] = W
        else:
            xp, Wp = x, W
        mulv = torch.empty(B, 2 * o, device=dev)
        ops.gemm_nt(xp, Wp, mulv, B, 2 * o, Kp, bias=b)
        z = torch.empty(B, o, device=dev); mu = torch.empty_like(z); logv = torch.empty_like(z)
        if eps is None:                     # drawn inside the launch (Lambda(noise="device"))
            seed, off = mod.noise_stream.take(B * o)
            eps = torch.empty(B, o, device=dev)
            ops.lambda_fwd(mulv, None, z, mu, logv, B, o, scale=mod.scale, seed=seed, offset=off, eps_out=eps)
        elif not torch.is_tensor(eps):      # models._HostNoise: a pinned host block, read in place by the launch
            host, eps = eps, torch.empty(B, o, device=dev)
            ops.lambda_fwd(mulv, host.buf, z, mu, logv, B, o, eps_out=eps)
            host.consumed()
        else:
            ops.lambda_fwd(mulv, eps, z, mu, logv, B, o)
        ctx.save_for_backward(x, W, mulv, eps)
        return z, mu, logv

    @staticmethod
    def backward(ctx, dz, dmu, dlogv):
        x, W, mulv, eps = ctx.saved_tensors
        B, K = x.shape
        o = W.shape[0] // 2
        dev = x.device
        dmulv = torch.empty(B, 2 * o, device=dev)
        c = lambda t: t.contiguous() if t is not None else None
        ops.lambda_bwd(mulv, eps, c(dz), c(dmu), c(dlogv), dmulv, B, o)
        Bp = (B + 3) // 4 * 4
        n2p = (2 * o + 3) // 4 * 4
        dmulvT = torch.zeros(2 * o, Bp, device=dev); xT = torch.zeros(K, Bp, device=dev)
        ops.cast_transpose(dmulv, B, 2 * o, dstT=dmulvT); ops.cast_transpose(x, B, K, dstT=xT)
        dW = torch.empty(2 * o, K, device=dev)
        ops.gemm_nt(dmulvT, xT, dW, 2 * o, K, Bp)
        db = torch.empty(2 * o, device=dev)
        ops.colsum(dmulv, B, 2 * o, db)
        WT = torch.zeros(K, n2p, device=dev)
        ops.cast_transpose(W, 2 * o, K, dstT=WT)
        dmp = torch.zeros(B, n2p, device=dev); dmp[:, :2 * o] = dmulv
        dx = torch.empty(B, K, device=dev)
        ops.gemm_nt(dmp, WT, dx, B, K, n2p)
        return None, dx, None, dW[:o], db[:o], dW[o:], db[o:]


def lambda_forward(mod, x, eps=None):
    """Stand-alone Lambda (models.py:89-94) for callers that use the head outside MolEncoder."""
    B = x.shape[0]
    o = mod.z_mean.out_features
    if eps is None and not (mod.noise == "device" and x.is_cuda):
        eps = mod.draw_eps(B, o, x.device)
    if torch.is_tensor(eps):
        eps = eps.contiguous().float().to(x.device)
    return _LambdaFn.apply(mod, x, eps, mod.z_mean.weight, mod.z_mean.bias,
                           mod.z_log_var.weight, mod.z_log_var.bias)
