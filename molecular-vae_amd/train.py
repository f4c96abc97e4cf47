"""Training-loop surface of train.py:81-104 / train_distributed.py:72-73, MI355X-native.

  * ``loss_function`` -- the reference's ELBO (train.py:31-38) as one fused HIP forward/backward pair.
  * ``FusedAdam``     -- ``clip_grad_norm_(params, max_norm)`` (train.py:102) + ``optim.Adam.step()`` (train.py:81,104)
                         as two HIP kernels over one flat fp32 buffer; a ``torch.optim.Optimizer`` so LR schedulers
                         (``ReduceLROnPlateau``, train.py:83) and ``state_dict()`` (train.py:173) keep working.
  * ``GradSync``      -- data parallelism: one process per GPU, bucketed all-reduce(SUM) of the flat gradient over RCCL
                         (backend "nccl" on ROCm) / gloo on CPU, replacing ``nn.DataParallel`` (train_distributed.py:72).
  * ``shard_batch`` / ``ShardedSampler`` -- contiguous per-rank shards (the commented-out DistributedSampler of
                         moses_train_distrib.py:176).
"""
import math

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops
from .functional import bce_kl_loss, make_loss_function  # noqa: F401  (re-exported)


# ------------------------------------------------------------------------------------------------ data parallel
class GradSync:
    """Bucketed all-reduce(SUM) of a flat gradient buffer; the 1/world scaling is folded into the optimiser kernel.

    Works on any device/backend (RCCL on GPUs, gloo on CPU for the world_size>1 tests).  Buckets are issued
    asynchronously in order; ``wait()`` blocks the current stream until every bucket has landed.
    On a fully connected 8-GPU xGMI node RCCL spreads each bucket over all 7 links; 32 MiB buckets keep each
    per-link transfer well above the latency floor while letting bucket k+1 overlap bucket k's reduction.
    """

    def __init__(self, bucket_bytes=32 << 20, group=None, early=True, compress=None, force=False):
        """force: issue every collective even when the group has ONE rank (an initialised process group is still required).  With backend
        "nccl" this runs the whole RCCL path -- ProcessGroupNCCL's internal communication stream, the event hand-off from the stream a
        collective is issued on (the side stream, for early ranges) and back at wait() -- on a single GPU; sums over one rank are the identity,
        so the step must equal the non-distributed one bit for bit (tests, `bench.py --force-comm`).
        compress="bf16": every bucket is all-reduced as bfloat16 (rounded copy out, sum, converted back into the fp32 buffer): half the
        bytes on the links, ~3 significant digits per gradient element -- what the bf16 training mode's decoder gradients carry anyway; the
        optimiser still accumulates in fp32.  None: fp32 on the wire (bit-reproducible sums)."""
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.group = group
        self.allow_early = early  # False: every range is reduced in step() (needed when parameter hooks clone gradients)
        self.compress = compress
        self.force = bool(force)
        self.handles = []
        self.early = []          # [(flat, lo, hi)] ranges whose all-reduce was started from inside backward (this step)
        self._staged = []        # compress: (fp32 view, bf16 copy) pairs to convert back in wait()
        self.stats = dict(buckets=0, early_ranges=0, bytes_early=0, bytes_rest=0)     # running totals (tests / logs / bench.py's `comm`)

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def active(self):
        """True when gradients are exchanged: more than one rank, or `force` under an initialised process group."""
        return self.world > 1 or (self.force and dist.is_available() and dist.is_initialized())

    def start(self, flat):
        if not self.active:
            return
        n = flat.numel()
        for off in range(0, n, self.bucket_elems):
            self.stats["buckets"] += 1
            piece = flat[off:min(n, off + self.bucket_elems)]
            if self.compress == "bf16":
                half = piece.to(torch.bfloat16)               # plumbing: a rounded copy for the wire
                self._staged.append((piece, half))
                piece = half
            self.handles.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def start_early(self, flat, lo, hi):
        """Called from a module's backward once flat[lo:hi] is final on the CURRENT stream (the collective is ordered after
        that stream's work): the all-reduce then runs under the rest of backward.  Every rank issues the same calls in the
        same order (same model, same code path)."""
        if not self.active or not self.allow_early:
            return
        self.start(flat[lo:hi])
        self.early.append((flat, lo, hi))
        self.stats["early_ranges"] += 1
        self.stats["bytes_early"] += (hi - lo) * (2 if self.compress == "bf16" else flat.element_size())

    def start_rest(self, flat):
        """All-reduce whatever part of `flat` start_early has not covered."""
        done = sorted((lo, hi) for f, lo, hi in self.early if f is flat)
        pos, n = 0, 0
        for lo, hi in done:
            if lo > pos:
                self.start(flat[pos:lo]); n += lo - pos
            pos = max(pos, hi)
        if pos < flat.numel():
            self.start(flat[pos:]); n += flat.numel() - pos
        if self.active:
            self.stats["bytes_rest"] += n * (2 if self.compress == "bf16" else flat.element_size())

    def wait(self):
        for h in self.handles:
            h.wait()
        for piece, half in self._staged:
            if half.is_cuda:
                half.record_stream(torch.cuda.current_stream())    # allocated on the stream start() ran on (early ranges: the side stream)
            piece.copy_(half)
        self.handles, self.early, self._staged = [], [], []

    # -- sharded form: reduce-scatter -> every rank updates its 1/world of the flat buffers -> all-gather of the parameters
    def reduce_scatter(self, flat, shard_elems):
        """flat [world * shard_elems] -> this rank's reduced shard (a view of `flat`): half the bytes of an all-reduce on every link."""
        w, r = self.world, dist.get_rank(self.group)
        assert flat.numel() == w * shard_elems
        out = flat[r * shard_elems:(r + 1) * shard_elems]
        self.stats["buckets"] += 1
        if out.device.type == "cpu":                           # gloo: no in-place aliasing of input and output
            tmp = torch.empty_like(out)
            dist.reduce_scatter_tensor(tmp, flat, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(tmp)
        else:
            dist.reduce_scatter_tensor(out, flat, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def all_gather(self, flat, shard_elems):
        """every rank's shard of `flat` (its own slice, updated in place) -> the whole buffer on every rank."""
        r = dist.get_rank(self.group)
        mine = flat[r * shard_elems:(r + 1) * shard_elems]
        if flat.device.type == "cpu":
            mine = mine.clone()
        dist.all_gather_into_tensor(flat, mine, group=self.group)

    def grad_scale(self):
        return 1.0 / self.world


def shard_batch(n_items, rank, world):
    """Contiguous shard [lo, hi) of a global batch for this rank (equal sizes; remainder dropped like drop_last)."""
    per = n_items // world
    return rank * per, (rank + 1) * per


class ShardedSampler(torch.utils.data.Sampler):
    """Per-epoch shuffled, rank-sharded index stream (seed + epoch), equal length on every rank."""

    def __init__(self, n, rank=0, world=1, seed=0, shuffle=True):
        self.n, self.rank, self.world, self.seed, self.shuffle, self.epoch = n, rank, world, seed, shuffle, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.n // self.world

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator(); g.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(self.n, generator=g)
        else:
            perm = torch.arange(self.n)
        per = self.n // self.world
        return iter(perm[self.rank * per:(self.rank + 1) * per].tolist())


# ------------------------------------------------------------------------------------------------ optimiser
class FusedAdam(torch.optim.Optimizer):
    """Adam (no weight decay / amsgrad, as train.py:81) with the global-norm clip of train.py:102 fused in.

    Parameters are flattened into one fp32 buffer (each ``p.data`` becomes a view of it); ``exp_avg`` / ``exp_avg_sq``
    are views of flat buffers too, so ``state_dict()`` has torch.optim.Adam's layout.  ``step()`` gathers the
    gradients into a flat buffer, all-reduces it when a process group is active, then runs
    ``mvae_sumsq`` + ``mvae_clip_adam`` -- no host synchronisation anywhere.
    """

    # torch.optim.Adam's remaining hyper-parameters at their inert values: kept in every param_group so that ``state_dict()`` loads into
    # ``torch.optim.Adam`` (train.py:81,173) and the reverse; step() refuses any other value.
    _ADAM_INERT = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                       decoupled_weight_decay=False)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.0, grad_sync=None, shard_optimizer=False):
        """shard_optimizer (needs a GradSync with world > 1): the gradient is reduce-SCATTERED, every rank runs the clip + Adam update on its
        1/world slice of the flat buffers only (exp_avg / exp_avg_sq of the other slices stay untouched: `gather_state()` collects them
        before a checkpoint), and the updated parameters are all-gathered -- half the collective bytes of the all-reduce form and 1/world
        of the 7 x 4 x P bytes of optimiser traffic.  The global gradient norm is formed from the same 64K-element partial sums in the same
        order as the all-reduce form, so both forms give bit-identical parameters.  No early (in-backward) ranges in this form."""
        defaults = dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm, **self._ADAM_INERT)
        super().__init__(params, defaults)
        self.grad_sync = grad_sync
        self.shard = bool(shard_optimizer) and grad_sync is not None and grad_sync.active
        self._moments_stale = False      # sharded form: True from step() until gather_state() -- the other ranks' slices of exp_avg / exp_avg_sq are old
        if self.shard:
            if grad_sync.compress is not None:
                raise ValueError("GradSync(compress=...) applies to the all-reduce form only: the reduce-scatter of shard_optimizer=True sends fp32")
            grad_sync.allow_early = False
        world = grad_sync.world if self.shard else 1
        self._flat = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            n = sum(p.numel() for p in ps)
            # One spare element behind the parameters in all four flat buffers: the POISON slot.  A persistent launch that gives up stores a
            # NaN into g[n] (mvae_rnn_*_desc.poison); mvae_sumsq covers the slot, so the norm becomes NaN and mvae_clip_adam skips the whole
            # update -- in data parallel on EVERY rank, because the slot travels with the last gradient bucket of the all-reduce (sharded
            # form: with the all-reduced partial sums).  It is zero otherwise (p[n], m[n], v[n] never leave zero) and adds nothing to the norm.
            # sharded form: equal slices whose boundaries fall on the 64K-element chunks of the gradient-norm partial sums (zero padding)
            chunk = 1 << 16
            shard_elems = ((n + 1 + world * chunk - 1) // (world * chunk)) * chunk if self.shard else n + 1
            n_alloc = shard_elems * world
            pflat = torch.zeros(n_alloc, dtype=torch.float32, device=dev)
            m = torch.zeros_like(pflat); v = torch.zeros_like(pflat); g = torch.zeros_like(pflat)
            off = 0
            for p in ps:
                k = p.numel()
                with torch.no_grad():
                    pflat[off:off + k].copy_(p.data.reshape(-1))
                    p.data = pflat[off:off + k].view(p.shape)
                self.state[p] = dict(step=torch.tensor(0.0), exp_avg=m[off:off + k].view(p.shape),
                                     exp_avg_sq=v[off:off + k].view(p.shape))
                off += k
            nparts = (n_alloc + (1 << 16) - 1) >> 16
            self._flat.append(dict(params=ps, p=pflat, m=m, v=v, g=g, partial=torch.zeros(nparts, device=dev),
                                   norm=torch.zeros(2, device=dev), step=0, n=n, shard_elems=shard_elems, poison=g[n:n + 1]))
            off = 0
            for p in ps:                 # modules may write their gradients straight into g (see _lib.register_grad_sink)
                L.register_grad_sink(p, self, g, off, poison=g[n:n + 1])
                off += p.numel()
        L.PARAM_EPOCH[0] += 1

    @property
    def last_grad_norm(self):
        """Device tensor holding the pre-clip global gradient norm of the last step (train.py:102's return value)."""
        return self._flat[0]["norm"][:1]

    @property
    def skipped_steps(self):
        """Device tensor: how many step() calls the optimiser kernel turned into no-ops because the global gradient norm was not finite (a
        persistent launch gave up and poisoned the step, or the gradients diverged).  Reading it synchronises; nothing in step() does."""
        return self._flat[0]["norm"][1:2]

    def gather_grads(self):
        """Copy every ``p.grad`` into the flat gradient buffer (missing grads count as zero); returns the flats."""
        outs = []
        early = self.grad_sync.early if self.grad_sync is not None else None
        for f in self._flat:
            if f is None:
                continue
            views, off = [], 0
            for p in f["params"]:
                k = p.numel()
                if p.grad is None:
                    f["g"][off:off + k].zero_()
                elif p.grad.data_ptr() != f["g"].data_ptr() + 4 * off or not p.grad.is_contiguous():
                    # autograd cloned the gradient (a tensor hook, a second reference) instead of adopting the sink view
                    if early and any(fl is f["g"] and lo < off + k and off < hi for fl, lo, hi in early):
                        raise L.MvaeError("a gradient inside a range whose all-reduce was already started from backward lives outside the "
                                          "flat gradient buffer (parameter hooks / retained grads clone it): the reduced and the local "
                                          "gradient would be mixed.  Remove the hook or build GradSync(early=False).")
                    views.append((off, k, p.grad))          # (a gradient written in place through the sink needs no copy)
                off += k
            if views:
                # one foreach copy: plumbing, not compute
                torch._foreach_copy_([f["g"][o:o + k] for o, k, _ in views], [g.reshape(-1) for _, _, g in views])
            outs.append(f["g"])
        return outs

    @torch.no_grad()
    @ops.traced("fused_adam_step")
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        # A launch with bounded spins that gave up during this step has poisoned the gradient buffer's spare slot ON THE DEVICE: the kernels below
        # then skip the update by themselves, on every rank -- no host wait here.  persist_check only reports what has already arrived.
        ops.persist_check()
        ops.join_pending()             # gradients produced on a side stream (decoder weight-gradient GEMMs)
        flats = self.gather_grads()
        sync = self.grad_sync
        if sync is not None and not self.shard:
            with ops._Timed("dp_allreduce_exposed" if sync.active else None):   # bench.py: what the overlap with backward did not hide
                for g in flats:
                    sync.start_rest(g)
                sync.wait()
        scale = sync.grad_scale() if sync is not None else 1.0
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            f["step"] += 1
            b1, b2 = group["betas"]
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise L.MvaeError("FusedAdam implements plain Adam (train.py:81): weight_decay / amsgrad / maximize are not supported")
            if f["p"].device.type != "cuda":
                raise L.MvaeError("FusedAdam.step runs on the MI355X only (no CPU fallback)")
            if self.shard:
                S, r = f["shard_elems"], dist.get_rank(sync.group)
                with ops._Timed("dp_allreduce_exposed"):
                    gs = sync.reduce_scatter(f["g"], S)
                sl = slice(r * S, (r + 1) * S)
                cps = S >> 16                                         # 64K-element chunks per shard: this rank's slice of the partial sums
                with ops._Timed("hbm_sumsq_clip_adam"):
                    f["partial"].zero_()
                    ops.sumsq(gs, f["partial"][r * cps:(r + 1) * cps])
                    dist.all_reduce(f["partial"], group=sync.group)    # disjoint slices + zeros: a gather, a few KB; same values on every rank
                    ops.clip_adam(f["p"][sl], gs, f["m"][sl], f["v"][sl], f["partial"], scale, group["max_grad_norm"], group["lr"], b1, b2,
                                  group["eps"], f["step"], f["norm"], poison_reset=f["poison"])
                with ops._Timed("dp_allreduce_exposed"):
                    sync.all_gather(f["p"], S)
                self._moments_stale = True
            else:
                with ops._Timed("hbm_sumsq_clip_adam"):
                    ops.sumsq(f["g"], f["partial"])
                    ops.clip_adam(f["p"], f["g"], f["m"], f["v"], f["partial"], scale, group["max_grad_norm"], group["lr"], b1, b2,
                                  group["eps"], f["step"], f["norm"], poison_reset=f["poison"])
            for p in f["params"]:
                self.state[p]["step"] += 1
        L.PARAM_EPOCH[0] += 1      # packed bf16 / transposed weight shadows must be refreshed
        return loss

    def gather_state(self):
        """Sharded form: every rank holds the Adam moments of its own slice only; collect all of them (before ``state_dict()`` / a checkpoint)."""
        if not self.shard:
            return
        for f in self._flat:
            if f is not None:
                self.grad_sync.all_gather(f["m"], f["shard_elems"])
                self.grad_sync.all_gather(f["v"], f["shard_elems"])
        self._moments_stale = False

    def state_dict(self):
        """torch.optim.Adam's layout.  Sharded form: after a step() this rank holds current moments for its own 1/world slice only, so the
        dictionary would be silently wrong for the rest -- refuse until EVERY rank has called ``gather_state()`` (a collective: call it on all
        ranks, then save on rank 0)."""
        if self.shard and self._moments_stale:
            raise L.MvaeError("FusedAdam(shard_optimizer=True).state_dict(): the Adam moments of the other ranks' slices are stale; call "
                              "optimizer.gather_state() on every rank first (a collective), then save")
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts a FusedAdam or a ``torch.optim.Adam`` state dict (train.py:173 ``optimizer_state_dict``): same ``state`` layout
        (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter index); hyper-parameters present in the dict override ours."""
        sd_groups = state_dict["param_groups"]
        if len(sd_groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for group, sg in zip(self.param_groups, sd_groups):
            if len(sg["params"]) != len(group["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for k in ("lr", "betas", "eps", "max_grad_norm", "initial_lr"):
                if k in sg:
                    group[k] = tuple(sg[k]) if k == "betas" else sg[k]
        idx = 0
        for group, f in zip(self.param_groups, self._flat):
            for p in group["params"]:
                st = state_dict["state"].get(idx)
                if st is not None and f is not None and p in self.state:
                    self.state[p]["exp_avg"].copy_(st["exp_avg"]); self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                    self.state[p]["step"] = torch.as_tensor(float(st["step"]))
                    f["step"] = int(float(st["step"]))
                idx += 1


class CosineAnnealingLRWithRestart:
    """A RESTATEMENT of moses_train_distrib.py:61-89 (the surface fixes the attribute names and the arithmetic; nothing here is MI355X-specific),
    same state machine: period 10 epochs, lr_end 1e-4, no period growth; constructing it performs the
    first ``step()`` (as ``_LRScheduler.__init__`` does), so the very first epoch already runs at the k = 1 point of the cosine."""

    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.n_period, self.n_mult, self.lr_end = 10, 1, 1e-4
        self.current_epoch, self.t_end = 0, self.n_period
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = -1
        self.step()

    def get_lr(self):
        return [cosine_lr_with_restart(b, self.current_epoch, self.t_end, self.lr_end) for b in self.base_lrs]

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        self.current_epoch += 1
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g["lr"] = lr
        if self.current_epoch == self.t_end:
            self.current_epoch = 0
            self.t_end = self.n_mult * self.t_end

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


# ------------------------------------------------------------------------------------------------ loop body
def train_step(model, optimizer, loss_function, data, ohe, eps=None):
    """train.py:95-104 minus the per-step ``loss.item()`` host sync: returns the loss as a device tensor.
    ``eps`` optionally injects the reparameterisation noise (parity tests); by default the model draws it as models.py:92 does."""
    optimizer.zero_grad(set_to_none=True)
    recon_batch, mu, logvar = model(data) if eps is None else model(data, eps=eps)
    loss = loss_function(recon_batch, ohe, mu, logvar)
    loss.backward()
    optimizer.step()           # clip (max_grad_norm) + Adam fused
    return loss.detach()


def exact_match_accuracy(recon_batch, data):
    """train.py:109-113: fraction of sequences whose arg-max reconstruction equals the input, computed on device."""
    preds = recon_batch.argmax(dim=2)
    return (preds == data).all(dim=1).float().mean()


def cosine_lr_with_restart(base_lr, epoch_in_period, period=10, lr_end=1e-4):
    """CosineAnnealingLRWithRestart.get_lr (moses_train_distrib.py:73-76)."""
    return lr_end + (base_lr - lr_end) * (1 + math.cos(math.pi * epoch_in_period / period)) / 2


class KLAnnealer:
    """A RESTATEMENT of moses_train_distrib.py:47-58 (ten lines of schedule arithmetic whose names the trainer surface fixes): linear 0 -> 1
    over n_epoch."""

    def __init__(self, n_epoch):
        self.i_start, self.w_start, self.w_max, self.n_epoch = 0, 0, 1, n_epoch
        self.inc = (self.w_max - self.w_start) / (self.n_epoch - self.i_start)

    def __call__(self, i):
        k = (i - self.i_start) if i >= self.i_start else 0
        return self.w_start + k * self.inc


# ------------------------------------------------------------------------------------------------ evaluation / checkpoints
@torch.no_grad()
def evaluate(model, loss_function, batches):
    """``test(epoch)`` of train.py:120-153 without the logging: forward-only over `batches` of ``(idx, ohe)``, returns
    ``(mean loss per batch, exact-match accuracy over all sequences)`` as Python floats (one host sync at the end).  Runs under
    ``no_grad``: the decoder then skips the 16 B per (row, unit, step) of saved gate / cell state it writes for backward."""
    was_training = model.training
    model.eval()
    total, right, n_seq, n = None, None, 0, 0
    for data, ohe in batches:
        recon, mu, logvar = model(data)
        loss = loss_function(recon, ohe, mu, logvar)
        acc = (recon.argmax(dim=2) == data).all(dim=1).sum()
        total = loss if total is None else total + loss
        right = acc if right is None else right + acc
        n_seq += data.shape[0]; n += 1
    model.train(was_training)
    if n == 0:
        return float("nan"), float("nan")
    return float(total) / n, float(right) / n_seq


@torch.no_grad()
def generate_from_latent(model, charset, n=None, z=None, batch_size=2000, generator=None):
    """``train_sample.py:29-45`` without the rdkit filter: decode latent vectors with ``model.decoder`` (forward-only pass: nothing is saved
    for backward), arg-max over the class axis (``torch.max(recon_batch, dim=2)``), map ids through ``charset`` and right-strip the padding.
    ``z`` [N, latent] or ``n`` draws ``torch.rand`` latents as the reference does (uniform on [0, 1)).  Returns (list of strings, z)."""
    from .data import indices_to_smiles
    dec = model.decoder if hasattr(model, "decoder") else model
    dev = next(dec.parameters()).device
    o = dec.latent_input[0].in_features
    if z is None:
        if n is None:
            raise ValueError("give n or z")
        z = torch.rand(n, o, device=dev, generator=generator)
    z = z.to(dev).float()
    out = []
    for lo in range(0, z.shape[0], batch_size):
        recon = dec(z[lo:lo + batch_size])
        out.extend(indices_to_smiles(recon.argmax(dim=2), charset))
    return out, z


def strip_module_prefix(state_dict):
    """Keys saved from ``nn.DataParallel(model)`` (train_distributed.py:72,145-151) carry a ``module.`` prefix; the strip of
    mosesanalyize.py:171-173, applied only to keys that have it."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def save_checkpoint(path, model, optimizer, epoch, charset, max_len, latent_size=None):
    """The dictionary of train.py:170-177 (``latent_size`` is absent in train_distributed.py:145-151).  With FusedAdam(shard_optimizer=True)
    every rank must have called ``optimizer.gather_state()`` since the last step (``state_dict()`` raises otherwise)."""
    lr = optimizer.param_groups[-1]["lr"]
    d = {"model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "epoch": epoch,
         "charset": charset, "max_len": max_len, "lr": lr}
    if latent_size is not None:
        d["latent_size"] = latent_size
    torch.save(d, path)
    return d


def load_checkpoint(path_or_dict, model, optimizer=None, map_location="cpu"):
    """Inverse of ``save_checkpoint``; also takes the reference's own files (same keys; ``module.``-prefixed model keys accepted).
    The reference has no resume path -- this is the loader its ``train_sample.py:16-19`` spells out by hand."""
    ck = torch.load(path_or_dict, map_location=map_location, weights_only=False) if isinstance(path_or_dict, (str, bytes)) or hasattr(path_or_dict, "read") else path_or_dict
    model.load_state_dict(strip_module_prefix(ck["model_state_dict"]))
    L.PARAM_EPOCH[0] += 1
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    return ck


# ------------------------------------------------------------------------------------------------ MOSES loop body
def moses_train_step(model, optimizer, kl_weight, batch, eps=None):
    """moses_train_distrib.py:287-299 in the 6-tuple form of moses_train_distrib_logp.py:321-345: ``loss = kl_weight * kl + recon``,
    backward, ``clip_grad_norm_(50)`` (the optimiser's ``max_grad_norm``) and ``Adam.step()`` -- without the per-step ``.item()``
    syncs.  Returns device scalars ``(loss, kl, recon)``."""
    optimizer.zero_grad(set_to_none=True)
    sync = getattr(optimizer, "grad_sync", None)
    if sync is not None and hasattr(model, "dp_group"):
        model.dp_group = sync.group          # the CE's global token count is reduced over the ranks the gradients are reduced over
        model.dp_force = sync.force
    kl_loss, recon_loss, _, _, _, _ = model(batch) if eps is None else model(batch, eps=eps)
    loss = kl_weight * kl_loss + recon_loss
    loss.backward()
    optimizer.step()
    return loss.detach(), kl_loss.detach(), recon_loss.detach()


def moses_train_epoch(model, epoch, batches, kl_weight, optimizer=None, log_every=0, log=print):
    """``_train_epoch`` (moses_train_distrib.py:200-258): train when an optimiser is given, else evaluate; returns the same ``postfix``
    dictionary (epoch means instead of the reference's 1000-entry circular-buffer means, which average over unwritten zeros)."""
    model.train(optimizer is not None)
    sums, n = None, 0
    for i, batch in enumerate(batches):
        if optimizer is not None:
            vals = moses_train_step(model, optimizer, kl_weight, batch)
        else:
            with torch.no_grad():
                kl, rec, _, _, _, _ = model(batch)
            vals = (kl_weight * kl + rec, kl, rec)
        v = torch.stack([x.float() for x in vals])
        sums = v if sums is None else sums + v
        n += 1
        if log_every and i % log_every == 0:
            cur = (sums / n).tolist()
            log(f"epoch {epoch} it {i}: loss={cur[0]:.5f} (kl={cur[1]:.5f} recon={cur[2]:.5f}) klw={kl_weight:.5f}")
    mean = (sums / max(n, 1)).tolist() if sums is not None else [float("nan")] * 3
    lr = optimizer.param_groups[0]["lr"] if optimizer is not None else None
    return {"epoch": epoch, "kl_weight": kl_weight, "lr": lr, "kl_loss": mean[1], "recon_loss": mean[2], "loss": mean[0],
            "mode": "Eval" if optimizer is None else "Train"}
