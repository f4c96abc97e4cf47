"""Host-side mirror of the reference's ``models.py`` module surface (models.py:97-165), MI355X-native underneath.

Same class names, constructor arguments, attribute names, ``forward`` signatures and ``state_dict`` keys as the
reference, so ``train.py:77-104`` runs against these modules unchanged.  The modules own
ordinary fp32 ``nn.Parameter``s; ``forward`` hands raw device pointers to the HIP kernels behind ``include/mvae.h``
through ``torch.autograd.Function``s.  There is no torch.nn compute and no CPU fallback on this path.

Differences that are deliberate and documented:
  * ``MolEncoder.forward(x, eps=None)`` / ``Lambda``: the reparameterisation noise may be injected (parity tests);
    when omitted it is drawn exactly as the reference does (``scale * torch.randn`` on the CPU default generator,
    models.py:92) into pinned memory and copied asynchronously.
  * ``dtype=torch.bfloat16`` (default) stores the decoder LSTM's weights/activations in bf16 with fp32 accumulation,
    cell state and master weights (BASELINE.json configs[1]); ``dtype=torch.float32`` is the exact-f32 MFMA path.
    The encoder always runs in f32 (mu / logvar parity).
"""
import math
import os
import weakref

import torch
import torch.nn as nn

from . import _lib as L
from . import ops


def _pad(n, m):
    return (n + m - 1) // m * m


# ----------------------------------------------------------------------------------------------- parameter holders
class _NoCompute(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} only holds parameters; the arithmetic runs in the fused HIP path "
                           "of the enclosing MolEncoder / MolDecoder")


def _holder_forward(*a, **k):
    raise RuntimeError("this nn.Linear only holds parameters; the arithmetic runs in the fused HIP path of the enclosing module")


def LinearWeights(in_features, out_features):
    """weight [out,in], bias [out] in a REAL ``torch.nn.Linear`` (same initialisation and RNG consumption as the reference's modules), so
    that hooks keyed on the exact type -- ``model.apply(init_weights)`` with ``type(m) == nn.Linear``, moses_train_distrib_logp.py:48-51,262 --
    find these layers.  Its ``forward`` is blocked: no torch compute on this path."""
    m = nn.Linear(in_features, out_features)
    m.forward = _holder_forward
    return m


class Conv1dWeights(_NoCompute):
    """weight [out,in,k], bias [out]; initialised like nn.Conv1d."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_channels * kernel_size)
        nn.init.uniform_(self.bias, -bound, bound)


class EmbeddingWeights(_NoCompute):
    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)


class RNNWeights(_NoCompute):
    """weight_ih_l{k}, weight_hh_l{k}, bias_ih_l{k}, bias_hh_l{k} with torch.nn.LSTM/GRU names, shapes, order, init."""

    def __init__(self, mode, input_size, hidden_size, num_layers):
        super().__init__()
        self.mode, self.input_size, self.hidden_size, self.num_layers = mode, input_size, hidden_size, num_layers
        g = {"LSTM": 4, "GRU": 3}[mode] * hidden_size
        for l in range(num_layers):
            inp = input_size if l == 0 else hidden_size
            self.register_parameter(f"weight_ih_l{l}", nn.Parameter(torch.empty(g, inp)))
            self.register_parameter(f"weight_hh_l{l}", nn.Parameter(torch.empty(g, hidden_size)))
            self.register_parameter(f"bias_ih_l{l}", nn.Parameter(torch.empty(g)))
            self.register_parameter(f"bias_hh_l{l}", nn.Parameter(torch.empty(g)))
        stdv = 1.0 / math.sqrt(hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


class SELU(nn.Module):
    """Marker for scale*ELU_alpha (models.py:58-68); fused into the producing GEMM's epilogue."""

    def __init__(self, alpha=1.6732632423543772848170429916717, scale=1.0507009873554804934193349852946, inplace=False):
        super().__init__()
        self.alpha, self.scale = alpha, scale


class Softmax(nn.Module):
    """Marker for the class-axis softmax of models.py:157-158; fused into the output head."""


def ConvSELU(i, o, kernel_size=3, padding=0, p=0.):
    """models.py:71-77.  padding / dropout are not on the training path (always 0 there)."""
    if padding != 0 or p > 0.:
        raise NotImplementedError("ConvSELU: only padding=0, p=0 (the configuration models.py:118-120 uses)")
    return nn.Sequential(Conv1dWeights(i, o, kernel_size), SELU(inplace=True))


class Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


class Repeat(nn.Module):
    """models.py:13-26.  The fused decoder never materialises the repeat (time-invariant layer-0 input)."""

    def __init__(self, rep):
        super().__init__()
        self.rep = rep

    def forward(self, x):
        return x.unsqueeze(1).expand(x.size(0), self.rep, *x.shape[1:]).contiguous()


class TimeDistributed(nn.Module):
    """models.py:29-55 container (key ``decoded_mean.module.0.*``)."""

    def __init__(self, module, batch_first=True):
        super().__init__()
        self.module = module
        self.batch_first = batch_first


# ----------------------------------------------------------------------------------------------- workspace / packing
class _Workspace:
    """Named device buffers, allocated once per shape and reused every step (zero-initialised: pad regions stay 0)."""

    def __init__(self):
        self.bufs = {}
        self.generation = 0

    def get(self, name, shape, dtype, device):
        key = (name, tuple(shape), dtype, device)
        b = self.bufs.get(key)
        if b is None:
            b = torch.zeros(shape, dtype=dtype, device=device)
            self.bufs[key] = b
        return b


class _SavedState:
    """Mixin: where a module's forward keeps what its backward reads.  By default ONE workspace (`_ws`, which also holds the packed weight
    shadows): a forward overwrites the previous forward's saved state, so `fwd, fwd, bwd(first)` raises -- at that backward, because only
    then is it known that the first forward is still wanted (train.py:120-153 runs forward after forward with gradients enabled and never
    calls backward).  `saved_state_depth = n` keeps a ring of n workspaces (n x the activation memory): up to n forward passes of the SAME
    module may be outstanding, backpropagated in any order (micro-batches interleaved by hand, two losses from two forwards)."""

    def _init_saved_state(self):
        self._ws = _Workspace()
        self.__dict__["_ring"] = [self._ws]
        self.__dict__["_turn"] = 0

    @property
    def saved_state_depth(self):
        return len(self.__dict__["_ring"])

    @saved_state_depth.setter
    def saved_state_depth(self, n):
        ring = self.__dict__["_ring"]
        if n < 1:
            raise ValueError("saved_state_depth must be >= 1")
        while len(ring) < n:
            ring.append(_Workspace())
        del ring[n:]
        self.__dict__["_turn"] = 0
        self.__dict__["_half_turns"] = {}

    def _next_saved_ws(self, half=""):
        """(slot, workspace) -- with `half`: (slot, workspace, generation) -- for the forward that is starting; bumps that slot's generation.
        `half`: a module whose forward can
        run in independently callable halves (mosesvae.VAE: encoder / decoder) keeps one turn and one generation count per half -- the halves
        use disjoint buffer names of the same workspaces, so `forward_encoder` followed by `forward_decoder` overwrites nothing."""
        ring = self.__dict__["_ring"]
        if not half:
            self.__dict__["_turn"] = slot = (self.__dict__["_turn"] + 1) % len(ring)
            ring[slot].generation += 1
            return slot, ring[slot]
        turns = self.__dict__.setdefault("_half_turns", {})
        turns[half] = slot = (turns.get(half, 0) + 1) % len(ring)
        gens = ring[slot].__dict__.setdefault("half_generation", {})
        gens[half] = gens.get(half, 0) + 1
        return slot, ring[slot], gens[half]

    def _saved_ws(self, slot, gen, what, half=""):
        ring = self.__dict__["_ring"]
        now = None
        if slot < len(ring):
            now = ring[slot].__dict__.get("half_generation", {}).get(half) if half else ring[slot].generation
        if now != gen:
            raise L.MvaeError(f"{what}: the state this forward saved was overwritten by a later forward of the same module (or it ran under "
                              f"no_grad); run backward before the next forward, or set `module.saved_state_depth` to the number of forward "
                              f"passes kept outstanding (now {len(ring)})")
        return ring[slot]



_LDPAD = 64   # elements added to power-of-two leading dimensions of the big recurrent operands
def _dyk(n_out):
    """K extent of the output-gradient product dy = dl . W_out (see MolDecoder): the class / vocabulary count rounded up to whole pairs of
    64-deep K-steps (so that the 4-way split backward can halve it)."""
    return _pad(n_out, 128)


def _apply_and_mark(self, fn):
    """nn.Module.apply, then invalidate the packed weight shadows: initialisation hooks often write through ``p.data`` (``m.bias.data.fill_``,
    moses_train_distrib_logp.py:51), which torch's version counters do not see."""
    out = nn.Module.apply(self, fn)
    L.PARAM_EPOCH[0] += 1
    return out


def _require_cuda(dev, what):
    if dev.type != "cuda":
        raise L.MvaeError(f"{what} runs on the MI355X only (no CPU fallback); move the module and inputs to cuda")


def _grad_buffer(params, dev):
    """Zeroed flat fp32 buffer for the gradients of `params` (in order): the optimiser's own flat gradient range when the
    parameters are registered with a FusedAdam and no gradient is pending accumulation, else a fresh tensor."""
    sink = L.grad_sink_range(params) if all(p.grad is None for p in params) else None
    if sink is not None and sink[1].device == dev:
        g = sink[1][sink[2]:sink[3]]
        g.zero_()
        return g, sink
    return torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev), None


def _params_key(params):
    return (L.PARAM_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in params)


DW_KSPLIT = "1"      # K-chunks of the grouped decoder weight-gradient launches (see _lstm_weight_grads)
WGRAD_CAPS_BIG = "0:0"   # workgroup caps of the two parts of the large-batch schedule (0: one workgroup per tile)
WGRAD_CAP = "160"      # workgroups of the grouped weight-gradient launches at small per-GPU batches (0: one per tile, released next to the encoder LSTM backward)


def _lstm_weight_grads(ws, grads, prefix, dt, dev, NL, Lq, B, H, dG, ldg, hs, ldh, layers=None, max_workgroups=0, batch=None):
    """dW_ih, dW_hh, db of every layer from the saved pre-activation gradients dG [T*B, 4H] and layer outputs hs [T*B, H].
    Both dtypes contract straight from the K-major buffers (bf16: hardware-transposed LDS reads; f32: exact-f32 TN kernel)."""
    G4, TB = 4 * H, Lq * B
    f32 = torch.float32
    layers = range(NL) if layers is None else layers
    # operands beyond one 2 GiB buffer descriptor (configs[4]: T * B = 524288 rows of 8320 bytes) are cut into K-chunks of whole time steps that
    # accumulate -- the grouped full-K kernel then serves them too (round 5; before, that size fell back to one split-K GEMM per matrix: 0.81 PFLOP/s)
    ks_auto = max(1, -(-(TB * max(ldg, ldh) * 2) // ((1 << 31) - (1 << 24))))
    k_chunk = TB - B if ks_auto == 1 else ((TB // ks_auto + B - 1) // B) * B
    if dt == torch.bfloat16 and Lq > 1 and ops.gemm_tn_grouped_supported(dG[0], G4, H, k_chunk, ldg, ldh) and H == 1024 and L.knob("MVAE_DW_GROUPED", "1") != "0":
        # ONE grouped launch for every dW_ih / dW_hh of the requested layers (64 tiles of 256 x 256 each, accumulated over the full K = T*B in
        # registers: no split-K slabs, no reduction launch); the bias gradient (column sums of dG) rides along one GEMM per layer.
        # K-chunks (MVAE_DW_KSPLIT, default 1): the same tiles as `ks` shorter launches that accumulate -- a 256 x 256 tile owns its CU for
        # its whole K (3.1 ms at B = 1024), so shorter launches give the dispatcher points at which the other stream's kernels get CUs
        ks = max(ks_auto, int(L.knob("MVAE_DW_KSPLIT", DW_KSPLIT)))
        step = ((TB // ks + B - 1) // B) * B if ks > 1 else TB            # whole time steps per chunk
        for k0 in range(0, TB, step):
            k1 = min(TB, k0 + step)
            first = k0 == 0
            probs = []
            for l in layers:
                a = dG[l].view(TB, ldg)
                db = grads[f"{prefix}.bias_ih_l{l}"]
                if l > 0:
                    probs.append(dict(A=a[k0:k1], B=hs[l - 1].view(TB, ldh)[k0:k1], out=grads[f"{prefix}.weight_ih_l{l}"], M=G4, N=H, K=k1 - k0, lda=ldg, ldb=ldh,
                                      colsum_out=db, accumulate=not first, colsum_accumulate=not first))
                    lo = max(k0, B)                                        # dW_hh pairs dG[t] with h[t - 1]: rows shifted by one time step
                    if k1 > lo:
                        probs.append(dict(A=a[lo:k1], B=hs[l].view(TB, ldh)[lo - B:k1 - B], out=grads[f"{prefix}.weight_hh_l{l}"], M=G4, N=H, K=k1 - lo, lda=ldg,
                                          ldb=ldh, accumulate=not first))
                else:
                    if first:
                        ops.colsum_t(a[:B], B, G4, db, ldx=ldg)             # the first time step's rows (the shifted GEMM skips them)
                    lo = max(k0, B)
                    if k1 > lo:
                        probs.append(dict(A=a[lo:k1], B=hs[l].view(TB, ldh)[lo - B:k1 - B], out=grads[f"{prefix}.weight_hh_l{l}"], M=G4, N=H, K=k1 - lo, lda=ldg,
                                          ldb=ldh, colsum_out=db, colsum_accumulate=True, accumulate=not first))
            ops.gemm_tn_grouped(probs, max_workgroups=max_workgroups)
        for l in layers:
            grads[f"{prefix}.bias_hh_l{l}"].copy_(grads[f"{prefix}.bias_ih_l{l}"])
        return
    if dt == torch.bfloat16:
        for l in layers:
            a = dG[l].view(TB, ldg)
            x = hs[l].view(TB, ldh)
            db = grads[f"{prefix}.bias_ih_l{l}"]
            have_db = False
            if l > 0:       # bias gradient = column sums of dG: produced by the W_ih GEMM's pass over dG when the fused kernel serves the shape
                have_db = ops.gemm_tn_colsum(a, hs[l - 1].view(TB, ldh), grads[f"{prefix}.weight_ih_l{l}"], db, G4, H, TB, lda=ldg, ldb=ldh)
                if not have_db:
                    ops.gemm_tn(a, hs[l - 1].view(TB, ldh), grads[f"{prefix}.weight_ih_l{l}"], G4, H, TB, lda=ldg, ldb=ldh)
            if Lq > 1:
                if not have_db and ops.gemm_tn_colsum_supported(a, G4, H, TB - B):
                    ops.colsum_t(a[:B], B, G4, db, ldx=ldg)                         # the first time step's rows (the shifted GEMM skips them)
                    ops.gemm_tn_colsum(a[B:], x, grads[f"{prefix}.weight_hh_l{l}"], db, G4, H, TB - B, lda=ldg, ldb=ldh, colsum_accumulate=True)
                    have_db = True
                else:
                    ops.gemm_tn(a[B:], x, grads[f"{prefix}.weight_hh_l{l}"], G4, H, TB - B, lda=ldg, ldb=ldh)
            if not have_db:
                ops.colsum_t(a, TB, G4, db, ldx=ldg)
            grads[f"{prefix}.bias_hh_l{l}"].copy_(db)
        return
    if batch is not None and dt == torch.float32 and H % 64:      # the narrow exact-f32 stack (the encoder): every contraction joins the caller's batch
        for l in layers:                      # (the caller copies bias_ih -> bias_hh after batch.run())
            a = dG[l].view(TB, ldg)
            x = hs[l].view(TB, ldh)
            if Lq > 1:
                batch.add(a[B:], x, grads[f"{prefix}.weight_hh_l{l}"], G4, H, TB - B, lda=ldg, ldb=ldh)
            if l > 0:
                batch.add(a, hs[l - 1].view(TB, ldh), grads[f"{prefix}.weight_ih_l{l}"], G4, H, TB, lda=ldg, ldb=ldh, colsum_out=grads[f"{prefix}.bias_ih_l{l}"])
            else:       # layer 0 has no W_ih product here (its input is the token table): the bias gradient = column sums of dG over ALL T * B rows
                # rides on a 4-column product whose output nobody reads
                batch.add(a, x, ws.get(f"{prefix}_dw_dummy", (G4, 4), torch.float32, dev), G4, 4, TB, lda=ldg, ldb=ldh, colsum_out=grads[f"{prefix}.bias_ih_l{l}"])
        return
    for l in layers:                          # f32: the exact-f32 TN kernel, same K-major operands
        a = dG[l].view(TB, ldg)
        x = hs[l].view(TB, ldh)
        if Lq > 1:
            ops.gemm_tn(a[B:], x, grads[f"{prefix}.weight_hh_l{l}"], G4, H, TB - B, lda=ldg, ldb=ldh)
        if l > 0 and H % 64:   # the bias gradient (column sums of dG over all T*B rows) rides along this GEMM as its virtual ones column, in the
            # slack of the last 64-wide tile (H = 72).  Not when H is a whole number of tiles (the f32 decoder, H = 1024: a 17th tile column
            # turns 4 full rounds of workgroups into 4.25 -- 12.4 -> 20.0 ms per GEMM, measured)
            ops.gemm_tn_f32_colsum(a, hs[l - 1].view(TB, ldh), grads[f"{prefix}.weight_ih_l{l}"], grads[f"{prefix}.bias_ih_l{l}"], G4, H, TB, lda=ldg, ldb=ldh)
        else:
            if l > 0:
                ops.gemm_tn(a, hs[l - 1].view(TB, ldh), grads[f"{prefix}.weight_ih_l{l}"], G4, H, TB, lda=ldg, ldb=ldh)
            ops.colsum(a, TB, G4, grads[f"{prefix}.bias_ih_l{l}"], ldx=ldg)
        grads[f"{prefix}.bias_hh_l{l}"].copy_(grads[f"{prefix}.bias_ih_l{l}"])


_EPS_RINGS = weakref.WeakKeyDictionary()


class _HostNoise:
    """A pinned host block of noise handed to mvae_lambda_fwd by address; consumed(): the launch that reads it has been enqueued (records the
    event its ring slot waits on before the block is rewritten)."""

    def __init__(self, buf, slot):
        self.buf, self.slot = buf, slot

    def consumed(self):
        ev = torch.cuda.Event(); ev.record()
        self.slot[1] = ev


class Lambda(nn.Module):
    """models.py:80-94: z_mean / z_log_var heads + reparameterisation; caches ``.mu`` / ``.log_v``.

    ``noise`` says where the draw of models.py:92 (``scale * randn``) happens when no ``eps`` is injected:
      * ``"device"`` (default): inside ``mvae_lambda_fwd`` itself, from the library's counter hash of an explicit (seed, element counter) --
        ``self.noise_stream`` (ops.NoiseStream; ``seed_noise(seed)`` fixes it, the default seed follows ``torch.manual_seed``).  No host work,
        no generator state on the device.
      * ``"cpu"``: the reference's RNG stream -- the CPU default generator, value for value what ``torch.randn(B, o)`` returns there -- drawn
        into a small ring of pinned host buffers which ``mvae_lambda_fwd`` reads IN PLACE (device-mapped host memory; the launch leaves a device
        copy for backward): no pinned allocation and no copy command per step.
    """
    EPS_RING = 4

    def __init__(self, i=435, o=292, scale=1E-2, noise="device"):
        super().__init__()
        if noise not in ("device", "cpu"):
            raise ValueError("Lambda(noise=...): 'device' or 'cpu'")
        self.scale = scale
        self.noise = noise
        self.z_mean = LinearWeights(i, o)
        self.z_log_var = LinearWeights(i, o)
        self.__dict__["noise_stream"] = ops.NoiseStream()

    def seed_noise(self, seed, counter=0):
        """Fix the device noise stream: the same (seed, counter) reproduces the same draws bit for bit."""
        self.noise_stream.reseed(seed, counter)

    def draw_eps(self, B, o, device):
        """The "cpu" source.  models.py:92: scale * randn(*size) on the CPU default generator, then type_as(log_v)."""
        # normal_(0, scale) IS scale * randn value for value (same generator consumption, same rounding) and runs on the calling thread; the
        # separate `.mul_(scale)` of round 4 was a parallel element-wise op: it woke torch's whole OpenMP team (128 threads on the GPU box),
        # whose spin-waiting starved the thread that enqueues the step's kernels -- the +2.5 ... +11.7 ms per step of VERDICT r04 weak #2
        # (tests/tuning/host/eps_cost.py: 36.7 vs 27.9 ms per step at B = 1024; the H2D copy command itself cost nothing).
        if device.type != "cuda":
            return torch.empty(B, o).normal_(0.0, self.scale)
        ring = _EPS_RINGS.setdefault(self, {}).setdefault((B, o, device.index), dict(slots=[], n=0))     # pinned buffers + events: not module state
        k = ring["n"] % self.EPS_RING
        ring["n"] += 1
        if k == len(ring["slots"]):
            ring["slots"].append([torch.empty(B, o, pin_memory=True), None])
        buf, ev = ring["slots"][k]
        if ev is not None:
            ev.synchronize()                    # the copy that read this buffer EPS_RING draws ago
        buf.normal_(0.0, self.scale)
        return _HostNoise(buf, ring["slots"][k])

    def forward(self, x, eps=None):
        from .functional import lambda_forward
        z, self.mu, self.log_v = lambda_forward(self, x, eps)
        return z, self.mu, self.log_v


# ----------------------------------------------------------------------------------------------- encoder
class MolEncoder(nn.Module, _SavedState):
    """models.py:109-135: Embedding -> LSTM(30->72, 3 layers) -> 3 x (Conv1d k=18 + SELU) -> Linear+SELU -> Lambda."""

    def __init__(self, i=120, o=292, c=35, word_embedding_size=30, h_size=72, num_lstm=3):
        super().__init__()
        self.i = i
        self.embedding = EmbeddingWeights(c, word_embedding_size)
        self.gru = RNNWeights("LSTM", word_embedding_size, h_size, num_lstm)      # attribute is named gru, is an LSTM (models.py:117)
        self.conv_1 = ConvSELU(i, 120, kernel_size=18)
        self.conv_2 = ConvSELU(120, 64, kernel_size=18)
        self.conv_3 = ConvSELU(64, 64, kernel_size=18)
        self.dense_1 = nn.Sequential(LinearWeights((h_size - (18 * 3) + 3) * 64, 512), SELU(inplace=True))
        self.lmbd = Lambda(512, o)
        self._init_saved_state()
        self.__dict__["_fork"] = ops.ForkState()   # side-stream work of the decoder paired with us (MolecularVAE shares one object between the two)
        self.fast_grad_gemms = False      # set by MolecularVAE in bf16 mode (conv input-gradient GEMMs as 3 x bf16 products)
        self._pack_key = None
        self._packed = {}

    apply = _apply_and_mark

    def forward(self, x, eps=None):
        B = x.shape[0]
        o = self.lmbd.z_mean.out_features
        if eps is None and not (self.lmbd.noise == "device" and x.is_cuda):
            eps = self.lmbd.draw_eps(B, o, x.device)
        params = list(self.parameters())
        # a training forward under FusedAdam: launches with bounded spins report a failure into the optimiser's poison slot (ops.PERSIST_DEFAULT)
        self.__dict__["_poison"] = L.grad_poison(params) if torch.is_grad_enabled() else None
        z, mu, logv = _EncoderFn.apply(self, x, eps, *params)        # eps None: drawn inside mvae_lambda_fwd from lmbd.noise_stream
        self.lmbd.mu, self.lmbd.log_v = mu, logv
        return z, mu, logv

    # -- packed weight shadows (refreshed when any parameter changed): ONE multi-tensor pack launch (ops.PackList) + the three conv packs
    def _pack(self, dev):
        params = list(self.parameters())
        key = _params_key(params)
        if key == self._pack_key:
            return self._packed
        ptrs = (dev,) + tuple((id(p), p.data_ptr()) for p in params)     # id: a deepcopy must rebuild its own job table
        if self.__dict__.get("_pack_ptrs") != ptrs:
            with torch.no_grad():                      # the job table keeps plain (non-autograd) views of the parameters
                self._build_pack(dev)
            self.__dict__["_pack_ptrs"] = ptrs
        with torch.no_grad():
            self.__dict__["_pack_list"].run()
            for n, conv in (("c1", self.conv_1[0]), ("c2", self.conv_2[0]), ("c3", self.conv_3[0])):
                Ci, Co, k = conv.in_channels, conv.out_channels, conv.kernel_size
                ops.conv1d_pack_weights(conv.weight, Ci, Co, k, _pad(Ci, 32), self._packed[n + "_wp"], _pad(Co, 32), self._packed[n + "_wq"])
        self._pack_key = key
        return self._packed

    def _build_pack(self, dev):
        """Allocate the shadows (zeroed: padding is never written) and record the jobs that fill them, each straight from a parameter."""
        g, ws = self.gru, self._ws
        H, NL, E, Cv = g.hidden_size, g.num_layers, g.input_size, self.embedding.num_embeddings
        P, pl = {}, ops.PackList()
        Ep = _pad(E, 4)
        P["E_p"] = ws.get("E_p", (Cv, Ep), torch.float32, dev)
        P["ET"] = ws.get("ET", (E, _pad(Cv, 4)), torch.float32, dev)
        pl.cast_transpose(self.embedding.weight, Cv, E, dst=P["E_p"], dstT=P["ET"])
        P["Wih0_p"] = ws.get("Wih0_p", (4 * H, Ep), torch.float32, dev)
        P["Wih0T"] = ws.get("Wih0T", (E, 4 * H), torch.float32, dev)
        pl.cast_transpose(g.weight_ih_l0, 4 * H, E, dst=P["Wih0_p"], dstT=P["Wih0T"])
        P["bias"] = []
        P["WihT"], P["WhhT"] = [None], []
        Hp = _pad(H, 32)                       # whole 128-byte K-steps (f32): zero-padded shadows -> LDS-direct main loop
        P["Hp"] = Hp
        P["Wih"], P["Whh"] = [None], []
        for l in range(NL):
            b = ws.get(f"bias{l}", (4 * H,), torch.float32, dev)
            pl.add(getattr(g, f"bias_ih_l{l}"), getattr(g, f"bias_hh_l{l}"), b)
            P["bias"].append(b)
            w = ws.get(f"Whh{l}", (4 * H, Hp), torch.float32, dev)
            t = ws.get(f"WhhT{l}", (H, 4 * H), torch.float32, dev)
            pl.cast_transpose(getattr(g, f"weight_hh_l{l}"), 4 * H, H, dst=w, dstT=t)
            P["Whh"].append(w); P["WhhT"].append(t)
            if l > 0:
                w = ws.get(f"Wih{l}", (4 * H, Hp), torch.float32, dev)
                t = ws.get(f"WihT{l}", (H, 4 * H), torch.float32, dev)
                pl.cast_transpose(getattr(g, f"weight_ih_l{l}"), 4 * H, H, dst=w, dstT=t)
                P["Wih"].append(w); P["WihT"].append(t)
        # conv stack, channels-last with channel counts padded to whole K-steps: packed weights for the sliding-window GEMMs
        for n, conv in (("c1", self.conv_1[0]), ("c2", self.conv_2[0]), ("c3", self.conv_3[0])):
            Ci, Co, k = conv.in_channels, conv.out_channels, conv.kernel_size
            ldx, ldo = _pad(Ci, 32), _pad(Co, 32)
            P[n + "_wp"] = ws.get(n + "_wp", (Co, k * ldx), torch.float32, dev)
            P[n + "_wq"] = ws.get(n + "_wq", (Ci, k * ldo), torch.float32, dev)
        d1 = self.dense_1[0]
        P["W1T"] = ws.get("W1T", (d1.in_features, d1.out_features), torch.float32, dev)
        pl.cast_transpose(d1.weight, d1.out_features, d1.in_features, dstT=P["W1T"])
        o = self.lmbd.z_mean.out_features
        P["Wml"] = ws.get("Wml", (2 * o, 512), torch.float32, dev)
        P["bml"] = ws.get("bml", (2 * o,), torch.float32, dev)
        P["WmlT"] = ws.get("WmlT", (512, _pad(2 * o, 4)), torch.float32, dev)
        for k, lin in enumerate((self.lmbd.z_mean, self.lmbd.z_log_var)):           # the stacked mu | logvar head and its transpose, each from the parameter
            pl.cast_transpose(lin.weight, o, 512, dst=P["Wml"][k * o:(k + 1) * o], dstT=P["WmlT"][:, k * o:(k + 1) * o])
            pl.copy(lin.bias, P["bml"][k * o:(k + 1) * o])
        self._packed = P
        self.__dict__["_pack_list"] = pl


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    @ops.traced("molvae_encoder_fwd")
    def forward(ctx, mod, idx, eps, *params):
        dev = idx.device
        _require_cuda(dev, "MolEncoder")
        idx = idx.contiguous()
        if eps is not None and not isinstance(eps, _HostNoise):
            eps = eps.contiguous().float().to(dev)
        B, Lq = idx.shape
        if Lq != mod.i:
            raise ValueError(f"sequence length {Lq} != encoder i={mod.i}")
        g = mod.gru
        H, NL, Cv = g.hidden_size, g.num_layers, mod.embedding.num_embeddings
        o = mod.lmbd.z_mean.out_features
        P = mod._pack(dev)
        slot, ws = mod._next_saved_ws()
        f32 = torch.float32
        G4 = 4 * H
        # K1: embedding folded into the layer-0 input projection: table = E . W_ih0^T + (b_ih0 + b_hh0)
        tbl = ws.get("tbl", (Cv, G4), f32, dev)
        Ep = P["E_p"].shape[1]
        ops.gemm_nt(P["E_p"], P["Wih0_p"], tbl, Cv, G4, Ep, bias=P["bias"][0])
        # (the gathered [T, B, 4H] sequence -- 141 MB at B = 1024 -- is never formed: layer 0 of the row-resident kernel keeps the 40 KB table
        # and its rows' ids in LDS; add_table / add_index of mvae_rnn_fwd_desc)
        # K2: 3-layer LSTM, f32 MFMA
        Hp = P["Hp"]
        hs = [ws.get(f"hs{l}", (Lq, B, Hp), f32, dev) for l in range(NL)]      # rows zero-padded to whole K-steps
        cs = [ws.get(f"cs{l}", (Lq, B, H), f32, dev) for l in range(NL)]
        gates = [ws.get(f"gates{l}", (Lq, B, G4), f32, dev) for l in range(NL)]
        cstate = [ws.get(f"cstate{l}", (2, B, H), f32, dev) for l in range(NL)]
        ops.rnn_fwd(L.CELL_LSTM, f32, Lq, B, H, None, 0, P["Wih"], [Hp] * NL, P["Whh"], [Hp] * NL, [None] + P["bias"][1:],
                    hs, Hp, cs, gates, cstate, zero_padded_k=True, tag="enc_lstm_fwd", add_table=tbl, add_index=idx,
                    poison=mod.__dict__.get("_poison"))
        # K3: conv stack over the hidden axis, sequence position = channel (models.py:129-131)
        c1, c2, c3 = mod.conv_1[0], mod.conv_2[0], mod.conv_3[0]
        k = c1.kernel_size
        W1, W2, W3 = H - k + 1, H - 2 * k + 2, H - 3 * k + 3
        # channels-last activations [b][w][c] with zero pad channels; conv_1 reads the LSTM output as x1[b][w][t] = hs[t][b][w]
        L1, O1, O2, O3 = _pad(Lq, 32), _pad(c1.out_channels, 32), _pad(c2.out_channels, 32), _pad(c3.out_channels, 32)
        if O3 != c3.out_channels:
            raise L.MvaeError("MolEncoder: conv_3 channel count must be a multiple of 32")
        x1 = ws.get("x1", (B * Hp, L1), f32, dev)
        ops.cast_transpose(hs[-1], Lq, B * Hp, dstT=x1, lds=B * Hp)
        y1 = ws.get("y1", (B * W1, O1), f32, dev)
        ops.conv1d_selu_fwd(x1, B, H, L1, Hp * L1, c1.out_channels, k, P["c1_wp"], c1.bias, y1, O1)
        y2 = ws.get("y2", (B * W2, O2), f32, dev)
        ops.conv1d_selu_fwd(y1, B, W1, O1, W1 * O1, c2.out_channels, k, P["c2_wp"], c2.bias, y2, O2)
        y3 = ws.get("y3", (B * W3, O3), f32, dev)
        ops.conv1d_selu_fwd(y2, B, W2, O2, W2 * O2, c3.out_channels, k, P["c3_wp"], c3.bias, y3, O3)
        # Flatten is channel-major (models.py:6-10): flat[b, co*W3 + w] = y3[(b*W3 + w), co]
        C3 = c3.out_channels
        flat = ws.get("flat", (B, C3 * W3), f32, dev)
        ops.permute021(y3, flat, B, W3, C3)
        # K4: dense + SELU
        d1 = mod.dense_1[0]
        d = ws.get("d", (B, d1.out_features), f32, dev)
        ops.gemm_nt(flat, d1.weight, d, B, d1.out_features, d1.in_features, bias=d1.bias, act=L.ACT_SELU)
        # K5: stacked heads + reparameterisation
        mulv = ws.get("mulv", (B, 2 * o), f32, dev)
        ops.gemm_nt(d, P["Wml"], mulv, B, 2 * o, 512, bias=P["bml"])
        z = torch.empty(B, o, dtype=f32, device=dev); mu = torch.empty_like(z); logv = torch.empty_like(z)
        if eps is None:                     # the draw of models.py:92 inside the launch; the drawn block stays in the saved state for backward
            seed, off = mod.lmbd.noise_stream.take(B * o)
            eps = ws.get("eps", (B, o), f32, dev)
            ops.lambda_fwd(mulv, None, z, mu, logv, B, o, scale=mod.lmbd.scale, seed=seed, offset=off, eps_out=eps)
        elif isinstance(eps, _HostNoise):   # the "cpu" source: the launch reads the pinned host block in place and leaves a device copy
            host, eps = eps, ws.get("eps", (B, o), f32, dev)
            ops.lambda_fwd(mulv, host.buf, z, mu, logv, B, o, eps_out=eps)
            host.consumed()
        else:
            ops.lambda_fwd(mulv, eps, z, mu, logv, B, o)
        ctx.mod, ctx.slot, ctx.gen, ctx.idx, ctx.eps = mod, slot, ws.generation, idx, eps
        return z, mu, logv

    @staticmethod
    @ops.traced("molvae_encoder_bwd")
    def backward(ctx, dz, dmu, dlogv):
        mod, idx, eps = ctx.mod, ctx.idx, ctx.eps
        ws = mod._saved_ws(ctx.slot, ctx.gen, "MolEncoder")
        fork = mod.__dict__["_fork"]
        dev = idx.device
        f32 = torch.float32
        B, Lq = idx.shape
        g = mod.gru
        H, NL, Cv, E = g.hidden_size, g.num_layers, mod.embedding.num_embeddings, g.input_size
        o = mod.lmbd.z_mean.out_features
        G4, TB = 4 * H, Lq * B
        Bp = _pad(B, 4)
        P = mod._packed
        params = list(mod.parameters())
        names = [n for n, _ in mod.named_parameters()]
        gflat, _ = _grad_buffer(params, dev)
        grads, off = {}, 0
        for n, p in zip(names, params):
            grads[n] = gflat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        dz = dz.contiguous() if dz is not None else None
        dmu = dmu.contiguous() if dmu is not None else None
        dlogv = dlogv.contiguous() if dlogv is not None else None
        W = lambda name, shape: ws.get(name, shape, f32, dev)
        # K5 backward
        mulv, d = W("mulv", (B, 2 * o)), W("d", (B, 512))
        dmulv = W("dmulv", (B, 2 * o))
        ops.lambda_bwd(mulv, eps, dz, dmu, dlogv, dmulv, B, o)
        # dW = dy^T . x straight from the batch-major operands (exact-f32 TN kernel), db = column sums of dy as the GEMM's virtual ones column;
        # one GEMM per head (the column halves of dmulv), written into the parameters' own gradient slots
        # Every GEMM that only produces PARAMETER gradients (heads, dense_1, the three convolutions, the LSTM stack: a dozen small contractions
        # nobody but the optimiser waits for) is collected and runs as ONE launch at the end of this backward pass (ops.TnF32Batch); the launches
        # left on the way are the input-gradient chain.  MVAE_ENC_DW_BATCH=0: one launch each, where it stands (A/B, tests).
        # (measured, ms per step: b = 128 5.47 -> 5.35, B = 512 17.14 -> 16.99; B = 1024 28.28 -> 28.66 -- there each of these GEMMs is big enough to
        # fill its launch and, issued where it stands, runs under the decoder's weight-gradient launches instead of behind the LSTM backward)
        bmode = L.knob("MVAE_ENC_DW_BATCH", "1" if B * Lq <= 65536 else "0")      # "2": only the LSTM stack's products + the table gradient (the tail)
        lstm_batch = ops.TnF32Batch(dev) if bmode != "0" else None
        batch = lstm_batch if bmode == "1" else None
        for k, nm in enumerate(("lmbd.z_mean", "lmbd.z_log_var")):
            if batch is not None:
                batch.add(dmulv[:, k * o:(k + 1) * o], d, grads[nm + ".weight"], o, 512, B, lda=2 * o, ldb=512, colsum_out=grads[nm + ".bias"])
            else:
                ops.gemm_tn_f32_colsum(dmulv[:, k * o:(k + 1) * o], d, grads[nm + ".weight"], grads[nm + ".bias"], o, 512, B, lda=2 * o, ldb=512)
        dd = W("dd", (B, 512))
        ops.gemm_nt(dmulv, P["WmlT"], dd, B, 512, 2 * o, ldb=P["WmlT"].stride(0))
        ops.selu_bwd(dd, d)
        # K4 backward
        d1 = mod.dense_1[0]
        F = d1.in_features
        flat = W("flat", (B, F))
        if batch is not None:
            batch.add(dd, flat, grads["dense_1.0.weight"], 512, F, B, colsum_out=grads["dense_1.0.bias"])
        else:
            ops.gemm_tn_f32_colsum(dd, flat, grads["dense_1.0.weight"], grads["dense_1.0.bias"], 512, F, B)
        dflat = W("dflat", (B, F))
        ops.gemm_nt(dd, P["W1T"], dflat, B, F, 512)
        fork.run_deferred(0)      # the decoder's upper-layer weight-gradient GEMMs: from here on our own launches are chip-filling too
        # K3 backward
        c1, c2, c3 = mod.conv_1[0], mod.conv_2[0], mod.conv_3[0]
        k = c1.kernel_size
        W1, W2, W3 = H - k + 1, H - 2 * k + 2, H - 3 * k + 3
        C3 = c3.out_channels
        dy3 = W("dy3", (B * W3, C3))
        ops.permute021(dflat, dy3, B, C3, W3)
        Hp = P["Hp"]
        L1, O1, O2 = _pad(Lq, 32), _pad(c1.out_channels, 32), _pad(c2.out_channels, 32)
        x1, y1, y2, y3 = W("x1", (B * Hp, L1)), W("y1", (B * W1, O1)), W("y2", (B * W2, O2)), W("y3", (B * W3, C3))
        dy2, dy1, dx1 = W("dy2", (B * W2, O2)), W("dy1", (B * W1, O1)), W("dx1", (B * H, L1))
        dzp3, dzp2, dzp1 = W("dzp3", (B * (W3 + 2 * k - 2), C3)), W("dzp2", (B * (W2 + 2 * k - 2), O2)), W("dzp1", (B * (W1 + 2 * k - 2), O1))
        # bf16 training mode: input-gradient GEMMs as 3 x bf16 products (forward conv stays exact fp32); MVAE_CONV_X3=0: A/B knob
        x3 = bool(mod.fast_grad_gemms) and L.knob("MVAE_CONV_X3", "1") != "0"
        for (n, Wi, Ci, ldx_, xbs, Co, ldo_, dy_, y_, x_, dzp_, dx_, lddx_) in (
                ("3", W2, c3.in_channels, O2, W2 * O2, C3, C3, dy3, y3, y2, dzp3, dy2, O2),
                ("2", W1, c2.in_channels, O1, W1 * O1, c2.out_channels, O2, dy2, y2, y1, dzp2, dy1, O1),
                ("1", H, Lq, L1, Hp * L1, c1.out_channels, O1, dy1, y1, x1, dzp1, dx1, L1)):
            gw, gb = grads[f"conv_{n}.0.weight"], grads[f"conv_{n}.0.bias"]
            if batch is not None:           # dz (padded) + the input gradient now; the weight / bias gradient joins the batch
                ops.conv1d_selu_bwd(B, Wi, Ci, ldx_, xbs, Co, ldo_, k, dy_, y_, x_, P[f"c{n}_wq"], dzp_, None, None, dx_, lddx_, x3=x3)
                batch.add_conv_dw(B, Wi, Ci, ldx_, xbs, Co, ldo_, k, dzp_, x_, W(f"dwp{n}", (Co, k * ldx_)), gw, gb, x3=x3)
            else:
                ops.conv1d_selu_bwd(B, Wi, Ci, ldx_, xbs, Co, ldo_, k, dy_, y_, x_, P[f"c{n}_wq"], dzp_, gw, gb, dx_, lddx_, x3=x3)
        dhs = W("dhs", (Lq, B, H))
        ops.cast_transpose(dx1, B * H, Lq, dstT=dhs.view(Lq, B * H), lds=L1)          # dhs[t][b][w] = dx1[(b, w)][t]
        # K2 backward (reverse wavefront) + weight gradients
        hs = [W(f"hs{l}", (Lq, B, Hp)) for l in range(NL)]
        cs = [W(f"cs{l}", (Lq, B, H)) for l in range(NL)]
        gates = [W(f"gates{l}", (Lq, B, G4)) for l in range(NL)]
        dG = [W(f"dG{l}", (Lq, B, G4)) for l in range(NL)]
        dstate = [W(f"dstate{l}", (2, B, H)) for l in range(NL)]
        fork.run_deferred(1)      # the decoder's remaining weight-gradient GEMMs: they fill the CUs the row-resident backward leaves idle
        ops.rnn_bwd(L.CELL_LSTM, f32, Lq, B, H, P["WhhT"], [G4] * NL, P["WihT"], [G4] * NL, dhs, H, hs, Hp, cs, gates,
                    dG, dstate, tag="enc_lstm_bwd", poison=L.grad_poison(params))
        batch = lstm_batch
        _lstm_weight_grads(ws, grads, "gru", f32, dev, NL, Lq, B, H, dG, G4, hs, Hp, batch=batch)
        # K1 backward: table gradient, then embedding / W_ih0
        dtbl = W("dtbl", (Cv, G4))
        if batch is not None:
            # dtbl[v] = sum of dG0[t, b] over the positions holding token v = onehot(idx)^T . dG0: one more problem of the batch (exact-f32 sums in
            # a fixed order).  The one-hot rows follow idx's own order (row b * L + t); the matching row of dG0 [T, B, 4H] is reached through the
            # B operand's row groups (group = L, stride between groups = one batch row, stride inside = one time step).
            oh = W("idx_onehot", (B * Lq, _pad(Cv, 4)))
            ops.onehot_f32(idx, oh, Cv)
            batch.add(oh, dG[0], dtbl, Cv, G4, B * Lq, lda=oh.stride(0), ldb=B * G4, b_group=Lq, b_gstride=G4)
            batch.run()
            for l in range(NL):
                grads[f"gru.bias_hh_l{l}"].copy_(grads[f"gru.bias_ih_l{l}"])
        else:
            ops.scatter_rows_tb(idx, dG[0], dtbl, B, Lq, Cv, G4)
        ops.gemm_nt(dtbl, P["Wih0T"], grads["embedding.weight"], Cv, E, G4)
        Cp = _pad(Cv, 4)
        dtblT = W("dtblT", (G4, Cp))
        ops.cast_transpose(dtbl, Cv, G4, dstT=dtblT)
        ops.gemm_nt(dtblT, P["ET"], grads["gru.weight_ih_l0"], G4, E, Cp)
        fork.join()               # the decoder's weight-gradient GEMMs that ran on the side stream under this backward
        return (None, None, None) + tuple(grads[n] for n in names)


# ----------------------------------------------------------------------------------------------- decoder
class MolDecoder(nn.Module, _SavedState):
    """models.py:148-165: Linear+SELU -> repeat L -> LSTM(292->1024, 4 layers) -> Linear(1024,C) -> softmax over C."""

    def __init__(self, i=292, o=120, c=35, num_gru=4, h_size=1024, dtype=torch.bfloat16):
        super().__init__()
        self.latent_input = nn.Sequential(LinearWeights(i, i), SELU(inplace=True))
        self.repeat_vector = Repeat(o)
        self.gru = RNNWeights("LSTM", i, h_size, num_gru)                           # models.py:156: named gru, is an LSTM
        self.decoded_mean = TimeDistributed(nn.Sequential(LinearWeights(h_size, c), Softmax()))
        self.compute_dtype = dtype
        self._init_saved_state()
        self._pack_key = None
        self._packed = {}
        self.__dict__["_peer"] = None          # weakref to the encoder whose backward runs after ours (set by MolecularVAE)
        self.__dict__["_z_from_peer"] = False  # True only inside MolecularVAE.forward: z is that encoder's output
        self.__dict__["_side"] = None
        self.overlap_weight_grads = True       # run the weight-gradient GEMMs on a side stream under the encoder's backward

    apply = _apply_and_mark

    def _side_stream(self, dev):
        return ops.side_stream(dev)       # process-wide, probed not to share a hardware queue with the main stream (stream priorities: no effect)

    def forward(self, x):
        # under no_grad (evaluation, train.py:120-153 / sampling from a latent, train_sample.py:32) nothing is saved for backward
        infer = not torch.is_grad_enabled()
        return _DecoderFn.apply(self, x, infer, *list(self.parameters()))

    def _pack(self, dev):
        params = list(self.parameters())
        key = _params_key(params) + (self.compute_dtype,)
        if key == self._pack_key:
            return self._packed
        ptrs = (dev, self.compute_dtype) + tuple((id(p), p.data_ptr()) for p in params)     # id: a deepcopy must rebuild its own job table
        if self.__dict__.get("_pack_ptrs") != ptrs:
            with torch.no_grad():                      # the job table keeps plain (non-autograd) views of the parameters
                self._build_pack(dev)
            self.__dict__["_pack_ptrs"] = ptrs
        with torch.no_grad():
            self.__dict__["_pack_list"].run()          # every shadow in ONE launch (ops.PackList)
        self._pack_key = key
        return self._packed

    def _build_pack(self, dev):
        g, ws, dt = self.gru, self._ws, self.compute_dtype
        H, NL, o = g.hidden_size, g.num_layers, g.input_size
        G4 = 4 * H
        f32 = torch.float32
        ldw, ldwT = H + _LDPAD, G4 + _LDPAD       # leading dimensions kept off powers of two (L2 / MALL set conflicts)
        P = {"Wih": [None], "WihT": [None], "Whh": [], "WhhT": [], "bias": [], "ldw": ldw, "ldwT": ldwT}
        pl = ops.PackList()
        li = self.latent_input[0]
        P["WliT"] = ws.get("WliT", (o, _pad(o, 4)), f32, dev)
        pl.cast_transpose(li.weight, o, o, dstT=P["WliT"])
        P["Wih0T"] = ws.get("Wih0T", (o, G4), f32, dev)
        pl.cast_transpose(g.weight_ih_l0, G4, o, dstT=P["Wih0T"])
        for l in range(NL):
            b = ws.get(f"bias{l}", (G4,), f32, dev)
            pl.add(getattr(g, f"bias_ih_l{l}"), getattr(g, f"bias_hh_l{l}"), b)
            P["bias"].append(b)
            w = ws.get(f"Whh{l}", (G4, ldw), dt, dev); wT = ws.get(f"WhhT{l}", (H, ldwT), dt, dev)
            pl.cast_transpose(getattr(g, f"weight_hh_l{l}"), G4, H, dst=w, dstT=wT)
            P["Whh"].append(w); P["WhhT"].append(wT)
            if l > 0:
                w = ws.get(f"Wih{l}", (G4, ldw), dt, dev); wT = ws.get(f"WihT{l}", (H, ldwT), dt, dev)
                pl.cast_transpose(getattr(g, f"weight_ih_l{l}"), G4, H, dst=w, dstT=wT)
                P["Wih"].append(w); P["WihT"].append(wT)
        om = self.decoded_mean.module[0]
        Cv = om.out_features
        Cp = _pad(Cv, 8)
        P["Wout"] = ws.get("Wout", (Cv, H), dt, dev)
        # bf16: W_out^T zero-padded to _dyk(C) columns -- the backward contracts the logit gradients with it inside the top LSTM cell
        P["WoutT"] = ws.get("WoutT", (H, _dyk(Cv) if dt == torch.bfloat16 else Cp), dt, dev)
        pl.cast_transpose(om.weight, Cv, H, dst=P["Wout"], dstT=P["WoutT"])
        self._packed = P
        self.__dict__["_pack_list"] = pl


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    @ops.traced("molvae_decoder_fwd")
    def forward(ctx, mod, z, infer, *params):
        dev = z.device
        _require_cuda(dev, "MolDecoder")
        z = z.contiguous().float()
        g, dt = mod.gru, mod.compute_dtype
        H, NL, o = g.hidden_size, g.num_layers, g.input_size
        B = z.shape[0]
        Lq = mod.repeat_vector.rep
        om = mod.decoded_mean.module[0]
        Cv = om.out_features
        G4, TB = 4 * H, Lq * B
        P = mod._pack(dev)
        slot, ws = mod._next_saved_ws()
        f32 = torch.float32
        li_mod = mod.latent_input[0]
        # K6: latent projection + SELU; layer-0 input is time-invariant -> its gate pre-activation is computed ONCE
        li = ws.get("li", (B, o), f32, dev)
        ops.gemm_nt(z, li_mod.weight, li, B, o, o, bias=li_mod.bias, act=L.ACT_SELU)
        gx0 = ws.get("gx0", (B, G4), f32, dev)
        ops.gemm_nt(li, g.weight_ih_l0, gx0, B, G4, o, bias=P["bias"][0])
        # K7: 4-layer LSTM wavefront
        ldh = H + _LDPAD
        hs = [ws.get(f"hs{l}", (Lq, B, ldh), dt, dev) for l in range(NL)]
        cs = [None if infer else ws.get(f"cs{l}", (Lq, B, H), dt, dev) for l in range(NL)]
        gates = [None if infer else ws.get(f"gates{l}", (Lq, B, G4), dt, dev) for l in range(NL)]
        cstate = [ws.get(f"cstate{l}", (2, B, H), f32, dev) for l in range(NL)]
        ops.rnn_fwd(L.CELL_LSTM, dt, Lq, B, H, gx0, 0, P["Wih"], [P["ldw"]] * NL, P["Whh"], [P["ldw"]] * NL, [None] + P["bias"][1:],
                    hs, ldh, cs, gates, cstate, tag="dec_lstm_fwd", poison=(None if infer else L.grad_poison(params)))
        # K8: output head + softmax over the class axis
        logits = ws.get("logits", (TB, Cv), f32, dev)
        ops.gemm_nt(hs[-1].view(TB, ldh), P["Wout"], logits, TB, Cv, H, bias=om.bias)
        recon = torch.empty(B, Lq, Cv, dtype=f32, device=dev)
        with ops._Timed("hbm_softmax_fwd"):
            ops.softmax_tb_fwd(logits, Cv, recon, B, Lq, Cv)
        ctx.mod, ctx.slot, ctx.gen, ctx.z = mod, slot, (-1 if infer else ws.generation), z     # a forward-only pass saved nothing: backward refuses
        ctx.from_peer = bool(mod.__dict__.get("_z_from_peer", False))
        ctx.save_for_backward(recon)
        return recon

    @staticmethod
    @ops.traced("molvae_decoder_bwd")
    def backward(ctx, drecon):
        mod, z = ctx.mod, ctx.z
        (recon,) = ctx.saved_tensors
        ws = mod._saved_ws(ctx.slot, ctx.gen, "MolDecoder")
        dev = z.device
        f32 = torch.float32
        g, dt = mod.gru, mod.compute_dtype
        H, NL, o = g.hidden_size, g.num_layers, g.input_size
        B = z.shape[0]
        Lq = mod.repeat_vector.rep
        om = mod.decoded_mean.module[0]
        Cv = om.out_features
        G4, TB = 4 * H, Lq * B
        Cp = _pad(Cv, 8)
        Bp = _pad(B, 4)
        P = mod._packed
        params = list(mod.parameters())
        names = [n for n, _ in mod.named_parameters()]
        gflat, sink = _grad_buffer(params, dev)
        grads, off, offs = {}, 0, {}
        for n, p in zip(names, params):
            grads[n] = gflat[off:off + p.numel()].view(p.shape)
            offs[n] = off
            off += p.numel()
        drecon = drecon.contiguous().float()
        W = lambda name, shape, d=f32: ws.get(name, shape, d, dev)
        ldh, ldg = H + _LDPAD, G4 + _LDPAD
        hs = [W(f"hs{l}", (Lq, B, ldh), dt) for l in range(NL)]
        cs = [W(f"cs{l}", (Lq, B, H), dt) for l in range(NL)]
        gates = [W(f"gates{l}", (Lq, B, G4), dt) for l in range(NL)]
        # K8 backward
        # bf16 (and a hidden size the LDS-direct loop serves): the top LSTM cell contracts dl . W_out itself (no [T, B, H] fp32 dy tensor)
        # The weights-resident dataflow backward (rnn_persist_bwd.hip: the per-rank shape of the 8-GPU configuration) wants dy as a tensor: one
        # [T*B, H] GEMM in front of it instead of the fused segment.
        fuse_ok = dt == torch.bfloat16 and (4 * H) % 64 == 0
        use_pb = fuse_ok and ops.rnn_bwd_persist_wanted(L.CELL_LSTM, dt, NL, B, H, ldg, dev)
        fuse_dy = fuse_ok and not use_pb
        ldl = _dyk(Cv) if fuse_ok else Cp           # pad columns are allocated zero and never written
        dl = W("dl", (TB + 8, ldl), dt)[:TB]        # +8 rows: the TN tile reads 256-byte row segments past the last row
        dlT = None
        if dt != torch.bfloat16:
            ldT = _pad(TB, 8) + 8
            dlT = W("dlT", (Cv, ldT), dt)
        with ops._Timed("hbm_softmax_bwd"):
            ops.softmax_tb_bwd(recon, drecon, dl, dlT, B, Lq, Cv)
        dy = None
        if not fuse_dy:
            dy = W("dy", (TB, H))
            ops.gemm_nt(dl, P["WoutT"], dy, TB, H, ldl if fuse_ok else Cp)
        # K7 backward
        dG = [W(f"dG{l}", (Lq, B, ldg), dt) for l in range(NL)]
        dstate = [W(f"dstate{l}", (2, B, H)) for l in range(NL)]
        ops.rnn_bwd(L.CELL_LSTM, dt, Lq, B, H, P["WhhT"], [P["ldwT"]] * NL, P["WihT"], [P["ldwT"]] * NL, dy, H, hs, ldh, cs, gates,
                    dG, dstate, ldg=ldg, tag="dec_lstm_bwd", dy_a=(dl if fuse_dy else None), dy_w=(P["WoutT"] if fuse_dy else None),
                    dy_k=(_dyk(Cv) if fuse_dy else 0), poison=L.grad_poison(params))

        # Everything that only produces parameter gradients (nothing on the path to dz) -- the output head + the LSTM weights -- is cut into
        # PARTS, each a set of layers whose gradients are one contiguous range of the flat gradient buffer, produced (and, in DP, all-reduced)
        # in reverse layer order.  Default: two parts (head + the upper layers | the two lowest layers: 256 / 192 tiles of the grouped GEMM, one
        # round of workgroups each).  Data parallel (an optimiser with a GradSync that takes early ranges): ONE PART PER LAYER, so that layer
        # 3's range is on the links while layers 2, 1, 0 are still being contracted and the encoder's backward runs -- four early ranges
        # instead of two (at the per-rank batch of a DP job the 128-tile launches cost little: K = T * b is short).
        gsync = getattr(sink[0], "grad_sync", None) if sink is not None else None
        dp_early = gsync is not None and gsync.active and gsync.allow_early
        # two parts (upper layers + head, lower layers) where the peer releases them at two points of its backward; ONE part at the small per-GPU
        # batches whose GEMMs are released at once on a capped grid (round 5, b = 128: 5.50 -> 5.37 ms per step with 160 workgroups -- the gap
        # between two grouped launches, ~0.1 ms of bias / column-sum launches, closes, and the main stream keeps 96 compute units)
        nlow = min(int(L.knob("MVAE_DEFER_LAYERS", "0" if TB <= int(L.knob("MVAE_WGRAD_LATE_TB", 32768)) else "2")), NL - 1)
        per_layer = L.knob("MVAE_WGRAD_PER_LAYER", "1" if dp_early else "0") == "1"
        if per_layer:
            parts = [[l] for l in range(NL - 1, -1, -1)]
        else:
            parts = [list(range(nlow, NL))] + ([list(range(nlow))] if nlow >= 1 else [])
        first_name = lambda l: f"gru.weight_ih_l{l}" if l >= 1 else "gru.weight_hh_l0"

        wg_cap = [0]             # workgroup cap of the grouped weight-gradient launches (0: one per tile), see the fork below

        def weight_grads(k):
            """part k of `parts` (None: everything at once, on the current stream)."""
            with ops._Timed("dec_lstm_wgrad" if (k is None or k == 0) else "dec_lstm_wgrad_deferred"):
                if k is None or k == 0:
                    if dt == torch.bfloat16:
                        ops.gemm_tn(dl, hs[-1].view(TB, ldh), grads["decoded_mean.module.0.weight"], Cv, H, TB, lda=ldl, ldb=ldh)
                        dbp = W("dbout_p", (Cp,))
                        ops.colsum_t(dl, TB, Cp, dbp, ldx=ldl)
                        grads["decoded_mean.module.0.bias"].copy_(dbp[:Cv])
                    else:
                        hsT = W("wg_hsT_out", (H, ldT), dt)
                        ops.cast_transpose(hs[-1].view(TB, ldh), TB, H, dstT=hsT, lds=ldh)
                        ops.gemm_nt(dlT, hsT, grads["decoded_mean.module.0.weight"], Cv, H, TB, lda=ldT, ldb=ldT)
                        ops.rowsum(dlT, Cv, TB, grads["decoded_mean.module.0.bias"])
                layers = range(NL) if k is None else parts[k]
                caps_big = [int(c) for c in L.knob("MVAE_WGRAD_CAPS", WGRAD_CAPS_BIG).split(":")]       # per part, the two-part schedule of large batches
                mw = wg_cap[0] if (wg_cap[0] or k is None) else caps_big[min(k, len(caps_big) - 1)]
                _lstm_weight_grads(ws, grads, "gru", dt, dev, NL, Lq, B, H, dG, ldg, hs, ldh, layers=layers, max_workgroups=mw)
                if k is not None and dp_early:
                    # What this part produced is final on this stream: all-reduce it now -- [first parameter of its lowest layer, start of
                    # the previous part) (part 0: to the end of our range, i.e. with the head).  weight_ih_l0 / latent_input, produced on
                    # the main stream later, go with the rest in step().  Fork path only: there the gradients are handed over by ASSIGNMENT
                    # (p.grad = view of the flat buffer), so what step() sees is the reduced buffer; through autograd's AccumulateGrad a
                    # clone could hide it (FusedAdam.gather_grads)
                    lo = sink[2] + offs[first_name(min(parts[k]))]
                    hi = sink[3] if k == 0 else sink[2] + offs[first_name(min(parts[k - 1]))]
                    gsync.start_early(sink[1], lo, hi)

        # Fork: the weight-gradient GEMMs are throughput-bound and independent of dz, while the encoder's backward that follows is a
        # latency-bound chain of small launches -> run them concurrently.  Only when our MolecularVAE peer will join the side stream
        # (its backward ends with ForkState.join) and no gradient accumulation is pending (p.grad is assigned, never added to).
        # The later parts are parked (ops.ForkState): the peer releases them right before its row-resident LSTM backward,
        # whose 128 workgroups leave half the CUs idle -- the GEMMs fill them instead of running alone later.
        peer = mod.__dict__["_peer"]() if mod.__dict__["_peer"] is not None else None
        fork = bool(mod.overlap_weight_grads and peer is not None and ctx.from_peer and ctx.needs_input_grad[1] and
                    all(p.grad is None for p in params))
        if fork:
            side = mod._side_stream(dev)
            gflat.record_stream(side)
            fstate = peer.__dict__["_fork"]                               # the model's own fork state (ops.ForkState), kept by the peer
            # first half of the parts: released by the peer after its head section; the rest next to its row-resident LSTM backward.  At small
            # per-GPU batches (K = T * B short: a group of GEMMs is a few hundred microseconds) EVERYTHING waits for that second point: released
            # earlier, the chip-filling 256 x 256 tiles only starve the peer's conv / dense backward (a 5 us bias column sum sat 308 us behind
            # them at b = 128), while next to the 32-workgroup row-resident kernel (0.7 ms) they are hidden completely.
            late_all = TB <= int(L.knob("MVAE_WGRAD_LATE_TB", 32768))      # round 3: b = 128 8.38 -> 8.22 ms / step, B = 256 +0.13; with the capped grid below B = 256 gains too (9.84 -> 9.72)
            # ... or, at those batches, released AT ONCE but with a capped grid (mvae_gemm_tn_grouped_capped: `cap` workgroups looping over the
            # tiles): the compute units left over serve the peer's dependent small launches without queueing behind chip-filling tiles
            cap = int(L.knob("MVAE_WGRAD_CAP", WGRAD_CAP)) if late_all else 0
            if cap > 0:
                wg_cap[0] = cap
                for k in range(len(parts)):
                    fstate.park(side, (lambda kk=k: weight_grads(kk)), -1)
                fstate.run_deferred(stage=-1)
            else:
                for k in range(len(parts)):
                    fstate.park(side, (lambda kk=k: weight_grads(kk)), 1 if (late_all or 2 * k >= len(parts)) else 0)
        else:
            weight_grads(None)
        # layer-0 input is time-invariant: its gradient is the time sum of dG[0]
        dgx0 = W("dgx0", (B, ldg))           # pad columns of dG are zero, so the padded time sum is too
        ops.timesum(dG[0], Lq, B, ldg, dgx0)
        li = W("li", (B, o))
        ops.gemm_tn(dgx0, li, grads["gru.weight_ih_l0"], G4, o, B, lda=ldg)
        dli = W("dli", (B, o))
        ops.gemm_nt(dgx0, P["Wih0T"], dli, B, o, G4)
        # K6 backward
        ops.selu_bwd(dli, li)
        ops.gemm_tn_f32_colsum(dli, z, grads["latent_input.0.weight"], grads["latent_input.0.bias"], o, o, B)
        dz = torch.empty(B, o, dtype=f32, device=dev)
        ops.gemm_nt(dli, P["WliT"], dz, B, o, o, ldb=P["WliT"].stride(0))
        if fork:
            # gradients are still being written on the side stream: hand them over by assignment (autograd must not touch them)
            for n, p in zip(names, params):
                p.grad = grads[n]
            return (None, dz, None) + (None,) * len(names)
        return (None, dz, None) + tuple(grads[n] for n in names)


# ----------------------------------------------------------------------------------------------- VAE
class MolecularVAE(nn.Module):
    """models.py:97-106."""

    def __init__(self, i=120, o=292, c=35, dtype=torch.bfloat16, noise="device"):
        """noise: where the reparameterisation noise is drawn -- "device" (inside mvae_lambda_fwd, counter hash) or "cpu" (the reference's
        CPU generator stream, models.py:92); see Lambda."""
        super().__init__()
        self.encoder = MolEncoder(i=i, o=o, c=c)
        self.encoder.lmbd.noise = noise
        if noise not in ("device", "cpu"):
            raise ValueError("MolecularVAE(noise=...): 'device' or 'cpu'")
        self.decoder = MolDecoder(i=o, o=i, c=c, dtype=dtype)
        self.decoder.__dict__["_peer"] = weakref.ref(self.encoder)      # the fork state the decoder parks work in is the encoder's (_fork)
        self.prepack_decoder = True       # refresh the decoder's weight shadows on the side stream beside the encoder's forward
        # bf16 mode: the encoder's conv INPUT-gradient GEMMs multiply as 3 x bf16 products (~16 mantissa bits, 3/16 of the f32 MFMA cycles);
        # the forward pass of the encoder (what mu / logvar / the ELBO are made of) is exact fp32 in both modes
        self.encoder.fast_grad_gemms = dtype == torch.bfloat16

    apply = _apply_and_mark

    def forward(self, x, eps=None):
        ev = None
        if x.is_cuda and self.prepack_decoder and L.knob("MVAE_PREPACK", "1") != "0":
            # the decoder's weight shadows (8 bf16 cast / transposes of 4096 x 1024) are independent of the encoder's forward: refresh them
            # on the side stream beside it (after everything issued so far: the optimiser update they read)
            side = self.decoder._side_stream(x.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.decoder._pack(x.device)
                ev = torch.cuda.Event(); ev.record()
        x, mu, logvar = self.encoder(x, eps) if eps is not None else self.encoder(x)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        self.decoder.__dict__["_z_from_peer"] = True       # this forward's z comes from our encoder: its backward will join / release side work
        try:
            return self.decoder(x), mu, logvar
        finally:
            self.decoder.__dict__["_z_from_peer"] = False
