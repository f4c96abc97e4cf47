"""Tensor-level wrappers over the C ABI (include/mvae.h).  torch is plumbing only: device memory + streams.

Every function launches HIP kernels on torch's current stream and raises on any non-zero status.
"""
import ctypes as C
import weakref

import torch

from . import _lib as L
from ._lib import ptr, check, stream_ptr, dt_code


def _pad(n, m):
    return (n + m - 1) // m * m


# Optional device-side timing of tagged launch groups (bench.py's roofline leg): {tag: [(start, end) events]}.
# Events are recorded on torch's current stream, which is the stream every kernel here is launched on.
PROFILE = None


# roctx ranges (tracing): with MVAE_ROCTX=1 in the environment (or ops.enable_roctx()) every tagged launch group and every phase
# of the step (encoder / decoder forward and backward, the optimiser, the MOSES halves) is bracketed by roctxRangePush / Pop, so that a
# `rocprofv3 --kernel-trace --marker-trace` timeline groups the kernels by kernel family.  Off by default: two ctypes calls per range.
_ROCTX = {"lib": None, "on": None}


def enable_roctx(on=True):
    if on and _ROCTX["lib"] is None:
        for name in ("libroctx64.so", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = C.CDLL(name)
                lib.roctxRangePushA.argtypes = [C.c_char_p]
                _ROCTX["lib"] = lib
                break
            except OSError:
                continue
    _ROCTX["on"] = bool(on) and _ROCTX["lib"] is not None
    return _ROCTX["on"]


def _roctx_on():
    if _ROCTX["on"] is None:
        import os
        enable_roctx(os.environ.get("MVAE_ROCTX", "0") not in ("", "0"))
    return _ROCTX["on"]


class trace_range:
    """with ops.trace_range("dec_lstm_fwd"): ...   -- a named roctx range around the launches enqueued inside (no-op unless enabled)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.live = bool(self.name) and _roctx_on()
        if self.live:
            _ROCTX["lib"].roctxRangePushA(self.name.encode())
        return self

    def __exit__(self, *a):
        if self.live:
            _ROCTX["lib"].roctxRangePop()
        return False


def traced(name):
    """Decorator form of trace_range (the autograd Functions' forward / backward)."""
    def deco(fn):
        def wrapper(*a, **k):
            with trace_range(name):
                return fn(*a, **k)
        wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
        return wrapper
    return deco


class _Timed:
    def __init__(self, tag):
        self.tag = tag
        self.range = trace_range(tag)

    def __enter__(self):
        self.range.__enter__()
        if PROFILE is not None and self.tag:
            self.s = torch.cuda.Event(enable_timing=True); self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if PROFILE is not None and self.tag:
            self.e.record()
            PROFILE.setdefault(self.tag, []).append((self.s, self.e))
        self.range.__exit__()
        return False


# One side stream per device for the whole process, picked so that it does NOT share a hardware queue with the main stream: HIP maps streams
# onto a few hardware queues (4 by default, the null stream holds one), further streams double up round-robin, and two streams on one queue
# execute strictly one after the other -- a fork onto such a stream is no fork at all (measured: the 4th stream a process creates turned the
# b = 128 step from 8.2 into 14.4 ms).  Nothing in the API tells which queue a stream got, so the pick is a 1 ms probe at first use -- the
# models probe at their first FORWARD (MolecularVAE's prepack fork, mosesvae.VAE.forward), not inside a backward pass.
_SIDE_STREAMS = {}
SIDE_STREAM_INFO = {}     # device index -> dict(tries, concurrent, ratio): which candidate was taken and whether it passed the probe (logs / bench line)


def side_stream(device):
    device = torch.device(device)
    s = _SIDE_STREAMS.get(device.index)
    if s is None:
        s = _SIDE_STREAMS[device.index] = _pick_concurrent_stream(device)
    return s


def _pick_concurrent_stream(device, tries=6):
    """Probe: ~0.4 ms of fills on the main stream, a tiny fill on the candidate issued behind them; the candidate runs concurrently iff its fill
    finishes long before the main stream's.  16 MB of scratch, freed on return.  When no candidate passes (a co-tenant or a profiler can skew
    the wall-clock test) the first one is used and SIDE_STREAM_INFO says so -- the results are the same, only the overlap is lost."""
    main = torch.cuda.current_stream(device)
    big = torch.empty(1 << 22, dtype=torch.float32, device=device)       # 16 MB: a fill takes ~5 us
    small = torch.empty(256, dtype=torch.float32, device=device)
    first, info = None, dict(tries=0, concurrent=False, ratio=None)
    pick = None
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device)
        first = first or cand
        info["tries"] += 1
        torch.cuda.synchronize(device)
        e0, e_main, e_c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(main)
        for _ in range(96):
            big.fill_(1.0)                                                # ~0.4 ms of work on the main stream
        e_main.record(main)
        with torch.cuda.stream(cand):
            small.fill_(0.0)                                              # independent of it: finishes at once -- unless it queues behind
            e_c.record(cand)
        torch.cuda.synchronize(device)
        info["ratio"] = e0.elapsed_time(e_c) / max(e0.elapsed_time(e_main), 1e-6)
        if info["ratio"] < 0.5:
            pick, info["concurrent"] = cand, True
            break
    del big, small
    SIDE_STREAM_INFO[device.index] = info
    return pick or first


class ForkState:
    """Side-stream bookkeeping of ONE model (a MolDecoder and the MolEncoder whose backward follows it share one): `pending` = events of work
    forked onto a side stream that nobody has joined yet (joined by the encoder's backward / FusedAdam.step); `deferred` = (side stream,
    callable, stage): throughput-bound work its owner parked so that the peer can release it at a chosen point of ITS launch sequence.
    The decoder parks its weight-gradient GEMMs in two stages: stage 0 is released by the encoder's backward once its head section (a dozen
    tiny launches that would otherwise each wait for a chip-filling GEMM workgroup to retire) is enqueued, stage 1 right before its
    row-resident LSTM backward, whose 128 workgroups leave half the CUs idle.  Whoever reaches run_deferred() / join() first runs it.
    Per model, not per process: two models stepping in one process never see each other's parked work."""
    _all = weakref.WeakSet()

    def __init__(self):
        self.pending, self.deferred = [], []
        ForkState._all.add(self)

    def park(self, side, fn, stage):
        self.deferred.append((side, fn, stage))

    def run_deferred(self, stage=None):
        """Launch the parked work of stages <= `stage` (None: all) on its side stream, ordered after everything issued so far on the current stream."""
        keep = []
        while self.deferred:
            side, fn, st = self.deferred.pop(0)
            if stage is not None and st > stage:
                keep.append((side, fn, st))
                continue
            ev = torch.cuda.Event(); ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                fn()
                e2 = torch.cuda.Event(); e2.record()
            self.pending.append(e2)
        self.deferred.extend(keep)

    def join(self):
        """Make the current stream wait for every forked piece of work of this model (weight-gradient GEMMs on the side stream)."""
        self.run_deferred()
        while self.pending:
            torch.cuda.current_stream().wait_event(self.pending.pop())


def join_pending():
    """Join the forked work of EVERY live model (FusedAdam.step: an optimiser may hold the parameters of several)."""
    for fs in list(ForkState._all):
        fs.join()


def release_caches():
    """Drop the op-internal scratch buffers and any forgotten fork bookkeeping (between benchmark configurations / models)."""
    join_pending()
    Scratch._bufs.clear()


class Scratch:
    """Grow-only byte scratch for split-K slabs and op-internal temporaries (one per device AND stream: two streams must
    never share split-K slabs)."""
    _bufs = {}

    @classmethod
    def get(cls, nbytes, device, tag="", zeroed=False):
        """zeroed: the buffer is allocated zero and only ever handed to call sites of this `tag` (kernels that keep a self-resetting
        counter in it, e.g. the loss's ticket)."""
        key = (device.type, device.index, torch.cuda.current_stream().cuda_stream if device.type == "cuda" else 0, tag)
        b = cls._bufs.get(key)
        if b is None or b.numel() < nbytes:
            if zeroed:
                b = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
            else:
                b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            cls._bufs[key] = b
        return b


def gemm_nt(A, B, out, M, N, K, lda=None, ldb=None, ldc=None, bias=None, act=L.ACT_NONE, accumulate=False, x3=False):
    """out[M,N] = act(A[M,K] . B[N,K]^T + bias).  A, B same dtype (f32 / bf16); out f32 or bf16.
    x3 (fp32 operands): products on the bf16 MFMA as hi.hi + hi.lo + lo.hi (MVAE_F32X3, ~16 mantissa bits)."""
    lib = L.load()
    dt = dt_code(A.dtype)
    assert B.dtype == A.dtype
    if x3:
        assert A.dtype == torch.float32
        dt = L.MVAE_F32X3
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    ldc = out.stride(0) if ldc is None else ldc
    need = lib.mvae_gemm_nt_workspace(M, N, K, dt)
    ws = Scratch.get(need, A.device) if need else None
    check(lib.mvae_gemm_nt(dt, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, dt_code(out.dtype), ptr(bias), act,
                           1 if accumulate else 0, ptr(ws), need, stream_ptr()), "mvae_gemm_nt")
    return out


def gemm_tn(A, B, out, M, N, K, lda=None, ldb=None, ldc=None, bias=None, act=L.ACT_NONE, accumulate=False):
    """out[M,N] = act(A[:K,:M]^T . B[:K,:N] + bias): A [K, lda], B [K, ldb] bf16, K-major (weight gradients)."""
    lib = L.load()
    dt = dt_code(A.dtype)
    assert B.dtype == A.dtype
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    ldc = out.stride(0) if ldc is None else ldc
    need = lib.mvae_gemm_tn_workspace(M, N, K, dt)
    ws = Scratch.get(need, A.device) if need else None
    check(lib.mvae_gemm_tn(dt, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, dt_code(out.dtype), ptr(bias), act,
                           1 if accumulate else 0, ptr(ws), need, stream_ptr()), "mvae_gemm_tn")
    return out


def gemm_tn_f32_colsum(A, B, out, colsum_out, M, N, K, lda=None, ldb=None, ldc=None, accumulate=False, colsum_accumulate=False):
    """fp32: out[M,N] (+)= A[:K,:M]^T . B[:K,:N]  and  colsum_out[m] (+)= sum_k A[k,m]  in ONE launch (the bias gradient as a virtual ones column)."""
    lib = L.load()
    assert A.dtype == B.dtype == torch.float32
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    ldc = out.stride(0) if ldc is None else ldc
    need = lib.mvae_gemm_tn_workspace(M, N, K, L.MVAE_F32)
    ws = Scratch.get(need, A.device) if need else None
    check(lib.mvae_gemm_tn_f32_colsum(M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, 1 if accumulate else 0, ptr(colsum_out),
                                      1 if colsum_accumulate else 0, ptr(ws), need, stream_ptr()), "mvae_gemm_tn_f32_colsum")
    return out


def gemm_tn_colsum_supported(A, M, N, K):
    return A.dtype == torch.bfloat16 and bool(L.load().mvae_gemm_tn_colsum_supported(M, N, K))


def gemm_tn_colsum(A, B, out, colsum_out, M, N, K, lda=None, ldb=None, colsum_accumulate=False):
    """bf16: out[M,N] = A[:K,:M]^T . B[:K,:N]  and  colsum_out[m] (+)= sum_k A[k,m] from the same pass over A.  Returns False (nothing
    launched) when the shape is not served by the fused kernel -- the caller then runs gemm_tn + colsum_t."""
    lib = L.load()
    if A.dtype != torch.bfloat16 or not lib.mvae_gemm_tn_colsum_supported(M, N, K):
        return False
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    need = lib.mvae_gemm_tn_colsum_workspace(M, N, K)
    ws = Scratch.get(need, A.device) if need else None
    check(lib.mvae_gemm_tn_colsum(M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), out.stride(0), 0, ptr(colsum_out),
                                  1 if colsum_accumulate else 0, ptr(ws), need, stream_ptr()), "mvae_gemm_tn_colsum")
    return True


def gemm_tn_grouped_supported(A, M, N, K, lda, ldb):
    return A.dtype == torch.bfloat16 and bool(L.load().mvae_gemm_tn_grouped_supported(M, N, K, lda, ldb))


def gemm_tn_grouped(problems, max_workgroups=0):
    """problems: list of dict(A, B, out, M, N, K, lda, ldb[, colsum_out, colsum_accumulate, accumulate]) -- bf16 K-major operands; ONE launch
    for all of them, each 256 x 256 tile accumulated over its full K (no split-K slabs / reduction launch).  max_workgroups > 0: that many
    workgroups at most, looping over the tiles (leaves compute units to another stream)."""
    lib = L.load()
    n = len(problems)
    arr = (L.GemmTnProblem * n)()
    for i, q in enumerate(problems):
        a = arr[i]
        a.M, a.N, a.K = q["M"], q["N"], q["K"]
        a.A, a.lda, a.B, a.ldb = q["A"].data_ptr(), q["lda"], q["B"].data_ptr(), q["ldb"]
        a.C, a.ldc, a.accumulate = q["out"].data_ptr(), q["out"].stride(0), 1 if q.get("accumulate") else 0
        cs = q.get("colsum_out")
        a.colsum_out = cs.data_ptr() if cs is not None else None
        a.colsum_accumulate = 1 if q.get("colsum_accumulate") else 0
    need = lib.mvae_gemm_tn_grouped_workspace(n, arr)
    ws = Scratch.get(need, problems[0]["A"].device, tag="tn_grouped") if need else None
    check(lib.mvae_gemm_tn_grouped_capped(n, arr, int(max_workgroups), ptr(ws), need, stream_ptr()), "mvae_gemm_tn_grouped_capped")


class TnF32Batch:
    """Collects exact-f32 (or 3 x bf16) TN contractions that only feed the optimiser -- the parameter gradients of an encoder's backward pass --
    and runs them as ONE launch + ONE split-K reduction (mvae_gemm_tn_f32_multi) instead of a dozen half-empty ones.  Operands must stay alive
    and unchanged until run(); run() of more than MVAE_TN_F32_MULTI_MAX problems issues several launches."""
    MAX = 16

    def __init__(self, device):
        self.device, self.probs, self.keep, self.after = device, [], [], []

    def add(self, A, B, out, M, N, K, lda=None, ldb=None, ldc=None, colsum_out=None, colsum_accumulate=False, accumulate=False, x3=False,
            b_group=0, b_gstride=0):
        """out[M, N] (+)= A[:K, :M]^T . B[:K, :N]; colsum_out[m] (+)= sum_k A[k, m].  b_group > 0: row r of B starts at element
        (r // b_group) * b_gstride + (r % b_group) * ldb (a permuted view of a [T, B, W] buffer, see MolEncoder's table gradient)."""
        assert A.dtype == B.dtype == out.dtype == torch.float32
        q = L.GemmTnF32Problem()
        q.M, q.N, q.K = M, N, K
        q.A, q.lda, q.a_group, q.a_gstride = A.data_ptr(), (A.stride(0) if lda is None else lda), 0, 0
        q.B, q.ldb, q.b_group, q.b_gstride = B.data_ptr(), (B.stride(0) if ldb is None else ldb), b_group, b_gstride
        q.C, q.ldc, q.accumulate = out.data_ptr(), (out.stride(0) if ldc is None else ldc), 1 if accumulate else 0
        q.colsum_out = colsum_out.data_ptr() if colsum_out is not None else None
        q.colsum_accumulate, q.x3 = (1 if colsum_accumulate else 0), (1 if x3 else 0)
        self.probs.append(q)
        self.keep += [A, B, out, colsum_out]

    def add_conv_dw(self, B, W, Cin, ldx, x_bs, Cout, ldo, k, dzp, x, dwp, dw, db, x3=False):
        """The weight / bias gradient of a channels-last Conv1d whose mvae_conv1d_act_bwd call skipped it (dw = None): dwp is the packed
        scratch [Cout, k * ldx]; after run() it is unpacked into the parameter layout dw [Cout, Cin, k]."""
        q = L.GemmTnF32Problem()
        check(L.load().mvae_conv1d_dw_problem(B, W, Cin, ldx, x_bs, Cout, ldo, k, ptr(dzp), ptr(x), ptr(dwp), ptr(db), 1 if x3 else 0, C.byref(q)),
              "mvae_conv1d_dw_problem")
        self.probs.append(q)
        self.keep += [dzp, x, dwp, db, dw]
        self.after.append(lambda: check(L.load().mvae_conv1d_unpack_dw(Cin, Cout, k, ptr(dwp), ldx, ptr(dw), stream_ptr()), "mvae_conv1d_unpack_dw"))

    def run(self):
        lib = L.load()
        for i in range(0, len(self.probs), self.MAX):
            part = self.probs[i:i + self.MAX]
            arr = (L.GemmTnF32Problem * len(part))(*part)
            need = lib.mvae_gemm_tn_f32_multi_workspace(len(part), arr)
            ws = Scratch.get(need, self.device, tag="tn_f32_multi") if need else None
            check(lib.mvae_gemm_tn_f32_multi(len(part), arr, ptr(ws), need, stream_ptr()), "mvae_gemm_tn_f32_multi")
        for fn in self.after:
            fn()
        self.probs, self.keep, self.after = [], [], []


def colsum_t(X, M, N, out, ldx=None):
    lib = L.load()
    need = lib.mvae_colsum_t_workspace(M, N)
    ws = Scratch.get(need, X.device)
    check(lib.mvae_colsum_t(dt_code(X.dtype), M, N, ptr(X), X.stride(0) if ldx is None else ldx, ptr(out), ptr(ws), need,
                            stream_ptr()), "mvae_colsum_t")


def cast_transpose(src, R, C_, dst=None, dstT=None, lds=None):
    lib = L.load()
    lds = src.stride(0) if lds is None else lds
    d = dst if dst is not None else dstT
    check(lib.mvae_cast_transpose(dt_code(src.dtype), dt_code(d.dtype), R, C_, ptr(src), lds,
                                  ptr(dst), dst.stride(0) if dst is not None else 0,
                                  ptr(dstT), dstT.stride(0) if dstT is not None else 0, stream_ptr()), "mvae_cast_transpose")


class PackList:
    """A list of weight-packing jobs run as ONE launch (mvae_pack_multi).  Build it once per set of buffers (`key`), call run() after every
    optimiser step: the job table stays in device memory.  Methods mirror cast_transpose / torch.add / copy_; every tensor handed in must
    stay alive and in place (parameters inside FusedAdam's flat buffer and workspace buffers do)."""

    def __init__(self):
        self.jobs, self.keep, self.table, self.total = [], [], None, 0

    def _job(self, kind, src, R, C_, lds, dst=None, ldd=0, dstT=None, ldt=0, src2=None, dst_dtype=None):
        j = L.PackJob()
        j.kind, j.R, j.C = kind, R, C_
        j.src_dtype = dt_code(src.dtype)
        d = dst if dst is not None else dstT
        j.dst_dtype = dt_code(d.dtype if dst_dtype is None else dst_dtype)
        j.src, j.lds = src.data_ptr(), lds
        j.dst, j.ldd = (dst.data_ptr() if dst is not None else None), ldd
        j.dstT, j.ldt = (dstT.data_ptr() if dstT is not None else None), ldt
        j.src2 = src2.data_ptr() if src2 is not None else None
        self.jobs.append(j)
        self.keep += [t for t in (src, dst, dstT, src2) if t is not None]
        self.table = None

    def cast_transpose(self, src, R, C_, dst=None, dstT=None, lds=None):
        if R <= 0 or C_ <= 0:
            return
        if src.stride(-1) != 1 or (dst is not None and dst.stride(-1) != 1) or (dstT is not None and dstT.stride(-1) != 1):
            raise L.MvaeError("PackList: innermost dimension must be contiguous")
        self._job(0, src, R, C_, src.stride(0) if lds is None else lds, dst, dst.stride(0) if dst is not None else 0,
                  dstT, dstT.stride(0) if dstT is not None else 0)

    def add(self, a, b, out):
        assert a.dtype == b.dtype == out.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
        self._job(1, a, 1, a.numel(), a.numel(), dst=out, ldd=out.numel(), src2=b)

    def copy(self, src, dst):
        """dst[r, c] = src[r, c] for 2-D (or 1-D) fp32 views with contiguous rows."""
        assert src.dtype == dst.dtype == torch.float32 and src.shape == dst.shape
        if src.dim() == 1:
            src, dst = src.unsqueeze(0), dst.unsqueeze(0)
        assert src.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
        self._job(2, src, src.shape[0], src.shape[1], src.stride(0) if src.shape[0] > 1 else src.shape[1],
                  dst=dst, ldd=dst.stride(0) if dst.shape[0] > 1 else dst.shape[1])

    def run(self):
        if not self.jobs:
            return
        lib = L.load()
        if self.table is None:
            n = len(self.jobs)
            arr = (L.PackJob * n)()
            tot = 0
            for i, j in enumerate(self.jobs):
                j.block0 = tot
                tot += lib.mvae_pack_job_blocks(C.byref(j))
                arr[i] = j
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev = self.keep[0].device
            self.table, self.total = host.to(dev), tot
        check(lib.mvae_pack_multi(len(self.jobs), ptr(self.table), self.total, stream_ptr()), "mvae_pack_multi")


def permute021(inp, out, N, A, Bd):
    check(L.load().mvae_permute021(N, A, Bd, ptr(inp), ptr(out), stream_ptr()), "mvae_permute021")


def onehot_tb(idx, out, B, Lq, nrows):
    """out[(t*B + b), c] = (idx[b, t] == c) for c < ld (bf16, time-major rows): the K-major operand that turns the table scatter
    dtable = sum_rows onehot^T . d into one TN GEMM."""
    check(L.load().mvae_onehot_tb(ptr(idx), B, Lq, nrows, ptr(out), out.stride(0), stream_ptr()), "mvae_onehot_tb")


def onehot_f32(idx, out, nrows):
    """out[r, c] = (idx.flat[r] == c), fp32, rows in idx's own order."""
    check(L.load().mvae_onehot_f32(ptr(idx), idx.numel(), nrows, ptr(out), out.stride(0), stream_ptr()), "mvae_onehot_f32")


def gather_rows_tb(idx, table, out, B, Lq, nrows, W, base=None):
    check(L.load().mvae_gather_rows_tb(ptr(idx), B, Lq, nrows, ptr(table), W, ptr(base), ptr(out), stream_ptr()), "mvae_gather_rows_tb")


def expand_indices(store, rows, idx, ohe, B, Lq, C_):
    check(L.load().mvae_expand_indices(ptr(store), ptr(rows), B, Lq, C_, ptr(idx), ptr(ohe), stream_ptr()), "mvae_expand_indices")


def relu_bwd(dy, y):
    check(L.load().mvae_relu_bwd(dy.numel(), ptr(dy), ptr(y), stream_ptr()), "mvae_relu_bwd")


def mask_rows_tb(buf, T, B, lengths):
    """rows (t*B + b) with t >= lengths[b] of the time-major [T*B, ld] buffer := 0."""
    check(L.load().mvae_mask_rows_tb(dt_code(buf.dtype), T, B, buf.stride(0), ptr(lengths), ptr(buf), stream_ptr()), "mvae_mask_rows_tb")


def permute102(inp, out, T, B, V):
    check(L.load().mvae_permute102(T, B, V, ptr(inp), ptr(out), stream_ptr()), "mvae_permute102")


def moses_sample_step(h_top, ldh, w_fc, bias, temp, seed, step, eos_id, table, base, add_out, x, end_pads, eos_mask, w_out, B, V, H):
    """One generated token behind the GRU step: head GEMV, temperature softmax, multinomial draw (counter hash of (seed, step, row)), the
    reference's eos / end-pad bookkeeping and the next step's layer-0 addend rows -- one launch (mvae_moses_sample_step)."""
    check(L.load().mvae_moses_sample_step(dt_code(h_top.dtype), B, V, H, ptr(h_top), ldh, ptr(w_fc), w_fc.stride(0), ptr(bias), float(temp),
                                          int(seed) & 0xFFFFFFFF, int(step), int(eos_id), ptr(table), table.shape[1], ptr(base), ptr(add_out),
                                          ptr(x), x.stride(0), ptr(end_pads), ptr(eos_mask), ptr(w_out), stream_ptr()), "mvae_moses_sample_step")


def sample_uniform(seed, step, B):
    """Host restatement of the sampling step's uniforms u(b) = hash(seed, step * B + b) / 2^32 (tests)."""
    import numpy as np
    idx = (np.uint64(step) * np.uint64(B) + np.arange(B, dtype=np.uint64)) & np.uint64(0xFFFFFFFF)
    h = ((idx * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) ^ np.uint64(seed & 0xFFFFFFFF)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h.astype(np.float64) / 4294967296.0


def moses_latent_fwd(mu, logvar, eps, z, kl, B, dz, seed=0, offset=0, eps_out=None):
    """eps None: the noise is drawn inside the launch from the counter hash of (seed, offset + i) and written to eps_out."""
    lib = L.load()
    need = lib.mvae_moses_latent_workspace(B)
    ws = Scratch.get(need, mu.device)
    check(lib.mvae_moses_latent_fwd(B, dz, ptr(mu), ptr(logvar), ptr(eps), int(seed) & 0xFFFFFFFF, int(offset), ptr(eps_out), ptr(z), ptr(kl),
                                    ptr(ws), need, stream_ptr()), "mvae_moses_latent_fwd")


def moses_latent_bwd(mu, logvar, eps, dz_in, dkl, dlogvar_ext, dmu, dlogvar, B, dz):
    check(L.load().mvae_moses_latent_bwd(B, dz, ptr(mu), ptr(logvar), ptr(eps), ptr(dz_in), ptr(dkl), ptr(dlogvar_ext), ptr(dmu), ptr(dlogvar),
                                         stream_ptr()), "mvae_moses_latent_bwd")


def ce_loss_fwd(logits, ldl, x, pad, loss2, B, T, V):
    lib = L.load()
    need = lib.mvae_ce_loss_workspace(B, T)
    ws = Scratch.get(need, logits.device)
    check(lib.mvae_ce_loss_fwd(B, T, V, ptr(logits), ldl, ptr(x), pad, ptr(loss2), ptr(ws), need, stream_ptr()), "mvae_ce_loss_fwd")


def ce_loss_bwd(logits, ldl, x, pad, loss2, grad_out, dy_ext, dl, B, T, V):
    check(L.load().mvae_ce_loss_bwd(dt_code(dl.dtype), B, T, V, ptr(logits), ldl, ptr(x), pad, ptr(loss2), ptr(grad_out), ptr(dy_ext), ptr(dl),
                                    dl.stride(0), stream_ptr()), "mvae_ce_loss_bwd")


def scatter_rows_tb(idx, d, dtable, B, Lq, nrows, W, ldd=None):
    lib = L.load()
    need = lib.mvae_scatter_rows_tb_workspace(B, Lq, nrows, W)
    ws = Scratch.get(need, d.device)
    check(lib.mvae_scatter_rows_tb(dt_code(d.dtype), ptr(idx), B, Lq, nrows, ptr(d), W if ldd is None else ldd, W, ptr(dtable), ptr(ws), need,
                                   stream_ptr()), "mvae_scatter_rows_tb")


def rowsum(X, R, C_, out, ldx=None, accumulate=False):
    check(L.load().mvae_rowsum(dt_code(X.dtype), R, C_, ptr(X), X.stride(0) if ldx is None else ldx, ptr(out),
                               1 if accumulate else 0, stream_ptr()), "mvae_rowsum")


def timesum(X, T, B, W, out):
    check(L.load().mvae_timesum(dt_code(X.dtype), T, B, W, ptr(X), ptr(out), stream_ptr()), "mvae_timesum")


def colsum(X, M, N, out, ldx=None):
    lib = L.load()
    need = lib.mvae_colsum_workspace(M, N)
    ws = Scratch.get(need, X.device)
    check(lib.mvae_colsum(M, N, ptr(X), X.stride(0) if ldx is None else ldx, ptr(out), ptr(ws), need, stream_ptr()), "mvae_colsum")


def selu_bwd(dy, y):
    check(L.load().mvae_selu_bwd(dy.numel(), ptr(dy), ptr(y), stream_ptr()), "mvae_selu_bwd")


def conv1d_pack_weights(w, Cin, Cout, k, ldx, wp, ldo=0, wq=None):
    """wp[o][j*ldx + c] = w[o][c][j];  wq[c][j*ldo + o] = w[o][c][k-1-j]  (pads zero)."""
    check(L.load().mvae_conv1d_pack_weights(Cin, Cout, k, ptr(w), ldx, ptr(wp), ldo, ptr(wq), stream_ptr()), "mvae_conv1d_pack_weights")


def conv1d_selu_fwd(x, B, W, ldx, x_bs, Cout, k, wp, bias, y, ldy, act=L.ACT_SELU):
    """Channels-last sliding-window conv + bias + activation: x[b, w, c] at b*x_bs + w*ldx + c (pad channels zero) -> y[(b*Wout + w), o] (ldy)."""
    lib = L.load()
    need = lib.mvae_conv1d_selu_fwd_workspace(B, W, ldx, Cout, k)
    ws = Scratch.get(need, x.device) if need else None
    check(lib.mvae_conv1d_act_fwd(act, B, W, ldx, x_bs, Cout, k, ptr(x), ptr(wp), ptr(bias), ptr(y), ldy, ptr(ws), need, stream_ptr()),
          "mvae_conv1d_act_fwd")


def conv1d_selu_bwd(B, W, Cin, ldx, x_bs, Cout, ldo, k, dy, y, x, wq, dzp, dw, db, dx, lddx, act=L.ACT_SELU, x3=False):
    """x3: the input-gradient GEMM multiplies its fp32 operands as 3 x bf16 products (MVAE_CONV_BWD_X3: gradients of the bf16 training mode).
    dw None: only dzp and dx are produced (the weight gradient joins a TnF32Batch: add_conv_dw)."""
    lib = L.load()
    if x3:
        act = act | L.CONV_BWD_X3
    need = lib.mvae_conv1d_selu_bwd_workspace(B, W, Cin, ldx, Cout, ldo, k)
    ws = Scratch.get(need, dy.device)
    check(lib.mvae_conv1d_act_bwd(act, B, W, Cin, ldx, x_bs, Cout, ldo, k, ptr(dy), ptr(y), ptr(x), ptr(wq), ptr(dzp), ptr(dw), ptr(db),
                                  ptr(dx), lddx, ptr(ws), need, stream_ptr()), "mvae_conv1d_act_bwd")


def lambda_fwd(mulv, eps, z, mu, logvar, B, o, scale=1.0, seed=0, offset=0, eps_out=None):
    """eps None: eps_out[i] = scale * n(seed, offset + i) is drawn inside the launch (the draw of models.py:92 as part of the op)."""
    check(L.load().mvae_lambda_fwd(B, o, ptr(mulv), ptr(eps), float(scale), int(seed) & 0xFFFFFFFF, int(offset), ptr(eps_out), ptr(z), ptr(mu),
                                   ptr(logvar), stream_ptr()), "mvae_lambda_fwd")


def normal_fill(out, scale, seed, offset):
    """out.flat[i] = scale * n(seed, offset + i): the library's counter-hash normal (randn_like without generator state)."""
    assert out.dtype == torch.float32 and out.is_contiguous()
    check(L.load().mvae_normal_fill(out.numel(), float(scale), int(seed) & 0xFFFFFFFF, int(offset), ptr(out), stream_ptr()), "mvae_normal_fill")
    return out


def normal_draw(seed, offset, n, scale=1.0):
    """Host restatement (numpy, float64) of the n normals the kernels draw for counters offset .. offset + n - 1 of stream `seed`
    (include/mvae.h: mvae_lambda_fwd).  The hash words are bit-exact (tests hold them to mvae_normal_words); the transform is in double."""
    import numpy as np
    M = np.uint64(0xFFFFFFFF)

    def H(seed_, idx):
        h = ((idx * np.uint64(0x9E3779B1)) & M) ^ seed_
        h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & M
        h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & M
        h ^= h >> np.uint64(16)
        return h
    c = np.uint64(offset) + np.arange(n, dtype=np.uint64)
    s = H(np.uint64(seed & 0xFFFFFFFF), (c >> np.uint64(31)) ^ np.uint64(0x6A09E667))
    lo = (c << np.uint64(1)) & M
    w1 = H(s, lo)
    w2 = H(s ^ np.uint64(0xBB67AE85), lo | np.uint64(1))
    u1 = ((w1 >> np.uint64(8)).astype(np.float64) + 0.5) / 16777216.0
    u2 = (w2 >> np.uint64(8)).astype(np.float64) / 16777216.0
    return scale * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2), w1.astype(np.uint32), w2.astype(np.uint32)


class NoiseStream:
    """One device-side normal stream: an explicit 32-bit seed and a running 64-bit element counter -- all the state there is (picklable; a
    checkpoint may carry `state()`).  take(n) hands out the (seed, offset) of the next n elements.  The default seed comes from
    torch.initial_seed() (so torch.manual_seed makes a run reproducible), the rank of the default process group (ranks draw different noise
    for their shards) and a per-process instance number (two models in one process draw different streams)."""
    _instances = [0]

    def __init__(self, seed=None):
        self.seed, self.counter = seed, 0

    def _default_seed(self):
        import torch.distributed as dist
        try:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        except (ValueError, RuntimeError):        # no default process group after all
            rank = 0
        NoiseStream._instances[0] += 1
        words = (C.c_uint32 * 2)()
        L.load().mvae_normal_words(torch.initial_seed() & 0xFFFFFFFF, (rank << 20) + NoiseStream._instances[0], words)
        return int(words[0])

    def reseed(self, seed, counter=0):
        self.seed, self.counter = int(seed) & 0xFFFFFFFF, int(counter)

    def take(self, n):
        if self.seed is None:
            self.seed = self._default_seed()
        off = self.counter
        self.counter += int(n)
        return self.seed, off

    def state(self):
        return dict(seed=self.seed, counter=self.counter)


def lambda_bwd(mulv, eps, dz, dmu, dlogvar, dmulv, B, o):
    check(L.load().mvae_lambda_bwd(B, o, ptr(mulv), ptr(eps), ptr(dz), ptr(dmu), ptr(dlogvar), ptr(dmulv), stream_ptr()),
          "mvae_lambda_bwd")


def softmax_tb_fwd(logits, ldl, recon, B, Lq, C_):
    check(L.load().mvae_softmax_tb_fwd(B, Lq, C_, ptr(logits), ldl, ptr(recon), stream_ptr()), "mvae_softmax_tb_fwd")


def softmax_tb_bwd(recon, drecon, dl, dlT, B, Lq, C_):
    check(L.load().mvae_softmax_tb_bwd(dt_code(dl.dtype), B, Lq, C_, ptr(recon), ptr(drecon), ptr(dl), dl.stride(0),
                                       ptr(dlT), dlT.stride(0) if dlT is not None else 0, stream_ptr()), "mvae_softmax_tb_bwd")


def bce_kl_loss_fwd(recon, target, mu, logvar, max_len, out3):
    lib = L.load()
    need = lib.mvae_bce_kl_loss_workspace(recon.numel(), mu.numel())
    ws = Scratch.get(need, recon.device, tag="bce_kl_ticket", zeroed=True)     # the kernel's ticket counter lives in its last 16 bytes
    check(lib.mvae_bce_kl_loss_fwd(recon.numel(), ptr(recon), ptr(target), mu.numel(), ptr(mu), ptr(logvar), float(max_len),
                                   ptr(out3), ptr(ws), need, stream_ptr()), "mvae_bce_kl_loss_fwd")


def bce_kl_loss_bwd(recon, target, mu, logvar, max_len, grad_out, drecon, dmu, dlogvar):
    check(L.load().mvae_bce_kl_loss_bwd(recon.numel(), ptr(recon), ptr(target), mu.numel(), ptr(mu), ptr(logvar), float(max_len),
                                        ptr(grad_out), ptr(drecon), ptr(dmu), ptr(dlogvar), stream_ptr()), "mvae_bce_kl_loss_bwd")


def _fill(arr, tensors):
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None


def dropout_keep_mask(seed, shape_lt_b_h, p):
    """Host restatement of the device-generated inter-layer dropout mask (mvae_dropout_keep): uint8 [layers-1... as given][T][B][H], element
    index = flat position.  Used by the tests / the oracle to draw exactly what the kernels draw."""
    import numpy as np
    n = int(np.prod(shape_lt_b_h))
    idx = np.arange(n, dtype=np.uint64)
    h = ((idx * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) ^ np.uint64(seed & 0xFFFFFFFF)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    thresh = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    return (h >= thresh).astype(np.uint8).reshape(shape_lt_b_h)


# Schedules with BOUNDED SPINS: the weights-resident dataflow passes of the decoder LSTM (rnn_persist*.hip; the per-rank shape of the 8-GPU
# configuration: 4 x 1024, b = 128 / 256, bf16) and the layer-concurrent row-resident encoder passes (rnn_rowres.hip).  Such a launch cannot hang,
# but it can GIVE UP (a workgroup not resident because something else held a compute unit): it then leaves a status record -- the library says
# where (status_out of mvae_rnn_fwd / mvae_rnn_bwd) -- and, when the caller passed a `poison` slot, a NaN there.  Two ways to survive that:
#   * poison given (a training step under FusedAdam): nothing waits.  The NaN sits in the spare slot behind the flat gradient buffer, so the
#     optimiser kernel skips the whole update of that step on the device (in data parallel the slot is all-reduced with the gradient: every rank
#     skips together); the status record is copied to pinned memory behind the launch and looked at later (persist_check).
#   * no poison (evaluation, plain torch optimisers): the call waits for its status record and, on a failure, runs the pass again on the
#     schedule without spins -- same buffers, every one of them is written as a whole by either schedule.
# Either way the failure is counted and reported once (warning); after PERSIST_MAX_FAILURES of them the spinning schedules are switched off
# for the rest of the process (`no_spin` in the descriptors).  PERSIST_DEFAULT: "1" = use them wherever the library serves the shape.
PERSIST_DEFAULT = "1"
PERSIST_MAX_FAILURES = 3
PERSIST_STATS = {"launches": 0, "rowres_pipe": 0, "bwd_launches": 0, "failures": 0, "reruns": 0, "disabled": False}     # passes that took a spinning schedule; what went wrong (tests / logs / bench line)
_PERSIST_PENDING = []      # [(pinned int32[4], event, what)]
_PERSIST_WARNED = [False]


def _spin_failure(what, rec):
    PERSIST_STATS["failures"] += 1
    if PERSIST_STATS["failures"] >= PERSIST_MAX_FAILURES and not PERSIST_STATS["disabled"]:
        PERSIST_STATS["disabled"] = True
    if not _PERSIST_WARNED[0] or PERSIST_STATS["disabled"]:
        import warnings
        warnings.warn(f"mvae: {what}: a launch with bounded spins gave up waiting for a hand-off (status {list(rec)}: code, block, layer) -- not "
                      "all of its workgroups were resident (another kernel held compute units).  Its outputs were discarded (training: the "
                      "optimiser skipped that step on the device; otherwise the pass was run again on the schedule without spins)."
                      + (f"  {PERSIST_STATS['failures']} such failures: these schedules are now off for this process." if PERSIST_STATS["disabled"] else ""),
                      RuntimeWarning, stacklevel=3)
        _PERSIST_WARNED[0] = True


def persist_check(sync=False):
    """Look at the status records that have arrived (sync: wait for all of them); a failed hand-off is counted and warned about (see above) --
    nothing is raised: the device has already kept the garbage out of the weights.  Returns the number of failures found by this call."""
    keep, found = [], 0
    for host, ev, what in _PERSIST_PENDING:
        if sync:
            ev.synchronize()
        if not ev.query():
            keep.append((host, ev, what))
            continue
        if int(host[0]) != 0:
            found += 1
            _spin_failure(what, host.tolist())
    _PERSIST_PENDING[:] = keep
    return found


_STATUS_RING = {"buf": None, "n": 0}


def _status_slot():
    """One int32[4] slot of a pinned ring allocated ONCE (a fresh pinned allocation per launch is a hipHostMalloc per launch, and that call
    waits for the device: with a 1.4 ms persistent kernel in flight it kept the host from enqueueing the launches behind it).  512 slots:
    persist_check() retires completed records at every call, so a slot is long done when the ring comes round."""
    r = _STATUS_RING
    if r["buf"] is None:
        r["buf"] = torch.zeros(512, 4, dtype=torch.int32).pin_memory()
    r["n"] += 1
    return r["buf"][r["n"] % 512]


def _status_view(addr, bufs):
    """int32[4] view of the status record at device address `addr`, which lies inside one of the scratch tensors `bufs`."""
    for b in bufs:
        if b is not None and b.data_ptr() <= addr and addr + 16 <= b.data_ptr() + b.numel() * b.element_size():
            off = addr - b.data_ptr()
            return b.view(torch.uint8)[off:off + 16].view(torch.int32)
    raise L.MvaeError("mvae_rnn_*: the status record the library reported lies outside the scratch buffers of the call")


def _after_spin_launch(what, addr, bufs, poison, rerun):
    """A launch with bounded spins was enqueued; its status record is at `addr`.  poison given: asynchronous bookkeeping only.  Otherwise wait
    for the record and run `rerun()` (the same pass with no_spin set) if the launch gave up."""
    rec = _status_view(addr, bufs)
    host = _status_slot()
    host.copy_(rec, non_blocking=True)
    ev = torch.cuda.Event(); ev.record()
    if poison is not None:
        _PERSIST_PENDING.append((host, ev, what))
        if len(_PERSIST_PENDING) > 64:
            persist_check()
        return
    ev.synchronize()
    if int(host[0]) != 0:
        _spin_failure(what, host.tolist())
        PERSIST_STATS["reruns"] += 1
        rerun()


def rnn_fwd(cell, dtype, T, B, H, add0, add0_tstride, w_ih, ldw_ih, w_hh, ldw_hh, bias, hs, ldh, cs, gates, cstate,
            x0=None, x0_ld=0, in0=0, h0=None, ldh0=0, lengths=None, zero_padded_k=False, tag=None,
            hdrop=None, drop_mask=None, drop_p=0.0, drop_seed=0, add_table=None, add_index=None, persist=None, poison=None):
    """add_table [rows, G*H] fp32 + add_index [B, L >= T] int64: layer 0 adds table row add_index[b, t] in its epilogue (no gathered copy).
    persist: True / False / None (= PERSIST_DEFAULT): the schedules with bounded spins (weights-resident dataflow pass / layer-concurrent
    row-resident pass) where the library serves the shape.  poison: the optimiser's poison slot (see the note above PERSIST_DEFAULT); None:
    the call verifies such a launch itself and re-runs the pass on a failure."""
    d = L.RnnFwdDesc()
    NL = len(w_hh)
    d.cell, d.dtype, d.layers, d.T, d.B, d.H, d.in0 = cell, dt_code(dtype), NL, T, B, H, in0
    d.x0 = x0.data_ptr() if x0 is not None else None
    d.x0_ld = x0_ld
    d.add0 = add0.data_ptr() if add0 is not None else None
    d.add0_tstride = add0_tstride
    if add_table is not None:
        assert add_table.dtype == torch.float32 and add_table.is_contiguous() and add_table.shape[1] == 4 * H
        assert add_index.dtype == torch.int64 and add_index.stride(1) == 1
        d.add_table, d.add_index = add_table.data_ptr(), add_index.data_ptr()
        d.add_index_ld, d.add_table_rows = add_index.stride(0), add_table.shape[0]
    _fill(d.w_ih, w_ih); _fill(d.w_hh, w_hh); _fill(d.bias, bias)
    for i in range(NL):
        d.ldw_ih[i] = ldw_ih[i]
        d.ldw_hh[i] = ldw_hh[i]
    if h0 is not None:
        _fill(d.h0, h0)
    d.ldh0 = ldh0
    d.lengths = lengths.data_ptr() if lengths is not None else None
    _fill(d.hs, hs); d.ldh = ldh
    if cs is not None:
        _fill(d.cs, cs)
    if gates is not None:
        _fill(d.gates, gates)
    if cstate is not None:
        _fill(d.cstate, cstate)
    d.zero_padded_k = 1 if zero_padded_k else 0
    if hdrop is not None:
        _fill(d.hdrop, hdrop)
        if drop_mask is not None:
            _fill(d.drop_mask, drop_mask)
        d.drop_p, d.drop_seed = float(drop_p), int(drop_seed) & 0xFFFFFFFF
    use_p = ((L.knob("MVAE_PERSIST", PERSIST_DEFAULT) != "0") if persist is None else bool(persist)) and not PERSIST_STATS["disabled"]
    pws = None
    if use_p:
        persist_check()
        need = L.load().mvae_rnn_fwd_persist_workspace(C.byref(d))
        if need:
            pws = Scratch.get(need, hs[0].device, tag="rnn_persist")
            d.persist_ws, d.persist_ws_bytes = pws.data_ptr(), need
        elif persist:
            raise L.MvaeError("rnn_fwd(persist=True): this shape / device is not served by the persistent schedule")
    d.no_spin = 0 if use_p else 1
    d.poison = poison.data_ptr() if poison is not None else None
    status = C.c_void_p()
    with _Timed(tag):
        check(L.load().mvae_rnn_fwd(C.byref(d), stream_ptr(), C.byref(status)), "mvae_rnn_fwd")
    if status.value:                 # the library took a schedule with bounded spins and says where its status record is
        PERSIST_STATS["launches" if dtype == torch.bfloat16 else "rowres_pipe"] += 1

        def rerun():
            d.no_spin, d.poison = 1, None
            check(L.load().mvae_rnn_fwd(C.byref(d), stream_ptr(), None), "mvae_rnn_fwd (again, without spins)")
        _after_spin_launch(tag or "mvae_rnn_fwd", status.value, (pws,), poison, rerun)


def rnn_bwd_persist_wanted(cell, dtype, NL, B, H, ldg, device):
    """Would rnn_bwd take the weights-resident dataflow schedule for this shape if it is given the output gradient as `dy`?  (The library has
    the last word -- mvae_rnn_bwd_persist_workspace -- this only lets the caller choose the gradient's form; a wrong guess costs speed, not
    correctness.)"""
    if L.knob("MVAE_PERSIST_BWD", PERSIST_DEFAULT) == "0" or PERSIST_STATS["disabled"]:
        return False
    if cell != L.CELL_LSTM or dtype != torch.bfloat16 or NL != 4 or H != 1024 or B not in (128, 256) or ldg != 4 * H + 64:
        return False
    return torch.cuda.get_device_properties(device).multi_processor_count == 256


def rnn_bwd(cell, dtype, T, B, H, w_hhT, ldw_hhT, w_ihT, ldw_ihT, dy, dy_ld, hs, ldh, cs, gates, dG, dstate,
            ldg=None, h0=None, ldh0=0, lengths=None, dh_last=None, dGh=None, dh0=None, tag=None,
            drop_mask=None, drop_p=0.0, drop_seed=0, dy_a=None, dy_w=None, dy_k=0, persist=None, poison=None):
    """dy_a [T*B, ld] / dy_w [H, ld] (dtype, zero-padded to dy_k columns): the output gradient as a product dy = dy_a . dy_w^T, contracted
    by the top layer's cell itself (no [T, B, H] fp32 dy tensor).
    persist: True / False / None (= PERSIST_DEFAULT): the weights-resident dataflow backward where the library serves the shape (it wants the
    output gradient as `dy`, see rnn_bwd_persist_served), and the layer-concurrent row-resident form of the narrow f32 stacks.  poison: as in
    rnn_fwd."""
    d = L.RnnBwdDesc()
    NL = len(w_hhT)
    d.cell, d.dtype, d.layers, d.T, d.B, d.H = cell, dt_code(dtype), NL, T, B, H
    _fill(d.w_hhT, w_hhT); _fill(d.w_ihT, w_ihT)
    for i in range(NL):
        d.ldw_hhT[i] = ldw_hhT[i]
        d.ldw_ihT[i] = ldw_ihT[i]
    d.lengths = lengths.data_ptr() if lengths is not None else None
    d.dy = dy.data_ptr() if dy is not None else None
    d.dy_ld = dy_ld
    if dy_a is not None:
        d.dy_a, d.dy_a_ld, d.dy_w, d.dy_w_ld, d.dy_k = dy_a.data_ptr(), dy_a.stride(0), dy_w.data_ptr(), dy_w.stride(0), dy_k
    if dh_last is not None:
        _fill(d.dh_last, dh_last)
    _fill(d.hs, hs); d.ldh = ldh
    if h0 is not None:
        _fill(d.h0, h0)
    d.ldh0 = ldh0
    if cs is not None:
        _fill(d.cs, cs)
    _fill(d.gates, gates); _fill(d.dG, dG)
    d.ldg = ldg if ldg is not None else {"LSTM": 4, "GRU": 3}["LSTM" if cell == L.CELL_LSTM else "GRU"] * H
    if dGh is not None:
        _fill(d.dGh, dGh)
    _fill(d.dstate, dstate)
    if dh0 is not None:
        _fill(d.dh0, dh0)
    if drop_p > 0.0:
        if drop_mask is not None:
            _fill(d.drop_mask, drop_mask)
        d.drop_p, d.drop_seed = float(drop_p), int(drop_seed) & 0xFFFFFFFF
    need = L.load().mvae_rnn_bwd_workspace(C.byref(d))     # scratch of the split-K schedules (used only when the shape qualifies)
    sws = Scratch.get(need, dy.device if dy is not None else dG[0].device, tag="rnn_split")
    d.split_ws, d.split_ws_bytes = sws.data_ptr(), need
    use_p = ((L.knob("MVAE_PERSIST_BWD", PERSIST_DEFAULT) != "0") if persist is None else bool(persist)) and not PERSIST_STATS["disabled"]
    pws = None
    if use_p:
        pneed = L.load().mvae_rnn_bwd_persist_workspace(C.byref(d))
        if pneed:
            persist_check()
            pws = Scratch.get(pneed, dG[0].device, tag="rnn_persist_bwd")
            d.persist_ws, d.persist_ws_bytes = pws.data_ptr(), pneed
        elif persist:
            raise L.MvaeError("rnn_bwd(persist=True): this shape / device / gradient form is not served by the persistent schedule")
    d.no_spin = 0 if use_p else 1
    d.poison = poison.data_ptr() if poison is not None else None
    status = C.c_void_p()
    with _Timed(tag):
        check(L.load().mvae_rnn_bwd(C.byref(d), stream_ptr(), C.byref(status)), "mvae_rnn_bwd")
    if status.value:                 # which schedule ran is the library's knowledge: it reports the status record of a launch with bounded spins
        if pws is not None and pws.data_ptr() <= status.value < pws.data_ptr() + pws.numel():
            PERSIST_STATS["bwd_launches"] += 1
        else:
            PERSIST_STATS["rowres_pipe"] += 1

        def rerun():
            d.no_spin, d.poison = 1, None
            check(L.load().mvae_rnn_bwd(C.byref(d), stream_ptr(), None), "mvae_rnn_bwd (again, without spins)")
        _after_spin_launch(tag or "mvae_rnn_bwd", status.value, (pws, sws), poison, rerun)


def sumsq(g, partial):
    check(L.load().mvae_sumsq(g.numel(), ptr(g), ptr(partial), stream_ptr()), "mvae_sumsq")


def clip_adam(p, g, m, v, partial, grad_scale, max_norm, lr, b1, b2, eps, step, norm_out, poison_reset=None):
    """norm_out [>= 1]: [0] = the pre-clip norm; [1] (when present) counts the steps the kernel skipped because the norm was not finite.
    poison_reset: the poison slot, set back to zero by the launch."""
    check(L.load().mvae_clip_adam(p.numel(), ptr(p), ptr(g), ptr(m), ptr(v), ptr(partial), partial.numel(), float(grad_scale),
                                  float(max_norm), float(lr), float(b1), float(b2), float(eps), int(step), ptr(norm_out), norm_out.numel(),
                                  ptr(poison_reset), stream_ptr()), "mvae_clip_adam")
