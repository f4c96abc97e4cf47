"""ctypes binding of libmvae_hip.so (C ABI in include/mvae.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, this module raises.
"""
import ctypes as C
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVAE_LIB: absolute path of a diagnostic build (csrc/build.sh tune -> tests/tuning/lib/libmvae_hip_tune.so: outside the package, never next to the product library) for timing decompositions; bench.py refuses to
# run with any MVAE_* variable set, so a measured number always comes from the product library.
LIB_PATH = ((os.environ.get("MVAE_LIB") if os.environ.get("MVAE_TUNING", "0") not in ("", "0") else None)
            or os.path.join(_HERE, "libmvae_hip.so"))      # MVAE_LIB too is honoured only under MVAE_TUNING=1

ABI_VERSION = 10
MVAE_F32, MVAE_BF16, MVAE_F32X3 = 0, 1, 2
CONV_BWD_X3 = 0x100
ACT_NONE, ACT_SELU, ACT_RELU = 0, 1, 2
CELL_LSTM, CELL_GRU = 0, 1
MAX_LAYERS = 8

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class RnnFwdDesc(C.Structure):
    _fields_ = [("cell", _i), ("dtype", _i), ("layers", _i), ("T", _i), ("B", _i), ("H", _i), ("in0", _i),
                ("x0", _vp), ("x0_ld", _i64), ("add0", _vp), ("add0_tstride", _i64),
                ("add_table", _vp), ("add_index", _vp), ("add_index_ld", _i64), ("add_table_rows", _i),
                ("w_ih", _vp * MAX_LAYERS), ("ldw_ih", _i64 * MAX_LAYERS),
                ("w_hh", _vp * MAX_LAYERS), ("ldw_hh", _i64 * MAX_LAYERS),
                ("bias", _vp * MAX_LAYERS),
                ("h0", _vp * MAX_LAYERS), ("ldh0", _i64),
                ("lengths", _vp),
                ("hs", _vp * MAX_LAYERS), ("ldh", _i64),
                ("cs", _vp * MAX_LAYERS),
                ("gates", _vp * MAX_LAYERS),
                ("cstate", _vp * MAX_LAYERS),
                ("zero_padded_k", _i),
                ("hdrop", _vp * MAX_LAYERS), ("drop_mask", _vp * MAX_LAYERS), ("drop_p", _f), ("drop_seed", C.c_uint32),
                ("persist_ws", _vp), ("persist_ws_bytes", _sz), ("poison", _vp), ("no_spin", _i)]


class RnnBwdDesc(C.Structure):
    _fields_ = [("cell", _i), ("dtype", _i), ("layers", _i), ("T", _i), ("B", _i), ("H", _i),
                ("w_hhT", _vp * MAX_LAYERS), ("ldw_hhT", _i64 * MAX_LAYERS),
                ("w_ihT", _vp * MAX_LAYERS), ("ldw_ihT", _i64 * MAX_LAYERS),
                ("lengths", _vp),
                ("dy", _vp), ("dy_ld", _i64),
                ("dy_a", _vp), ("dy_a_ld", _i64), ("dy_w", _vp), ("dy_w_ld", _i64), ("dy_k", _i),
                ("dh_last", _vp * MAX_LAYERS),
                ("hs", _vp * MAX_LAYERS), ("ldh", _i64),
                ("h0", _vp * MAX_LAYERS), ("ldh0", _i64),
                ("cs", _vp * MAX_LAYERS),
                ("gates", _vp * MAX_LAYERS),
                ("dG", _vp * MAX_LAYERS), ("ldg", _i64),
                ("dGh", _vp * MAX_LAYERS),
                ("dstate", _vp * MAX_LAYERS),
                ("dh0", _vp * MAX_LAYERS),
                ("split_ws", _vp), ("split_ws_bytes", _sz),
                ("drop_mask", _vp * MAX_LAYERS), ("drop_p", _f), ("drop_seed", C.c_uint32),
                ("persist_ws", _vp), ("persist_ws_bytes", _sz), ("poison", _vp), ("no_spin", _i)]


class GemmTnProblem(C.Structure):
    _fields_ = [("M", _i), ("N", _i), ("K", _i64), ("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64),
                ("C", _vp), ("ldc", _i64), ("accumulate", _i), ("colsum_out", _vp), ("colsum_accumulate", _i)]


class GemmTnF32Problem(C.Structure):
    _fields_ = [("M", _i), ("N", _i), ("K", _i64), ("A", _vp), ("lda", _i64), ("a_group", _i), ("a_gstride", _i64),
                ("B", _vp), ("ldb", _i64), ("b_group", _i), ("b_gstride", _i64), ("C", _vp), ("ldc", _i64), ("accumulate", _i),
                ("colsum_out", _vp), ("colsum_accumulate", _i), ("x3", _i)]


class PackJob(C.Structure):
    _fields_ = [("kind", _i), ("src_dtype", _i), ("dst_dtype", _i), ("R", _i), ("C", _i), ("block0", _i),
                ("src", _vp), ("lds", _i64), ("dst", _vp), ("ldd", _i64), ("dstT", _vp), ("ldt", _i64), ("src2", _vp)]


# name -> (restype, argtypes); mirrors include/mvae.h one to one (tests check every symbol is exported)
SIGNATURES = {
    "mvae_abi_version": (_i, []),
    "mvae_struct_size": (_sz, [_i]),
    "mvae_status_string": (C.c_char_p, [_i]),
    "mvae_knob_int": (_i, [C.c_char_p, _i]),
    "mvae_gemm_nt_workspace": (_sz, [_i, _i, _i, _i]),
    "mvae_gemm_nt": (_i, [_i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "mvae_gemm_tn_workspace": (_sz, [_i, _i, _i, _i]),
    "mvae_gemm_tn": (_i, [_i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "mvae_cast_transpose": (_i, [_i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "mvae_pack_job_blocks": (_i, [C.POINTER(PackJob)]),
    "mvae_pack_multi": (_i, [_i, _vp, _i, _vp]),
    "mvae_permute021": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "mvae_gather_rows_tb": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "mvae_scatter_rows_tb": (_i, [_i, _vp, _i, _i, _i, _vp, _i64, _i, _vp, _vp, _sz, _vp]),
    "mvae_scatter_rows_tb_workspace": (_sz, [_i, _i, _i, _i]),
    "mvae_onehot_tb": (_i, [_vp, _i, _i, _i, _vp, _i64, _vp]),
    "mvae_onehot_f32": (_i, [_vp, _i64, _i, _vp, _i64, _vp]),
    "mvae_rnn_fwd": (_i, [C.POINTER(RnnFwdDesc), _vp, C.POINTER(_vp)]),
    "mvae_rnn_fwd_persist_workspace": (_sz, [C.POINTER(RnnFwdDesc)]),
    "mvae_dropout_keep": (_i, [C.c_uint32, C.c_uint32, _f]),
    "mvae_gemm_tn_f32_colsum": (_i, [_i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _i, _vp, _sz, _vp]),
    "mvae_gemm_tn_colsum_supported": (_i, [_i, _i, _i]),
    "mvae_gemm_tn_grouped_supported": (_i, [_i, _i, _i64, _i64, _i64]),
    "mvae_gemm_tn_grouped_workspace": (_sz, [_i, C.POINTER(GemmTnProblem)]),
    "mvae_gemm_tn_grouped": (_i, [_i, C.POINTER(GemmTnProblem), _vp, _sz, _vp]),
    "mvae_gemm_tn_grouped_capped": (_i, [_i, C.POINTER(GemmTnProblem), _i, _vp, _sz, _vp]),
    "mvae_gemm_tn_f32_multi_workspace": (_sz, [_i, C.POINTER(GemmTnF32Problem)]),
    "mvae_gemm_tn_f32_multi": (_i, [_i, C.POINTER(GemmTnF32Problem), _vp, _sz, _vp]),
    "mvae_conv1d_dw_problem": (_i, [_i, _i, _i, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _i, C.POINTER(GemmTnF32Problem)]),
    "mvae_conv1d_unpack_dw": (_i, [_i, _i, _i, _vp, _i, _vp, _vp]),
    "mvae_gemm_tn_colsum_workspace": (_sz, [_i, _i, _i]),
    "mvae_gemm_tn_colsum": (_i, [_i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _i, _vp, _sz, _vp]),
    "mvae_rnn_bwd": (_i, [C.POINTER(RnnBwdDesc), _vp, C.POINTER(_vp)]),
    "mvae_rnn_bwd_workspace": (_sz, [C.POINTER(RnnBwdDesc)]),
    "mvae_rnn_bwd_persist_workspace": (_sz, [C.POINTER(RnnBwdDesc)]),
    "mvae_rowsum": (_i, [_i, _i, _i, _vp, _i64, _vp, _i, _vp]),
    "mvae_timesum": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "mvae_colsum_workspace": (_sz, [_i, _i]),
    "mvae_colsum": (_i, [_i, _i, _vp, _i64, _vp, _vp, _sz, _vp]),
    "mvae_colsum_t_workspace": (_sz, [_i, _i]),
    "mvae_colsum_t": (_i, [_i, _i, _i, _vp, _i64, _vp, _vp, _sz, _vp]),
    "mvae_conv1d_pack_weights": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "mvae_conv1d_selu_fwd_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "mvae_conv1d_selu_fwd": (_i, [_i, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "mvae_conv1d_selu_bwd_workspace": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "mvae_conv1d_selu_bwd": (_i, [_i, _i, _i, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "mvae_conv1d_act_fwd": (_i, [_i, _i, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "mvae_conv1d_act_bwd": (_i, [_i, _i, _i, _i, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "mvae_selu_bwd": (_i, [_i64, _vp, _vp, _vp]),
    "mvae_lambda_fwd": (_i, [_i, _i, _vp, _vp, _f, C.c_uint32, C.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    "mvae_normal_fill": (_i, [_i64, _f, C.c_uint32, C.c_uint64, _vp, _vp]),
    "mvae_normal_words": (None, [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32)]),
    "mvae_lambda_bwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvae_softmax_tb_fwd": (_i, [_i, _i, _i, _vp, _i64, _vp, _vp]),
    "mvae_softmax_tb_bwd": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "mvae_bce_kl_loss_workspace": (_sz, [_i64, _i64]),
    "mvae_bce_kl_loss_fwd": (_i, [_i64, _vp, _vp, _i64, _vp, _vp, _f, _vp, _vp, _sz, _vp]),
    "mvae_bce_kl_loss_bwd": (_i, [_i64, _vp, _vp, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "mvae_expand_indices": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "mvae_moses_latent_workspace": (_sz, [_i]),
    "mvae_moses_latent_fwd": (_i, [_i, _i, _vp, _vp, _vp, C.c_uint32, C.c_uint64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mvae_moses_latent_bwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvae_ce_loss_workspace": (_sz, [_i, _i]),
    "mvae_ce_loss_fwd": (_i, [_i, _i, _i, _vp, _i64, _vp, _i, _vp, _vp, _sz, _vp]),
    "mvae_ce_loss_bwd": (_i, [_i, _i, _i, _i, _vp, _i64, _vp, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mvae_permute102": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "mvae_moses_sample_step": (_i, [_i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _f, C.c_uint32, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "mvae_relu_bwd": (_i, [_i64, _vp, _vp, _vp]),
    "mvae_mask_rows_tb": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp]),
    "mvae_sumsq_workspace": (_sz, [_i64]),
    "mvae_sumsq": (_i, [_i64, _vp, _vp, _vp]),
    "mvae_clip_adam": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _i, _vp, _i, _vp, _vp]),
}

_lib = None


class MvaeError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvaeError(
            f"{LIB_PATH} not found: build it with molecular-vae_amd/csrc/build.sh "
            "(or __graft_entry__.build()).  There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.mvae_abi_version() != ABI_VERSION:
        raise MvaeError("libmvae_hip.so ABI version mismatch")
    for which, st in enumerate((RnnFwdDesc, RnnBwdDesc, GemmTnProblem, PackJob, GemmTnF32Problem)):
        if lib.mvae_struct_size(which) != C.sizeof(st):
            raise MvaeError(f"{st.__name__}: ctypes mirror is {C.sizeof(st)} bytes, the library's struct {lib.mvae_struct_size(which)}")
    _lib = lib
    return lib


_KNOB_SAID = [False]


def knob(name, default):
    """Host-side schedule knob (string): the environment variable `name` is honoured ONLY while MVAE_TUNING=1 is set as well (tests, A/B
    scripts) -- a training run ignores stray MVAE_* variables, as the library does (mvae_knob_int).  The first one honoured is reported once."""
    if os.environ.get("MVAE_TUNING", "0") in ("", "0"):
        return default
    v = os.environ.get(name)
    if v is None:
        return default
    if not _KNOB_SAID[0]:
        _KNOB_SAID[0] = True
        import sys
        print(f"mvae: MVAE_TUNING=1 -- schedule knob {name}={v} honoured (further knobs not reported)", file=sys.stderr)
    return v


def check(rc, what=""):
    if rc != 0:
        msg = load().mvae_status_string(rc).decode()
        raise MvaeError(f"{what or 'mvae call'} failed: {rc} ({msg})")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def dt_code(dtype):
    if dtype == torch.float32:
        return MVAE_F32
    if dtype == torch.bfloat16:
        return MVAE_BF16
    raise MvaeError(f"unsupported dtype {dtype}")


# Bumped by in-place parameter updates that bypass torch's version counters (FusedAdam.step, load_state_dict) so that the packed
# (bf16 / transposed) weight shadows are refreshed.  Code that edits parameter storage by hand (``p.data[...] = ...`` through a view
# torch does not track) must do ``_lib.PARAM_EPOCH[0] += 1`` itself.
PARAM_EPOCH = [0]

# Gradient sinks.  train.FusedAdam registers, ON each parameter object (attribute ``_mvae_sink``), where that parameter's gradient
# lives inside the optimiser's flat fp32 gradient buffer: (weakref(optimizer), flat, offset, numel).  The modules' backward passes
# then write parameter gradients straight into that buffer (no gather copy) and can start the data-parallel all-reduce of a finished
# range while the rest of backward is still running.  The entry dies with the parameter; a dead optimiser's entry is dropped on sight.
_SINK_ATTR = "_mvae_sink"


def register_grad_sink(p, owner, flat, off, poison=None):
    """poison: the spare fp32 slot behind the flat gradient buffer (FusedAdam): a persistent launch that gives up stores a NaN there, which
    makes mvae_clip_adam skip the step (mvae_rnn_*_desc.poison)."""
    setattr(p, _SINK_ATTR, (weakref.ref(owner), flat, off, p.numel(), poison))


def grad_poison(params):
    """The poison slot of the optimiser the first registered parameter of `params` belongs to, or None."""
    for p in params:
        e = getattr(p, _SINK_ATTR, None)
        if e is not None and e[0]() is not None:
            return e[4]
    return None


def clear_grad_sink(p):
    if hasattr(p, _SINK_ATTR):
        delattr(p, _SINK_ATTR)


def _sink(p):
    e = getattr(p, _SINK_ATTR, None)
    if e is None:
        return None
    owner = e[0]()
    if owner is None:
        delattr(p, _SINK_ATTR)
        return None
    return owner, e[1], e[2], e[3]


def grad_sink_range(params):
    """(owner, flat, lo, hi) when `params` occupy one contiguous, in-order range of one registered flat buffer; else None."""
    if not params:
        return None
    first = _sink(params[0])
    if first is None:
        return None
    owner, flat, lo, _ = first
    off = lo
    for q in params:
        e = _sink(q)
        if e is None or e[0] is not owner or e[1] is not flat or e[2] != off:
            return None
        off += e[3]
    return owner, flat, lo, off
