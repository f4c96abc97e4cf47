"""CPU tests of the host side: C-ABI library loads and exports every symbol of include/mvae.h, module surface /
state-dict parity with the reference layout, workspace bookkeeping (kernels stubbed out), sharding helpers."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import molecular_vae_amd as mv
from molecular_vae_amd import _lib as L
from molecular_vae_amd import models as M
from molecular_vae_amd import ops
from oracle import initparams as ip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "mvae.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mvae_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mvae.h but not exported"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    assert lib.mvae_abi_version() == L.ABI_VERSION == int(re.search(r"#define MVAE_ABI_VERSION (\d+)", header).group(1))
    assert lib.mvae_status_string(-2) == b"workspace too small"


def test_struct_layout_matches_header():
    # sizes computed by hand from include/mvae.h (LP64): guards the ctypes mirror against drift
    assert ctypes.sizeof(L.RnnFwdDesc) == 7 * 4 + 4 + 8 * 4 + (8 * 3 + 8) + 8 * (8 * 4) + 8 * 8 + 8 * 8 + 8 + 8 + 8 * 8 + 8 + 8 * 8 * 3 + 8 + (8 * 8 * 2 + 8) + 16 + 16
    assert ctypes.sizeof(L.RnnBwdDesc) == 6 * 4 + 8 * (8 * 4) + 8 + 16 + (8 * 4 + 8) + 8 * 8 + 8 * 8 + 8 + 8 * 8 + 8 + 8 * 8 * 3 + 8 + 8 * 8 * 3 + 16 + (8 * 8 + 8) + 16 + 16
    # ... and against what the compiler laid out (the loader refuses a mismatch as well)
    lib = L.load()
    for which, st in enumerate((L.RnnFwdDesc, L.RnnBwdDesc, L.GemmTnProblem)):
        assert lib.mvae_struct_size(which) == ctypes.sizeof(st)
    assert lib.mvae_struct_size(99) == 0
    # the host restatement of the device dropout hash agrees with the library's own (C) restatement
    lib = L.load()
    m = ops.dropout_keep_mask(12345, (2, 3, 5, 8), 0.2).reshape(-1)
    assert [lib.mvae_dropout_keep(12345, i, 0.2) for i in range(m.size)] == m.tolist()
    assert 0.7 < ops.dropout_keep_mask(7, (1, 50, 40, 64), 0.2).mean() < 0.9


def test_counter_normal_host_restatement_and_distribution():
    """The reparameterisation noise the kernels draw (mvae_lambda_fwd with eps == NULL): the numpy restatement's hash words equal the ones the
    library's own host-compiled function returns (same source as the device code), across the 2^31 counter fold; the normals they make pass
    moment / Kolmogorov-Smirnov / lag-correlation checks; streams of different seeds and disjoint offsets are uncorrelated."""
    from scipy import stats
    lib = L.load()
    w = (ctypes.c_uint32 * 2)()
    for seed, off in ((0, 0), (42, 7), (0xDEADBEEF, (1 << 31) - 3), (5, (1 << 40) + 12345), (0xFFFFFFFF, (1 << 63) + 9)):
        _, w1, w2 = ops.normal_draw(seed, off, 8)
        for k in range(8):
            lib.mvae_normal_words(seed, off + k, w)
            assert (int(w[0]), int(w[1])) == (int(w1[k]), int(w2[k])), (seed, off, k)
    n = 1 << 20
    x, _, _ = ops.normal_draw(1234, 0, n)
    assert abs(x.mean()) < 4 / np.sqrt(n) and abs(x.std() - 1) < 4e-3
    assert abs(stats.skew(x)) < 0.01 and abs(stats.kurtosis(x)) < 0.02
    assert stats.kstest(x, "norm").pvalue > 1e-3
    assert np.abs(x).max() < 5.9
    for lag in (1, 2, 292, 1024):                       # neighbouring elements, neighbouring rows (o = 292), a power of two
        assert abs(np.corrcoef(x[:-lag], x[lag:])[0, 1]) < 5 / np.sqrt(n)
    y, _, _ = ops.normal_draw(1235, 0, n)               # neighbouring seed
    z_, _, _ = ops.normal_draw(1234, n, n)              # the next block of the same stream
    assert abs(np.corrcoef(x, y)[0, 1]) < 5 / np.sqrt(n) and abs(np.corrcoef(x, z_)[0, 1]) < 5 / np.sqrt(n)
    # the Box-Muller pair (u1, u2) fills the unit square evenly, squares are uncorrelated at the lags that matter (volatility clustering would
    # not show in the plain correlations above)
    _, w1, w2 = ops.normal_draw(1234, 0, n)
    Hh, _, _ = np.histogram2d((w1 >> 8) / 2.0 ** 24, (w2 >> 8) / 2.0 ** 24, bins=32)
    chi = ((Hh - n / 1024) ** 2 / (n / 1024)).sum()
    assert 1e-4 < 1 - stats.chi2.cdf(chi, 1023) < 1 - 1e-4, chi
    for lag in (1, 292, 128 * 292):
        assert abs(np.corrcoef(x[:-lag] ** 2, x[lag:] ** 2)[0, 1]) < 5 / np.sqrt(n)
    a, _, _ = ops.normal_draw(1234, 100, 50, scale=1e-2)
    assert np.allclose(a, 1e-2 * x[100:150], rtol=1e-14, atol=0)     # a draw depends on (seed, counter) only


def test_noise_stream_and_cpu_noise_source():
    """ops.NoiseStream hands out consecutive counter ranges of one seeded stream; Lambda(noise="cpu") consumes the CPU default generator exactly
    as models.py:92 does (scale * torch.randn(B, o))."""
    torch.manual_seed(7)
    s1 = ops.NoiseStream(); a = s1.take(10); b = s1.take(5)
    assert a[0] == b[0] and (a[1], b[1]) == (0, 10) and s1.state() == dict(seed=a[0], counter=15)
    torch.manual_seed(7)
    s2 = ops.NoiseStream()
    assert s2.take(1)[0] != a[0]                         # a second stream of the same process: another seed
    s2.reseed(99, 1 << 40); assert s2.take(3) == (99, 1 << 40) and s2.take(1) == (99, (1 << 40) + 3)
    lam = M.Lambda(8, 5, noise="cpu")
    torch.manual_seed(3); e = lam.draw_eps(4, 5, torch.device("cpu"))
    torch.manual_seed(3); ref = 1e-2 * torch.randn(4, 5)
    assert torch.equal(e, ref)
    with pytest.raises(ValueError):
        M.Lambda(8, 5, noise="gpu")
    assert M.Lambda(8, 5).noise == "device" and mv.MolecularVAE().encoder.lmbd.noise == "device"
    assert mv.MolecularVAE(noise="cpu").encoder.lmbd.noise == "cpu"


def test_state_dict_keys_and_shapes_match_reference_layout():
    m = mv.MolecularVAE()
    sd = m.state_dict()
    shapes = ip.molvae_shapes()          # SURVEY.md section 8a listing
    assert set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert sum(p.numel() for p in m.parameters()) == 32285105


def test_seeded_init_is_reproducible_and_uses_reference_order():
    torch.manual_seed(42); a = mv.MolecularVAE().state_dict()
    torch.manual_seed(42); b = mv.MolecularVAE().state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    # same RNG consumption as torch.nn's own modules constructed in models.py order
    torch.manual_seed(42)
    e = torch.nn.Embedding(35, 30); l = torch.nn.LSTM(30, 72, 3, batch_first=True); c = torch.nn.Conv1d(120, 120, 18)
    assert torch.equal(a["encoder.embedding.weight"], e.weight)
    assert torch.equal(a["encoder.gru.weight_hh_l2"], l.weight_hh_l2)
    assert torch.equal(a["encoder.conv_1.0.bias"], c.bias)


def test_product_path_refuses_cpu():
    m = mv.MolecularVAE(i=24, o=16, c=12)
    with pytest.raises(L.MvaeError):
        m(torch.zeros(2, 24, dtype=torch.long))
    with pytest.raises(L.MvaeError):
        mv.bce_kl_loss(torch.rand(2, 3, 4), torch.rand(2, 3, 4), torch.rand(2, 5), torch.rand(2, 5), 3)


def test_forward_backward_bookkeeping_with_stubbed_kernels(monkeypatch):
    """Runs the whole host-side orchestration on CPU tensors with every kernel launch replaced by a no-op:
    checks argument bookkeeping (shapes, workspace names, gradient slots), not arithmetic."""
    calls = []

    def stub(name):
        def f(*a, **k):
            calls.append(name)
        return f
    for name in ("gemm_nt", "gemm_tn", "gemm_tn_f32_colsum", "colsum_t", "cast_transpose", "permute021", "gather_rows_tb", "scatter_rows_tb", "onehot_f32", "rowsum", "timesum", "colsum",
                 "selu_bwd", "conv1d_pack_weights", "conv1d_selu_fwd", "conv1d_selu_bwd", "lambda_fwd", "lambda_bwd", "softmax_tb_fwd", "softmax_tb_bwd",
                 "rnn_fwd", "rnn_bwd"):
        monkeypatch.setattr(ops, name, stub(name))
    monkeypatch.setattr(ops.PackList, "run", lambda self: calls.append("pack_multi"))      # the multi-tensor pack launch
    monkeypatch.setattr(ops.TnF32Batch, "run", lambda self: calls.append(f"tn_f32_multi[{len(self.probs)}]"))       # the encoder's parameter-gradient GEMMs: ONE launch
    monkeypatch.setattr(ops.TnF32Batch, "add_conv_dw", lambda self, *a, **k: self.probs.append("conv"))
    monkeypatch.setattr(M, "_require_cuda", lambda dev, what: None)
    enc = mv.MolEncoder(i=24, o=16, c=12, h_size=56, num_lstm=2)
    dec = mv.MolDecoder(i=16, o=24, c=12, num_gru=2, h_size=32, dtype=torch.float32)
    idx = torch.zeros(3, 24, dtype=torch.long)
    z, mu, logvar = enc(idx, torch.zeros(3, 16))
    recon = dec(z)
    assert recon.shape == (3, 24, 12) and mu.shape == (3, 16)
    (recon.sum() + mu.sum() + logvar.sum()).backward()
    for p in list(enc.parameters()) + list(dec.parameters()):
        assert p.grad is not None and p.grad.shape == p.shape
    assert calls.count("rnn_fwd") == 2 and calls.count("rnn_bwd") == 2 and calls.count("conv1d_selu_bwd") == 3
    # heads (2) + dense_1 + 3 convolutions + LSTM(2 layers): dW_hh x 2, dW_ih of layer 1, the layer-0 bias product -- all in the one batch
    assert calls.count("tn_f32_multi[11]") == 1                     # ... and the token-table gradient
    assert calls.count("pack_multi") == 2 and len(enc._pack_list.jobs) >= 10 and len(dec._pack_list.jobs) >= 8     # one pack launch per module
    # a second forward invalidates the saved workspace of the first
    z2, _, _ = enc(idx, torch.zeros(3, 16))
    z3, _, _ = enc(idx, torch.zeros(3, 16))
    with pytest.raises(L.MvaeError):
        z2.sum().backward()


def test_shard_helpers():
    assert mv.shard_batch(1024, 3, 8) == (384, 512)
    seen = []
    for r in range(4):
        s = mv.ShardedSampler(103, rank=r, world=4, seed=5)
        s.set_epoch(2)
        seen += list(iter(s))
        assert len(s) == 25
    assert len(set(seen)) == 100


def test_no_kernel_spills_to_scratch():
    """hipcc's per-kernel resource remarks (csrc/build.sh keeps them under csrc/build/): a kernel that spills or keeps an array in scratch
    runs several times slower without failing any numerical test (it happened: LDS-DMA descriptors selected at run time)."""
    import glob
    files = glob.glob(os.path.join(ROOT, "molecular-vae_amd", "csrc", "build", "*.usage.txt"))
    if not files:
        pytest.skip("no build remarks here (library built elsewhere)")
    kernels, bad = 0, []
    for f in files:
        name = None
        for line in open(f, errors="replace"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1); kernels += 1
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) > 0:
                bad.append((name, int(m.group(1))))
            m = re.search(r"VGPRs Spill: (\d+)", line)
            if m and int(m.group(1)) > 0:
                bad.append((name, "spill", int(m.group(1))))
    assert kernels > 50 and not bad, bad


def test_c_abi_rejects_bad_arguments_before_touching_the_device():
    """Error behaviour of the boundary (include/mvae.h: int status, no exceptions, nothing launched): every call below must come back with
    a negative status from the argument checks alone -- this runs on the GPU-less build container."""
    import ctypes as C
    lib = L.load()
    INV, WSP, UNS = -1, -2, -3
    assert lib.mvae_status_string(0) == b"ok" and lib.mvae_status_string(INV) == b"invalid argument"
    assert lib.mvae_status_string(UNS) == b"unsupported combination" and lib.mvae_status_string(-77) == b"unknown status"
    buf = (C.c_float * 4096)()
    p = C.cast(buf, C.c_void_p)
    # GEMMs: null operands, unknown dtype, accumulate into a bf16 output
    assert lib.mvae_gemm_nt(L.MVAE_F32, 8, 8, 8, None, 8, p, 8, p, 8, L.MVAE_F32, None, 0, 0, None, 0, None) == INV
    assert lib.mvae_gemm_nt(7, 8, 8, 8, p, 8, p, 8, p, 8, L.MVAE_F32, None, 0, 0, None, 0, None) == INV
    assert lib.mvae_gemm_nt(L.MVAE_F32, 8, 8, 8, p, 8, p, 8, p, 8, L.MVAE_BF16, None, 0, 1, None, 0, None) == INV
    assert lib.mvae_gemm_nt(L.MVAE_F32, 0, 8, 8, None, 8, None, 8, None, 8, L.MVAE_F32, None, 0, 0, None, 0, None) == 0      # empty output: nothing to do
    assert lib.mvae_gemm_tn(7, 8, 8, 8, p, 8, p, 8, p, 8, L.MVAE_F32, None, 0, 0, None, 0, None) == UNS
    assert lib.mvae_gemm_tn(L.MVAE_F32, 8, 8, 8, p, 8, p, 8, p, 8, L.MVAE_F32, p, 0, 0, None, 0, None) == UNS                # exact-f32 TN: no fused bias
    assert lib.mvae_gemm_tn_grouped(0, None, None, 0, None) == INV
    assert lib.mvae_gemm_tn_grouped_supported(4096, 1000, 61440, 4160, 1088) == 0                                        # N not a multiple of 256
    assert lib.mvae_gemm_tn_grouped_supported(4096, 1024, 61440, 4160, 1088) == 1
    assert lib.mvae_gemm_tn_colsum(8, 8, 8, p, 8, p, 8, p, 8, 0, None, 0, None, 0, None) == INV
    # recurrent stacks: empty descriptor, unknown cell, zero sizes, a missing save buffer
    d = L.RnnFwdDesc()
    assert lib.mvae_rnn_fwd(None, None, None) == INV
    d.cell = 9
    assert lib.mvae_rnn_fwd(C.byref(d), None, None) == UNS
    d.cell, d.dtype, d.layers, d.T, d.B, d.H = L.CELL_LSTM, L.MVAE_BF16, 1, 0, 4, 8
    assert lib.mvae_rnn_fwd(C.byref(d), None, None) == INV
    d.T = 3
    assert lib.mvae_rnn_fwd(C.byref(d), None, None) == INV            # no weights / outputs
    d.layers = L.MAX_LAYERS + 1
    assert lib.mvae_rnn_fwd(C.byref(d), None, None) == INV
    b = L.RnnBwdDesc()
    assert lib.mvae_rnn_bwd(None, None, None) == INV
    b.cell, b.dtype, b.layers, b.T, b.B, b.H = L.CELL_LSTM, 5, 1, 3, 4, 8
    assert lib.mvae_rnn_bwd(C.byref(b), None, None) == INV
    # helpers
    assert lib.mvae_onehot_tb(p, 4, 4, 30, p, 30, None) == INV             # leading dimension not a multiple of 8
    assert lib.mvae_onehot_tb(p, 4, 4, 40, p, 32, None) == INV             # table wider than the row
    assert lib.mvae_colsum_t(L.MVAE_BF16, 16, 8, p, 12, p, p, 1 << 20, None) == INV       # ldx % 8
    assert lib.mvae_colsum_t(L.MVAE_BF16, 16, 8, p, 16, p, None, 0, None) == WSP
    assert lib.mvae_scatter_rows_tb(L.MVAE_F32, None, 4, 4, 30, p, 8, 8, p, None, 0, None) == INV
    assert lib.mvae_timesum(L.MVAE_F32, 0, 4, 8, p, p, None) == INV
    assert lib.mvae_rowsum(L.MVAE_F32, 0, 4, p, 4, p, 0, None) == INV
    assert lib.mvae_relu_bwd(8, None, p, None) == INV


def test_gru_rowres_time_loops_issue_exactly_the_counted_memory_operations(tmp_path):
    """the counted s_waitcnt vmcnt(N) of gru_rowres_fwd / _bwd (NLD / NST in rnn_rowres.hip) are right only while the compiler
    emits exactly that many vector-memory instructions per time step -- no merged, split or predicated-away access, no scratch traffic.
    Disassemble the shipped library and count the global loads / stores inside each kernel's time loop (the widest backward branch)."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    lib = os.path.join(str(tmp_path), "lib.so")
    shutil.copy(L.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], cwd=str(tmp_path), capture_output=True, check=True)
    text = None
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" in f:
            d = subprocess.run([objdump, "-d", os.path.join(str(tmp_path), f)], capture_output=True, text=True).stdout
            if "gru_rowres_fwd_kernel" in d:
                text = d
    assert text is not None, "gru_rowres kernels not found in the library's gfx950 code objects"
    want = {"gru_rowres_fwd_kernelILi256ELb1E": (6, 10), "gru_rowres_fwd_kernelILi256ELb0E": (6, 2), "gru_rowres_bwd_kernelILi256E": (10, 8)}
    for key, (nld, nst) in want.items():
        m = re.search(r"<[^>]*" + key + r"[^>]*>:\n(.*?)s_endpgm", text, flags=re.S)
        assert m, key
        ins = []                                    # (address, mnemonic, operand text)
        for line in m.group(1).splitlines():
            mm = re.match(r"\s*(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):", line)
            if mm:
                ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
        assert not any(i[1].startswith("scratch_") for i in ins), key
        loops = []
        for a, mn, ops_ in ins:
            if mn.startswith("s_cbranch") or mn == "s_branch":
                off = int(ops_.split()[0])
                if off >= 32768:
                    loops.append((a + 4 + (off - 65536) * 4, a))       # backward branch: (target, branch address)
        assert loops, key
        lo, hi = max(loops, key=lambda t: t[1] - t[0])                # the time loop
        body = [i for i in ins if lo <= i[0] <= hi]
        loads = sum(1 for i in body if i[1].startswith(("global_load", "buffer_load")))
        stores = sum(1 for i in body if i[1].startswith(("global_store", "buffer_store")))
        steps = sum(1 for i in body if i[1] == "s_barrier")            # one workgroup barrier per time step (the compiler may unroll the loop)
        assert steps >= 1 and (loads, stores) == (nld * steps, nst * steps), (key, loads, stores, steps)


def test_no_wide_buffer_store_takes_its_offset_from_an_sgpr(tmp_path):
    """Round-4 finding (rnn_persist_bwd.hip store_b128_wt): on gfx950 a buffer_store_dwordx4 whose soffset is an SGPR gets no wait
    state from the compiler before a VALU instruction overwrites its data registers, and dword 1 of the stored vector was then the NEXT value.
    Every >64-bit buffer store of the shipped library must carry a literal soffset (the compiler's hazard rule then applies); the persistent
    kernels must also be free of scratch traffic (a scratch reload inside their slot loops would break the counted vmcnt waits)."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    lib = os.path.join(str(tmp_path), "lib.so")
    shutil.copy(L.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], cwd=str(tmp_path), capture_output=True, check=True)
    wide, seen_persist = 0, 0
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" not in f:
            continue
        d = subprocess.run([objdump, "-d", os.path.join(str(tmp_path), f)], capture_output=True, text=True).stdout
        for line in d.splitlines():
            mm = re.match(r"\s*(buffer_store_dwordx[34]|buffer_store_format_xyzw?)\s+(.*?)//", line)
            if mm:
                wide += 1
                ops_ = mm.group(2).replace(",", " ").split()
                soff = [o for o in ops_ if not o.startswith(("v", "s[", "off", "idx", "sc", "nt", "lds"))]
                assert soff and not soff[0].startswith("s"), f"SGPR soffset on a wide buffer store: {line.strip()}"
        for key in ("lstm_persist_fwd_kernel", "lstm_persist_bwd_kernel"):
            for m in re.finditer(r"<[^>]*" + key + r"[^>]*>:\n(.*?)s_endpgm", d, flags=re.S):
                seen_persist += 1
                assert "scratch_" not in m.group(1), key
    assert wide >= 8 and seen_persist >= 3      # the backward kernel's exchange / dG stores exist and were checked; fwd<save>, fwd<no save>, bwd


def test_persistent_schedule_queries_answer_not_served_without_a_256_cu_device():
    """mvae_rnn_fwd_persist_workspace / mvae_rnn_bwd_persist_workspace: 0 = "this shape / device is not served" -- for a shape outside the one the
    dataflow kernels serve, for a NULL descriptor, and (here: no GPU) for the served shape on a machine without a 256-CU device.  The callers
    (ops.rnn_fwd / rnn_bwd) then keep the launch-per-diagonal schedules; asking for persist=True explicitly raises instead."""
    lib = L.load()
    assert lib.mvae_rnn_fwd_persist_workspace(None) == 0 and lib.mvae_rnn_bwd_persist_workspace(None) == 0
    buf = (ctypes.c_char * 64)()
    addr = ctypes.addressof(buf) & ~15
    f = L.RnnFwdDesc(); b = L.RnnBwdDesc()
    for d in (f, b):
        d.cell, d.dtype, d.layers, d.T, d.B, d.H = L.CELL_LSTM, L.MVAE_BF16, 4, 120, 128, 1024
    for l in range(4):
        f.hs[l] = f.w_hh[l] = f.w_ih[l] = f.cstate[l] = addr
        f.ldw_hh[l] = f.ldw_ih[l] = 1088
        b.w_hhT[l] = b.w_ihT[l] = b.cs[l] = b.gates[l] = b.dG[l] = addr
        b.ldw_hhT[l] = b.ldw_ihT[l] = 4160
    f.ldh = 1088
    b.ldg, b.dy, b.dy_ld = 4160, addr, 1024
    served = (lib.mvae_rnn_fwd_persist_workspace(ctypes.byref(f)), lib.mvae_rnn_bwd_persist_workspace(ctypes.byref(b)))
    import torch
    if not torch.cuda.is_available():
        assert served == (0, 0)                 # no device: not served, never an error
    else:
        assert served[0] in (0, 64 + 4 * 120 * 64 * 4) and (served[1] == 0 or served[1] > 16 * 1024 * 1024)
    f.H = b.H = 512                             # another hidden size: never
    assert lib.mvae_rnn_fwd_persist_workspace(ctypes.byref(f)) == 0 and lib.mvae_rnn_bwd_persist_workspace(ctypes.byref(b)) == 0
    b.H = 1024; b.dy = None                     # the output gradient as a product only (dy_a): the wavefront form
    assert lib.mvae_rnn_bwd_persist_workspace(ctypes.byref(b)) == 0


def test_persistent_kernels_time_loops_issue_exactly_the_counted_memory_operations(tmp_path):
    """The operand rings of rnn_persist.hip / rnn_persist_bwd.hip wait with counted s_waitcnt vmcnt(N): N is right only while one time step
    issues exactly 4 LDS-DMA instructions per slot (128 per step with an x part, 64 without) and the stores the counts name -- no merged,
    split or predicated-away access.  Disassemble the shipped library and count inside each kernel's time loops."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    lib = os.path.join(str(tmp_path), "lib.so")
    shutil.copy(L.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], cwd=str(tmp_path), capture_output=True, check=True)
    found = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" not in f:
            continue
        d = subprocess.run([objdump, "-d", os.path.join(str(tmp_path), f)], capture_output=True, text=True).stdout
        for key in ("lstm_persist_fwd_kernelILb1E", "lstm_persist_fwd_kernelILb0E", "lstm_persist_bwd_kernel"):
            m = re.search(r"<[^>]*" + key + r"[^>]*>:\n(.*?)s_endpgm", d, flags=re.S)
            if not m:
                continue
            ins = []
            for line in m.group(1).splitlines():
                mm = re.match(r"\s*(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):", line)
                if mm:
                    ins.append((int(mm.group(3), 16), mm.group(1), mm.group(2)))
            loops = []
            for a, mn, ops_ in ins:
                if mn.startswith("s_cbranch") or mn == "s_branch":
                    try:
                        off = int(ops_.split()[0])
                    except ValueError:
                        continue
                    if off >= 32768:
                        loops.append((a + 4 + (off - 65536) * 4, a))
            best = {}                                # MFMAs per step -> the widest loop with that many (the time loop of that body)
            for lo, hi in loops:
                body = [i for i in ins if lo <= i[0] <= hi]
                mf = sum(1 for i in body if i[1].startswith("v_mfma"))
                if mf in (256, 512) and (mf not in best or hi - lo > best[mf][0]):
                    best[mf] = (hi - lo, body)
            found[key] = {mf: dict(dma=sum(1 for i in b if i[1].startswith("buffer_load") and " lds" in i[2]),
                                   st4=sum(1 for i in b if i[1].startswith("buffer_store_dwordx4")),
                                   st2=sum(1 for i in b if i[1].startswith("buffer_store_dwordx2")),
                                   scratch=sum(1 for i in b if i[1].startswith("scratch_")))
                          for mf, (_, b) in best.items()}
    assert set(found) == {"lstm_persist_fwd_kernelILb1E", "lstm_persist_fwd_kernelILb0E", "lstm_persist_bwd_kernel"}, found.keys()
    for key, loops_ in found.items():
        # the body with an x part (512 MFMAs per step) must be there; the one without (layer 0 / top layer: 256) when the compiler closed its
        # loop with a short branch this scan can see
        assert 512 in loops_ and set(loops_) <= {256, 512}, (key, loops_.keys())
        for mf, c in loops_.items():
            assert c["dma"] == mf // 4 and c["scratch"] == 0, (key, mf, c)
    # stores per step: forward with saved state 2 x (h + 4 gates + c) 8-byte stores, forward-only 2 h stores; backward 8 partial + 4 dG 16-byte stores
    assert all(c["st2"] == 12 and c["st4"] == 0 for c in found["lstm_persist_fwd_kernelILb1E"].values()), found
    assert all(c["st2"] == 2 and c["st4"] == 0 for c in found["lstm_persist_fwd_kernelILb0E"].values()), found
    assert all(c["st4"] == 12 and c["st2"] == 0 for c in found["lstm_persist_bwd_kernel"].values()), found
