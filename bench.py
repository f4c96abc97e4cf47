#!/usr/bin/env python3
"""bench.py -- molecules/s of one full ELBO training step (fwd + loss + bwd + clip + Adam [+ grad all-reduce]).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches it under
torch.distributed.run with one rank per GPU (RCCL); when it is started directly with --gpus N > 1 and no WORLD_SIZE it starts
those N ranks itself as a child `python -m torch.distributed.run ...` (decided before anything touches the GPU; never re-execs)
and fails loudly when the box has fewer than N GPUs or fewer than N ranks join.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json: "molecules/s (ELBO fwd+bwd+step) ... batch 1024, 1/2/4/8 MI355X"): a "step" is train.py:95-104 on one
synthetic minibatch: idx = randint(0, 35, (b, 120)), one-hot target, MolecularVAE(i=120, o=292, c=35) with the reference's default
init under manual_seed(42), bf16 decoder-LSTM storage / fp32 accumulate + fp32 master weights.  The GLOBAL batch is fixed at 1024
(configs[2]) and split over the ranks (b = 1024 / N: strong scaling); `--batch b` fixes the per-GPU batch instead (weak scaling).
Inputs are resident in HBM before the timed region.

Extra objects on the line:
  roofline     -- the decoder LSTM wavefront step kernel with the larger share of the step (at B = 1024 the fused backward step), priced against
                  the BINDING roof of that launch: `bound` = "hbm" when its algorithmic bytes / 8 TB/s exceed its algorithmic FLOPs / 2.5 PF/s
                  (B = 1024: 250 FLOP/B, below the ridge), else "mfma"; algorithmic bytes (or FLOPs) per launch / average launch duration
                  measured live with HIP events on the launching stream.  Both roofs' figures under `mfma` / `hbm`, the other direction's
                  kernel under `other`.
  cpu_baseline -- oracle/torch_ref.py (the reference architecture on stock torch.nn, CPU) timed on this host's cores on a bounded
                  sample (rank 0, N = 1 only).
  secondary    -- N = 1 only: the same step at configs[1] (B=512 bf16), at the per-rank shape of configs[2] (b=128 bf16, with its own
                  roofline object), in the exact-f32 parity mode (B=1024), the mosesvae.VAE step of configs[3] (B=1024) and the long-sequence
                  stress of configs[4] (L=256, C=64, B=2048, with the GB/s of its HBM-bound kernels), each on fewer steps.
  input_pipeline -- N = 1: the main workload's step fed through DeviceDataset.batches (250k-molecule corpus in HBM) instead of one resident batch.
  comm         -- N > 1 only: allreduce_exposed_ms (HIP-event time the main stream waited for the gradient all-reduce in the optimiser
                  step), the world size and backend torch.distributed reported, gradient bytes per step.
`--model moses` / `--model models2d` bench the mosesvae.VAE path (configs[3]) and the models2d.VAE variant the same way.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_SEQ, VOCAB, LATENT = 120, 35, 292
H_DEC, NL_DEC = 1024, 4
BF16_DENSE_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md, Peak BF16 MFMA dense
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


def flops_per_molecule(L=L_SEQ, C=VOCAB, o=LATENT):
    """exact MACs * 2, bwd = 2 x fwd."""
    macs = (L * (4 * 72 * (30 + 72) + 2 * 4 * 72 * 144) + 18 * (120 * 55 * L + 64 * 38 * 120 + 64 * 21 * 64) + 1344 * 512
            + 2 * 512 * o + o * o + L * (4 * 1024 * (o + 1024) + 3 * 4 * 1024 * 2048) + L * 1024 * C)
    return 3 * 2 * macs


def lstm_step_flops(B, T):
    """Algorithmic FLOPs of the decoder wavefront per LAUNCH (averaged over the T+3 launches of one pass).
    fwd cell (l,t): 2*B*4H*(K_x + H) with K_x = H for l>=1; layer 0's x-part is hoisted (time-invariant) so it is
    NOT counted here.  bwd cell: 2*B*H*(4H [W_hh] + 4H [W_ih of the layer above, l<3])."""
    H = H_DEC
    per_t_fwd = 2 * B * 4 * H * (H + 3 * 2 * H)
    per_t_bwd = 2 * B * H * (4 * 4 * H + 3 * 4 * H)
    n_launch = T + NL_DEC - 1
    return per_t_fwd * T / n_launch, per_t_bwd * T / n_launch, n_launch


def lstm_step_bytes(B):
    """Algorithmic HBM bytes of ONE full wavefront diagonal (4 cells) of the decoder LSTM, bf16 storage:
    forward: 7 weight panels [4H, H] + x (3) / h (4) operands [B, H] + previous cell state fp32 (4) read; h, c (fp32 + bf16), gates written;
    backward: 7 transposed weight panels + 7 dG operands [B, 4H] + saved gates, c, c_prev, carried dc (fp32) read; dG and dc written."""
    H = H_DEC
    w = 7 * 4 * H * H * 2
    fwd = w + 7 * B * H * 2 + 4 * B * H * 4 + 4 * (B * H * 2 + B * H * 4 + B * H * 2 + B * 4 * H * 2)
    bwd = w + 7 * B * 4 * H * 2 + 4 * (B * 4 * H * 2 + 2 * B * H * 2 + B * H * 4) + 4 * (B * 4 * H * 2 + B * H * 4)
    return fwd, bwd


def lstm_persist_bytes(B):
    """The same for the weights-resident dataflow passes (rnn_persist.hip / rnn_persist_bwd.hip: b = 128 / 256): the weights never move, so a
    diagonal's algorithmic bytes are the hand-off operands and the saved state only -- forward: 7 x / h operands read, h + c (bf16) + gates
    written; backward: 7 dG operands + saved gates, c, c_prev + dy (fp32, top layer) read, dG written.  (Not counted: the K-quarter partial
    sums of the backward pass, 8 MB out + 8 MB in per diagonal at b = 128 -- traffic of this kernel's structure, not of the math; the PMC
    figure in `traffic` has them.)"""
    H = H_DEC
    fwd = 7 * B * H * 2 + 4 * (B * H * 2 + B * H * 2 + B * 4 * H * 2)
    bwd = 7 * B * 4 * H * 2 + 4 * (B * 4 * H * 2 + 2 * B * H * 2) + B * H * 4 + 4 * B * 4 * H * 2
    return fwd, bwd


def host_cores(cap=32):
    """Usable host cores: the affinity mask clipped by the cgroup CPU quota (an over-subscribed OpenMP team spins)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def die(msg, code=2):
    print(f"bench.py: error: {msg}", file=sys.stderr, flush=True)
    sys.exit(code)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=1024, help="global batch, split over the ranks (BASELINE.json: batch 1024; strong scaling)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch; when given the per-GPU work is fixed instead (weak scaling)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="molvae", choices=["molvae", "moses", "models2d"])
    ap.add_argument("--seq-len", type=int, default=L_SEQ, help="molvae: sequence length (BASELINE configs[4]: 256)")
    ap.add_argument("--vocab", type=int, default=VOCAB, help="molvae: vocabulary size (BASELINE configs[4]: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the B=512 / b=128 / f32 secondary measurements (N = 1)")
    ap.add_argument("--cpu-steps", type=int, default=16)
    ap.add_argument("--shard-optimizer", action="store_true", help="N > 1: reduce-scatter + sharded clip/Adam + all-gather instead of all-reduce")
    ap.add_argument("--grad-compress", default=None, choices=[None, "bf16"], help="N > 1: all-reduce the gradient as bfloat16")
    ap.add_argument("--force-comm", action="store_true", help="N = 1: initialise a one-rank RCCL process group and issue every collective of the "
                                                              "N-rank step anyway (GradSync(force=True)); prints `comm` for the one GPU at hand")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only to rehearse the multi-rank "
                                                       "path on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as a child torch.distributed.run and relay its exit code.  Nothing in this
    process has touched the GPU (device_count() does not initialise HIP on this image)."""
    import torch
    have = torch.cuda.device_count()
    if args.backend == "nccl" and have < args.gpus:
        die(f"--gpus {args.gpus} requested but this box has {have} GPU(s); refusing to report a smaller job as n_gpus={args.gpus}")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("starting " + " ".join(cmd))
    sys.exit(subprocess.run(cmd, env=env).returncode)


# ------------------------------------------------------------------------------------------------------------------ workloads
class MolVaeWorkload:
    name = "MolecularVAE(i=120,o=292,c=35) ELBO train step: fwd+loss+bwd+clip(3.0)+Adam(8e-4), synthetic one-hot SMILES [B,120,35]"

    def __init__(self, B, dtype, dev, rank, sync, L=L_SEQ, C=VOCAB, noise="device"):
        import torch
        import molecular_vae_amd as mv
        self.mv, self.B, self.L, self.C = mv, B, L, C
        self.noise = noise
        if (L, C) != (L_SEQ, VOCAB):
            self.name = self.name.replace("i=120,o=292,c=35", f"i={L},o=292,c={C}").replace("[B,120,35]", f"[B,{L},{C}]")
        torch.manual_seed(42)                                     # train.py:73
        # Reparameterisation noise: the PRODUCT default (noise="device") draws 1e-2 * N(0, 1) inside mvae_lambda_fwd from the library's counter
        # hash -- no bench-side override; noise="cpu" (the reference's CPU generator stream of models.py:92 through a pinned ring) is timed as
        # secondary.headline_B1024_cpu_eps.  config.eps_source says which one a line used.
        self.model = mv.MolecularVAE(i=L, o=LATENT, c=C, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, noise=noise).to(dev)
        self.optimizer = mv.FusedAdam(self.model.parameters(), lr=0.0008, max_grad_norm=3.0, grad_sync=sync, shard_optimizer=bool(getattr(sync, "shard_optimizer", False)))   # train.py:81,102
        self.loss_function = mv.make_loss_function(L)
        self.n_params = sum(p.numel() for p in self.model.parameters())
        g = torch.Generator().manual_seed(1234 + rank)
        self.data = torch.randint(0, C, (B, L), generator=g).to(dev)
        self.ohe = torch.nn.functional.one_hot(self.data, C).float()    # the (idx, ohe) pair MoleLoader yields, resident in HBM

    def step(self):
        return self.mv.train_step(self.model, self.optimizer, self.loss_function, self.data, self.ohe)

    def flops_per_step(self):
        return flops_per_molecule(self.L, self.C) * self.B

    def pipeline_steps(self, steps, warmup, n_corpus=250000):
        """The same step fed THROUGH the device-side input pipeline: a 250k-molecule uint8 corpus resident in HBM (30 MB),
        per-epoch shuffle, mvae_expand_indices to the (idx, ohe) pair of data_loader.py:26-31 -- instead of one resident batch.  ms / step."""
        import torch
        from molecular_vae_amd import data as D
        g = torch.Generator().manual_seed(99)
        store = torch.randint(0, self.C, (n_corpus, self.L), generator=g, dtype=torch.uint8).numpy()
        ds = D.DeviceDataset(store, self.C, self.data.device)
        it = ds.batches(self.B, epoch=1, seed=0)
        for _ in range(warmup):
            d, o = next(it); self.mv.train_step(self.model, self.optimizer, self.loss_function, d, o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            d, o = next(it); self.mv.train_step(self.model, self.optimizer, self.loss_function, d, o)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    def hbm_kernels(self, tag_ms):
        """HBM-bound kernels of the step: algorithmic bytes = one read + one write of their operands; achieved GB/s from
        the HIP-event time of the launches (BASELINE configs[4] asks for these against the 8 TB/s roof)."""
        n, m, P = self.B * self.L * self.C, self.B * LATENT, self.n_params
        algo = {"hbm_bce_kl_loss_fwd": 4 * (2 * n + 2 * m), "hbm_bce_kl_loss_bwd": 4 * (3 * n + 4 * m), "hbm_softmax_fwd": 4 * 2 * n,
                "hbm_softmax_bwd": 4 * 2 * n + 2 * n, "hbm_sumsq_clip_adam": 4 * P + 7 * 4 * P}
        out = {}
        for k, b in algo.items():
            if k in tag_ms and tag_ms[k] > 0:
                gbs = b / (tag_ms[k] * 1e-3) / 1e9
                out[k[4:]] = dict(bytes=b, us=round(1e3 * tag_ms[k], 1), achieved_GBps=round(gbs, 1), frac_of_8TBps=round(gbs / HBM_PEAK_GBS, 4))
        return out

    def roofline(self, tag_ms, dtype):
        """`kernel` is the wavefront step kernel with the larger share of the step's time -- at B >= 1024 the fused backward step
        (lstm_step_bwd_kernel<bf16,128,128,4,WS>: one launch per diagonal), at smaller batches the backward PAIR (split GEMM launch +
        element-wise launch) or the forward step, whichever is slower; the other direction's figures stand beside it (`other`)."""
        fwd_f, bwd_f, n_launch = lstm_step_flops(self.B, self.L)
        fwd_us = 1e3 * tag_ms.get("dec_lstm_fwd", float("nan")) / n_launch
        bwd_us = 1e3 * tag_ms.get("dec_lstm_bwd", float("nan")) / n_launch
        peak = BF16_DENSE_PEAK_TFLOPS if dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
        fused_bwd = dtype == "bf16" and self.B >= 1024 and self.B % 128 == 0
        from molecular_vae_amd import ops as _ops
        persist = dtype == "bf16" and self.B in (128, 256) and (self.L, H_DEC) == (self.L, 1024) and _ops.PERSIST_STATS["launches"] > 0
        persist_b = dtype == "bf16" and self.B in (128, 256) and _ops.PERSIST_STATS["bwd_launches"] > 0
        leg = {"fwd": dict(kernel=("lstm_persist_fwd_kernel (weights-resident dataflow pass: ONE launch, time per diagonal)" if persist else
                                   "lstm_step_fwd_gm_kernel" if dtype == "bf16" and (self.B >= 1024 or self.B <= 128) else "lstm_step_fwd_kernel"),
                           us=fwd_us, flops=fwd_f, pmc_key=("lstm_persist_fwd" if persist else "lstm_step_fwd")),
               "bwd": dict(kernel=("lstm_persist_bwd_kernel (weights-resident dataflow pass: ONE launch, time per diagonal)" if persist_b else
                                   "lstm_step_bwd_kernel (fused gate-derivative epilogue)" if fused_bwd or dtype != "bf16" else
                                   "lstm_step_bwd_kernel + lstm_bwd_epi_kernel (launch pair)"), us=bwd_us, flops=bwd_f, pmc_key=("lstm_persist_bwd" if persist_b else "lstm_step_bwd"))}
        by_f, by_b = lstm_step_bytes(self.B)
        pby_f, pby_b = lstm_persist_bytes(self.B)
        leg["fwd"]["bytes"] = (pby_f if persist else by_f) * self.L / n_launch
        leg["bwd"]["bytes"] = (pby_b if persist_b else by_b) * self.L / n_launch
        for v in leg.values():
            v["tflops"] = v["flops"] / (v["us"] * 1e-6) / 1e12
            # BOTH floors of the launch: MFMA time of its algorithmic FLOPs, HBM time of its algorithmic bytes (bf16: intensity 250 FLOP/B at
            # B = 1024, below the 312 FLOP/B ridge -- the HBM floor is the higher one)
            v["floor_us"] = dict(mfma=round(v["flops"] / (peak * 1e12) * 1e6, 2), hbm=round(v["bytes"] / (HBM_PEAK_GBS * 1e9) * 1e6, 2))
        dom = "bwd" if not (bwd_us < fwd_us) else "fwd"
        oth = "fwd" if dom == "bwd" else "bwd"
        d, o = leg[dom], leg[oth]
        traffic, src = pmc_traffic(d["pmc_key"], self.B, dtype)
        hbm_bound = dtype == "bf16" and d["floor_us"]["hbm"] > d["floor_us"]["mfma"]
        # `bound` names the BINDING roof of the dominant launch -- the higher of its two floors: at B = 1024 the algorithmic bytes / 8 TB/s exceed
        # the algorithmic FLOPs / 2.5 PF/s (intensity 250 FLOP/B, below the 312 FLOP/B ridge), so the launch is priced against HBM there and
        # `achieved` / `peak` / `unit` / `frac` are GB/s; where the MFMA floor is the higher one (persistent passes, small batches) they are
        # TFLOP/s.  The other roof's figures stand beside it under `mfma` / `hbm` on every line.
        gbps = d["bytes"] / (d["us"] * 1e-6) / 1e9
        mfma_obj = dict(achieved=round(d["tflops"], 2), peak=peak, unit="TFLOP/s", frac=round(d["tflops"] / peak, 4))
        hbm_obj = dict(achieved=round(gbps, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbps / HBM_PEAK_GBS, 4))
        head = dict(bound="hbm", **hbm_obj) if hbm_bound else dict(bound="mfma", **mfma_obj)
        return dict(**head, kernel=d["kernel"], mfma=mfma_obj, hbm=hbm_obj,
                    floors_us=d["floor_us"], higher_floor=("hbm" if hbm_bound else "mfma"),
                    frac_of_higher_floor=round(max(d["floor_us"].values()) / d["us"], 4), algorithmic_bytes_per_launch=int(d["bytes"]),
                    achieved_GBps_algorithmic=round(gbps, 1),
                    traffic=traffic, traffic_source=src, launches_per_pass=n_launch,
                    share_of_step_ms=round(d["us"] * n_launch * 1e-3, 3),
                    avg_launch_us=dict(lstm_step_fwd=round(fwd_us, 2), lstm_step_bwd=round(bwd_us, 2)),
                    flops_per_launch=dict(lstm_step_fwd=fwd_f, lstm_step_bwd=bwd_f),
                    other=dict(kernel=o["kernel"], achieved=round(o["tflops"], 2), frac=round(o["tflops"] / peak, 4), floors_us=o["floor_us"],
                               share_of_step_ms=round(o["us"] * n_launch * 1e-3, 3)),
                    hbm_bound_kernels=self.hbm_kernels(tag_ms))

    def config(self, world):
        return dict(workload=self.name, per_gpu_batch=self.B, global_batch=self.B * world, seq_len=self.L, vocab=self.C, latent=LATENT,
                    parallelism=f"dp{world}",
                    eps_source=("product default, MolecularVAE(noise='device'): drawn inside mvae_lambda_fwd (counter hash + Box-Muller, explicit seed / element counter)"
                                if self.noise == "device" else
                                "MolecularVAE(noise='cpu'): CPU default generator -> pinned ring -> async H2D (models.py:92's RNG stream)"))


def pmc_traffic(kernel_key, B, dtype):
    """(HBM bytes per launch, source) of the dominant kernel from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, KB -> bytes).  NOT from this run: collected off-line with tests/collect_pmc.sh on a micro-benchmark of the same
    kernel at the same batch (bench.py cannot run the profiler on itself) -- T = 120, the timed pass's own length, where such a profile exists
    (the persistent kernels: a launch covers the whole pass and its traffic is reported per diagonal, like its time, so the fill / drain
    diagonals weigh exactly as they do in the timed run), else T = 16; (None, None) when no profile of this shape exists."""
    if dtype != "bf16":
        return None, None
    for name, T_ in ((f"r05_pmc_kernels_T120_B{B}.json", 120), (f"r04_pmc_kernels_T16_B{B}.json", 16), (f"r03_pmc_kernels_T16_B{B}.json", 16),
                     (f"r02_pmc_kernels_T16_B{B}.json", 16), (f"r01_v9_pmc_kernels_T16_B{B}.json", 16)):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        for k, v in pm.items():
            if kernel_key in k and "hbm_read_MB_corrected" in v and "hbm_write_MB" in v:
                return (int((v["hbm_read_MB_corrected"] + v["hbm_write_MB"]) * 1024 * 1024 / v.get("diagonals_per_launch", 1)),
                        f"profiles/{name}: off-line PMC passes over a T={T_} micro-benchmark of this kernel at B={B}"
                        + (f", one launch = {v['diagonals_per_launch']} diagonals, traffic per diagonal" if "diagonals_per_launch" in v else ", full-grid launches"))
    return None, None


def make_workload(model, B, dtype, dev, rank, sync, L=L_SEQ, C=VOCAB, noise="device"):
    if model == "molvae":
        return MolVaeWorkload(B, dtype, dev, rank, sync, L, C, noise=noise)
    if model == "moses":
        from bench_extra import MosesWorkload
        return MosesWorkload(B, dtype, dev, rank, sync)
    from bench_extra import Models2dWorkload
    return Models2dWorkload(B, dtype, dev, rank, sync)


def measure(model, B, dtype, steps, warmup, dev, rank, world, sync, label, L=L_SEQ, C=VOCAB, pipeline=False, noise="device"):
    """W untimed warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
    import torch
    import torch.distributed as dist
    from molecular_vae_amd import ops
    ops.PERSIST_STATS.update(launches=0, bwd_launches=0, rowres_pipe=0, failures=0, reruns=0)      # per measured configuration
    wl = make_workload(model, B, dtype, dev, rank, sync, L, C, noise=noise)
    log(f"[{label}] model built, per-GPU batch {B}, dtype {dtype}, world {world}")
    for i in range(warmup):
        loss = wl.step()
        if i == 0:
            torch.cuda.synchronize()
            log(f"[{label}] first step done, loss={float(loss):.5f}")
    torch.cuda.synchronize()
    # a full Python garbage collection NOW, outside the timed region: the previous configurations of this process leave a large heap
    # behind, and a generation-2 pass landing inside the 20 timed steps showed up as +3 ms per step on the b = 128 line (identical device
    # phase times, host stalled)
    import gc
    gc.collect()
    if world > 1:
        dist.barrier()
    ops.PROFILE = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = wl.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    # launches with bounded spins that gave up (ops.PERSIST_DEFAULT): the optimiser skipped those steps on the device -- the run stays valid but
    # the line says so, per rank (a rank that fell back to the wavefront schedules explains a slow SCALE point)
    ops.persist_check(sync=True)
    sched = dict(persist_fwd=ops.PERSIST_STATS["launches"], persist_bwd=ops.PERSIST_STATS["bwd_launches"], rowres_pipe=ops.PERSIST_STATS["rowres_pipe"],
                 failures=ops.PERSIST_STATS["failures"], disabled=bool(ops.PERSIST_STATS["disabled"]),
                 skipped_steps=int(float(getattr(wl, "optimizer", None).skipped_steps)) if hasattr(getattr(wl, "optimizer", None), "skipped_steps") else 0)
    rank_ms = [1e3 * dt / steps]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_ms = [1e3 * float(x) / steps for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        st = torch.tensor([sched["persist_fwd"], sched["persist_bwd"], sched["rowres_pipe"], sched["failures"], sched["skipped_steps"]], device=dev, dtype=torch.float64)
        alls = [torch.zeros_like(st) for _ in range(world)]
        dist.all_gather(alls, st)
        sched["per_rank"] = [dict(zip(("persist_fwd", "persist_bwd", "rowres_pipe", "failures", "skipped_steps"), [int(v) for v in x.tolist()])) for x in alls]
    ms = 1e3 * dt / steps
    log(f"[{label}] timed region done: {ms:.2f} ms/step")
    tag_ms = {k: sum(s.elapsed_time(e) for s, e in v) / steps for k, v in prof.items()}      # per STEP (a tag may be recorded several times in one)
    roof = wl.roofline(tag_ms, dtype)
    roof["phase_ms"] = {k: round(v, 3) for k, v in tag_ms.items()}
    comm = None
    if world > 1 or (sync is not None and sync.active):
        # time the main stream spent blocked on the gradient all-reduce in FusedAdam.step (HIP events around start_rest + wait): what the
        # overlap with backward did NOT hide.  MAX over ranks, like the step time.
        t = torch.tensor([tag_ms.get("dp_allreduce_exposed", 0.0)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        nst = max(1, steps + warmup)
        comm = dict(allreduce_exposed_ms=round(float(t), 3), ms_per_step_min_over_ranks=round(min(rank_ms), 3), ms_per_step_max_over_ranks=round(max(rank_ms), 3),
                    bytes_early_per_step=int(sync.stats["bytes_early"] / nst), bytes_rest_per_step=int(sync.stats["bytes_rest"] / nst), form=("reduce-scatter + sharded clip/Adam + all-gather" if getattr(sync, "shard_optimizer", False) else
                                                                  "all-reduce, early ranges from inside backward") + (", bf16 on the wire" if sync.compress else ""), world_size_reported=dist.get_world_size(), backend=dist.get_backend(),
                    gradient_bytes=4 * getattr(wl, "n_params", 0), early_ranges_per_step=(sync.stats["early_ranges"] / max(1, steps + warmup)) if sync else 0)
    roof["whole_step_tflops"] = round(wl.flops_per_step() * steps / dt / 1e12, 2)
    res = dict(value=round(B * world * steps / dt, 1), ms_per_step=round(ms, 3), steps=steps, warmup=warmup, dtype=dtype,
               final_loss=round(float(loss), 6), roofline=roof, config=wl.config(world), comm=comm, schedules=sched)
    if pipeline and hasattr(wl, "pipeline_steps"):
        pms = wl.pipeline_steps(max(5, steps // 2), 2)
        res["input_pipeline"] = dict(ms_per_step=round(pms, 3), value=round(B * world / (pms * 1e-3), 1), unit="molecules/s",
                                     overhead_vs_resident_batch=round(pms / ms - 1.0, 4),
                                     what="same step, batches drawn from a 250k-molecule uint8 corpus resident in HBM (DeviceDataset.batches: "
                                          "per-epoch shuffle + mvae_expand_indices to (idx, one-hot))")
        log(f"[{label}] through the input pipeline: {pms:.2f} ms/step")
    del wl
    ops.release_caches()
    torch.cuda.empty_cache()
    return res


def measure_generation(dev, b_size=2000, reps=5):
    """train_sample.py:29-45 / hugesample.py:94-95 ("Samples per second"): latents -> MolDecoder (forward-only pass, nothing saved for backward)
    -> arg-max -> strings through the charset, batches of 2000 as the reference draws them.  Random-init weights; samples/s incl. the host
    side string assembly, and the device part alone."""
    import torch
    import molecular_vae_amd as mv
    from molecular_vae_amd import ops
    torch.manual_seed(42)
    model = mv.MolecularVAE(i=L_SEQ, o=LATENT, c=VOCAB).to(dev).eval()
    charset = {i: ch for i, ch in enumerate(" #()+-123=@BCFHNOS[]clnors"[:VOCAB].ljust(VOCAB, "x"))}
    mv.generate_from_latent(model, charset, n=b_size, batch_size=b_size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        strings, _ = mv.generate_from_latent(model, charset, n=b_size, batch_size=b_size)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    z = torch.rand(b_size, LATENT, device=dev)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        model.decoder(z); s.record()
        for _ in range(reps):
            model.decoder(z).argmax(dim=2)
        e.record(); torch.cuda.synchronize()
    dev_ms = s.elapsed_time(e) / reps
    del model
    ops.release_caches(); torch.cuda.empty_cache()
    return dict(value=round(b_size / dt, 1), unit="samples/s", batch=b_size, device_only_samples_per_s=round(b_size / (dev_ms * 1e-3), 1),
                device_ms_per_batch=round(dev_ms, 3), what="MolDecoder from uniform latents, arg-max, charset strings (train_sample.py:29-45)")


def measure_moses_sample(dev, B=1024, max_len=100, reps=3):
    """mosesvae.VAE.sample (mosesvae.py:214-262; hugesample.py "Samples per second"): B latents from the prior -> autoregressive decoding at
    temperature 1 on the GRU step kernels + one sampling launch per token -> strings.  Random-init weights; samples/s incl. the host-side
    string assembly, and the device part alone (HIP events around the token loop's launches)."""
    import torch
    from molecular_vae_amd import mosesvae as MV, vocab as VC, ops
    v = VC.OneHotVocab([chr(ord("a") + i) for i in range(26)])
    torch.manual_seed(42)
    model = MV.VAE(v).to(dev).eval()
    model.sample(B, max_len=max_len, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        model.sample(B, max_len=max_len, seed=2 + r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for r in range(reps):
        model.sample(B, max_len=max_len, seed=9 + r, return_tokens=True)
    e.record(); torch.cuda.synchronize()
    dev_ms = s.elapsed_time(e) / reps
    del model
    ops.release_caches(); torch.cuda.empty_cache()
    return dict(value=round(B / dt, 1), unit="samples/s", batch=B, max_len=max_len, ms_per_batch=round(1e3 * dt, 2),
                us_per_token_step=round(1e3 * dev_ms / (max_len - 1), 2), launches_per_token=4,
                what="mosesvae.VAE.sample: 3-layer GRU wavefront pass (T = 1) + ONE sampling launch (head GEMV, temperature softmax, multinomial from a "
                     "counter hash, eos bookkeeping, next input rows) per token; strings assembled on the host")


def main():
    args = parse_args()
    if args.gpus < 1:
        die("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                           # never returns
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        die(f"--gpus {args.gpus} but WORLD_SIZE={world}: every GPU of the job must run one rank")
    bad = sorted(k for k in os.environ if k.startswith("MVAE_"))
    if bad:
        die(f"tuning variables set in the environment ({', '.join(bad)}): the bench measures the default build only")
    import torch
    if args.backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())      # rehearsal: ranks may share a GPU
    elif torch.cuda.device_count() <= local_rank:
        die(f"rank {rank}: no GPU {local_rank} on this box ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sync = None
    force = bool(args.force_comm) and world == 1
    if world > 1 or force:
        import torch.distributed as dist
        kw = {}
        if force:
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
            kw = dict(init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend, **kw)
        if dist.get_world_size() != args.gpus:
            die(f"{dist.get_world_size()} ranks joined, --gpus {args.gpus}")
        import molecular_vae_amd as mv
        sync = mv.GradSync(compress=args.grad_compress, force=force)
        sync.shard_optimizer = args.shard_optimizer
    if args.batch > 0:
        B, scaling = args.batch, "weak"
    else:
        if args.global_batch % world:
            die(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
        B, scaling = args.global_batch // world, "strong"

    main_res = measure(args.model, B, args.dtype, args.steps, args.warmup, dev, rank, world, sync, "main", args.seq_len, args.vocab,
                       pipeline=(world == 1 and not force and args.model == "molvae" and not args.no_secondary))
    metric = {"molvae": "molecules/s (ELBO fwd+bwd+step)", "moses": "molecules/s (mosesvae.VAE KL+CE fwd+bwd+step)",
              "models2d": "molecules/s (models2d.VAE ELBO fwd+bwd+step)"}[args.model]
    cfg = main_res["config"]; cfg["final_loss"] = main_res["final_loss"]
    out = dict(metric=metric, value=main_res["value"], unit="molecules/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=main_res["ms_per_step"], higher_is_better=True, scaling=scaling, vs_baseline=None, dtype=args.dtype,
               data="synthetic", config=cfg, roofline=main_res["roofline"])
    if main_res.get("comm") is not None:
        out["comm"] = main_res["comm"]
    out["schedules"] = main_res["schedules"]
    if main_res.get("input_pipeline") is not None:
        out["input_pipeline"] = main_res["input_pipeline"]

    if (world == 1 and not force and args.model == "molvae" and not args.no_secondary and args.batch == 0 and args.dtype == "bf16"
            and (args.seq_len, args.vocab) == (L_SEQ, VOCAB)):
        # every BASELINE.json config gets a number on this line: configs[1] (B=512), the per-rank shape of configs[2] (b=128), the exact-f32
        # parity mode, configs[3] (mosesvae.VAE, B=1024) and configs[4] (L=256, C=64, B=2048 with the GB/s of its HBM-bound kernels)
        sec = {}
        r = measure("molvae", B, "bf16", 10, 3, dev, rank, 1, None, "headline_B1024_cpu_eps", noise="cpu")
        sec["headline_B1024_cpu_eps"] = dict(value=r["value"], unit="molecules/s", ms_per_step=r["ms_per_step"], steps=10, warmup=3, dtype="bf16",
                                                 per_gpu_batch=B, eps_source=r["config"]["eps_source"])
        for label, mdl, b, dt_, st, wu, Lq, Cq in (("configs1_B512_bf16", "molvae", 512, "bf16", 10, 3, L_SEQ, VOCAB),
                                                   ("configs2_per_rank_b128_bf16", "molvae", 128, "bf16", 20, 5, L_SEQ, VOCAB),
                                                   ("parity_mode_B1024_f32", "molvae", args.global_batch, "f32", 10, 2, L_SEQ, VOCAB),
                                                   ("configs3_moses_B1024", "moses", 1024, "bf16", 20, 5, L_SEQ, VOCAB),
                                                   ("configs4_L256_C64_B2048", "molvae", 2048, "bf16", 5, 2, 256, 64)):
            r = measure(mdl, b, dt_, st, wu, dev, rank, 1, None, label, Lq, Cq)
            sec[label] = dict(value=r["value"], unit="molecules/s", ms_per_step=r["ms_per_step"], steps=st, warmup=wu, dtype=dt_,
                              per_gpu_batch=b, final_loss=r["final_loss"], roofline=r["roofline"], workload=r["config"]["workload"],
                              schedules=r["schedules"])
        sec["generation_from_latent_b2000"] = measure_generation(dev)
        sec["moses_sample_b1024"] = measure_moses_sample(dev)
        out["secondary"] = sec

    if rank == 0 and world == 1 and not force and not args.no_cpu_baseline:
        from oracle import torch_ref
        cores = host_cores()
        log(f"cpu baseline on {cores} threads ...")
        r = torch_ref.time_cpu_training(batch=32, steps=args.cpu_steps, warmup=2, threads=cores, log=log)
        out["cpu_baseline"] = dict(value=round(r["molecules_per_s"], 2), unit="molecules/s", cores=r["threads"], kind="port",
                                   cpu_model=cpu_model(),
                                   sample=f"{args.cpu_steps} train steps (after 2 warm-up) of the MolecularVAE step at batch 32 "
                                          f"(BASELINE.json configs[0]), torch.nn CPU modules, {r['s_per_step']:.2f} s/step")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
