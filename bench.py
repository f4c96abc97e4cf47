#!/usr/bin/env python3
"""bench.py -- molecules/s of one full ELBO training step (fwd + loss + bwd + clip + Adam [+ grad all-reduce]).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run with one
rank per GPU (RCCL).  Rank 0 prints ONE JSON line.  A "step" is train.py:95-104 on one synthetic minibatch of
BASELINE.json's shape: idx = randint(0, 35, (B, 120)), one-hot target, MolecularVAE(i=120, o=292, c=35) with the
reference's default init under manual_seed(42), bf16 decoder-LSTM storage / fp32 accumulate + fp32 master weights
(configs[1]).  Per-GPU batch is fixed (weak scaling); inputs are resident in HBM before the timed region.

Extra objects on the line:
  roofline     -- the dominant kernel family (decoder LSTM wavefront steps, MFMA bound): algorithmic FLOPs per launch /
                  average launch duration measured live with HIP events on the launching stream.
  cpu_baseline -- oracle/torch_ref.py (the reference architecture on stock torch.nn, CPU) timed on this host's cores
                  on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_SEQ, VOCAB, LATENT = 120, 35, 292
H_DEC, NL_DEC = 1024, 4
BF16_DENSE_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md, Peak BF16 MFMA dense
F32_MFMA_PEAK_TFLOPS = 157.3


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


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE.json configs[1]: 512)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only to rehearse the multi-rank "
                                                       "path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if args.backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())      # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    import molecular_vae_amd as mv
    from molecular_vae_amd import ops

    B = args.batch
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(42)                                     # train.py:73
    model = mv.MolecularVAE(i=L_SEQ, o=LATENT, c=VOCAB, dtype=dtype).to(dev)
    sync = mv.GradSync() if world > 1 else None
    optimizer = mv.FusedAdam(model.parameters(), lr=0.0008, max_grad_norm=3.0, grad_sync=sync)   # train.py:81,102
    loss_function = mv.make_loss_function(L_SEQ)
    g = torch.Generator().manual_seed(1234 + rank)
    data = torch.randint(0, VOCAB, (B, L_SEQ), generator=g).to(dev)
    ohe = torch.nn.functional.one_hot(data, VOCAB).float()    # the (idx, ohe) pair MoleLoader yields, resident in HBM
    model.encoder.lmbd.draw_eps = lambda b, o, d: 1e-2 * torch.randn(b, o, device=d)   # device noise: no H2D copy in the step

    def step():
        return mv.train_step(model, optimizer, loss_function, data, ohe)

    log(f"model built, B={B} dtype={args.dtype} world={world}")
    for i in range(args.warmup):
        loss = step()
        if i == 0:
            torch.cuda.synchronize()
            log(f"first step done, loss={float(loss):.5f}")
    torch.cuda.synchronize()
    log("warm-up done")
    if world > 1:
        dist.barrier()
    ops.PROFILE = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    log(f"timed region done: {1e3 * dt / args.steps:.2f} ms/step")
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = 1e3 * dt / args.steps
    value = B * world * args.steps / dt

    # roofline of the dominant kernel family, measured live (HIP events on the launching stream)
    fwd_f, bwd_f, n_launch = lstm_step_flops(B, L_SEQ)
    tag_ms = {k: sum(s.elapsed_time(e) for s, e in v) / len(v) for k, v in prof.items()}
    fwd_us = 1e3 * tag_ms.get("dec_lstm_fwd", float("nan")) / n_launch
    bwd_us = 1e3 * tag_ms.get("dec_lstm_bwd", float("nan")) / n_launch
    peak = BF16_DENSE_PEAK_TFLOPS if args.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
    # Dominant SINGLE kernel (profiles/r01_v9_kernel_stats.csv): the forward wavefront step.  The backward step is two kernels per launch
    # since the split-segment schedule (partial-tile GEMM + element-wise gate-derivative kernel), each smaller than the forward step;
    # its pair time is reported alongside.
    dom = "lstm_step_fwd_kernel"
    ach = fwd_f / (fwd_us * 1e-6) / 1e12
    ach_bwd = bwd_f / (bwd_us * 1e-6) / 1e12
    # HBM traffic of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KB -> bytes) on the same
    # kernel at the same shape, collected off-line with tests/bench_kernels.py and committed under profiles/ (bench.py cannot run
    # the profiler on itself); null when the shape differs from the profiled one
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_v9_pmc_kernels_T16_B512.json")))
        if B == 512 and args.dtype == "bf16":
            for k, v in pm.items():
                if dom.replace("_kernel", "") in k:
                    traffic = int((v["hbm_read_MB_corrected"] + v["hbm_write_MB"]) * 1024 * 1024)
    except Exception:
        traffic = None
    roofline = dict(bound="mfma", kernel=dom, achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                    traffic=traffic,
                    launches_per_pass=n_launch, avg_launch_us=dict(lstm_step_fwd=round(fwd_us, 2), lstm_step_bwd=round(bwd_us, 2)),
                    flops_per_launch=dict(lstm_step_fwd=fwd_f, lstm_step_bwd=bwd_f),
                    bwd_pair_tflops=round(ach_bwd, 2),
                    phase_ms={k: round(v, 3) for k, v in tag_ms.items()},
                    whole_step_tflops=round(flops_per_molecule() * B * args.steps / dt / 1e12 / world * world, 2))

    out = dict(metric="molecules/s (ELBO fwd+bwd+step)", value=round(value, 1), unit="molecules/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
               config=dict(workload="MolecularVAE(i=120,o=292,c=35) ELBO train step: fwd+loss+bwd+clip(3.0)+Adam(8e-4), "
                                    "synthetic one-hot SMILES [B,120,35]", per_gpu_batch=B, global_batch=B * world,
                           seq_len=L_SEQ, vocab=VOCAB, parallelism=f"dp{world}", final_loss=round(float(loss), 6)),
               roofline=roofline)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_ref
        cores = host_cores()
        log(f"cpu baseline on {cores} threads ...")
        r = torch_ref.time_cpu_training(batch=32, steps=args.cpu_steps, warmup=2, threads=cores, log=log)
        out["cpu_baseline"] = dict(value=round(r["molecules_per_s"], 2), unit="molecules/s", cores=r["threads"], kind="port",
                                   sample=f"{args.cpu_steps} train steps (after 2 warm-up) of the same model at batch 32 "
                                          f"(BASELINE.json configs[0]), torch.nn CPU modules, {r['s_per_step']:.2f} s/step")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
