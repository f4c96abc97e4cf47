#!/usr/bin/env python3
"""Data-parallel equivalence rehearsal on ONE GPU box: K training steps of the full model
  (a) single process, global batch 2b;   (b) two ranks (gloo, sharing cuda:0), b each, GradSync + early all-reduce
must give the same losses and parameters (up to fp32 re-association).  Usage on the GPU box:
  python tests/dp_equiv.py --out gpurun_out/dp1.json
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tests/dp_equiv.py --out gpurun_out/dp2.json
  python tests/dp_equiv.py --compare gpurun_out/dp1.json gpurun_out/dp2.json"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--out")
ap.add_argument("--compare", nargs=2)
ap.add_argument("--b", type=int, default=64)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="f32: re-association noise only (sharp check); bf16: the bench configuration")
ap.add_argument("--model", default="molvae", choices=["molvae", "moses"])
ap.add_argument("--shard", action="store_true", help="world > 1: reduce-scatter + sharded clip / Adam + all-gather (FusedAdam(shard_optimizer=True))")
ap.add_argument("--compress", default=None, choices=[None, "bf16"], help="world > 1: gradient all-reduce as bfloat16")
ap.add_argument("--backend", default="gloo", help="gloo (ranks sharing the GPU) or nccl (= RCCL; one rank per GPU)")
ap.add_argument("--poison-rank", type=int, default=-1, help="this rank's persistent launches give up (MVAE_PERSIST_SPIN=1) at --poison-step: every rank must skip that update")
ap.add_argument("--poison-step", type=int, default=1)
ap.add_argument("--force-comm", action="store_true", help="ONE rank: initialise a one-rank process group on --backend and issue every collective anyway (GradSync(force=True))")
args = ap.parse_args()
if args.compare:
    a, b = (json.load(open(f)) for f in args.compare)
    ok = True
    for k in ("loss", "psum", "gnorm"):
        for x, y in zip(a[k], b[k]):
            rel = abs(x - y) / (abs(x) + 1e-30)
            print(f"{k}: {x:.8f} vs {y:.8f}  rel {rel:.2e}")
            ok = ok and rel < 2e-6
    print("EQUIVALENT" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)

import molecular_vae_amd as mv
rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
force = args.force_comm and world == 1
if world > 1 or force:
    import torch.distributed as dist
    if force:
        import socket
        so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
        kw = dict(device_id=dev) if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
    else:
        dist.init_process_group(args.backend)
if args.model == "moses":
    # mosesvae.VAE step (moses_train_distrib.py:287-299): variable-length batch, the CE mean runs over the GLOBAL non-pad token count
    import numpy as np
    from molecular_vae_amd import mosesvae as MV, vocab as VC
    v = VC.OneHotVocab([chr(ord("a") + i) for i in range(26)])
    torch.manual_seed(42)
    model = MV.VAE(v, dtype=torch.float32 if args.dtype == "f32" else torch.bfloat16).to(dev)
    model.d_dropout = 0.0                     # the dropout hash is indexed by batch position, which differs between the two layouts
    sync = mv.GradSync(force=force) if (world > 1 or force) else None
    opt = mv.FusedAdam(model.parameters(), lr=3e-4, max_grad_norm=50.0, grad_sync=sync)
    rs = np.random.RandomState(9)
    gb = 2 * args.b
    lens = rs.randint(6, 40, size=gb)
    strs = [torch.tensor([v.bos] + rs.randint(0, 26, size=n).tolist() + [v.eos]) for n in lens]
    eps_all = torch.from_numpy(rs.standard_normal((args.steps, gb, 160)).astype("float32"))
    mine = list(range(rank, gb, world))       # interleaved shards: different token counts per rank
    order = sorted(mine, key=lambda i: -len(strs[i]))
    out = dict(loss=[], psum=[], gnorm=[], world=world)
    for s_ in range(args.steps):
        loss, kl, rec = mv.moses_train_step(model, opt, 0.5, [strs[i] for i in order], eps=eps_all[s_, order].to(dev))
        lt = loss.detach().clone()
        if world > 1:
            dist.all_reduce(lt); lt /= world
        out["loss"].append(float(lt)); out["gnorm"].append(float(opt.last_grad_norm))
        out["psum"].append(float(sum(p.detach().abs().sum() for p in model.parameters())))
    out["early_ranges"] = 0
    out["buckets"] = sync.stats["buckets"] if sync is not None else 0
    out["pcheck"] = [float(p.detach().double().sum()) for p in model.parameters()]
    if rank == 0 and args.out:
        json.dump(out, open(args.out, "w")); print(out)
    if world > 1 or force:
        dist.destroy_process_group()
    sys.exit(0)
L_SEQ, VOCAB, LATENT = 120, 35, 292
torch.manual_seed(42)
model = mv.MolecularVAE(i=L_SEQ, o=LATENT, c=VOCAB, dtype=torch.float32 if args.dtype == "f32" else torch.bfloat16).to(dev)
sync = mv.GradSync(compress=args.compress, force=force) if (world > 1 or force) else None
opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0, grad_sync=sync, shard_optimizer=args.shard)
loss_fn = mv.make_loss_function(L_SEQ)
g = torch.Generator().manual_seed(7)
gb = 2 * args.b
data_all = torch.randint(0, VOCAB, (gb, L_SEQ), generator=g)
eps_all = 1e-2 * torch.randn(args.steps, gb, LATENT, generator=g)
per = gb // world
sl = slice(rank * per, (rank + 1) * per)
data = data_all[sl].to(dev)
ohe = torch.nn.functional.one_hot(data, VOCAB).float()
out = dict(loss=[], psum=[], gnorm=[], world=world)
for s in range(args.steps):
    eps = eps_all[s, sl].to(dev)
    if args.poison_rank == rank:
        if s == args.poison_step:
            os.environ["MVAE_TUNING"] = "1"; os.environ["MVAE_PERSIST_SPIN"] = "1"
        else:
            os.environ.pop("MVAE_PERSIST_SPIN", None)
    opt.zero_grad(set_to_none=True)
    recon, mu, lv = model(data, eps=eps)
    loss = loss_fn(recon, ohe, mu, lv)
    loss.backward()
    opt.step()
    lt = loss.detach().clone()
    if world > 1:
        dist.all_reduce(lt); lt /= world
    out["loss"].append(float(lt))
    out["gnorm"].append(float(opt.last_grad_norm))
    out["psum"].append(float(sum(p.detach().abs().sum() for p in model.parameters())))
out["early_ranges"] = sync.stats["early_ranges"] if sync is not None else 0
out["buckets"] = sync.stats["buckets"] if sync is not None else 0
if args.poison_rank >= 0:
    import warnings
    from molecular_vae_amd import ops as _ops
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _ops.persist_check(sync=True)
    mine = dict(psum=out["psum"], skipped=float(opt.skipped_steps), failures=_ops.PERSIST_STATS["failures"], persist=_ops.PERSIST_STATS["launches"] + _ops.PERSIST_STATS["bwd_launches"])
    allr = [None] * world
    if world > 1:
        dist.all_gather_object(allr, mine)
    else:
        allr = [mine]
    out["ranks"] = allr
out["pcheck"] = [float(p.detach().double().sum()) for p in list(model.parameters())[:6]] + [float(opt.state[next(iter(model.decoder.gru.parameters()))]["exp_avg"].double().abs().sum())]
if args.shard and (world > 1 or force):
    # a state_dict() taken now would hold stale moments for the other ranks' slices -- it must refuse until gather_state() ran
    try:
        opt.state_dict(); out["stale_state_dict_refused"] = False
    except mv._lib.MvaeError:
        out["stale_state_dict_refused"] = True
    opt.gather_state()
    out["mcheck"] = float(sum(opt.state[p]["exp_avg"].double().abs().sum() for p in model.parameters()))
    # ... and afterwards the dictionary loads into an UNSHARDED optimiser with every moment in place
    sd = opt.state_dict()
    m2 = mv.MolecularVAE(i=L_SEQ, o=LATENT, c=VOCAB, dtype=model.decoder.compute_dtype).to(dev)
    o2 = mv.FusedAdam(m2.parameters(), lr=8e-4, max_grad_norm=3.0)
    o2.load_state_dict(sd)
    out["mcheck_reloaded"] = float(sum(o2.state[p]["exp_avg"].double().abs().sum() for p in m2.parameters()))
    out["vcheck"] = float(sum(opt.state[p]["exp_avg_sq"].double().sum() for p in model.parameters()))
    out["vcheck_reloaded"] = float(sum(o2.state[p]["exp_avg_sq"].double().sum() for p in m2.parameters()))
elif world >= 1:
    out["mcheck"] = float(sum(opt.state[p]["exp_avg"].double().abs().sum() for p in model.parameters()))
if rank == 0 and args.out:
    json.dump(out, open(args.out, "w"))
    print(out)
if world > 1 or force:
    dist.destroy_process_group()
