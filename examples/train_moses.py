#!/usr/bin/env python3
"""The MOSES character-VAE trainer (moses_train_distrib.py:160-356 surface) on the MI355X path: char vocabulary from the corpus, sorted
collate, KL annealing, cosine LR with restarts, clip 50 + Adam(3e-4), rank-0 checkpoint + vocabulary pickle + samples per epoch.  One
process per GPU under torch.distributed.run (the reference's DistributedDataParallel / DistributedSampler lines are commented out there;
here the gradient all-reduce is FusedAdam's GradSync and the data shard is ShardedSampler).

    python examples/train_moses.py --smi data/moses_train.smi --epochs 2 -b 128
"""
import argparse
import os
import pickle
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molecular_vae_amd as mv                          # noqa: E402
from molecular_vae_amd import data as D, mosesvae as MV, vocab as VC   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)     # moses_train_distrib.py:27
ap.add_argument("--smi", default=None)
ap.add_argument("-b", "--batch_size", default=128, type=int)                                # moses_train_distrib.py:177
ap.add_argument("--epochs", default=100, type=int)
ap.add_argument("--n_samples", default=1024, type=int)
ap.add_argument("--n_synth", default=8192, type=int, help="size of the synthetic corpus when no --smi is given")
ap.add_argument("--out_dir", default=".")
ap.add_argument("--report", default=None, help="write a JSON summary (the per-epoch postfix dictionaries + samples) here")
args = ap.parse_args()

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(args.local_rank)
dev = torch.device("cuda", args.local_rank)
if world > 1:
    torch.distributed.init_process_group("nccl", device_id=dev)

if args.smi:
    smiles = D.load_smiles(args.smi)
else:                                                    # synthetic MOSES-like corpus (no data ships with the reference)
    rs = np.random.RandomState(0)
    lens = np.clip(rs.normal(38, 8, size=args.n_synth), 10, 57).astype(int)
    smiles = [D.synthetic_smiles(1, seed=1000 + i, lo=int(n), hi=int(n) + 1)[0] for i, n in enumerate(lens)]
vocab = VC.OneHotVocab.from_data(smiles)                 # moses_train_distrib.py:166
collate = VC.get_padded_collate_fn(vocab, pin_memory=True)
sampler = mv.ShardedSampler(len(smiles), rank=rank, world=world, seed=0)
loader = torch.utils.data.DataLoader(smiles, batch_size=args.batch_size, sampler=sampler, collate_fn=collate, drop_last=True)

torch.manual_seed(42)
model = MV.VAE(vocab).to(dev)
sync = mv.GradSync() if world > 1 else None
optimizer = mv.FusedAdam(model.parameters(), lr=3e-4, max_grad_norm=50.0, grad_sync=sync)   # :188, :227
kl_annealer = mv.KLAnnealer(args.epochs)                                                     # :47-58, :185
lr_annealer = mv.CosineAnnealingLRWithRestart(optimizer)                                     # :61-89, :195

report = dict(vocab=len(vocab), n=len(smiles), epochs=[], samples=[])
for epoch in range(args.epochs):
    sampler.set_epoch(epoch)
    kl_weight = kl_annealer(epoch)
    post = mv.moses_train_epoch(model, epoch, (b.to(dev) for b in loader), kl_weight, optimizer, log_every=100 if rank == 0 else 0)
    if rank == 0:                                        # :342-353
        print(post, flush=True)
        report["epochs"].append(post)
        torch.save(model.state_dict(), os.path.join(args.out_dir, "trained_save.pt"))
        with open(os.path.join(args.out_dir, "vocab.pkl"), "wb") as f:
            pickle.dump(vocab, f)
        model.eval()
        res, _ = model.sample(min(args.n_samples, 64), max_len=60)
        model.train()
        for s in res[:5]:
            print("sample:", s)
        report["samples"] = res[:16]
    lr_annealer.step()                                   # :355
if rank == 0 and args.report:
    import json
    json.dump(report, open(args.report, "w"))
if world > 1:
    torch.distributed.destroy_process_group()
