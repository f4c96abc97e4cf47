#!/usr/bin/env python3
"""train.py of the reference (train.py:15-19, 41-177), re-hosted on the MI355X path.  Same flags (-b, -o, -l), same loop body,
same checkpoint dictionary; the data set is a .smi/CSV file (or synthetic SMILES when none is given -- the reference's hard-coded
/vol/ml/... path does not exist anywhere else).  One process per GPU under torch.distributed.run for data parallelism.

    python examples/train_zinc.py --smi data/250k_rndm_zinc_drugs_clean.smi -b 512 -l 292 --epochs 1
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molecular_vae_amd as mv                     # noqa: E402
from molecular_vae_amd import data as D            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-b", "--batch_size", default=128, type=int)
ap.add_argument("-o", "--optimizer", default="adam", type=str)       # label only, as in the reference (always Adam)
ap.add_argument("-l", "--latent_size", default=292, type=int)
ap.add_argument("--smi", default=None)
ap.add_argument("--max_len", default=120, type=int)
ap.add_argument("--epochs", default=1, type=int)
ap.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)    # moses_train_distrib.py:27
ap.add_argument("--n_synth", default=4096, type=int, help="size of the synthetic corpus when no --smi is given")
ap.add_argument("--out_dir", default=".", help="where the checkpoint goes")
ap.add_argument("--report", default=None, help="write a JSON summary (per-epoch losses, molecules/s through the input pipeline) here")
args = ap.parse_args()

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(args.local_rank)
dev = torch.device("cuda", args.local_rank)
if world > 1:
    torch.distributed.init_process_group("nccl", device_id=dev)

if args.smi:
    smiles = [s for s in D.load_smiles(args.smi) if len(s) < args.max_len]
else:                                               # synthetic corpus of the ZINC alphabet (motif strings: learnable in one epoch)
    smiles = D.synthetic_smiles(args.n_synth, seed=0)
max_len = args.max_len
vocab = D.build_vocab(smiles, max_len)              # train.py:45-60, sorted
charset = {i: c for c, i in vocab.items()}
enc = D.encode_smiles(smiles, vocab, max_len)
msk = np.random.RandomState(1).rand(len(enc)) < 0.8                       # train.py:62-64
train_ds, test_ds = D.DeviceDataset(enc[msk], len(vocab), dev), D.DeviceDataset(enc[~msk], len(vocab), dev)

torch.manual_seed(42)                               # train.py:73
model = mv.MolecularVAE(i=max_len, c=len(vocab), o=args.latent_size).to(dev)
sync = mv.GradSync() if world > 1 else None
optimizer = mv.FusedAdam(model.parameters(), lr=0.0008, max_grad_norm=3.0, grad_sync=sync)         # train.py:81,102
scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, "min", factor=0.9, patience=10, threshold=1e-3, cooldown=5)
loss_function = mv.make_loss_function(max_len)

report = dict(batch_size=args.batch_size, n_train=len(train_ds), n_test=len(test_ds), vocab=len(vocab), epochs=[])
ckpt_path = os.path.join(args.out_dir, f"save_{args.batch_size}_{args.optimizer}_{args.latent_size}.pt")
for epoch in range(1, args.epochs + 1):
    model.train()
    total, n, nb = torch.zeros((), device=dev), 0, 0
    first = None
    t_beg, t_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_beg.record()
    for batch_idx, (data, ohe) in enumerate(train_ds.batches(args.batch_size, epoch=epoch, seed=0, rank=rank, world=world)):
        loss = mv.train_step(model, optimizer, loss_function, data, ohe)           # train.py:95-104
        total += loss; n += data.shape[0]; nb += 1
        if first is None:
            first = loss
        if batch_idx % 100 == 0 and rank == 0:
            with torch.no_grad():
                recon, _, _ = model(data)
                acc = mv.exact_match_accuracy(recon, data)                          # train.py:109-113
            print(f"train: {epoch} / {batch_idx}\t{float(loss):.4f}\tacc {float(acc):.3f}", flush=True)
    t_end.record(); torch.cuda.synchronize()
    epoch_ms = t_beg.elapsed_time(t_end)
    # test(epoch), train.py:120-153: forward-only (no saved state is written under no_grad), every sequence of the shard (drop_last=False)
    val, acc = mv.evaluate(model, loss_function, test_ds.batches(args.batch_size, shuffle=False, rank=rank, world=world, drop_last=False))
    if world > 1:                                        # every rank must take the same ReduceLROnPlateau decision
        t = torch.tensor([val, acc], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t); val, acc = (t / world).tolist()
    scheduler.step(val)                                  # train.py:165
    if rank == 0:
        print(f"epoch {epoch}: train {float(total) / max(n, 1):.5f}  val {val:.5f}  acc {acc:.4f}  lr {optimizer.param_groups[0]['lr']:.2e}", flush=True)
        mv.save_checkpoint(ckpt_path, model, optimizer, epoch, charset, max_len, latent_size=args.latent_size)      # train.py:170-177
        report["epochs"].append(dict(epoch=epoch, first_batch_loss=float(first), mean_batch_loss=float(total) / max(nb, 1), val_loss=val, val_acc=acc,
                                     molecules_per_s_through_pipeline=n * world / (epoch_ms * 1e-3)))
if rank == 0 and args.report:
    import json
    report["checkpoint"] = ckpt_path
    json.dump(report, open(args.report, "w"))
if world > 1:
    torch.distributed.destroy_process_group()
