"""world_size=2 CPU (gloo) tests of the data-parallel path: bucketed flat-gradient all-reduce + 1/world scaling, shard
helpers, and the identity the DP design rests on: mean over a global batch == average of per-shard means (oracle)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import np_oracle as O
from oracle import initparams as ip

G1 = dict(i=24, o=16, c=12, emb=30, h_enc=56, n_enc=2, h_dec=32, n_dec=2, seed=101, gain=2.0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import molecular_vae_amd as mv
    # 1) GradSync: many small buckets, uneven tail
    sync = mv.GradSync(bucket_bytes=4 * 1000)
    flat = torch.arange(10007, dtype=torch.float32) * (rank + 1)
    sync.start(flat); sync.wait()
    expect = torch.arange(10007, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(flat, expect)
    assert sync.grad_scale() == 1.0 / world and sync.world == world
    # 1b) early range (started from inside backward) + the complement in step(): every element reduced exactly once
    flat = torch.arange(10007, dtype=torch.float32) * (rank + 1)
    sync.start_early(flat, 3000, 9001)
    sync.start_early(flat, 100, 200)
    sync.start_rest(flat); sync.wait()
    assert torch.equal(flat, expect) and sync.early == [] and sync.handles == []
    # 1c) bf16 on the wire: values that are exact in bfloat16 survive exactly; others are rounded once per rank
    sync16 = mv.GradSync(bucket_bytes=4 * 1000, compress="bf16")
    flat = torch.arange(2048, dtype=torch.float32).remainder(64) * (rank + 1)
    sync16.start(flat); sync16.wait()
    assert torch.equal(flat, torch.arange(2048, dtype=torch.float32).remainder(64) * sum(r + 1 for r in range(world))) and sync16._staged == []
    # 1d) the sharded form's collectives: reduce-scatter leaves this rank's reduced slice in place, all-gather rebuilds the whole buffer
    S = 4096
    flat = torch.arange(world * S, dtype=torch.float32) * (rank + 1)
    mine = sync.reduce_scatter(flat, S)
    want = torch.arange(world * S, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert mine.data_ptr() == flat[rank * S:].data_ptr() and torch.equal(mine, want[rank * S:(rank + 1) * S])
    mine.mul_(0.5)                                            # "the update" on this rank's slice
    sync.all_gather(flat, S)
    assert torch.equal(flat, 0.5 * want)
    # 2) per-shard oracle gradients, all-reduced and scaled, equal the global-batch gradients
    shapes = ip.molvae_shapes(G1["i"], G1["o"], G1["c"], G1["emb"], G1["h_enc"], G1["n_enc"], G1["h_dec"], G1["n_dec"])
    p = ip.init_params(shapes, G1["seed"], G1["gain"], np.float64)
    B = 4
    idx = ip.seeded_indices(7, B, G1["i"], G1["c"]); eps = ip.seeded_eps(7, B, G1["o"], dtype=np.float64)
    lo, hi = mv.shard_batch(B, rank, world)
    r = O.molvae_loss_and_grads(p, idx[lo:hi], eps[lo:hi], max_len=G1["i"], num_lstm=G1["n_enc"], num_gru=G1["n_dec"])
    names = sorted(r["grads"])
    flat = torch.from_numpy(np.concatenate([r["grads"][k].reshape(-1) for k in names]))
    sync = mv.GradSync(bucket_bytes=1 << 16)
    sync.start(flat); sync.wait()
    flat *= sync.grad_scale()
    loss = torch.tensor([r["loss"]]); dist.all_reduce(loss); loss /= world
    if rank == 0:
        full = O.molvae_loss_and_grads(p, idx, eps, max_len=G1["i"], num_lstm=G1["n_enc"], num_gru=G1["n_dec"])
        ref = np.concatenate([full["grads"][k].reshape(-1) for k in names])
        err = float(np.abs(flat.numpy() - ref).max() / np.abs(ref).max())
        np.save(os.path.join(out_dir, "result.npy"), np.array([err, abs(float(loss) - full["loss"]) / abs(full["loss"])]))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    err, loss_err = np.load(os.path.join(str(tmp_path), "result.npy"))
    assert err < 1e-10 and loss_err < 1e-12


def test_single_process_gradsync_is_identity():
    import molecular_vae_amd as mv
    s = mv.GradSync()
    x = torch.ones(10)
    s.start(x); s.wait()
    assert torch.equal(x, torch.ones(10)) and s.grad_scale() == 1.0
