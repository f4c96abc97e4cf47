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


def _worker8(rank, world, port, out_dir):
    """The bookkeeping of the N-rank step at the world size the job will run at (8): no GPU, no kernels -- ranges, shard boundaries, the
    complement of the early ranges, the poison slot's way through the collectives."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import molecular_vae_amd as mv
    tot = sum(r + 1 for r in range(world))
    # (a) a MolecularVAE-shaped parameter list (reduced widths, same parameter ORDER and count: 43 tensors) under FusedAdam, both forms
    enc = mv.MolEncoder(i=24, o=16, c=12, h_size=56, num_lstm=3)
    dec = mv.MolDecoder(i=16, o=24, c=12, num_gru=4, h_size=64, dtype=torch.float32)
    params = list(enc.parameters()) + list(dec.parameters())
    n = sum(p.numel() for p in params)
    sync = mv.GradSync(bucket_bytes=4 * 5000)
    opt = mv.FusedAdam(params, lr=1e-3, max_grad_norm=3.0, grad_sync=sync)
    f = opt._flat[0]
    assert f["g"].numel() == n + 1 and f["poison"].data_ptr() == f["g"][n:].data_ptr() and f["partial"].numel() == (n + 1 + 65535) >> 16
    # the decoder backward's four per-layer early ranges (reverse layer order, layer 3 with the head), then step()'s complement
    names = [k for k, _ in enc.named_parameters()] + ["D." + k for k, _ in dec.named_parameters()]
    offs, off = {}, 0
    for k, p_ in zip(names, params):
        offs[k] = off; off += p_.numel()
    g = f["g"]
    g.copy_(torch.arange(n + 1, dtype=torch.float32).remainder(1000) * (rank + 1))
    first = lambda l: "D.gru.weight_ih_l%d" % l if l >= 1 else "D.gru.weight_hh_l0"
    hi = offs["D.decoded_mean.module.0.bias"] + dec.decoded_mean.module[0].bias.numel()
    assert hi == n
    for l in (3, 2, 1, 0):
        lo = offs[first(l)]
        sync.start_early(g, lo, hi); hi = lo
    if rank == 5:
        f["poison"].fill_(float("nan"))                     # a persistent launch on ONE rank gave up
    sync.start_rest(g); sync.wait()
    want = torch.arange(n + 1, dtype=torch.float32).remainder(1000) * tot
    assert torch.equal(g[:n], want[:n]) and bool(torch.isnan(g[n])), "every element reduced exactly once; the poison reaches every rank"
    assert sync.stats["early_ranges"] == 4 and sync.stats["bytes_early"] + sync.stats["bytes_rest"] == 4 * (n + 1)
    assert sync.stats["bytes_early"] == 4 * (n - offs[first(0)])
    for p_ in params:
        mv._lib.clear_grad_sink(p_)
    # (b) the sharded form: equal slices on 64K-element boundaries, the poison slot inside the allocation, partial-sum slices per rank
    sync2 = mv.GradSync()
    enc2 = mv.MolEncoder(i=24, o=16, c=12, h_size=56, num_lstm=3)
    dec2 = mv.MolDecoder(i=16, o=24, c=12, num_gru=4, h_size=64, dtype=torch.float32)
    p2 = list(enc2.parameters()) + list(dec2.parameters())
    opt2 = mv.FusedAdam(p2, lr=1e-3, grad_sync=sync2, shard_optimizer=True)
    f2 = opt2._flat[0]
    S = f2["shard_elems"]
    assert opt2.shard and S % 65536 == 0 and world * S >= n + 1 and (world * S - (n + 1)) < world * 65536
    assert f2["g"].numel() == world * S == f2["p"].numel() and f2["partial"].numel() == world * (S >> 16)
    assert sync2.allow_early is False                         # no in-backward ranges in this form
    f2["g"].copy_(torch.arange(world * S, dtype=torch.float32).remainder(977) * (rank + 1))
    if rank == 2:
        f2["poison"].fill_(float("inf"))
    mine = sync2.reduce_scatter(f2["g"], S)
    w2 = torch.arange(world * S, dtype=torch.float32).remainder(977) * tot
    owner = n // S                                            # the rank whose slice holds the poison slot
    ok = torch.equal(mine[:n - owner * S] if rank == owner else mine, (w2[rank * S:(rank + 1) * S])[:n - owner * S] if rank == owner else w2[rank * S:(rank + 1) * S])
    assert ok
    if rank == owner:
        assert not bool(torch.isfinite(mine[n - owner * S]))
    # the norm: this rank's partial-sum slice, others zero, all-reduced -> the same (non-finite) total everywhere
    part = torch.zeros(world * (S >> 16)); cps = S >> 16
    part[rank * cps:(rank + 1) * cps] = (mine.double() ** 2).view(cps, 65536).sum(1).float()
    dist.all_reduce(part)
    assert not bool(torch.isfinite(part.sum())), "every rank sees the poisoned norm and skips the update"
    # (c) the reparameterisation noise streams of the ranks: same torch.manual_seed everywhere, different device-noise seeds (the rank is folded in)
    torch.manual_seed(42)
    from molecular_vae_amd import ops
    seed = ops.NoiseStream().take(1)[0]
    seeds = [None] * world
    dist.all_gather_object(seeds, seed)
    assert len(set(seeds)) == world, seeds
    if rank == 0:
        np.save(os.path.join(out_dir, "ok8.npy"), np.array([1.0]))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_eight_ranks_gloo_bookkeeping(tmp_path):
    """World size 8 (the job's size): early ranges + complement, byte counters, sharded slices on 64K-element boundaries
    and the poison slot's propagation, over gloo on CPU tensors."""
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok8.npy"))


def test_single_process_gradsync_is_identity():
    import molecular_vae_amd as mv
    s = mv.GradSync()
    x = torch.ones(10)
    s.start(x); s.wait()
    assert torch.equal(x, torch.ones(10)) and s.grad_scale() == 1.0
