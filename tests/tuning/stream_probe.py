#!/usr/bin/env python3
"""Does the b=128 step time depend on WHICH torch stream the decoder's side stream is?  (HIP maps streams onto a few hardware queues; a side
stream that shares the main stream's queue serialises with it.)   python tests/tuning/stream_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import molecular_vae_amd as mv
from molecular_vae_amd import ops
dev = torch.device("cuda")
def run(B, burn):
    keep = [torch.cuda.Stream(device=dev) for _ in range(burn)]        # advance torch's stream pool
    torch.manual_seed(42)
    model = mv.MolecularVAE().to(dev)
    opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
    loss_fn = mv.make_loss_function(120)
    g = torch.Generator().manual_seed(1234)
    data = torch.randint(0, 35, (B, 120), generator=g).to(dev)
    ohe = torch.nn.functional.one_hot(data, 35).float()
    model.encoder.lmbd.draw_eps = lambda b, o, d: 1e-2 * torch.randn(b, o, device=d)
    for _ in range(5):
        mv.train_step(model, opt, loss_fn, data, ohe)
    import gc
    if os.environ.get('PROBE_GC', '1') == '1': gc.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.PROFILE = {}
    for _ in range(20):
        mv.train_step(model, opt, loss_fn, data, ohe)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    prof, ops.PROFILE = ops.PROFILE, None
    print("   ", {k: round(sum(a.elapsed_time(b) for a, b in v) / 20, 3) for k, v in prof.items() if "hbm" not in k})
    side = model.decoder._side_stream(dev)
    ws = model.decoder._ws.bufs
    ptrs = {k[0]: f"{v.data_ptr():#x}" for k, v in ws.items() if k[0] in ("Whh0", "WhhT0", "hs0", "gates0", "dG0", "cs0")}
    print(f"B={B} streams burnt before={burn}: {ms:.3f} ms/step  side stream id {side.stream_id}  flat p {opt._flat[0]['p'].data_ptr():#x} {ptrs}", flush=True)
    del model, opt
    ops.release_caches(); torch.cuda.empty_cache()
for burn in (0, 0, 0, 0, 0):
    run(128, burn)
