#!/usr/bin/env python3
"""Where does the reparameterisation noise cost time?  (VERDICT r04 missing #1 / weak #2: the product-default CPU draw ran 40.06 ms / step on the
driver box against 28.34 with device noise.)  Times the B = 1024 step with: the library draw (noise="device"), the CPU stream through the
pinned ring (noise="cpu"), and the round-4 form (a fresh torch.randn(pin_memory=True) per step); and the host pieces alone while the GPU is busy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import molecular_vae_amd as mv

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = 12


def run(noise, old=None, threads=None):
    torch.manual_seed(42)
    if threads:
        torch.set_num_threads(threads)
    model = mv.MolecularVAE(noise=noise).to(dev)
    lam = model.encoder.lmbd
    pbuf = torch.empty(B, 292, pin_memory=True); dbuf = torch.zeros(B, 292, device=dev)
    ring = [torch.empty(B, 292, pin_memory=True) for _ in range(8)]; cnt = [0]
    if old == "fresh":
        def draw(b, o, d):
            e = torch.randn(b, o, pin_memory=True); e.mul_(1e-2)
            return e.to(d, non_blocking=True)
        lam.draw_eps = draw
    elif old == "ring_copy":
        def draw(b, o, d):
            e = ring[cnt[0] % 8]; cnt[0] += 1
            torch.randn(b, o, out=e); e.mul_(1e-2)
            return e.to(d, non_blocking=True)
        lam.draw_eps = draw
    elif old == "copy_only":
        lam.draw_eps = lambda b, o, d: pbuf.to(d, non_blocking=True)
    elif old == "randn_only":
        def draw(b, o, d):
            torch.randn(b, o, out=pbuf); pbuf.mul_(1e-2)
            return dbuf
        lam.draw_eps = draw
    opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
    lf = mv.make_loss_function(120)
    data = torch.randint(0, 35, (B, 120), generator=torch.Generator().manual_seed(1)).to(dev)
    ohe = torch.nn.functional.one_hot(data, 35).float()
    for _ in range(4):
        mv.train_step(model, opt, lf, data, ohe)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        mv.train_step(model, opt, lf, data, ohe)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del model, opt
    from molecular_vae_amd import ops
    ops.release_caches(); torch.cuda.empty_cache()
    return 1e3 * dt, 1e3 * host / steps


print("torch threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)), flush=True)
for label, kw in (("device (library draw)", dict(noise="device")), ("cpu, pinned ring read in place (product)", dict(noise="cpu")),
                  ("cpu, fresh pinned tensor + copy per step (round 4)", dict(noise="cpu", old="fresh")),
                  ("cpu, pinned ring + copy command", dict(noise="cpu", old="ring_copy")),
                  ("copy command only (no randn)", dict(noise="cpu", old="copy_only")),
                  ("randn only (no copy)", dict(noise="cpu", old="randn_only")),
                  ("randn only, 1 torch thread", dict(noise="cpu", old="randn_only", threads=1)),
                  ("cpu product, 1 torch thread", dict(noise="cpu", threads=1)),
                  ("device again", dict(noise="device"))):
    ms, host = run(**kw)
    print(f"B={B} {label:48s} {ms:7.2f} ms/step   host enqueue {host:6.2f} ms/step", flush=True)

# the host pieces alone, with the GPU kept busy by a long kernel queue
big = torch.empty(1 << 28, device=dev)
for name, fn in (("torch.randn(B, 292)", lambda: torch.randn(B, 292)),
                 ("torch.randn(B, 292, pin_memory=True)", lambda: torch.randn(B, 292, pin_memory=True)),
                 ("torch.empty(B, 292, pin_memory=True)", lambda: torch.empty(B, 292, pin_memory=True)),
                 ("randn(out=pinned)", None)):
    buf = torch.empty(B, 292, pin_memory=True)
    if fn is None:
        fn = lambda: torch.randn(B, 292, out=buf)
    for _ in range(200):
        big.fill_(1.0)                # ~0.2 ms each: the GPU stays busy for the whole measurement
    keep = []
    t0 = time.perf_counter()
    for _ in range(20):
        t = fn()
        keep.append(t.to(dev, non_blocking=True))     # as the draw does: the pinned block stays in use until the copy ran
    dt = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    print(f"host piece {name:40s} {1e3 * dt:7.3f} ms per call (GPU busy)", flush=True)
