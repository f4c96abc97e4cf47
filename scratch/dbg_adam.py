import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import molecular_vae_amd as mv
dev = torch.device("cuda")
torch.manual_seed(7); m1 = mv.MolecularVAE(i=24, o=16, c=12, dtype=torch.float32).to(dev)
torch.manual_seed(7); m2 = mv.MolecularVAE(i=24, o=16, c=12, dtype=torch.float32).to(dev)
fa = mv.FusedAdam(m1.parameters(), lr=8e-4, max_grad_norm=3.0)
loss_fn = mv.make_loss_function(24)
g = torch.Generator().manual_seed(1)
idx = torch.randint(0, 12, (6, 24), generator=g).to(dev)
ohe = torch.nn.functional.one_hot(idx, 12).float()
eps = (1e-2 * torch.randn(6, 16, generator=g)).to(dev)
for _ in range(2):
    mv.train_step(m1, fa, loss_fn, idx, ohe, eps=eps)
m2.load_state_dict(m1.state_dict())
ta = torch.optim.Adam(m2.parameters(), lr=1.0)
ta.load_state_dict(fa.state_dict())
for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
    assert torch.equal(a, b), k
m2.zero_grad(set_to_none=True)
recon, mu, lv = m2(idx, eps); loss_fn(recon, ohe, mu, lv).backward(); torch.cuda.synchronize()
g2 = {k: p.grad.clone() for k, p in m2.named_parameters()}
n2 = torch.nn.utils.clip_grad_norm_(m2.parameters(), 3.0)
pre = {k: p.detach().clone() for k, p in m1.named_parameters()}
mv.train_step(m1, fa, loss_fn, idx, ohe, eps=eps)
torch.cuda.synchronize()
print("norms", float(n2), float(fa.last_grad_norm))
off = 0
gflat = fa._flat[0]["g"]
for k, p in m1.named_parameters():
    gg = gflat[off:off + p.numel()].view(p.shape); off += p.numel()
    d = (gg - g2[k]).abs().max().item()
    print(f"{k:40s} grad maxdiff {d:.3e}  gmax {g2[k].abs().max().item():.3e}")
ta.step()
for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
    print(f"{k:40s} param diff/lr {((a - b).abs().max() / 8e-4).item():.4f}  moved/lr m1 {((a - pre[k]).abs().max() / 8e-4).item():.3f} m2 {((b - pre[k]).abs().max() / 8e-4).item():.3f}")
for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
    sa, sb = fa.state[a], ta.state[b]
    print(f"{k:40s} m diff {(sa['exp_avg'] - sb['exp_avg']).abs().max().item():.3e} v diff {(sa['exp_avg_sq'] - sb['exp_avg_sq']).abs().max().item():.3e} steps {float(sa['step'])} {float(sb['step'])}")
    break
