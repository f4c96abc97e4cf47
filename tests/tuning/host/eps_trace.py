#!/usr/bin/env python3
"""A few B = 1024 steps with the CPU noise source, for a rocprofv3 --kernel-trace --memory-copy-trace timeline (where is the gap?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import molecular_vae_amd as mv
dev = torch.device("cuda", 0)
noise = sys.argv[1] if len(sys.argv) > 1 else "cpu"
torch.manual_seed(42)
model = mv.MolecularVAE(noise=noise).to(dev)
opt = mv.FusedAdam(model.parameters(), lr=8e-4, max_grad_norm=3.0)
lf = mv.make_loss_function(120)
data = torch.randint(0, 35, (1024, 120), generator=torch.Generator().manual_seed(1)).to(dev)
ohe = torch.nn.functional.one_hot(data, 35).float()
for _ in range(6):
    mv.train_step(model, opt, lf, data, ohe)
torch.cuda.synchronize()
