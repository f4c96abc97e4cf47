"""bench.py workloads beside the headline one: the mosesvae.VAE path (BASELINE.json configs[3]) and the models2d.VAE variant."""
import numpy as np
import torch

BF16_DENSE_PEAK_TFLOPS = 2500.0
F32_MFMA_PEAK_TFLOPS = 157.3


class MosesWorkload:
    """configs[3] / SURVEY section 8d: V = 30 (26 characters + 4 specials), lengths ~ N(38, 8) clipped to [10, 57] + <bos>/<eos>, batch
    sorted by length descending; step = moses_train_distrib.py:287-299 in train mode (decoder dropout 0.2 drawn on device): kl_weight *
    KL + CE, backward, clip 50, Adam(3e-4)."""
    name = ("mosesvae.VAE(V=30) train step: GRU enc(256) + 3xGRU dec(512) fwd + KL + token CE + bwd + clip(50) + Adam(3e-4), synthetic "
            "MOSES-like SMILES, lengths ~N(38,8) in [10,57] + 2 specials, dropout 0.2 (train mode)")

    def __init__(self, B, dtype, dev, rank, sync):
        import molecular_vae_amd as mv
        from molecular_vae_amd import mosesvae as MV, vocab as VC
        self.mv, self.B = mv, B
        v = VC.OneHotVocab([chr(ord("a") + i) for i in range(26)])
        self.V = len(v)
        torch.manual_seed(42)
        self.model = MV.VAE(v, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32).to(dev).train()
        self.optimizer = mv.FusedAdam(self.model.parameters(), lr=3e-4, max_grad_norm=50.0, grad_sync=sync, shard_optimizer=bool(getattr(sync, "shard_optimizer", False)))
        rs = np.random.RandomState(1234 + rank)
        lens = np.clip(np.rint(rs.normal(38, 8, size=B)), 10, 57).astype(int)
        lens = np.sort(lens)[::-1]
        seqs = [torch.tensor([v.bos] + rs.randint(0, 26, size=n).tolist() + [v.eos]) for n in lens]
        self.batch = VC.pad_batch(seqs, v.pad).to(dev)                  # resident in HBM before the timed region
        self.T = int(self.batch.x_pad.shape[1])
        self.tokens = int(sum(lens) + 2 * B)
        self.kl_weight = 0.5

    def step(self):
        loss, _, _ = self.mv.moses_train_step(self.model, self.optimizer, self.kl_weight, self.batch)
        return loss

    def _macs_per_mol(self, T):
        V = self.V
        return (T * 3 * 256 * (V + 256) + 2 * (256 * 256 + 256 * 160) + 160 * 512 + T * (3 * 512 * (V + 160 + 512) + 2 * 3 * 512 * 1024) + T * 512 * V)

    def flops_per_step(self):
        return 3 * 2 * self._macs_per_mol(self.T) * self.B     # padded length (what the kernels compute over; SURVEY 8d formula)

    def roofline(self, tag_ms, dtype):
        # dominant kernel: the decoder GRU wavefront step (3 x 512): algorithmic FLOPs per launch = 2 B 3H (H [x, l >= 1] + H [h]) per cell
        H, NL, T, B = 512, 3, self.T, self.B
        per_t = 2 * B * 3 * H * (H * NL + H * (NL - 1))          # layer 0's input projection is a table gather + a hoisted z part
        n_launch = T + NL - 1
        f = per_t * T / n_launch
        us = 1e3 * tag_ms.get("moses_dec_fwd", float("nan")) / n_launch
        usb = 1e3 * tag_ms.get("moses_dec_bwd", float("nan")) / (n_launch + 1)
        peak = BF16_DENSE_PEAK_TFLOPS if dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
        ach = f / (us * 1e-6) / 1e12
        return dict(bound="mfma", kernel="lstm_step_fwd_kernel (GRU cell)", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                    traffic=None, launches_per_pass=n_launch, avg_launch_us=dict(gru_step_fwd=round(us, 2), gru_step_bwd=round(usb, 2)),
                    flops_per_launch=dict(gru_step_fwd=f), note="B x 512 per cell: latency / launch bound, far from either roof")

    def config(self, world):
        return dict(workload=self.name, per_gpu_batch=self.B, global_batch=self.B * world, vocab=self.V, padded_len=self.T,
                    tokens_per_batch=self.tokens, parallelism=f"dp{world}")


class Models2dWorkload:
    """models2d.VAE (models2d.py:8-52; SURVEY section 8f row 4 -- the architecture north_star's words describe literally: Conv1d encoder
    over the one-hot block [B,120,35], 2-d latent, GRU(2 -> 501, 3 layers) decoder) stepped like train.py:95-104: fwd + train.py:31-38
    loss + bwd + clip(3.0) + Adam(8e-4)."""
    name = ("models2d.VAE ELBO train step: Conv1d(120->9,k9)/(9->9,k9)/(9->10,k11)+ReLU encoder over the one-hot block, GRU(2->501,3) decoder, "
            "fwd+loss+bwd+clip(3.0)+Adam(8e-4), synthetic one-hot SMILES [B,120,35]")

    def __init__(self, B, dtype, dev, rank, sync):
        import molecular_vae_amd as mv
        from molecular_vae_amd import models2d as M2
        self.mv, self.B = mv, B
        torch.manual_seed(42)
        self.model = M2.VAE(dtype=torch.bfloat16 if dtype == "bf16" else torch.float32).to(dev).train()
        self.optimizer = mv.FusedAdam(self.model.parameters(), lr=8e-4, max_grad_norm=3.0, grad_sync=sync, shard_optimizer=bool(getattr(sync, "shard_optimizer", False)))
        self.loss_function = mv.make_loss_function(120)
        g = torch.Generator().manual_seed(1234 + rank)
        idx = torch.randint(0, 35, (B, 120), generator=g).to(dev)
        self.x = torch.nn.functional.one_hot(idx, 35).float()            # the one-hot block, resident in HBM

    def step(self):
        self.optimizer.zero_grad(set_to_none=True)
        recon, mu, logvar = self.model(self.x)
        loss = self.loss_function(recon, self.x, mu, logvar)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _macs_per_mol(self):
        H, L = 501, 120
        conv = 27 * 9 * 120 * 9 + 19 * 9 * 9 * 9 + 9 * 10 * 9 * 11
        return conv + 90 * 435 + 2 * 435 * 2 + 4 + 3 * H * 2 + L * (3 * H * H * 3 + 3 * H * H * 2) + L * H * 35

    def flops_per_step(self):
        return 3 * 2 * self._macs_per_mol() * self.B

    def roofline(self, tag_ms, dtype):
        H, NL, T, B = 501, 3, 120, self.B
        per_t = 2 * B * 3 * H * (H * NL + H * (NL - 1))          # algorithmic (unpadded, 3 gates); layer 0's input projection is hoisted
        n_launch = T + NL - 1
        f = per_t * T / n_launch
        us = 1e3 * tag_ms.get("m2d_gru_fwd", float("nan")) / n_launch
        usb = 1e3 * tag_ms.get("m2d_gru_bwd", float("nan")) / n_launch
        peak = BF16_DENSE_PEAK_TFLOPS if dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
        ach = f / (us * 1e-6) / 1e12
        return dict(bound="mfma", kernel="lstm_step_fwd_kernel (GRU cell, H 501 padded to 512)", achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                    frac=round(ach / peak, 4), traffic=None, launches_per_pass=n_launch,
                    avg_launch_us=dict(gru_step_fwd=round(us, 2), gru_step_bwd=round(usb, 2)), flops_per_launch=dict(gru_step_fwd=f))

    def config(self, world):
        return dict(workload=self.name, per_gpu_batch=self.B, global_batch=self.B * world, seq_len=120, vocab=35, latent=2, parallelism=f"dp{world}")
