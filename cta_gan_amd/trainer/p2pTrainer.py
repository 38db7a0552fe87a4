"""Step body of the reference's `P2p_Trainer` (trainer/p2pTrainer.py:55-312) on the HIP path: pix2pix with the CycleGan
`Generator` and a `Discriminator(2 * input_nc)` that judges the channel-concatenated (input, output) pair."""
from __future__ import annotations

import torch

from ..Model.HdGan import DataPrefetcher
from .. import dp, optim, synth
from ..Model.CycleGan import Discriminator, Generator
from ..nets import l1_loss
from .HdTrainer import run_epoch_steps, _frozen, resume_epoch, run_test_loop, save_epoch, validate_if_due, synced_losses


class P2p_Trainer:
    def __init__(self, config):
        self.config = config
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.netG_A2B = Generator(config["input_nc"], config["output_nc"]).to(dev)
        self.netD_B = Discriminator(config["input_nc"] * 2).to(dev)                      # p2pTrainer.py:61
        dp.broadcast_params(self.netG_A2B, self.netD_B)
        self.optimizer_D_B = optim.Adam(self.netD_B.parameters(), lr=config["lr"], betas=(0.5, 0.999))
        self.optimizer_G = optim.Adam(self.netG_A2B.parameters(), lr=config["lr"], betas=(0.5, 0.999))
        self.last = {}

    def update_learning_rate(self):
        """p2pTrainer.py:107-116."""
        lrd = self.config["lr"] / self.config["decay_epoch"]
        lr = self.config["lr"] - lrd
        for g in self.optimizer_D_B.param_groups:
            g["lr"] = lr
        for g in self.optimizer_G.param_groups:
            g["lr"] = lr
        self.config["lr"] = lr

    def train_step(self, batch, sync_losses: bool = False):
        """p2pTrainer.py:122-148 on a dict batch of device tensors A, B (each (B, nc, S, S) fp32)."""
        cfg = self.config
        real_A, real_B = batch["A"], batch["B"]

        def mse(p, t):
            return ((p - t) ** 2).mean()

        self.optimizer_G.zero_grad()
        fake_B = self.netG_A2B(real_A)
        loss_L1 = l1_loss(fake_B, real_B) * cfg["P2P_lamda"]
        # D's own weight gradients of this pass are zeroed before anything reads them (p2pTrainer.py:139): not computed
        with _frozen(self.netD_B):
            pred_fake = self.netD_B(torch.cat((real_A.to(fake_B.dtype), fake_B), 1))
        loss_GAN_A2B = mse(pred_fake, 1.0) * cfg["Adv_lamda"]
        toal_loss = loss_L1 + loss_GAN_A2B
        toal_loss.backward()
        dp.allreduce_grads(self.netG_A2B.parameters())
        self.optimizer_G.step()

        self.optimizer_D_B.zero_grad()
        with torch.no_grad():
            fake_B = self.netG_A2B(real_A)
        # (A, fake) and (A, real) pairs as ONE pass over the concatenated batch: D is per-sample (InstanceNorm), so the halves
        # equal the reference's two calls (p2pTrainer.py:143-144); there the loss weight scales the PREDICTION
        nb = real_A.shape[0]
        a = real_A.to(fake_B.dtype)
        pairs = torch.cat([torch.cat((a, fake_B), 1), torch.cat((a, real_B.to(fake_B.dtype)), 1)], 0)
        pred = self.netD_B(pairs) * cfg["Adv_lamda"]
        loss_D_B = mse(pred[:nb], 0.0) + mse(pred[nb:], 1.0)
        loss_D_B.backward()
        dp.allreduce_grads(self.netD_B.parameters())
        self.optimizer_D_B.step()
        self.last = dict(L1=loss_L1, GAN_A2B=loss_GAN_A2B, total=toal_loss, loss_D=loss_D_B, fake_B=fake_B)
        if sync_losses:
            return synced_losses(self.last)
        return None

    def synthetic_batch(self, seed=1234):
        b, s = self.config["batchSize"], self.config["size"]
        tag = "_r%d" % dp.rank() if dp.world_size() > 1 else ""      # replicas train on different slices
        return {k: synth.synth_images("p2p_%s_%d%s" % (k, seed, tag), b, s).to(self.device) for k in ("A", "B")}

    def train(self, dataloader=None, val_dataloader=None):
        """Epoch loop of p2pTrainer.py:118-148 (see Hd_Trainer_x2.train)."""
        for epoch in range(self.config["epoch"] + 1, self.config["n_epochs"] + 1 + self.config["decay_epoch"]):
            if epoch > self.config["n_epochs"]:
                self.update_learning_rate()
            it = dataloader if dataloader is not None else (
                self.synthetic_batch(i) for i in range(self.config.get("synthetic_steps", 4)))
            if dataloader is not None:
                # host batches: pinned, double-buffered H2D on a copy stream, one batch ahead of the step that trains
                it = DataPrefetcher(it, device=self.device)
            run_epoch_steps(self, it)
            val = validate_if_due(self, epoch, dataloader, val_dataloader, ("A", "B"))      # p2pTrainer.py:153-174
            save_epoch(self, epoch, self._ckpt_files(), self._ckpt_optimizers(), val=val)

    def _ckpt_files(self):   # p2pTrainer.py:179-184
        return {"netG_A2B_": self.netG_A2B, "netD_B_": self.netD_B}

    def _ckpt_optimizers(self):
        return {"G": self.optimizer_G, "D_B": self.optimizer_D_B}

    def resume(self, epoch):
        resume_epoch(self, epoch, self._ckpt_files(), self._ckpt_optimizers())

    def test(self, dataloader=None):
        """p2pTrainer.py:186-312 (generator inference + windowed / raw MAE, PSNR, UQI with its `bb = b`, `cc = c`
        aliasing at :233-243) and SSIM / SSIMw; LPIPS and the DICOM export are not part of this build."""
        return run_test_loop(self, dataloader, ("A", "B"), "netG_A2B.pth", aliased=True)
