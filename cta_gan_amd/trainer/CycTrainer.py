"""Step body of the reference's `Cyc_Trainer` (trainer/CycTrainer.py:60-200) on the HIP path."""
from __future__ import annotations

import itertools

import torch

from ..Model.HdGan import DataPrefetcher
from .. import dp, optim, synth
from ..Model.CycleGan import Discriminator, Generator
from ..nets import l1_loss
from . import HdTrainer as _hd
from .HdTrainer import run_epoch_steps, _frozen, resume_epoch, run_test_loop, save_epoch, validate_if_due, side_branch, synced_losses
from .utils import ReplayBuffer


class Cyc_Trainer:
    def __init__(self, config):
        self.config = config
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.netG_A2B = Generator(config["input_nc"], config["output_nc"]).to(dev)
        self.netD_B = Discriminator(config["input_nc"]).to(dev)
        self.netG_B2A = Generator(config["input_nc"], config["output_nc"]).to(dev)
        self.netD_A = Discriminator(config["input_nc"]).to(dev)
        dp.broadcast_params(self.netG_A2B, self.netD_B, self.netG_B2A, self.netD_A)
        self.optimizer_D_B = optim.Adam(self.netD_B.parameters(), lr=config["lr"], betas=(0.5, 0.999))
        self.optimizer_G = optim.Adam(itertools.chain(self.netG_A2B.parameters(), self.netG_B2A.parameters()),
                                      lr=config["lr"], betas=(0.5, 0.999))
        self.optimizer_D_A = optim.Adam(self.netD_A.parameters(), lr=config["lr"], betas=(0.5, 0.999))
        self.fake_A_buffer = ReplayBuffer()
        self.fake_B_buffer = ReplayBuffer()
        self.last = {}

    def update_learning_rate(self):
        """CycTrainer.py:117-126 (only D_B and G decay; D_A is left out there too)."""
        lrd = self.config["lr"] / self.config["decay_epoch"]
        lr = self.config["lr"] - lrd
        for g in self.optimizer_D_B.param_groups:
            g["lr"] = lr
        for g in self.optimizer_G.param_groups:
            g["lr"] = lr
        self.config["lr"] = lr

    def train_step(self, batch, sync_losses: bool = False):
        cfg = self.config
        real_A, real_B = batch["A"], batch["B"]

        def mse(p, t):
            return ((p - t) ** 2).mean()

        self.optimizer_G.zero_grad()
        # the discriminators' weight gradients of the G step are zeroed before use (CycTrainer.py:165,182): skipped
        fake_B = self.netG_A2B(real_A)
        # the two adversarial branches only share fake_B / fake_A with the generator passes that follow: second stream
        with side_branch(self) as br_b:
            with _frozen(self.netD_B):
                loss_GAN_A2B = cfg["Adv_lamda"] * mse(self.netD_B(fake_B), 1.0)
        fake_A = self.netG_B2A(real_B)
        with side_branch(self) as br_a:
            with _frozen(self.netD_A):
                loss_GAN_B2A = cfg["Adv_lamda"] * mse(self.netD_A(fake_A), 1.0)
        recovered_A = self.netG_B2A(fake_B)
        loss_cycle_ABA = cfg["Cyc_lamda"] * l1_loss(recovered_A, real_A)
        recovered_B = self.netG_A2B(fake_A)
        loss_cycle_BAB = cfg["Cyc_lamda"] * l1_loss(recovered_B, real_B)
        br_b.join()
        br_a.join()
        loss_Total = loss_GAN_A2B + loss_GAN_B2A + loss_cycle_ABA + loss_cycle_BAB
        loss_Total.backward()
        dp.allreduce_grads(itertools.chain(self.netG_A2B.parameters(), self.netG_B2A.parameters()))
        self.optimizer_G.step()

        self.optimizer_D_A.zero_grad()
        fake_A_b = self.fake_A_buffer.push_and_pop(fake_A)
        # real and buffered-fake halves in one pass: D is per-sample (InstanceNorm), so this equals the reference's two
        # calls (CycTrainer.py:168-173)
        nb = real_A.shape[0]

        def d_pair(net, real, fake):
            if _hd._NO_D_BATCH:     # A/B and test switch: the reference's two separate calls
                return net(real), net(fake.detach().to(real.dtype))
            pred = net(torch.cat([real, fake.detach().to(real.dtype)], 0))
            return pred[:nb], pred[nb:]
        pred_real, pred_fake = d_pair(self.netD_A, real_A, fake_A_b)
        loss_D_A = cfg["Adv_lamda"] * mse(pred_real, 1.0) + cfg["Adv_lamda"] * mse(pred_fake, 0.0)
        loss_D_A.backward()
        dp.allreduce_grads(self.netD_A.parameters())
        self.optimizer_D_A.step()

        self.optimizer_D_B.zero_grad()
        fake_B_b = self.fake_B_buffer.push_and_pop(fake_B)
        pred_real, pred_fake = d_pair(self.netD_B, real_B, fake_B_b)
        loss_D_B = cfg["Adv_lamda"] * mse(pred_real, 1.0) + cfg["Adv_lamda"] * mse(pred_fake, 0.0)
        loss_D_B.backward()
        dp.allreduce_grads(self.netD_B.parameters())
        self.optimizer_D_B.step()
        self.last = dict(GAN_A2B=loss_GAN_A2B, GAN_B2A=loss_GAN_B2A, cyc_ABA=loss_cycle_ABA, cyc_BAB=loss_cycle_BAB,
                         total=loss_Total, loss_D_A=loss_D_A, loss_D_B=loss_D_B, fake_B=fake_B, fake_A=fake_A)
        if sync_losses:
            return synced_losses(self.last)
        return None

    def synthetic_batch(self, seed=1234):
        b, s = self.config["batchSize"], self.config["size"]
        tag = "_r%d" % dp.rank() if dp.world_size() > 1 else ""      # replicas train on different slices
        return {k: synth.synth_images("cyc_%s_%d%s" % (k, seed, tag), b, s).to(self.device) for k in ("A", "B")}

    def train(self, dataloader=None, val_dataloader=None):
        for epoch in range(self.config["epoch"] + 1, self.config["n_epochs"] + 1 + self.config["decay_epoch"]):
            if epoch > self.config["n_epochs"]:
                self.update_learning_rate()
            it = dataloader if dataloader is not None else (
                self.synthetic_batch(i) for i in range(self.config.get("synthetic_steps", 4)))
            if dataloader is not None:
                # host batches: pinned, double-buffered H2D on a copy stream, one batch ahead of the step that trains
                it = DataPrefetcher(it, device=self.device)
            run_epoch_steps(self, it)
            val = validate_if_due(self, epoch, dataloader, val_dataloader, ("A", "B"))      # CycTrainer.py:203-226
            save_epoch(self, epoch, self._ckpt_files(), self._ckpt_optimizers(), val=val)

    def _ckpt_files(self):   # CycTrainer.py:233-236: the A2B generator's file has no stem
        return {"": self.netG_A2B, "netD_B_": self.netD_B, "netG_B2A_": self.netG_B2A, "netD_A_": self.netD_A}

    def _ckpt_optimizers(self):
        return {"G": self.optimizer_G, "D_A": self.optimizer_D_A, "D_B": self.optimizer_D_B}

    def resume(self, epoch):
        resume_epoch(self, epoch, self._ckpt_files(), self._ckpt_optimizers())

    def test(self, dataloader=None):
        """Inference + metrics loop of CycTrainer.py:238-398 (see Hd_Trainer_x2.test): batches are dicts with 'A', 'B'
        (B,1,S,S) and optionally 'WC' / 'WW'.  The windowed metrics reproduce the reference's aliasing (`bb = b`,
        `cc = c` at :288-298), i.e. they compare the two +-1 foreground masks.  SSIM / SSIMw (`ops.window_ssim`) are reported as well; LPIPS, DICOM export: not built."""
        return run_test_loop(self, dataloader, ("A", "B"), "aa.pth", aliased=True)
