"""Step body of the reference's `Reg_Trainer` (trainer/RegTrainer.py:90-380) on the HIP path.

Term for term its step (:170-198) is the stage-1 CTA-GAN step (HdTrainer.py:192-228) with the CycleGan `Generator` /
`Discriminator`, batches keyed A / B and loss weights read from Corr_lamda / Adv_lamda / Smooth_lamda, so it runs on
`Hd_Trainer_x1`'s step with those names mapped."""
from __future__ import annotations

from .. import synth
from .HdTrainer import Hd_Trainer_x1, run_test_loop


class Reg_Trainer(Hd_Trainer_x1):
    def __init__(self, config):
        # the user's dict stays the live config (update_learning_rate writes 'lr' back into it, :159); the stage-1 names are added
        config.setdefault("Adv_lamda1", config["Adv_lamda"])
        config.setdefault("Corr_lamda1", config["Corr_lamda"])
        config.setdefault("lrd", config["lr"])            # one learning rate for all three optimisers (:97-101)
        super().__init__(config)

    def update_learning_rate(self):
        """RegTrainer.py:148-159: all three optimisers, D included, follow 'lr'."""
        lrd = self.config["lr"] / self.config["decay_epoch"]
        lr = self.config["lr"] - lrd
        for opt in (self.optimizer_D_B, self.optimizer_R_A, self.optimizer_G):
            for g in opt.param_groups:
                g["lr"] = lr
        self.config["lr"] = lr

    def train_step(self, batch, sync_losses: bool = False):
        """RegTrainer.py:170-198 on a dict batch of device tensors A, B."""
        return super().train_step(dict(A2=batch["A"], B2=batch["B"], B1=batch["B"]), sync_losses)

    def synthetic_batch(self, seed=1234):
        b, s = self.config["batchSize"], self.config["size"]
        return {k: synth.synth_images("reg_%s_%d" % (k, seed), b, s).to(self.device) for k in ("A", "B")}

    _val_keys = ("A", "B")         # RegTrainer.py:211-219
    _val_suffix = ".pth"           # RegTrainer.py:225-231

    def _ckpt_files(self):   # RegTrainer.py:236-240
        return {"netG_A2B_": self.netG_A2B, "R_A_": self.R_A, "netD_B_": self.netD_B}

    def test(self, dataloader=None):
        """RegTrainer.py:242-380: its masks are deep copies (:299,308), i.e. the Hd_Trainer_x2.test arithmetic."""
        return run_test_loop(self, dataloader, ("A", "B"), "netG_A2B_5b.pth", aliased=False)
