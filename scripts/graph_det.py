"""Run-to-run spread of the eager step vs the hipGraph step (6 steps, fp32, 256^2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import synth
from cta_gan_amd.trainer import Hd_Trainer_x2
for graph in (False, False, True, True):
    cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=1, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
               Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
    tr = Hd_Trainer_x2(cfg)
    synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
    hist = []
    for i in range(6):
        batch = {k: synth.synth_smooth_images("gr%d_%s" % (i, k), 1, 256).cuda() for k in ("A2", "B1", "B2")}
        l = tr.train_step(batch, sync_losses=True)
        hist.append("%.5f/%.5f" % (l["SM"], l["loss_D"]))
    print("graph" if graph else "eager", " ".join(hist))
    del tr
