"""Spread of loss trajectories: eager x2 vs graph x2 (B=4, 256^2, 24 steps), to tell chaos from a bug."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
nets.set_default_compute_dtype(torch.bfloat16)
B, S, N = 4, 256, 24
for graph in (False, False, True, True):
    cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
               Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
    torch.manual_seed(0)
    tr = Hd_Trainer_x2(cfg)
    synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
    hist = []
    for i in range(N):
        batch = {k: synth.synth_smooth_images("sc%d_%s" % (i % 4, k), B, S).cuda() for k in ("A2", "B1", "B2")}
        out = tr.train_step(batch, sync_losses=(i % 4 == 3))
        if out is not None:
            hist.append("%.3f/%.3f" % (out["SR"], out["loss_D"]))
    print("graph" if graph else "eager", " ".join(hist), flush=True)
    del tr
