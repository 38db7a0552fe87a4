"""Eager vs hipGraph-replayed training at the bench shape: same init, same batches, losses step by step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
nets.set_default_compute_dtype(torch.bfloat16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 512)

def run(graph):
    cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
               Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
    tr = Hd_Trainer_x2(cfg)
    synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
    out = []
    for i in range(N):
        batch = {k: synth.synth_smooth_images("gve%d_%s" % (i % 4, k), B, S).cuda() for k in ("A2", "B1", "B2")}
        out.append(tr.train_step(batch, sync_losses=True))
    return out

g = run(True); e = run(False)
for i, (a, b) in enumerate(zip(e, g)):
    print(i, " ".join("%s %.4f/%.4f" % (k, a[k], b[k]) for k in ("SR", "adv", "SM", "loss_D")), "  max rel %.2e" % max(
        abs(a[k] - b[k]) / max(abs(a[k]), 1e-3) for k in a))
