"""Eager vs hipGraph-replayed HdGan stage-2 step at small batches (the reference's yaml uses batchSize 1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
nets.set_default_compute_dtype(torch.bfloat16)
S = 512
MODES = [bool(int(c)) for c in (sys.argv[2] if len(sys.argv) > 2 else "01")]
for B in [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,16").split(",")]:
    res = {False: (1, 0, 0), True: (1, 0, 0)}
    for graph in MODES:
        cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
                   Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
        torch.manual_seed(0)
        tr = Hd_Trainer_x2(cfg)
        synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
        batch = {k: synth.synth_images("g_" + k, B, S).cuda() for k in ("A2", "B1", "B2")}
        for _ in range(6):
            tr.train_step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            tr.train_step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[graph] = (dt, float(tr.last["loss_D"]), float(tr.netG_A2B.model_tail._modules["7"].bias.detach()[0]))
        del tr
        torch.cuda.empty_cache()
    print("B=%2d eager %.2f ms  graph %.2f ms  (%.2fx)   loss_D %.6f / %.6f   tail bias %.8f / %.8f" % (
        B, res[False][0] * 1e3, res[True][0] * 1e3, res[False][0] / res[True][0], res[False][1], res[True][1],
        res[False][2], res[True][2]))
