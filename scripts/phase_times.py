"""Per-phase GPU time of one HdGan stage-2 step (HIP events, B=16, 512x512, bf16)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
from cta_gan_amd.nets import l1_loss, masked_l1_loss
from cta_gan_amd.trainer.utils import smooothing_loss
nets.set_default_compute_dtype(torch.bfloat16)
B, S = 16, 512
cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
           Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1)
tr = Hd_Trainer_x2(cfg)
batch = {k: synth.synth_images("p_" + k, B, S).cuda() for k in ("A2", "B1", "B2")}
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
acc = {}
for it in range(3):
    marks.clear()
    A2, B2 = batch["A2"], batch["B2"]
    tr.optimizer_R_A.zero_grad(); tr.optimizer_G.zero_grad()
    mark("start")
    fake = tr.netG_A2B(A2); mark("G fwd (grad)")
    flow = tr.R_A(fake, B2); mark("Reg fwd")
    warped = tr.spatial_transform(fake, flow)
    sm = 10 * smooothing_loss(flow); sr = 20 * l1_loss(warped, B2); mark("STN+losses fwd")
    pf = tr.netD_B(fake); adv = tr.criterionGAN(pf, True); mark("D fwd (grad to G)")
    sr2 = 2 * masked_l1_loss(warped, B2, batch["B1"])
    total = sm + adv + sr + sr2
    total.backward(); mark("backward D+STN+Reg+G")
    tr.optimizer_R_A.step(); tr.optimizer_G.step(); mark("Adam R,G")
    tr.optimizer_D_B.zero_grad()
    with torch.no_grad():
        fake2 = tr.netG_A2B(A2)
    mark("G fwd (no_grad)")
    p1 = tr.netD_B(fake2); p2 = tr.netD_B(B2); mark("2x D fwd")
    ld = (tr.criterionGAN(p1, False) + tr.criterionGAN(p2, True)) / 2
    ld.backward(); mark("D backward x2")
    tr.optimizer_D_B.step(); mark("Adam D")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc.setdefault(n1, []).append(e0.elapsed_time(e1))
tot = 0
for k, v in acc.items():
    m = sorted(v)[len(v) // 2]; tot += m
    print("%-28s %7.2f ms" % (k, m))
print("%-28s %7.2f ms" % ("sum", tot))
