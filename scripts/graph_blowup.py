"""Graph-mode training at the bench shape: size of the generator's per-step update (a reset-like jump marks a bug)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
nets.set_default_compute_dtype(torch.bfloat16)
N, B, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
graph = (sys.argv[4] == "graph") if len(sys.argv) > 4 else True
cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
           Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
tr = Hd_Trainer_x2(cfg)
synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
prev = {k: flat(m) for k, m in (("G", tr.netG_A2B), ("R", tr.R_A), ("D", tr.netD_B))}
out = []
for i in range(N):
    batch = {k: synth.synth_smooth_images("gve%d_%s" % (i % 4, k), B, S).cuda() for k in ("A2", "B1", "B2")}
    tr.train_step(batch, sync_losses=True)
    cur = {k: flat(m) for k, m in (("G", tr.netG_A2B), ("R", tr.R_A), ("D", tr.netD_B))}
    og = tr.optimizer_G
    ps = [p for p in tr.netG_A2B.parameters() if og.state.get(p)]
    m = float(torch.cat([og.state[p]["exp_avg"].reshape(-1) for p in ps]).norm())
    v = float(torch.cat([og.state[p]["exp_avg_sq"].reshape(-1) for p in ps]).sum())
    gn = float(torch.cat([p.grad.reshape(-1).float() for p in ps if p.grad is not None]).norm())
    out.append(" ".join("%s %.3e" % (k, float((cur[k] - prev[k]).norm())) for k in cur) + "  G: m %.3e v %.3e grad %.3e" % (m, v, gn))
    prev = cur
    if not (gn == gn and gn < 1e6) and not os.environ.get("GB_QUIET"):
        print("step", i, "non-finite / huge grads per G parameter:")
        for name, p in tr.netG_A2B.named_parameters():
            if p.grad is None: continue
            g = p.grad.float()
            bad = int((~torch.isfinite(g)).sum()); big = int((g.abs() > 1e3).sum())
            if bad or big:
                print("   %-36s shape %-18s nonfinite %8d  |g|>1e3 %8d  of %d" % (name, tuple(g.shape), bad, big, g.numel()))
        break
print("\n".join(out))
