"""Where do eager and hipGraph-replayed training part?  Per-step parameter deltas of a few tensors + Adam device state."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
nets.set_default_compute_dtype(torch.bfloat16)
N, B, S = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (9, 2, 256)
KEYS = [("G", "model_head.1.weight"), ("G", "model_body.4.conv_block.1.weight"), ("G", "model_tail.7.weight"),
        ("R", None), ("D", None)]

def run(graph):
    cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
               Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
    tr = Hd_Trainer_x2(cfg)
    synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
    mods = {"G": tr.netG_A2B, "R": tr.R_A, "D": tr.netD_B}
    def snap():
        out = {}
        for m, k in KEYS:
            ps = dict(mods[m].named_parameters())
            if k is None:
                out[m + ":all"] = torch.cat([p.detach().reshape(-1) for p in ps.values()]).clone()
            else:
                out[m + ":" + k] = ps[k].detach().clone()
        return out
    hist = []
    prev = snap()
    for i in range(N):
        batch = {k: synth.synth_smooth_images("gve%d_%s" % (i % 4, k), B, S).cuda() for k in ("A2", "B1", "B2")}
        losses = tr.train_step(batch, sync_losses=True)
        if os.environ.get("GVE_LOSS"):
            print("   %s step %d" % ("graph" if graph else "eager", i), {k: round(v, 4) for k, v in losses.items()}, flush=True)
        cur = snap()
        hist.append({k: (cur[k] - prev[k]).float() for k in cur})
        prev = cur
        if graph:
            st = [o._dev_state.get(0) for o in (tr.optimizer_G, tr.optimizer_R_A, tr.optimizer_D_B)]
            print("  graph step", i, "dev step counters", [None if s is None else float(s[0]) for s in st],
                  "host", tr.optimizer_G.state[next(iter(tr.netG_A2B.parameters()))]["step"])
    return hist

if os.environ.get("GVE_ORDER") == "graph_first":
    g = run(True); e = run(False)
else:
    e, g = run(False), run(os.environ.get("GVE_MODE", "graph") == "graph")
for i, (a, b) in enumerate(zip(e, g)):
    print(i, " ".join("%s |d| %.3e/%.3e cos %.3f" % (k.split(":")[0] + ":" + k.split(":")[1][-14:], a[k].norm(), b[k].norm(),
                                                     float((a[k] * b[k]).sum() / (a[k].norm() * b[k].norm() + 1e-30))) for k in a))
