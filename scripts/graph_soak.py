"""300-step soak of the hipGraph-replayed HdGan step at the reference's shipped batch sizes (Yaml/HdGan.yaml:19: batchSize 1,
train 4) against the eager step on the same batches: losses finite, generator gradients finite, and the two loss
trajectories stay together (medians of SR and total over windows of 20 steps within 15 %: single steps are chaotic, see
tests/test_step_parity_gpu.py).  python scripts/graph_soak.py [B] [steps] [bf16|bf16x3|bf16x3f]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
MODE = sys.argv[3] if len(sys.argv) > 3 else "bf16"
nets.set_default_compute_dtype(torch.bfloat16 if MODE == "bf16" else MODE)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
S = 512
runs = {}
for graph in (False, True):
    cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
               Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
    torch.manual_seed(0)
    tr = Hd_Trainer_x2(cfg)
    synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
    hist = []
    for i in range(N):
        batch = {k: synth.synth_smooth_images("gs%d_%s" % (i % 8, k), B, S).cuda() for k in ("A2", "B1", "B2")}
        out = tr.train_step(batch, sync_losses=True)
        assert all(v == v and abs(v) < 1e6 for v in out.values()), (graph, i, out)
        if i % 50 == 49:
            g = torch.cat([p.grad.reshape(-1).float() for p in tr.netG_A2B.parameters() if p.grad is not None])
            assert bool(torch.isfinite(g).all()), (graph, i)
        hist.append(out)
    runs[graph] = hist
    print("graph" if graph else "eager", "B=%d" % B, "last", {k: round(v, 4) for k, v in hist[-1].items()}, flush=True)
    del tr
    torch.cuda.empty_cache()
import statistics
for w0 in range(0, N - 19, 20):
    for k in ("SR", "total"):      # (loss_D hovers near 0 with spikes in both runs -- GAN dynamics -- and is only printed)
        a = statistics.median(h[k] for h in runs[False][w0:w0 + 20])
        b = statistics.median(h[k] for h in runs[True][w0:w0 + 20])
        assert abs(a - b) <= 0.15 * max(abs(a), abs(b)) + 1e-3, (w0, k, a, b)
    if w0 % 60 == 0:
        print("steps %3d-%3d  eager / graph medians:" % (w0, w0 + 19), {k: (round(statistics.median(h[k] for h in runs[False][w0:w0 + 20]), 4),
              round(statistics.median(h[k] for h in runs[True][w0:w0 + 20]), 4)) for k in ("SR", "total", "loss_D")})
print("graph soak ok: B=%d, %d steps, windowed medians of SR / total within 15 %% of the eager run" % (B, N))
