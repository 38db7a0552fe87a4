"""Soak: N training steps at the bench configuration; losses must stay finite and allocated memory flat."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
nets.set_default_compute_dtype(torch.bfloat16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
B, S = 16, 512
cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
           Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
tr = Hd_Trainer_x2(cfg)
mem = []
for i in range(N):
    batch = {k: synth.synth_smooth_images("soak%d_%s" % (i % 4, k), B, S).cuda() for k in ("A2", "B1", "B2")}
    out = tr.train_step(batch, sync_losses=(i % 10 == 9 or i == N - 1))
    if out is not None:
        assert all(v == v and abs(v) < 1e6 for v in out.values()), out
        mem.append(torch.cuda.memory_allocated() / 2**30)
        print(i, {k: round(v, 4) for k, v in out.items()}, "alloc %.2f GiB  peak %.2f GiB" % (mem[-1], torch.cuda.max_memory_allocated() / 2**30), flush=True)
assert max(mem) - min(mem) < 0.5, mem
print("soak ok", "graph" if graph else "eager")
