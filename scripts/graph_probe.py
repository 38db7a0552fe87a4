"""Capture one component of the step into a hipGraph and replay it (find what the runtime cannot capture)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth, optim
from cta_gan_amd.Model.HdGan import Generator, Discriminator_m, GANLoss
from cta_gan_amd.trainer.reg import Reg
from cta_gan_amd.trainer.transformer import Transformer_2D
from cta_gan_amd.trainer.utils import smooothing_loss
from cta_gan_amd.nets import l1_loss, masked_l1_loss
nets.set_default_compute_dtype(torch.bfloat16)
which = sys.argv[1]
B, S = 2, 256
a = synth.synth_images("gp_a", B, S).cuda()
b = synth.synth_images("gp_b", B, S).cuda()
G = Generator(1, 1).cuda(); D = Discriminator_m(1).cuda(); R = Reg(S, S, 1, 1).cuda(); T = Transformer_2D()
crit = GANLoss()
opt = optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999), capturable=True)

def body():
    if which == "gfwd":
        with torch.no_grad():
            return G(a)
    if which == "gbwd":
        G.zero_grad(); (G(a) * b).sum().backward(); return None
    if which == "dbwd":
        D.zero_grad(); crit(D(a), True).backward(); return None
    if which == "reg":
        R.zero_grad(); R(a, b).sum().backward(); return None
    if which == "stn":
        ar = a.clone().requires_grad_(True)
        fl = (torch.zeros(B, 2, S, S, device="cuda") + 0.5).requires_grad_(True)
        (T(ar, fl).sum() + smooothing_loss(fl) + l1_loss(ar, b) + masked_l1_loss(ar, b, b)).backward(); return None
    if which == "adam":
        G.zero_grad(); (G(a) * b).sum().backward(); opt.step(); return None
    raise SystemExit("?")

for _ in range(3):
    body()
torch.cuda.synchronize()
for m in (G, D, R):
    for sub in [m] + list(getattr(m, "_scales", [])):
        sub._cache.store.clear()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay(); g.replay()
torch.cuda.synchronize()
print("captured + replayed:", which)
