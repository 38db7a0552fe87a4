"""Run one network fwd+bwd in isolation (for rocprofv3 --stats): python scripts/net_profile.py reg|gen|disc"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.Model.HdGan import Generator, Discriminator_m
from cta_gan_amd.trainer.reg import Reg
nets.set_default_compute_dtype(torch.bfloat16)
which = sys.argv[1]
B, S = 16, 512
a = synth.synth_images("np_a", B, S).cuda().requires_grad_(True)
b = synth.synth_images("np_b", B, S).cuda()
if which == "reg":
    net = Reg(S, S, 1, 1).cuda(); run = lambda: net(a, b)
elif which == "gen":
    net = Generator(1, 1).cuda(); run = lambda: net(a)
else:
    net = Discriminator_m(1).cuda(); run = lambda: net(a)[0][-1]
for i in range(4):
    out = run()
    out.float().sum().backward()
torch.cuda.synchronize()
