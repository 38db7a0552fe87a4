"""Run each network's forward + backward twice on the same inputs at the bench shape: every output and parameter gradient
must be bit-identical (no atomics inside the networks; only the STN scatter is order-dependent)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.Model.HdGan import Generator, Discriminator_m
from cta_gan_amd.trainer.reg import Reg
nets.set_default_compute_dtype(torch.bfloat16)
B, S = 16, 512
a = synth.synth_smooth_images("det_a", B, S).cuda()
b = synth.synth_smooth_images("det_b", B, S).cuda()
for name in ("gen", "reg", "disc"):
    if name == "reg":
        net = synth.fill_module(Reg(S, S, 1, 1), seed=4).cuda(); run = lambda x: net(x, b)
    elif name == "gen":
        net = synth.fill_module(Generator(1, 1), seed=0).cuda(); run = lambda x: net(x)
    else:
        net = synth.fill_module(Discriminator_m(1), seed=1).cuda(); run = lambda x: net(x)[0][-1]
    res = []
    for rep in range(3):
        for p in net.parameters(): p.grad = None
        x = a.clone().requires_grad_(True)
        y = run(x)
        (y.float() * torch.linspace(0.5, 1.5, y.numel(), device=y.device).view_as(y)).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    bad = []
    for rep in (1, 2):
        if not torch.equal(res[0][0], res[rep][0]): bad.append("output")
        if not torch.equal(res[0][1], res[rep][1]): bad.append("input grad")
        for k in res[0][2]:
            if not torch.equal(res[0][2][k], res[rep][2][k]):
                d = (res[0][2][k] - res[rep][2][k]).abs().max().item()
                bad.append("%s (max diff %.3e of %.3e)" % (k, d, res[0][2][k].abs().max().item()))
    print(name, "bitwise repeatable" if not bad else "DIFFERS: " + "; ".join(bad[:8]))
    del net, res
    torch.cuda.empty_cache()
