"""Gradients of the generator at a size where every fused path is live (B=2, 256^2: 64^2 residual-block maps), bf16x3f and bf16
against bf16x3 (itself 5e-3 ... 9e-3 from the fp32 oracle): rel-L2 per parameter tensor and of the input gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cta_gan_amd import nets, synth  # noqa: E402
from cta_gan_amd.Model.HdGan import Generator  # noqa: E402


def run(mode):
    nets.set_default_compute_dtype({"bf16": torch.bfloat16}.get(mode, mode))
    g = synth.fill_module(Generator(1, 1), seed=0).cuda()
    x = synth.synth_smooth_images("gx", 2, 256).cuda().requires_grad_(True)
    y = g(x)
    (y.float() * torch.linspace(0.5, 1.5, y.numel(), device=y.device).view_as(y)).sum().backward()
    return {"x": x.grad.double().cpu(), **{k: p.grad.double().cpu() for k, p in g.named_parameters() if p.grad is not None}}


ref = run("bf16x3")
for mode in ("bf16x3f", "bf16"):
    got = run(mode)
    errs = {k: float((got[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-30)) for k in ref if k.endswith("weight") or k == "x"}
    worst = max(errs, key=errs.get)
    print(mode, "input gradient %.2e; weights: median %.2e, worst %.2e (%s)" % (
        errs["x"], sorted(errs.values())[len(errs) // 2], errs[worst], worst))
