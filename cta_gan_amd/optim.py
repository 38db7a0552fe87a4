"""torch.optim.Adam(lr, betas=(0.5, 0.999)) as ONE multi-tensor HIP kernel per optimiser step.

Mirrors the update order of torch's Adam (the reference's only optimiser: trainer/HdTrainer.py:612-616,
trainer/CycTrainer.py:67-73) so that parameters track the reference step for step; state layout
(`exp_avg`, `exp_avg_sq`, `step` per parameter) follows torch so checkpoints stay interchangeable.
"""
from __future__ import annotations

import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise NotImplementedError("weight_decay is not used on the CTA-GAN path")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            step = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("cta_gan_amd.optim.Adam: parameters must live on the GPU (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                if not p.is_contiguous() or p.dtype != torch.float32:
                    raise RuntimeError("parameters must be contiguous fp32")
                if step is None:
                    step = st["step"]
                if st["step"] != step:  # parameters that joined later get their own launch
                    ops.adam_step([p], [g], [st["exp_avg"]], [st["exp_avg_sq"]], group["lr"], group["betas"][0],
                                  group["betas"][1], group["eps"], st["step"])
                    p._ctg_version = getattr(p, "_ctg_version", 0) + 1
                    continue
                ps.append(p); gs.append(g); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            if ps:
                ops.adam_step(ps, gs, ms, vs, group["lr"], group["betas"][0], group["betas"][1], group["eps"], step)
                for p in ps:
                    p._ctg_version = getattr(p, "_ctg_version", 0) + 1
        return None
