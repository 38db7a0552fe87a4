"""torch.optim.Adam(lr, betas=(0.5, 0.999)) as ONE multi-tensor HIP kernel per optimiser step.

Mirrors the update order of torch's Adam (the reference's only optimiser: trainer/HdTrainer.py:612-616,
trainer/CycTrainer.py:67-73) so that parameters track the reference step for step; state layout
(`exp_avg`, `exp_avg_sq`, `step` per parameter) follows torch so checkpoints stay interchangeable.
"""
from __future__ import annotations

import torch

from . import ops


class Adam(torch.optim.Optimizer):
    """`capturable=True` keeps the step counter / bias corrections in device memory (advanced by a 1-thread kernel
    inside the stream), so `step()` issues no per-step host constant and can be recorded into a hipGraph."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, capturable=False):
        if weight_decay != 0:
            raise NotImplementedError("weight_decay is not used on the CTA-GAN path")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = capturable
        self._dev_state = {}

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        for gi, group in enumerate(self.param_groups):
            if self.capturable:
                self._step_capturable(gi, group)
                continue
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

    def _ensure_dev_state(self, gi, group):
        if gi not in self._dev_state:
            # resume from the host-side count if eager steps were taken before
            ps = group["params"]
            n0 = max((self.state[p].get("step", 0) for p in ps if self.state.get(p)), default=0)
            b1, b2 = group["betas"]
            self._dev_state[gi] = torch.tensor([float(n0), 1.0 - b1 ** max(n0, 1), (1.0 - b2 ** max(n0, 1)) ** 0.5],
                                               dtype=torch.float32, device=ps[0].device)

    def prepare_capture(self):
        """Create the device-side step counters NOW (an H2D copy must not happen inside a stream capture). The moment
        buffers must already exist too: take at least one eager step before capturing."""
        for gi, group in enumerate(self.param_groups):
            self._ensure_dev_state(gi, group)

    def note_replayed(self):
        """A captured step() was replayed: advance the host mirrors (state_dict `step`, packed-weight versions)."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] += 1
                    p._ctg_version = getattr(p, "_ctg_version", 0) + 1

    def load_state_dict(self, state_dict):
        """Accepts this class's files and those of `torch.optim.Adam` (the reference's optimiser, whose `step` is a
        tensor): `step` becomes a Python int, the moments contiguous fp32 on the parameter's device."""
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            if not st:
                continue
            st["step"] = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])
            for k in ("exp_avg", "exp_avg_sq"):
                st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
        self._dev_state.clear()      # rebuilt from the loaded `step` on the next step()

    def _step_capturable(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return
        self._ensure_dev_state(gi, group)
        capturing = torch.cuda.is_current_stream_capturing()
        gs, ms, vs = [], [], []
        for p in ps:
            st = self.state[p]
            if not st:
                if capturing:
                    raise RuntimeError("Adam state would be allocated inside a stream capture: take an eager step first")
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
            if not capturing:    # host mirror (state_dict); the kernels read the device copy. A capture runs nothing:
                st["step"] += 1  # note_replayed() counts the replays instead
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            gs.append(g); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
        state = self._dev_state[gi]
        ops.adam_tick(state, group["betas"][0], group["betas"][1])
        ops.adam_step(ps, gs, ms, vs, group["lr"], group["betas"][0], group["betas"][1], group["eps"], 0,
                      dev_state=state)
        for p in ps:
            p._ctg_version = getattr(p, "_ctg_version", 0) + 1
