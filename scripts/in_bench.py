"""Micro-benchmark of the InstanceNorm elementwise kernels at the residual-block shape (16 x 128 x 128 x 256, bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops
dev = torch.device("cuda:0")
B, S, C = 16, 128, 256
x = torch.randn(B, S, S, C, device=dev).bfloat16()
r = torch.randn(B, S, S, C, device=dev).bfloat16()
g = torch.randn(B, S, S, C, device=dev).bfloat16()
gp = torch.randn(B, S + 2, S + 2, C, device=dev).bfloat16()
o = torch.empty_like(x)
mean, rstd = ops.in_stats(x)
mb = x.numel() * 2 / 1e6

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, fn, passes in (("in_apply relu", lambda: ops.in_apply(x, mean, rstd, 1, None, o), 2),
                         ("in_apply +res", lambda: ops.in_apply(x, mean, rstd, 0, r, o), 3),
                         ("in_stats", lambda: ops.in_stats(x), 1),
                         ("in_bwd pad0", lambda: ops.in_bwd(x, g, 0, mean, rstd, 1, o), 5),
                         ("in_bwd pad1", lambda: ops.in_bwd(x, gp, 1, mean, rstd, 1, o), 5),
                         ("grad_combine", lambda: ops.grad_combine(g, gp, 1, None, 0, o), 3)):
    us = t(fn)
    print("%-14s %7.1f us  %5.2f TB/s (%d tensor passes)" % (name, us, passes * mb / us, passes))
