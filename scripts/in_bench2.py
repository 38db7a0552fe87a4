"""InstanceNorm elementwise kernels against a plain device copy of the same bytes, bf16 and split pair, at the three big shapes of
the Hd step (B=16): 128^2 x 256, 256^2 x 128, 512^2 x 64.    python scripts/in_bench2.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops, nets
dev = torch.device("cuda:0")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for mode in ("bf16", "x3"):
    nets.set_default_compute_dtype("bf16x3" if mode == "x3" else torch.bfloat16)
    for (S, C) in ((128, 256), (256, 128), (512, 64)):
        B = 16
        x32 = torch.randn(B, S, S, C, device=dev)
        if mode == "x3":
            x, r, g = ops.to_pair(x32), ops.to_pair(x32 * 0.5), ops.to_pair(x32 * 0.25)
            o = ops.empty_act((B, S, S, C), torch.bfloat16, dev)
            esz = 4
        else:
            x, r, g = x32.bfloat16(), (x32 * 0.5).bfloat16(), (x32 * 0.25).bfloat16()
            o = torch.empty_like(x)
            esz = 2
        mb = B * S * S * C * esz / 1e6
        src = torch.empty(int(mb * 1e6) // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        src2 = torch.empty_like(src).normal_()
        mean, rstd = ops.in_stats(x)
        rows = [("device copy (torch copy_)", lambda: dst.copy_(src), 2),
                ("torch add (2 in, 1 out)", lambda: torch.add(src, src2, out=dst), 3),
                ("in_apply relu", lambda: ops.in_apply(x, mean, rstd, 1, None, o), 2),
                ("in_apply +res", lambda: ops.in_apply(x, mean, rstd, 0, r, o), 3),
                ("in_stats", lambda: ops.in_stats(x), 1),
                ("in_bwd (sums + apply)", lambda: ops.in_bwd(x, g, 0, mean, rstd, 1, o), 5)]
        for name, fn, passes in rows:
            us = t(fn)
            print("%-5s %dx%dx%d  %-26s %7.1f us  %5.2f TB/s (%d tensor passes of %.0f MB)" % (mode, S, S, C, name, us, passes * mb / us, passes, mb))
