"""The statistics pass of the InstanceNorm backward (moments_partial_kernel<T, 1>) alone on a residual-block-sized map, against the
elementwise pass that follows it and a torch copy of the same bytes:  python scripts/experiments/r06_in_bwd_partial_bench.py [mode]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cta_gan_amd import nets, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype(torch.bfloat16 if mode == "bf16" else mode)
B, H, W, C = 16, 128, 128, 256
x32 = torch.randn(B, H, W, C, device="cuda")
g32 = torch.randn(B, H, W, C, device="cuda")
x = ops.to_pair(x32) if ops.PAIR else x32.bfloat16()
g = ops.to_pair(g32) if ops.PAIR else g32.bfloat16()
mean = torch.zeros(B, C, device="cuda")
rstd = torch.ones(B, C, device="cuda")
dx = ops.empty_like_act(x)
bytes_x = x32.numel() * (4 if ops.PAIR else 2)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for act in (0, 1):
    t = timeit(lambda: ops.in_bwd_partial(x, g, 0, mean, rstd, act))
    print("%s in_bwd_partial act=%d: %.1f us = %.2f TB/s (reads 2 tensors)" % (mode, act, t, 2 * bytes_x / t / 1e6))
part = ops.in_bwd_partial(x, g, 0, mean, rstd, 1)
t = timeit(lambda: ops.in_bwd_stats(x, g, mean, rstd, 1, dx, part))
print("%s in_bwd_stats (finalize + elementwise pass): %.1f us = %.2f TB/s (3 tensors)" % (mode, t, 3 * bytes_x / t / 1e6))
t = timeit(lambda: ops.in_partial(x))
print("%s in_partial (forward moments): %.1f us = %.2f TB/s (1 tensor)" % (mode, t, bytes_x / t / 1e6))
y = torch.empty_like(x32.bfloat16())
src = x32.bfloat16()
t = timeit(lambda: y.copy_(src))
print("torch copy of one bf16 tensor: %.1f us = %.2f TB/s (read + write)" % (t, 2 * src.numel() * 2 / t / 1e6))
