"""Micro-benchmark of the generator's stride-2 layers in the split-bf16 mode (B=16): the sliding-window kernels of
csrc/conv_pair_strips.hip against the halo-kernel forms they replace (CTG_NO_STRIPTP=1, ...).

    python scripts/pair_strips_bench.py [t|s2|s2w ...]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, ops
from cta_gan_amd.engine import _convT_classes
from cta_gan_amd.ops import pack_tap

nets.set_default_compute_dtype("bf16x3")
B = int(os.environ.get("B", "16"))
which = sys.argv[1:] or ["t", "s2", "s2w", "tw"]
g = torch.Generator().manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def report(name, ms, flop, nbytes):
    print("%-34s %7.1f us   %5.0f TFLOP/s (x3 executed: %5.0f)   %.2f TB/s algorithmic" % (
        name, ms * 1e3, flop / ms / 1e9, 3 * flop / ms / 1e9, nbytes / ms / 1e9), flush=True)


taps33 = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
for k in which:
    if k in ("t", "tw"):
        cin, cout, S = (128, 64, 256) if k == "t" else (256, 128, 128)
        x = ops.to_pair(torch.randn(B, S, S, cin, generator=g).cuda().relu_())
        wp = (torch.randn(9, max(cout, 64), cin, generator=g) * 0.05).cuda()
        y = ops.empty_act((B, 2 * S, 2 * S, cout), torch.bfloat16, x.device)
        classes = _convT_classes(3, 1)
        ms = timed(lambda: ops.conv_igemm_classes(x, wp, wp.shape[1], y, None, cout, S, S, classes, ops.PAD_ZERO, 0, want_stats=True))
        report("convT %d->%d @%d^2 -> %d^2" % (cin, cout, S, 2 * S), ms, 2.0 * B * S * S * 9 * cin * cout, (x.numel() + y.numel()) * 4)
    else:
        cin, cout, S = (64, 128, 256) if k == "s2" else (128, 256, 128)
        x = ops.to_pair(torch.randn(B, 2 * S, 2 * S, cin, generator=g).cuda().relu_())
        wp = (torch.randn(9, cout, cin, generator=g) * 0.05).cuda()
        y = ops.empty_act((B, S, S, cout), torch.bfloat16, x.device)
        ms = timed(lambda: ops.conv_igemm(x, wp, cout, y, None, cout, S, S, 0, 0, 1, 2, ops.PAD_ZERO, 0, taps33, want_stats=True))
        report("conv s2 %d->%d @%d^2 -> %d^2" % (cin, cout, 2 * S, S), ms, 2.0 * B * S * S * 9 * cin * cout, (x.numel() + y.numel()) * 4)
