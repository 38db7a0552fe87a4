"""Micro-benchmark of the generator's second up-sampling layer (ConvTranspose2d(128, 64, 3, s2): [16,256,256,128] -> [16,512,512,64], bf16,
155 GFLOP, 805 MB) as `ctg_conv_igemm_classes` launches it: the sliding-window kernel of csrc/conv_stript.h, or (CTG_NO_STRIPT=1) the
merged parity classes on conv_halo_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops
from cta_gan_amd.engine import _convT_classes
B, S = 16, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, S, S, 128, generator=g).cuda().to(torch.bfloat16).relu_()
wp = (torch.randn(9, 64, 128, generator=g) * 0.05).cuda().to(torch.bfloat16)
y = torch.empty(B, 2 * S, 2 * S, 64, dtype=torch.bfloat16, device="cuda")
classes = _convT_classes(3, 1)
fn = lambda: ops.conv_igemm_classes(x, wp, 64, y, None, 64, S, S, classes, ops.PAD_ZERO, 0, want_stats=True)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 20
for _ in range(n):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flop = 2.0 * B * S * S * 9 * 128 * 64
nbytes = x.numel() * 2 + y.numel() * 2
print("convT 128->64 @ %d^2 -> %d^2: %.1f us   %.0f TFLOP/s   %.2f TB/s algorithmic" % (S, 2 * S, ms * 1e3, flop / ms / 1e9, nbytes / ms / 1e9))
