"""Micro-benchmark of the generator's first down-sampling layer (Conv2d(64, 128, 3, s2): [16,512,512,64] -> [16,256,256,128], bf16,
155 GFLOP, 805 MB) as `ctg_conv_igemm` launches it: the sliding-window kernel of csrc/conv_strips2.h, or (CTG_NO_STRIPS2=1) the
gather kernel conv_igemm_kernel<256,128>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops
B, S = 16, 256
CI, CO = 64, 128
if len(sys.argv) > 1 and sys.argv[1] == "wide":      # d2: Conv2d(128, 256, 3, s2) [16,256,256,128] -> [16,128,128,256] (conv_strips2w.h)
    S, CI, CO = 128, 128, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 2 * S, 2 * S, CI, generator=g).cuda().to(torch.bfloat16)
wp = (torch.randn(9, CO, CI, generator=g) * 0.05).cuda().to(torch.bfloat16)
y = torch.empty(B, S, S, CO, dtype=torch.bfloat16, device="cuda")
taps = [ops.pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
fn = lambda: ops.conv_igemm(x, wp, CO, y, None, CO, S, S, 0, 0, 1, 2, ops.PAD_ZERO, 0, taps, want_stats=True)
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
flop = 2.0 * B * S * S * 9 * CI * CO
nbytes = x.numel() * 2 + y.numel() * 2
print("conv s2 %d->%d @ %d^2 -> %d^2: %.1f us   %.0f TFLOP/s   %.2f TB/s algorithmic" % (CI, CO, 2 * S, S, ms * 1e3, flop / ms / 1e9, nbytes / ms / 1e9))
