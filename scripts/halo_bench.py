"""Micro-benchmark of the dominant kernel: 256->256 3x3 reflect conv, B=16, 128x128, bf16 (for --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops
from cta_gan_amd.engine import PAD_REFLECT, ACT_NONE, pack_tap
dev = torch.device("cuda:0")
B, S, C = 16, 128, 256
x = torch.randn(B, S, S, C, device=dev).bfloat16()
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
wp = ops.weight_pack(w, torch.bfloat16, 9, C, C, C, C, C * 9, 9, 1)
y = torch.empty(B, S, S, C, dtype=torch.bfloat16, device=dev)
taps = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
f = lambda: ops.conv_igemm(x, wp, C, y, None, C, S, S, 0, 0, 1, 1, PAD_REFLECT, ACT_NONE, taps, want_stats=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print("conv 256->256 3x3: %.1f us  %.0f TF" % (us, 2.0 * B * S * S * C * C * 9 / us / 1e6))
