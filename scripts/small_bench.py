"""Micro-benchmark of ctg_conv_smallcin on the Generator head (1 -> 64, 7x7 reflect, B=16, 512^2, bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops
from cta_gan_amd.engine import PAD_REFLECT, ACT_NONE
B, S = 16, 512
dev = torch.device("cuda:0")
s0 = torch.randn(B, S, S, device=dev)
w = torch.randn(64, 1, 7, 7, device=dev)
wp = ops.weight_pack(w, torch.bfloat16, 1, 64, 49, 64, 64, 49, 1, 0)
y = torch.empty(B, S, S, 64, dtype=torch.bfloat16, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    ops.conv_smallcin(s0, None, 7, 1, 3, PAD_REFLECT, wp, 64, None, ACT_NONE, y, 64, want_stats=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.conv_smallcin(s0, None, 7, 1, 3, PAD_REFLECT, wp, 64, None, ACT_NONE, y, 64, want_stats=True)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print("conv_small head: %.1f us  -> %.2f TB/s of output" % (us, y.numel() * 2 / us / 1e6))
