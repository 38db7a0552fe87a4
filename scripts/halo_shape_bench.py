"""Micro-benchmark of one stride-1 full-window conv launch on any shape, bf16 or split pair ("x3"):
    python scripts/halo_shape_bench.py MODE CIN COUT K SIZE [B] [PADMODE reflect|zero] [realistic]
prints us per launch, effective TFLOP/s and us per (channel slice, tap) step of a workgroup round."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops, nets
from cta_gan_amd.engine import PAD_REFLECT, PAD_ZERO, ACT_NONE, pack_tap

mode, cin, cout, k, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 16
padm = PAD_REFLECT if (len(sys.argv) > 7 and sys.argv[7] == "reflect") else PAD_ZERO
realistic = len(sys.argv) > 8
dev = torch.device("cuda:0")
if mode == "x3":
    nets.set_default_compute_dtype("bf16x3")
pad = (k - 1) // 2
ho = size + 2 * pad - k + 1
x32 = torch.randn(B, size, size, cin, device=dev)
if realistic:
    x32 = torch.relu(x32)
w = torch.randn(cout, cin, k, k, device=dev) * 0.02
if mode == "x3":
    x = ops.to_pair(x32)
    wp = ops.weight_pack(w, torch.float32, k * k, cout, cin, cout, cin, cin * k * k, k * k, 1)
    y = ops.empty_act((B, ho, ho, cout), torch.bfloat16, dev)
else:
    x = x32.bfloat16()
    wp = ops.weight_pack(w, torch.bfloat16, k * k, cout, cin, cout, cin, cin * k * k, k * k, 1)
    y = torch.empty(B, ho, ho, cout, dtype=torch.bfloat16, device=dev)
taps = [pack_tap(ky - pad, kx - pad, ky * k + kx) for ky in range(k) for kx in range(k)]
f = lambda: ops.conv_igemm(x, wp, cout, y, None, cout, ho, ho, 0, 0, 1, 1, padm, ACT_NONE, taps, want_stats=True)
n = 20
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    f()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
wgs = B * ((ho + 15) // 16) ** 2 * ((cout + 127) // 128)
rounds = (wgs + 511) // 512
steps = k * k * (cin // 32 if mode == "x3" else cin // 64)
print("%s %d->%d %dx%d @%d B=%d: %.1f us  %.0f TF  %d workgroups = %.2f rounds of 512, %d steps: %.2f us / step"
      % (mode, cin, cout, k, k, size, B, us, 2.0 * B * ho * ho * cin * cout * k * k / us / 1e6, wgs, wgs / 512.0, steps,
         us / max(rounds, 1) / steps))
