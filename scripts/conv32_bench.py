"""Micro-benchmark of the 32 -> 32 channel 3x3 reflect conv at [16, 512, 512] (Reg's full-resolution level):
    python scripts/conv32_bench.py [bf16|bf16x3]     (CTG_NO_STRIP=1: the halo-resident kernel instead of the sliding window)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, ops
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype(torch.bfloat16 if mode == "bf16" else mode)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x32 = torch.randn(16, 512, 512, 32, generator=g).to(dev)
x = ops.to_pair(x32) if ops.PAIR else x32.to(torch.bfloat16)
wp = (torch.randn(9, 32, 32, generator=g) * 0.08).to(dev)
if not ops.PAIR:
    wp = wp.to(torch.bfloat16)
taps = [ops.pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
y = ops.empty_act((16, 512, 512, 32), torch.bfloat16, dev)
for stats in (False, True):
    for _ in range(3):
        ops.conv_igemm(x, wp, 32, y, None, 32, 512, 512, 0, 0, 1, 1, ops.PAD_REFLECT, ops.ACT_NONE, taps, want_stats=stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv_igemm(x, wp, 32, y, None, 32, 512, 512, 0, 0, 1, 1, ops.PAD_REFLECT, ops.ACT_NONE, taps, want_stats=stats)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = 2 * 16 * 512 * 512 * 32 * (4 if ops.PAIR else 2)
    print("%s stats=%d: %.1f us per launch, %.2f TB/s algorithmic (CTG_NO_STRIP=%s CTG_STRIP_BAND=%s)" % (
        mode, stats, us, nbytes / us / 1e6, os.environ.get("CTG_NO_STRIP"), os.environ.get("CTG_STRIP_BAND")))
