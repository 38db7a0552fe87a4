"""Micro-benchmark of the stride-2 convs (G down-sampling d1 / d2, D's 4x4 s2) through ops.conv_igemm, bf16, B=16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import ops
from cta_gan_amd.engine import PAD_ZERO, ACT_NONE, pack_tap
dev = torch.device("cuda:0")
B = 16
for name, cin, cout, k, S in (("d1 64->128 3x3", 64, 128, 3, 512), ("d2 128->256 3x3", 128, 256, 3, 256),
                              ("D 64->128 4x4", 64, 128, 4, 256), ("D 128->256 4x4", 128, 256, 4, 128)):
    x = torch.randn(B, S, S, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    npad = (cout + 127) // 128 * 128
    wp = ops.weight_pack(w, torch.bfloat16, k * k, cout, cin, npad, cin, cin * k * k, k * k, 1)
    ho = (S + 2 - k) // 2 + 1
    y = torch.empty(B, ho, ho, cout, dtype=torch.bfloat16, device=dev)
    taps = [pack_tap(ky - 1, kx - 1, ky * k + kx) for ky in range(k) for kx in range(k)]
    f = lambda: ops.conv_igemm(x, wp, npad, y, None, cout, ho, ho, 0, 0, 1, 2, PAD_ZERO, ACT_NONE, taps)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gf = 2.0 * B * ho * ho * cout * cin * k * k / 1e9
    print("%-16s %7.1f us  %6.0f TF" % (name, us, gf / us * 1e3))
