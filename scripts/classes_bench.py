"""The parity-class launches (transposed convs / stride-2 backward-data) of the step, one merged launch each (ctg_conv_igemm_classes),
us per launch:   python scripts/classes_bench.py [bf16|bf16x3]      (CTG_NO_MC4=1: one workgroup per class instead of all classes in one)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import engine as E, ops, nets
from cta_gan_amd.engine import ConvSpec, PackCache
MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype(torch.bfloat16 if MODE == "bf16" else MODE)
dt = torch.bfloat16
B = 16
for (cin, cout, hi, k, pad) in ((128, 64, 256, 3, 1), (256, 128, 128, 3, 1), (128, 64, 128, 4, 1), (256, 128, 64, 4, 1)):
    spec = ConvSpec(cin, cout, k, 2, pad, transposed=True, use_bias=False)
    w = torch.randn(cin, cout, k, k, device="cuda") * 0.05
    x32 = torch.randn(B, hi, hi, cin, device="cuda").relu_()
    x = ops.to_pair(x32) if ops.PAIR else x32.to(dt)
    y = ops.empty_act((B, 2 * hi, 2 * hi, cout), dt, x32.device)
    cache = PackCache()
    wp, npad = E._pack_fwd(cache, spec, w, dt)
    classes = E._convT_classes(k, pad)
    f = lambda: ops.conv_igemm_classes(x, wp, npad, y, None, cout, hi, hi, classes, ops.PAD_ZERO, 0, want_stats=True)
    assert f() is not None
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%s classes %d->%d %dx%d @%d^2: %.1f us  %.0f TF" % (MODE, cin, cout, k, k, hi, us, 2.0 * B * hi * hi * cin * cout * k * k / us / 1e6))
