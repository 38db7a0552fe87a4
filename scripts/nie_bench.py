"""Micro-benchmark: one residual block's no-grad forward (conv + InstanceNorm fused into the conv launch vs three launches),
B=16, 256 channels, 128x128:   python scripts/nie_bench.py [bf16|bf16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, ops, synth
from cta_gan_amd.Model.HdGan import ResidualBlock
MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype(torch.bfloat16 if MODE == "bf16" else MODE)
blk = synth.fill_module(ResidualBlock(256), seed=1).cuda()
x = torch.randn(16, 256, 128, 128, device="cuda")
ops.NIE_MAX_WGS = ops.NIE_MAX_WGS_PAIR = 1 << 20      # the policy limit (ops.conv_in_fusable) lifted: this script measures what it is based on


def run(n=20):
    with torch.no_grad():
        for _ in range(3):
            blk(x)
        torch.cuda.synchronize()
        ops.OP_LOG = []
        for _ in range(n):
            blk(x)
        torch.cuda.synchronize()
    log, ops.OP_LOG = ops.OP_LOG, None
    agg = {}
    for label, flop, nbytes, e0, e1 in log:
        a = agg.setdefault(label, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    return {k: 1e3 * v[1] / v[0] for k, v in agg.items()}


for name, flag in (("fused", False), ("unfused", True)):
    ops._NO_NIE = flag
    print(MODE, name, {k: "%.1f us" % v for k, v in run().items()})
print("nie failures", ops.nie_failures())
