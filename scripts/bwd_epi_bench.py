"""What the fused epilogue of the residual blocks' backward-data launch costs (256 -> 256, 3x3, B=16, 128^2): the plain conv, + skip
gradient (res), + frame fold, + InstanceNorm-backward sums -- us per launch, HIP events.   python scripts/bwd_epi_bench.py [bf16|bf16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, ops
from cta_gan_amd.engine import PAD_ZERO, ACT_NONE, ACT_RELU, pack_tap
MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype(torch.bfloat16 if MODE == "bf16" else MODE)
dev = torch.device("cuda:0")
B, S, C = 16, 128, 256
pdt = torch.float32 if ops.PAIR else torch.bfloat16


def act(*shape):
    t = torch.randn(*shape, device=dev).relu_()
    return ops.to_pair(t) if ops.PAIR else t.bfloat16()


x = act(B, S, S, C)
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
wp = ops.weight_pack(w, pdt, 9, C, C, C, C, C * 9, 9, 1)
y = ops.empty_act((B, S, S, C), torch.bfloat16, dev)
res, z = act(B, S, S, C), act(B, S, S, C)
fold = act(B, S + 2, S + 2, C)
mean, rstd = torch.zeros(B, C, device=dev), torch.ones(B, C, device=dev)
taps = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
variants = {
    "plain": dict(),
    "+stats (forward)": dict(want_stats=True),
    "+res": dict(res=res),
    "+res+fold": dict(res=res, fold=fold),
    "+res+INsums": dict(res=res, in_bwd=(z, mean, rstd, ACT_RELU)),
    "+res+fold+INsums": dict(res=res, fold=fold, in_bwd=(z, mean, rstd, ACT_RELU)),
}
for name, kw in list(variants.items()) * 2:      # two passes: the first variant of a cold chip reads long
    f = lambda: ops.conv_igemm(x, wp, C, y, None, C, S, S, 0, 0, 1, 1, PAD_ZERO, ACT_NONE, taps, **kw)
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
    print("%s %-20s %7.1f us  %5.0f TF" % (MODE, name, us, 2.0 * B * S * S * C * C * 9 / us / 1e6))
