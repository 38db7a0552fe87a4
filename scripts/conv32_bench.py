"""Micro-benchmark of the narrow full-resolution convs of Reg (32 -> 32 channels, 3x3 reflect, [16, 512, 512, 32] bf16: 537 MB per
launch): forward with InstanceNorm moments and the fused backward-data form.  `both`: with the persistent weights-resident kernel of
scripts/experiments/r03_persistent_narrow_conv.patch applied, runs it and (CTG_NO_PW=1) the regular halo kernel in turn.
Also the target of --pmc runs (scripts/pmc_conv32.sh)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "both":
    for env in ({}, {"CTG_NO_PW": "1"}):
        print("---", env or "default (persistent)", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[2:], env=dict(os.environ, **env), check=True)
    sys.exit(0)
import torch
from cta_gan_amd import ops
B, S, C = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(B, S, S, C, generator=g).to(dev).to(torch.bfloat16).relu_()
wp = (torch.randn(9, C, C, generator=g) * 0.05).to(dev).to(torch.bfloat16)
y = torch.empty(B, S, S, C, dtype=torch.bfloat16, device=dev)
res = torch.randn(B, S, S, C, generator=g).to(dev).to(torch.bfloat16)
fold = torch.randn(B, S + 2, S + 2, C, generator=g).to(dev).to(torch.bfloat16)
mean, rstd = ops.in_stats(x)
taps = [ops.pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


nb = x.numel() * 2
for name, fn, passes in (
        ("fwd + IN moments", lambda: ops.conv_igemm(x, wp, C, y, None, C, S, S, 0, 0, 1, 1, ops.PAD_REFLECT, 0, taps, want_stats=True), 2),
        ("bwd-data FUSE (res + fold + IN-bwd sums)", lambda: ops.conv_igemm(x, wp, C, y, None, C, S, S, 0, 0, 1, 1, ops.PAD_ZERO, 0, taps,
                                                                            res=res, fold=fold, in_bwd=(x, mean, rstd, 1)), 4)):
    ms = t(fn)
    print("%-44s %7.1f us   %6.2f TB/s algorithmic (%d tensor passes)" % (name, ms * 1e3, passes * nb / ms / 1e9, passes))
