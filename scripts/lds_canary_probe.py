"""Run the LDS canary on a second stream beside one conv layer on the main stream and print what it caught.
    python scripts/lds_canary_probe.py probe_<cin>_<cout>_<k>_<size>[_f32out][_fwd]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cta_gan_amd import nets, ops
from cta_gan_amd.engine import ConvSpec
import test_kernels_gpu as K
nets.set_default_compute_dtype(torch.bfloat16)
what = sys.argv[1]
parts = what.split("_")
cin, cout, k, size = int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4])
spec = ConvSpec(cin, cout, k, 1, (k - 1) // 2, use_bias=True, out_f32="f32out" in parts)
net = K._make_probe(spec, None).cuda()
x = torch.randn(16, cin, size, size, device="cuda").requires_grad_(True)
side = torch.cuda.Stream()
for _ in range(2):
    with torch.no_grad():
        net(x)
torch.cuda.synchronize()
rep = ops.lds_canary(blocks=512, spins=12000, cap=48, stream=side)
for _ in range(60):
    if "fwd" in parts:
        with torch.no_grad():
            net(x)
    else:
        net(x).float().mean().backward()
torch.cuda.synchronize()
r = rep.cpu().view(torch.int32)
n = int(r[0])
print(what, "foreign LDS writes seen:", n, "| broadcast-read mismatches (lane-events):", int(r[1]))
ev = r[4:4 + 4 * min(max(n, int(r[1])), 48)].view(-1, 4).tolist()
for (i, v, s, b) in ev[:24]:
    print("   word %4d (byte %5d)  value 0x%08x  spin %d  workgroup %d" % (i, 4 * i, v & 0xffffffff, s, b))
