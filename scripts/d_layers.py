"""Per-launch table of the PatchGAN discriminator alone (B=16, 512^2; forward + backward with weight gradients), HIP events around
every conv-class launch (cta_gan_amd.ops.OP_LOG):   python scripts/d_layers.py [bf16|bf16x3]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cta_gan_amd import nets, ops, synth
from cta_gan_amd.Model.HdGan import Discriminator_m
MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype(torch.bfloat16 if MODE == "bf16" else MODE)
peak = bench.PEAK_TFLOPS[MODE] * 1e12
torch.manual_seed(1)
net = Discriminator_m(1).cuda()
net.patch_only = True
a = synth.synth_images("np_a", 16, 512).cuda().requires_grad_(True)


def step():
    for p in net.parameters():
        p.grad = None
    a.grad = None
    net(a)[0][-1].float().sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    step()
e1.record()
torch.cuda.synchronize()
print("%s D fwd+bwd %.3f ms" % (MODE, e0.elapsed_time(e1) / 5))
ops.OP_LOG = []
for _ in range(3):
    step()
torch.cuda.synchronize()
log, ops.OP_LOG = ops.OP_LOG, None
agg = collections.OrderedDict()
for label, flop, nbytes, b0, b1 in log:
    r = agg.setdefault(label, [0, 0.0, flop, nbytes])
    r[0] += 1
    r[1] += b0.elapsed_time(b1)
for label, (n, ms, flop, nbytes) in agg.items():
    us = 1e3 * ms / n
    frac = max(flop / peak, nbytes / 8e12) / (us * 1e-6)
    print("%-62s x%d %7.1f us %6.0f TF %5.2f TB/s  %.2f" % (label, n // 3, us, flop / us / 1e6, nbytes / us / 1e6, frac))
