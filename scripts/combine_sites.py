"""Which call sites still launch grad_combine / fold in one fwd+bwd of a network, and on how many bytes."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth, ops
from cta_gan_amd.Model.HdGan import Generator, Discriminator_m
from cta_gan_amd.trainer.reg import Reg
nets.set_default_compute_dtype(torch.bfloat16)
which = sys.argv[1]
B, S = 16, 512
a = synth.synth_images("np_a", B, S).cuda().requires_grad_(True)
b = synth.synth_images("np_b", B, S).cuda()
if which == "reg":
    net = Reg(S, S, 1, 1).cuda(); run = lambda: net(a, b)
elif which == "gen":
    net = Generator(1, 1).cuda(); run = lambda: net(a)
else:
    net = Discriminator_m(1).cuda(); run = lambda: net(a)[0][-1]
sites = collections.Counter(); vol = collections.Counter()
orig = ops.grad_combine
def spy(*args, **kw):
    st = traceback.extract_stack(limit=4)
    key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[:-1]))
    out = args[-1]
    sites[key] += 1; vol[key] += out.numel() * out.element_size()
    return orig(*args, **kw)
ops.grad_combine = spy
out = run(); out.float().sum().backward(); torch.cuda.synchronize()
for k, n in sites.most_common():
    print("%3d  %7.1f MB out  %s" % (n, vol[k] / 1e6, k))
