import sys, collections
sys.path.insert(0, "/root/repo")
import torch
from cta_gan_amd import nets, ops, synth
from cta_gan_amd.Model.HdGan import Generator
for mode in ("bf16x3", torch.bfloat16):
    nets.set_default_compute_dtype(mode)
    g = synth.fill_module(Generator(1, 1), seed=0).cuda()
    x = synth.synth_images("p", 16, 512).cuda()
    with torch.no_grad():
        g(x)
    ops.OP_LOG = []
    with torch.no_grad():
        g(x)
    torch.cuda.synchronize()
    c = collections.Counter(r[0] for r in ops.OP_LOG)
    ops.OP_LOG = None
    print(mode, dict(c))
    print("refused", ops._NIE_REFUSED)
