"""fwd / fwd+bwd time of each network in isolation (B=16, 512^2), HIP events:  python scripts/net_times.py [bf16|bf16x3|fp32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.Model.HdGan import Generator, Discriminator_m
from cta_gan_amd.trainer.reg import Reg
MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype({"bf16": torch.bfloat16, "fp32": torch.float32}.get(MODE, MODE))
B, S = 16, 512
a = synth.synth_images("np_a", B, S).cuda()
b = synth.synth_images("np_b", B, S).cuda()

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for name in ("gen", "reg", "disc"):
    if name == "reg":
        net = Reg(S, S, 1, 1).cuda(); run = lambda: net(a, b)
    elif name == "gen":
        net = Generator(1, 1).cuda(); run = lambda: net(a)
    else:
        net = Discriminator_m(1).cuda(); run = lambda: net(a)[0][-1]
    def fwd():
        with torch.no_grad(): run()
    def fb():
        for p in net.parameters(): p.grad = None
        run().float().sum().backward()
    tf, tfb = timeit(fwd), timeit(fb)
    print(MODE, "%-5s fwd %.2f ms   fwd+bwd %.2f ms   (bwd %.2f)" % (name, tf, tfb, tfb - tf))
