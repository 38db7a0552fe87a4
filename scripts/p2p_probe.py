"""Diagnose the p2p D-step gradient: HIP D vs oracle D on the SAME inputs (smooth pairs, 128^2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from cta_gan_amd import synth
from hip_ns import hip_namespace
from oracle import golden_cases

ns, ons = hip_namespace(), golden_cases.oracle_namespace()
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
A = synth.synth_smooth_images("p2p_A", 2, 128); B = synth.synth_smooth_images("p2p_B", 2, 128)
F = synth.synth_smooth_images("p2p_F", 2, 128) * 0.3
for tag, mk in (("uniform", lambda n: synth.synth_images(n, 2, 128)), ("smooth", lambda n: synth.synth_smooth_images(n, 2, 128))):
    A, B, F = mk("p2p_A"), mk("p2p_B"), mk("p2p_F") * 0.3
    res = {}
    for name, n_, dev in (("hip", ns, "cuda"), ("ora", ons, "cpu")):
        D = synth.fill_module(n_.Discriminator(2), seed=7).to(dev)
        pf = D(torch.cat((A, F), 1).to(dev)); pr = D(torch.cat((A, B), 1).to(dev))
        loss = ((pf - 0.0) ** 2).mean() + ((pr - 1.0) ** 2).mean()
        loss.backward()
        res[name] = (float(loss), {k: (p.grad.detach().float().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)) for k, p in D.named_parameters()},
                     pf.detach().cpu().numpy(), pr.detach().cpu().numpy())
    print(tag, "loss", res["hip"][0], res["ora"][0], "pf", res["hip"][2].ravel(), res["ora"][2].ravel())
    for k in res["hip"][1]:
        print("   %-18s rel-L2 %.3e  |g| %.3e" % (k, rel(res["hip"][1][k], res["ora"][1][k]), np.linalg.norm(res["ora"][1][k])))
