"""Gradient accuracy of the golden cases in a compute mode, reported (not asserted): rel-L2 of every array and relative deviation of
every gradient norm against the reference-generated goldens.
    python scripts/experiments/r06_x3f_goldens.py MODE     (bf16x3 | bf16x3f | bf16 | fp32)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cta_gan_amd import nets  # noqa: E402
from hip_ns import hip_namespace  # noqa: E402
from oracle import golden_cases  # noqa: E402
import test_parity_gpu as P  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3f"
nets.set_default_compute_dtype({"fp32": torch.float32, "bf16": torch.bfloat16}.get(mode, mode))
print("mode", nets.compute_mode())
for name in ["generator_64", "resblock_256x12", "discriminator_64", "discriminator_m2_128", "nlayer_d_bn_64", "reg_256"]:
    want = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    got = golden_cases.CASES[name](hip_namespace())
    rep = {}
    for key in want.files:
        w, g = want[key], got[key]
        if w.dtype.kind in "US" or key.startswith("shape_"):
            continue
        if key == "gradnorm_vals":
            keys = [str(k) for k in want["gradnorm_keys"]]
            dev = [(abs(gv - wv) / (abs(wv) + 1e-12), k) for k, gv, wv in zip(keys, np.asarray(g), w) if not P._is_dead_bias(k)]
            rep["gradnorm_worst"] = "%.2e (%s)" % max(dev)
            continue
        if np.ndim(w) == 0:
            rep[key] = "%.2e" % (abs(float(g) - float(w)) / max(abs(float(w)), 1e-6))
        elif "stats" not in key:
            rep[key] = "%.2e" % P.rel_l2(g, w)
    print(name, rep)
