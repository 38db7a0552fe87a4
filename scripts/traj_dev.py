"""Per-step deviation of the 5-step Hd trajectory (tests/golden/hd_traj5_stage2_256.npz) in every compute mode:
    python scripts/traj_dev.py [modes...]      (default: fp32 bf16x3)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_step_parity_gpu as S
from cta_gan_amd import nets
want = np.load(os.path.join(ROOT, "tests", "golden", "hd_traj5_stage2_256.npz"))
for mode in (sys.argv[1:] or ["fp32", "bf16x3"]):
    nets.set_default_compute_dtype(torch.float32 if mode == "fp32" else mode)
    tr = S.make_hd()
    rows = []
    for i in range(5):
        losses = tr.train_step(S.hd_batch("traj%d_" % i), sync_losses=True)
        dev = {k: abs(losses[k] - float(want["losses"][i, j])) / max(abs(float(want["losses"][i, j])), 1e-6) for j, k in enumerate(S.HD_KEYS)}
        rows.append(max(dev.values()))
        worst = max(dev, key=dev.get)
        print(mode, "step", i, "max rel dev %.2e (%s)" % (rows[-1], worst), {k: "%.1e" % v for k, v in dev.items()}, flush=True)
