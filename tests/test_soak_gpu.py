"""GPU, slow (about a minute): does the benchmarked precision TRAIN like the reference's fp32?

Single steps of a ReLU network under Adam are chaotic (tests/test_step_parity_gpu.py bounds them per step); windowed medians of
the loss terms over a longer run are not.  `scripts/precision_soak.py` takes the same 120 HdGan stage-2 steps (B=4, 256^2, the
same batches and initial weights) in fp32, bf16x3, bf16x3f and bf16 and asserts, for every 20-step window, that the medians of SR and of
the total loss of the three bf16-family modes stay within 15 % of the fp32 run (and that every loss of every step is finite)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_bf16_family_modes_train_like_fp32_over_120_steps():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "precision_soak.py"), "120"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "precision soak ok: 120 steps" in r.stdout, r.stdout[-3000:]
    print(r.stdout)
