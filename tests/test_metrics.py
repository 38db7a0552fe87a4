"""Evaluation path: to_windowdata + masks + MAE / PSNR / UQI.

CPU: the numpy oracle (oracle/ref_metrics.py) against fixtures produced by the reference's own functions
(oracle/make_golden_metrics.py).  GPU: the HIP kernels (csrc/metrics.hip) through the C ABI against the same fixtures
and against the oracle on larger seeded slices.  Tolerances: windowed image bit-exact (float32 op-for-op); metrics
1e-5 relative (the reference reduces in float32 pairwise sums, the HIP path in fp64)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import ref_metrics

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "metrics_*.npz")))


def _close(a, b, rel=1e-5, abs_=1e-9):
    return np.all(np.abs(np.asarray(a) - np.asarray(b)) <= rel * np.abs(np.asarray(b)) + abs_)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_reference_fixtures(path):
    z = np.load(path)
    assert len(GOLD) >= 4
    win = ref_metrics.to_windowdata(z["real"].copy(), float(z["wc"]), float(z["ww"]))
    assert np.array_equal(win.astype(np.float32), z["win_real"])
    got = ref_metrics.slice_metrics(z["fake"].copy(), z["real"].copy(), float(z["wc"]), float(z["ww"]))
    assert _close(got, z["metrics"], rel=1e-6), (got, z["metrics"])
    got = ref_metrics.slice_metrics_cyc(z["fake"].copy(), z["real"].copy(), float(z["wc"]), float(z["ww"]))
    assert _close(got, z["metrics_cyc"], rel=1e-6), (got, z["metrics_cyc"])


@pytest.mark.gpu
def test_hip_window_and_metrics_match_reference_fixtures():
    from cta_gan_amd import ops
    for path in GOLD:
        z = np.load(path)
        fake = torch.from_numpy(z["fake"]).cuda()[None, None]
        real = torch.from_numpy(z["real"]).cuda()[None, None]
        wc, ww = float(z["wc"]), float(z["ww"])
        win = ops.to_windowdata(real, wc, ww)[0, 0].cpu().numpy()
        assert np.array_equal(win, z["win_real"]), path          # bit-exact: masks hang on exact comparisons
        got = ops.window_metrics(fake, real, wc, ww)[0].cpu().numpy()
        assert _close(got, z["metrics"]), (path, got, z["metrics"])
        got = ops.window_metrics(fake, real, wc, ww, aliased=True)[0].cpu().numpy()     # CycTrainer.py variant
        assert _close(got, z["metrics_cyc"]), (path, got, z["metrics_cyc"])


@pytest.mark.gpu
def test_hip_metrics_batch_vs_oracle_512():
    """A batch of full-size slices with per-slice windows against the numpy oracle."""
    from cta_gan_amd import ops, synth
    b, s = 3, 512
    real = synth.synth_smooth_images("mt_real", b, s)
    fake = (real + 0.1 * synth.synth_smooth_images("mt_noise", b, s)).clamp(-1, 1)
    real[:, :, :20] = -1
    wc = [40.0, 60.0, 300.0]
    ww = [400.0, 300.0, 1500.0]
    got = ops.window_metrics(fake.cuda(), real.cuda(), wc, ww).cpu().numpy()
    for i in range(b):
        want = ref_metrics.slice_metrics(fake[i, 0].numpy().copy(), real[i, 0].numpy().copy(), wc[i], ww[i])
        assert _close(got[i], want, rel=2e-5), (i, got[i], want)
    win = ops.to_windowdata(real.cuda(), wc, ww).cpu().numpy()
    for i in range(b):
        assert np.array_equal(win[i, 0], ref_metrics.to_windowdata(real[i, 0].numpy().copy(), wc[i], ww[i]).astype(np.float32))


@pytest.mark.gpu
def test_trainer_test_loop_reports_metrics():
    """trainer.test(): generator inference + device metrics over a small dataloader equals per-slice oracle numbers
    computed from the same generator outputs."""
    from cta_gan_amd import synth
    from cta_gan_amd.trainer import Hd_Trainer_x2
    cfg = dict(input_nc=1, output_nc=1, size=64, batchSize=2, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
               Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, WC=40.0, WW=400.0)
    tr = Hd_Trainer_x2.__new__(Hd_Trainer_x2)
    # only the generator is needed for test(); build it without Reg (which needs >= 256)
    from cta_gan_amd.Model.HdGan import Generator
    tr.config, tr.device = cfg, torch.device("cuda:0")
    tr.netG_A2B = Generator(1, 1).cuda()
    synth.fill_module(tr.netG_A2B, seed=0)
    batches = [{"A2": synth.synth_smooth_images("tt_a%d" % i, 2, 64), "B2": synth.synth_smooth_images("tt_b%d" % i, 2, 64)}
               for i in range(2)]
    out = tr.test(batches)
    assert out["num"] == 4
    acc = np.zeros((2, 3))
    with torch.no_grad():
        for bt in batches:
            fk = tr.netG_A2B(bt["A2"].cuda()).float().cpu().numpy()
            for i in range(2):
                acc += ref_metrics.slice_metrics(fk[i, 0].copy(), bt["B2"][i, 0].numpy().copy(), 40.0, 400.0)
    acc /= 4
    got = np.array([[out["MAEw"], out["PSNRw"], out["UQIw"]], [out["MAE"], out["PSNR"], out["UQI"]]])
    assert _close(got, acc, rel=2e-5), (got, acc)
