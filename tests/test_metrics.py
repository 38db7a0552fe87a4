"""Evaluation path: to_windowdata + masks + MAE / PSNR / UQI, and SSIM.

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


def test_oracle_ssim_closed_forms_and_bruteforce():
    """The SSIM restatement (skimage.measure.compare_ssim defaults; no reference fixture exists: "parity unpinned") against what
    can be known without skimage: identical images give exactly 1; two constant images a, b give (2ab + C1) / (a^2 + b^2 + C1);
    and an explicit 49-pixel-sum evaluation agrees to 1e-12 on random images."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (19, 23)).astype(np.float32)
    assert ref_metrics.ssim(x, x) == pytest.approx(1.0, abs=1e-12)
    a, b = np.full((9, 11), 0.25, np.float32), np.full((9, 11), -0.5, np.float32)
    c1 = (0.01 * 2) ** 2
    assert ref_metrics.ssim(a, b) == pytest.approx((2 * 0.25 * -0.5 + c1) / (0.25 ** 2 + 0.5 ** 2 + c1), rel=1e-9)
    y = (x + 0.3 * rng.standard_normal(x.shape)).astype(np.float32)
    assert ref_metrics.ssim(x, y) == pytest.approx(ref_metrics.ssim_bruteforce(x, y), rel=1e-12)
    assert 0.0 < ref_metrics.ssim(x, y) < 1.0
    one = rng.uniform(-1, 1, (7, 7)).astype(np.float32)          # a single window position
    assert ref_metrics.ssim(one, one * 0.5) == pytest.approx(ref_metrics.ssim_bruteforce(one, one * 0.5), rel=1e-12)
    with pytest.raises(ValueError):
        ref_metrics.ssim(one[:6], one[:6])


@pytest.mark.gpu
def test_hip_ssim_matches_the_oracle():
    """ctg_ssim (csrc/metrics.hip) against the oracle: plain pairs on ragged sizes (one window position, sizes that are not
    multiples of the 16x16 tile, 512^2), per-slice results of a batch, and the two masked pairs of the test() loop (both
    aliasing variants).  float64 on both sides; the summation orders differ: 1e-9."""
    from cta_gan_amd import ops, synth
    rng = np.random.default_rng(7)
    for (h, w) in ((7, 7), (7, 40), (23, 38), (64, 64), (130, 97)):
        x = rng.uniform(-1, 1, (2, 1, h, w)).astype(np.float32)
        y = (x + 0.2 * rng.standard_normal(x.shape)).astype(np.float32)
        got = ops.ssim(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()).cpu().numpy()
        for i in range(2):
            assert got[i] == pytest.approx(ref_metrics.ssim(x[i, 0], y[i, 0]), rel=1e-9, abs=1e-12), (h, w, i)
    same = ops.ssim(torch.from_numpy(x).cuda(), torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.allclose(same, 1.0, atol=1e-12)
    with pytest.raises(ValueError):
        ops.ssim(torch.zeros(1, 6, 9).cuda(), torch.zeros(1, 6, 9).cuda())
    b, s = 3, 512
    real = synth.synth_smooth_images("mt_real", b, s)
    fake = (real + 0.1 * synth.synth_smooth_images("mt_noise", b, s)).clamp(-1, 1)
    real[:, :, :20] = -1
    wc, ww = [40.0, 60.0, 300.0], [400.0, 300.0, 1500.0]
    plain = ops.ssim(fake.cuda(), real.cuda()).cpu().numpy()
    for aliased in (False, True):
        got = ops.window_ssim(fake.cuda(), real.cuda(), wc, ww, aliased=aliased).cpu().numpy()
        for i in range(b):
            want = ref_metrics.slice_ssim(fake[i, 0].numpy().copy(), real[i, 0].numpy().copy(), wc[i], ww[i], aliased=aliased)
            assert np.allclose(got[i], want, rtol=1e-9, atol=1e-12), (aliased, i, got[i], want)
    for i in range(b):
        assert plain[i] == pytest.approx(ref_metrics.ssim(fake[i, 0].numpy(), real[i, 0].numpy()), rel=1e-9)
    # the trainers' PSNR(fake, real) of the validation pass
    psnr = ops.val_psnr(fake.cuda(), real.cuda()).cpu().numpy()
    for i in range(b):
        assert psnr[i] == pytest.approx(ref_metrics.psnr(fake[i, 0].numpy(), real[i, 0].numpy()), rel=1e-5)


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
    acc_ssim = np.zeros(2)
    with torch.no_grad():
        for bt in batches:
            fk = tr.netG_A2B(bt["A2"].cuda()).float().cpu().numpy()
            for i in range(2):
                acc += ref_metrics.slice_metrics(fk[i, 0].copy(), bt["B2"][i, 0].numpy().copy(), 40.0, 400.0)
                acc_ssim += ref_metrics.slice_ssim(fk[i, 0].copy(), bt["B2"][i, 0].numpy().copy(), 40.0, 400.0)
    acc /= 4
    acc_ssim /= 4
    got = np.array([[out["MAEw"], out["PSNRw"], out["UQIw"]], [out["MAE"], out["PSNR"], out["UQI"]]])
    assert _close(got, acc, rel=2e-5), (got, acc)
    assert np.allclose([out["SSIMw"], out["SSIM"]], acc_ssim, rtol=1e-8, atol=1e-10), (out["SSIMw"], out["SSIM"], acc_ssim)
