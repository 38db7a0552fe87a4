"""ORACLE (test infrastructure only -- never imported by the product path): numpy restatement of the evaluation
arithmetic of the reference's test() loop.

  to_windowdata      trainer/HdTrainer.py:41-64 (== trainer/CycTrainer.py:34-57)
  slice_metrics      trainer/HdTrainer.py:1008-1050 (masking) with MAE/PSNR/UQI of :1089-1125
Pinned by tests/golden/metrics_*.npz, produced by oracle/make_golden_metrics.py from the reference's own functions.
SSIM (skimage) and LPIPS (lpips) of the same loop are outside this build (dependencies absent from the image).
"""
import numpy as np


def to_windowdata(image, WC, WW):
    """HdTrainer.py:41-64: [-1,1] -> HU -> window (WC, WW) -> 8-bit levels -> [-1,1]; float32 throughout."""
    image = (image + 1) * 0.5 * 4095
    image[image == 0] = -2000
    image = image - 1024
    win_min = (2 * WC - WW) / 2.0 + 0.5
    win_max = (2 * WC + WW) / 2.0 + 0.5
    d_factor = 255.0 / (win_max - win_min)
    image = image - win_min
    image = np.trunc(image * d_factor)
    image[image > 255] = 255
    image[image < 0] = 0
    image = image / 255
    return (image - 0.5) / 0.5


def psnr(fake, real):
    """HdTrainer.py:1089-1104 (background = pixels of `real` equal to -1)."""
    m = real != -1
    if not m.any():
        mse = np.mean(((fake + 1) / 2. - (real + 1) / 2.) ** 2) + 1e-10
    else:
        mse = np.mean(((fake[m] + 1) / 2. - (real[m] + 1) / 2.) ** 2)
    if mse < 1.0e-10:
        return 100
    return 20 * np.log10(1 / (np.sqrt(mse) + 1e-10))


def mae(fake, real):
    """HdTrainer.py:1106-1117."""
    m = real != -1
    if not m.any():
        v = np.nanmean(np.abs(fake - real)) + 1e-10
    else:
        v = np.nanmean(np.abs(fake[m] - real[m]))
    return v / 2


def uqi(fake, real):
    """HdTrainer.py:1119-1125."""
    meanf, meanr = np.mean(fake), np.mean(real)
    m, n = np.shape(fake)
    varf = np.sqrt(np.sum((fake - meanf) ** 2) / (m * n - 1))
    varr = np.sqrt(np.sum((real - meanr) ** 2) / (m * n - 1))
    cov = np.sum((fake - meanf) * (real - meanr)) / (m * n - 1)
    return 4 * meanf * meanr * cov / ((meanf ** 2 + meanr ** 2) * (varf ** 2 + varr ** 2) + 1e-10)


def slice_metrics(fake_B, real_B, WC, WW, fns=None):
    """One iteration of the test loop, HdTrainer.py:1008-1050, on 2-D float32 arrays: returns
    [[MAEw, PSNRw, UQIw], [MAE, PSNR, UQI]].  `fns` lets the golden generator plug the reference's own functions."""
    win, f_mae, f_psnr, f_uqi = fns or (to_windowdata, mae, psnr, uqi)
    b = win(real_B, WC, WW)
    bb = b.copy()
    bb[bb < 0.3] = 0
    bb[bb >= 0.3] = 1
    b = b * bb
    b[b == 0] = -1
    c = win(fake_B, WC, WW) * bb
    cc = c.copy()
    cc[cc < 0.3] = 0
    cc[cc >= 0.3] = 1
    c = c * cc
    c[c == 0] = -1
    windowed = [f_mae(c, b), f_psnr(c, b), f_uqi(c, b)]
    real_m = real_B * bb
    real_m[real_m == 0] = -1
    fake_m = fake_B * cc
    fake_m[fake_m == 0] = -1
    raw = [f_mae(fake_m, real_m), f_psnr(fake_m, real_m), f_uqi(fake_m, real_m)]
    return np.array([windowed, raw], dtype=np.float64)


def slice_metrics_cyc(fake_B, real_B, WC, WW, fns=None):
    """One iteration of trainer/CycTrainer.py:284-331: as slice_metrics, but `bb = b` and `cc = c` are ALIASES there
    (no copy), so thresholding the masks in place also turns b and c into the masks."""
    win, f_mae, f_psnr, f_uqi = fns or (to_windowdata, mae, psnr, uqi)
    b = win(real_B, WC, WW)
    bb = b
    bb[bb < 0.3] = 0
    bb[bb >= 0.3] = 1
    b = b * bb
    b[b == 0] = -1
    c = win(fake_B, WC, WW) * bb
    cc = c
    cc[cc < 0.3] = 0
    cc[cc >= 0.3] = 1
    c = c * cc
    c[c == 0] = -1
    windowed = [f_mae(c, b), f_psnr(c, b), f_uqi(c, b)]
    real_m = real_B * bb
    real_m[real_m == 0] = -1
    fake_m = fake_B * cc
    fake_m[fake_m == 0] = -1
    raw = [f_mae(fake_m, real_m), f_psnr(fake_m, real_m), f_uqi(fake_m, real_m)]
    return np.array([windowed, raw], dtype=np.float64)
