"""ORACLE (test infrastructure only -- never imported by the product path): numpy restatement of the evaluation
arithmetic of the reference's test() loop.

  to_windowdata      trainer/HdTrainer.py:41-64 (== trainer/CycTrainer.py:34-57)
  slice_metrics      trainer/HdTrainer.py:1008-1050 (masking) with MAE/PSNR/UQI of :1089-1125
Pinned by tests/golden/metrics_*.npz, produced by oracle/make_golden_metrics.py from the reference's own functions.

  ssim / slice_ssim  `skimage.measure.compare_ssim(x, y)` as the reference calls it (HdTrainer.py:256, 779, 1028, 1053;
                     CycTrainer.py:216; p2pTrainer.py:164; RegTrainer.py:219).  scikit-image is a third-party dependency that is
                     neither vendored in /root/reference nor installed here, and the reference pins no version (the name
                     `compare_ssim` exists in scikit-image 0.12-0.17); this restates that function's published algorithm with its
                     defaults.  PARITY UNPINNED: the reference holds no SSIM fixture and cannot produce one here; the restatement is
                     checked against closed-form cases and a brute-force evaluation (tests/test_metrics.py).
LPIPS (lpips: a pretrained AlexNet) of the same loop is outside this build.
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


def ssim(x, y, data_range=None):
    """skimage.measure.compare_ssim(X, Y) with its defaults on two 2-D arrays (scikit-image 0.12-0.17,
    measure/_structural_similarity.py): win_size 7, uniform filter (scipy.ndimage.uniform_filter, as skimage itself calls it),
    K1 0.01, K2 0.03, use_sample_covariance, data_range from the dtype when not given (floats: dtype_range = (-1, 1), i.e. 2),
    float64 arithmetic, mean over the map cropped by (win_size - 1) // 2 on every side."""
    from scipy.ndimage import uniform_filter
    if x.shape != y.shape or x.ndim != 2:
        raise ValueError("two 2-D images of one shape")
    win_size, K1, K2 = 7, 0.01, 0.03
    if min(x.shape) < win_size:
        raise ValueError("win_size exceeds image extent")
    if data_range is None:
        if not np.issubdtype(x.dtype, np.floating):
            raise ValueError("integer images: pass data_range")
        data_range = 2.0
    X, Y = x.astype(np.float64), y.astype(np.float64)
    NP = win_size ** X.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(X, size=win_size), uniform_filter(Y, size=win_size)
    uxx, uyy, uxy = uniform_filter(X * X, size=win_size), uniform_filter(Y * Y, size=win_size), uniform_filter(X * Y, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
    S = (A1 * A2) / (B1 * B2)
    pad = (win_size - 1) // 2
    return float(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean())


def ssim_bruteforce(x, y, data_range=2.0):
    """The same number from explicit 49-pixel sums (a check of `ssim` that shares no filtering code with it)."""
    X, Y = x.astype(np.float64), y.astype(np.float64)
    h, w = X.shape
    tot = 0.0
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    for i in range(h - 6):
        for j in range(w - 6):
            a, b = X[i:i + 7, j:j + 7], Y[i:i + 7, j:j + 7]
            ux, uy = a.mean(), b.mean()
            vx, vy = ((a - ux) ** 2).sum() / 48.0, ((b - uy) ** 2).sum() / 48.0
            vxy = ((a - ux) * (b - uy)).sum() / 48.0
            tot += ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    return tot / ((h - 6) * (w - 6))


def slice_ssim(fake_B, real_B, WC, WW, aliased=False):
    """[SSIMw, SSIM] of one test() iteration (HdTrainer.py:1008-1053): compare_ssim of the windowed masked pair (c, b) and of the
    raw masked pair, built as in slice_metrics (aliased: as in slice_metrics_cyc)."""
    b = to_windowdata(real_B, WC, WW)
    bb = b if aliased else b.copy()
    bb[bb < 0.3] = 0
    bb[bb >= 0.3] = 1
    b = b * bb
    b[b == 0] = -1
    c = to_windowdata(fake_B, WC, WW) * bb
    cc = c if aliased else c.copy()
    cc[cc < 0.3] = 0
    cc[cc >= 0.3] = 1
    c = c * cc
    c[c == 0] = -1
    real_m = real_B * bb
    real_m[real_m == 0] = -1
    fake_m = fake_B * cc
    fake_m[fake_m == 0] = -1
    return np.array([ssim(c, b), ssim(fake_m, real_m)], dtype=np.float64)
