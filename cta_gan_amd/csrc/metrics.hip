// Evaluation path of the trainers' test()/validation loops on the device:
//   to_windowdata            trainer/HdTrainer.py:41-64 == trainer/CycTrainer.py:34-57  (CT window -> [-1, 1])
//   threshold masks          HdTrainer.py:1008-1023, 1041-1047   (>= 0.3 foreground, background := -1)
//   MAE / PSNR / UQI         HdTrainer.py:1089-1125
// The reference pulls every slice to the host and runs numpy; here one launch windows, masks and reduces a batch of
// slices, a second turns the sums into the six numbers per slice.  All elementwise arithmetic follows the
// reference's float32 operation order with explicitly rounded operations (no FMA contraction), so the masks --
// which hang on exact comparisons (== 0, >= 0.3) -- are identical; the reductions accumulate in fp64.
#include "common.h"

#define MET_NSUM 20   // per (x, y) pair: n_m, sum|d|_m, sum d^2_m, sum|d|, sum d^2, sx, sy, sxx, syy, sxy

struct WinParams { float wmin, dfac; };

__device__ __forceinline__ WinParams win_params(float wc, float ww) {
    // python floats in the reference: win_min = (2*c - w)/2.0 + 0.5, dFactor = 255.0 / (win_max - win_min);
    // a float32 array combined with them rounds each to float32 first
    const double c = (double)wc, w = (double)ww;
    const double wmin = (2.0 * c - w) / 2.0 + 0.5, wmax = (2.0 * c + w) / 2.0 + 0.5;
    WinParams p;
    p.wmin = (float)wmin;
    p.dfac = (float)(255.0 / (wmax - wmin));
    return p;
}

__device__ __forceinline__ float window_one(float v, const WinParams p) {
    float t = __fmul_rn(__fmul_rn(__fadd_rn(v, 1.0f), 0.5f), 4095.0f);
    if (t == 0.0f) t = -2000.0f;
    t = __fsub_rn(t, 1024.0f);
    t = __fsub_rn(t, p.wmin);
    t = truncf(__fmul_rn(t, p.dfac));
    if (t > 255.0f) t = 255.0f;
    if (t < 0.0f) t = 0.0f;
    t = __fdiv_rn(t, 255.0f);
    return __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
}

__global__ __launch_bounds__(256) void to_windowdata_kernel(const float* __restrict__ img, const float* __restrict__ wc,
                                                            const float* __restrict__ ww, float* __restrict__ out,
                                                            long HW) {
    const int n = blockIdx.y;
    const WinParams p = win_params(wc[n], ww[n]);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256)
        out[n * HW + i] = window_one(img[n * HW + i], p);
}

struct PairAcc {
    float v[10];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = 0.f;
    }
    // x = generated, y = reference ("real"); background of y is exactly -1
    __device__ __forceinline__ void add(float x, float y) {
        const float d = x - y, ad = fabsf(d), d2 = d * d;
        const bool m = y != -1.0f;
        v[0] += m ? 1.f : 0.f;
        v[1] += m ? ad : 0.f;
        v[2] += m ? d2 : 0.f;
        v[3] += ad;
        v[4] += d2;
        v[5] += x;
        v[6] += y;
        v[7] += x * x;
        v[8] += y * y;
        v[9] += x * y;
    }
};

__global__ __launch_bounds__(256) void window_metrics_partial_kernel(const float* __restrict__ fake,
                                                                     const float* __restrict__ real,
                                                                     const float* __restrict__ wc,
                                                                     const float* __restrict__ ww, long HW,
                                                                     double* __restrict__ part, int aliased) {
    __shared__ float red[4][MET_NSUM];
    const int n = blockIdx.y;
    const WinParams p = win_params(wc[n], ww[n]);
    PairAcc aw, ar;   // windowed pair (c, b) and raw pair (fake*cc, real*bb)
    aw.clear();
    ar.clear();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const float f = fake[n * HW + i], r = real[n * HW + i];
        // b = W(real); bb = b >= 0.3; b = b*bb; b[b == 0] = -1
        float b = window_one(r, p);
        const float bb = b >= 0.3f ? 1.f : 0.f;
        b = __fmul_rn(b, bb);
        if (b == 0.f) b = -1.f;
        // c = W(fake)*bb; cc = c >= 0.3; c = c*cc; c[c == 0] = -1
        float c = __fmul_rn(window_one(f, p), bb);
        const float cc = c >= 0.3f ? 1.f : 0.f;
        c = __fmul_rn(c, cc);
        if (c == 0.f) c = -1.f;
        if (aliased) {
            // trainer/CycTrainer.py:288-298 writes `bb = b` / `cc = c` WITHOUT a copy, so thresholding the masks also
            // thresholds b and c: its windowed pair is the two binary masks mapped to +-1
            b = bb != 0.f ? 1.f : -1.f;
            c = cc != 0.f ? 1.f : -1.f;
        }
        aw.add(c, b);
        // raw maps under the same masks
        float rm = __fmul_rn(r, bb);
        if (rm == 0.f) rm = -1.f;
        float fm = __fmul_rn(f, cc);
        if (fm == 0.f) fm = -1.f;
        ar.add(fm, rm);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float a = wave_sum(aw.v[k]), b = wave_sum(ar.v[k]);
        if (lane == 0) { red[wave][k] = a; red[wave][10 + k] = b; }
    }
    __syncthreads();
    if (threadIdx.x < MET_NSUM) {
        const int k = threadIdx.x;
        part[((size_t)n * gridDim.x + blockIdx.x) * MET_NSUM + k] =
            (double)red[0][k] + (double)red[1][k] + (double)red[2][k] + (double)red[3][k];
    }
}

__global__ void window_metrics_final_kernel(const double* __restrict__ part, int nblk, long HW, int B,
                                            double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (slice, pair)
    if (i >= B * 2) return;
    const int n = i >> 1, pair = i & 1;
    double s[10];
    for (int k = 0; k < 10; ++k) s[k] = 0.0;
    for (int b = 0; b < nblk; ++b)
        for (int k = 0; k < 10; ++k) s[k] += part[((size_t)n * nblk + b) * MET_NSUM + pair * 10 + k];
    const double N = (double)HW;
    // MAE (HdTrainer.py:1106-1117): masked mean |x - y| / 2; all-background slices fall back to the full mean + 1e-10
    const double mae = (s[0] > 0.0 ? s[1] / s[0] : s[3] / N + 1e-10) / 2.0;
    // PSNR (:1089-1104) on (x+1)/2 - (y+1)/2 = (x - y)/2
    const double mse = s[0] > 0.0 ? s[2] / (4.0 * s[0]) : s[4] / (4.0 * N) + 1e-10;
    const double psnr = mse < 1.0e-10 ? 100.0 : 20.0 * log10(1.0 / (sqrt(mse) + 1e-10));
    // UQI (:1119-1125) over all pixels, unbiased (N - 1) moments
    const double mf = s[5] / N, mr = s[6] / N;
    const double vf = (s[7] - N * mf * mf) / (N - 1.0), vr = (s[8] - N * mr * mr) / (N - 1.0);
    const double cov = (s[9] - N * mf * mr) / (N - 1.0);
    const double uqi = 4.0 * mf * mr * cov / ((mf * mf + mr * mr) * (vf + vr) + 1e-10);
    out[i * 3 + 0] = mae;
    out[i * 3 + 1] = psnr;
    out[i * 3 + 2] = uqi;
}

extern "C" int ctg_to_windowdata(const float* img, const float* wc, const float* ww, float* out, int B, long HW,
                                 void* stream) {
    CTG_ENTER();
    if (img == nullptr || wc == nullptr || ww == nullptr || out == nullptr || B < 1 || HW < 1) return CTG_EINVAL;
    const int blocks = (int)((HW + 255) / 256 < 1024 ? (HW + 255) / 256 : 1024);
    hipLaunchKernelGGL(to_windowdata_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, img, wc, ww, out, HW);
    return ctg_launch_status();
}

extern "C" int ctg_window_metrics(const float* fake, const float* real, const float* wc, const float* ww, int B,
                                  long HW, int nblk, int aliased, double* part, double* out, void* stream) {
    CTG_ENTER();
    if (fake == nullptr || real == nullptr || wc == nullptr || ww == nullptr || part == nullptr || out == nullptr)
        return CTG_EINVAL;
    if (B < 1 || HW < 2 || nblk < 1 || nblk > 4096) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(window_metrics_partial_kernel, dim3(nblk, B), dim3(256), 0, st, fake, real, wc, ww, HW, part,
                       aliased);
    hipLaunchKernelGGL(window_metrics_final_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, st, part, nblk, HW, B, out);
    return ctg_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Mean structural similarity of slice pairs: what `skimage.measure.compare_ssim(x, y)` returns for two float images with its
// defaults -- the call of the reference's validation pass (trainer/HdTrainer.py:256,779, CycTrainer.py:216, p2pTrainer.py:164,
// RegTrainer.py:219) and of its test() loops (HdTrainer.py:1028,1053).  scikit-image is not in this image and not pinned by the
// reference; the function's published algorithm (Wang et al. 2004, as implemented by scikit-image 0.12-0.17): 7x7 uniform window,
// local means / unbiased (sample) variances and covariance in float64, C1 = (0.01 R)^2, C2 = (0.03 R)^2 with R = data_range (2
// for float images), S = (2 ux uy + C1)(2 vxy + C2) / ((ux^2 + uy^2 + C1)(vx + vy + C2)), averaged over the (H-6) x (W-6)
// positions whose window lies inside the image.
// mode 0: the pair (fake, real) as given -> out[B][1].  mode 1: the two masked pairs of the test() loop, built per pixel exactly
// as in window_metrics_partial_kernel -> out[B][2] = {windowed (c, b), raw (fake cc, real bb)}.
#define SSIM_T 16
#define SSIM_W 7
#define SSIM_HT (SSIM_T + SSIM_W - 1)

__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ fake, const float* __restrict__ real,
                                                           const float* __restrict__ wc, const float* __restrict__ ww, int H,
                                                           int W, int mode, int aliased, double data_range,
                                                           double* __restrict__ part) {
    __shared__ float tile[4][SSIM_HT * SSIM_HT];      // pair p: planes 2p (x), 2p + 1 (y)
    __shared__ double red[256];
    const int n = blockIdx.y;
    const int tx_n = (W - SSIM_W + 1 + SSIM_T - 1) / SSIM_T;
    const int y0 = (blockIdx.x / tx_n) * SSIM_T, x0 = (blockIdx.x % tx_n) * SSIM_T;
    const int npair = mode ? 2 : 1;
    WinParams p = {0.f, 0.f};
    if (mode) p = win_params(wc[n], ww[n]);
    const float* __restrict__ F = fake + (size_t)n * H * W;
    const float* __restrict__ R = real + (size_t)n * H * W;
    for (int i = threadIdx.x; i < SSIM_HT * SSIM_HT; i += 256) {
        const int ty = i / SSIM_HT, tx = i - ty * SSIM_HT;
        const int iy = y0 + ty, ix = x0 + tx;
        float f = 0.f, r = 0.f;
        if (iy < H && ix < W) { f = F[(size_t)iy * W + ix]; r = R[(size_t)iy * W + ix]; }
        if (!mode) {
            tile[0][i] = f;
            tile[1][i] = r;
        } else {
            float b = window_one(r, p);
            const float bb = b >= 0.3f ? 1.f : 0.f;
            b = __fmul_rn(b, bb);
            if (b == 0.f) b = -1.f;
            float c = __fmul_rn(window_one(f, p), bb);
            const float cc = c >= 0.3f ? 1.f : 0.f;
            c = __fmul_rn(c, cc);
            if (c == 0.f) c = -1.f;
            if (aliased) {
                b = bb != 0.f ? 1.f : -1.f;
                c = cc != 0.f ? 1.f : -1.f;
            }
            float rm = __fmul_rn(r, bb);
            if (rm == 0.f) rm = -1.f;
            float fm = __fmul_rn(f, cc);
            if (fm == 0.f) fm = -1.f;
            tile[0][i] = c;
            tile[1][i] = b;
            tile[2][i] = fm;
            tile[3][i] = rm;
        }
    }
    __syncthreads();
    const int ly = threadIdx.x / SSIM_T, lx = threadIdx.x % SSIM_T;
    const bool valid = (y0 + ly + SSIM_W <= H) && (x0 + lx + SSIM_W <= W);
    const double NP = (double)(SSIM_W * SSIM_W), cov_norm = NP / (NP - 1.0);
    const double C1 = (0.01 * data_range) * (0.01 * data_range), C2 = (0.03 * data_range) * (0.03 * data_range);
    for (int q = 0; q < npair; ++q) {
        double S = 0.0;
        if (valid) {
            double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
            for (int wy = 0; wy < SSIM_W; ++wy)
#pragma unroll
                for (int wx = 0; wx < SSIM_W; ++wx) {
                    const double x = (double)tile[2 * q][(ly + wy) * SSIM_HT + lx + wx];
                    const double y = (double)tile[2 * q + 1][(ly + wy) * SSIM_HT + lx + wx];
                    sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
                }
            const double ux = sx / NP, uy = sy / NP;
            const double vx = cov_norm * (sxx / NP - ux * ux), vy = cov_norm * (syy / NP - uy * uy);
            const double vxy = cov_norm * (sxy / NP - ux * uy);
            S = ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
        }
        red[threadIdx.x] = S;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {      // fixed tree: the same bits on every run
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[((size_t)n * gridDim.x + blockIdx.x) * 2 + q] = red[0];
        __syncthreads();
    }
}

__global__ void ssim_final_kernel(const double* __restrict__ part, int nblk, int npair, int B, double count,
                                  double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (slice, pair)
    if (i >= B * npair) return;
    const int n = i / npair, q = i - n * npair;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[((size_t)n * nblk + b) * 2 + q];
    out[i] = s / count;
}

extern "C" int ctg_ssim(const float* fake, const float* real, const float* wc, const float* ww, int B, int H, int W, int mode,
                        int aliased, double data_range, double* part, double* out, void* stream) {
    CTG_ENTER();
    if (fake == nullptr || real == nullptr || part == nullptr || out == nullptr) return CTG_EINVAL;
    if (mode != 0 && mode != 1) return CTG_EINVAL;
    if (mode == 1 && (wc == nullptr || ww == nullptr)) return CTG_EINVAL;
    if (B < 1 || B > 65535 || H < SSIM_W || W < SSIM_W || !(data_range > 0.0)) return CTG_EINVAL;   // (skimage: win_size exceeds image extent)
    const long nblk = (long)((H - SSIM_W + 1 + SSIM_T - 1) / SSIM_T) * ((W - SSIM_W + 1 + SSIM_T - 1) / SSIM_T);
    if (nblk > (1L << 30)) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_partial_kernel, dim3((unsigned)nblk, B), dim3(256), 0, st, fake, real, wc, ww, H, W, mode, aliased,
                       data_range, part);
    const int npair = mode ? 2 : 1;
    hipLaunchKernelGGL(ssim_final_kernel, dim3((B * npair + 63) / 64), dim3(64), 0, st, part, (int)nblk, npair, B,
                       (double)(H - SSIM_W + 1) * (double)(W - SSIM_W + 1), out);
    return ctg_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Input pipeline arithmetic: read_ori_w (trainer/datasets.py:36-71) after the DICOM
// read -- raw HU (SimpleITK convention: pydicom value - 1024) -> the two normalised images of a training pair --
// and Resize (trainer/utils.py:13-32 = F.interpolate(mode='nearest')).  The reference does the first in float64
// numpy and casts to float32 in the transform; so does this kernel.
__global__ __launch_bounds__(256) void hu_to_inputs_kernel(const short* __restrict__ hu, double wmin, double dfac,
                                                           float* __restrict__ win, float* __restrict__ full, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double d1 = (double)hu[i];
        // image1: CT window -> 8-bit levels -> [-1, 1]
        double t = trunc((d1 - wmin) * dfac);
        t = t > 255.0 ? 255.0 : t;
        t = t < 0.0 ? 0.0 : t;
        t = t / 255.0;
        win[i] = (float)((t - 0.5) / 0.5);
        // image2: full 12-bit range -> [-1, 1]
        double f = d1 + 1024.0;
        f = f < 0.0 ? 0.0 : f;
        f = f / 4095.0;
        full[i] = (float)((f - 0.5) / 0.5);
    }
}

__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ src, int Hi, int Wi,
                                                             float* __restrict__ dst, int Ho, int Wo, float sh, float sw) {
    const int n = blockIdx.y;
    const int total = Ho * Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int oy = i / Wo, ox = i - oy * Wo;
        int iy = (int)floorf(__fmul_rn((float)oy, sh)), ix = (int)floorf(__fmul_rn((float)ox, sw));
        iy = iy < Hi - 1 ? iy : Hi - 1;
        ix = ix < Wi - 1 ? ix : Wi - 1;
        dst[(size_t)n * total + i] = src[((size_t)n * Hi + iy) * Wi + ix];
    }
}

extern "C" int ctg_hu_to_inputs(const short* hu, float wc, float ww, float* win, float* full, long n, void* stream) {
    CTG_ENTER();
    if (hu == nullptr || win == nullptr || full == nullptr || n < 1 || ww <= 0.f) return CTG_EINVAL;
    const double c = (double)wc, w = (double)ww;
    const double wmin = (2.0 * c - w) / 2.0 + 0.5, wmax = (2.0 * c + w) / 2.0 + 0.5;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(hu_to_inputs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hu, wmin,
                       255.0 / (wmax - wmin), win, full, n);
    return ctg_launch_status();
}

extern "C" int ctg_resize_nearest(const float* src, int B, int Hi, int Wi, float* dst, int Ho, int Wo, void* stream) {
    CTG_ENTER();
    if (src == nullptr || dst == nullptr || B < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return CTG_EINVAL;
    if ((long)Ho * Wo >= (1L << 31) || (long)Hi * Wi >= (1L << 31)) return CTG_EINVAL;
    const int total = Ho * Wo;
    const int blocks = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, src, Hi, Wi, dst, Ho,
                       Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return ctg_launch_status();
}
