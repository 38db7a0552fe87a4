// InstanceNorm (affine-free, eps 1e-5, biased variance) forward/backward with the
// neighbouring activation, residual add and reflection-pad "fold" fused in, plus
// the gradient-combine and bias-gradient kernels.  All HBM-bound: every tensor
// is read in 16-byte chunks (4 fp32 / 8 bf16 channels of one pixel), per-channel
// statistics are fp32 partials per (sample, pixel-slab) summed in a fixed order
// (deterministic, no float atomics).
//
// Replaces nn.InstanceNorm2d + nn.ReLU/LeakyReLU + the residual add at
// Model/HdGan.py:55-63,71-72,79-80,94-95,124-133 and trainer/layers.py:14,282-300
// and their autograd backward.
//
// "fold": the backward-data pass of a reflection-padded conv produces the
// gradient on the PADDED grid (H+2p, W+2p); consumers read it through
// fold_load(), which adds the mirrored border rows/columns back onto the
// interior (the transpose of nn.ReflectionPad2d), so no separate pass exists.
#include "common.h"

// IN_EPS: common.h
#define MAX_SLABS 128

__device__ __forceinline__ int fold_srcs(int y, int H, int p, int* s) {
    int k = 0;
    s[k++] = y;
    if (y >= 1 && y <= p) s[k++] = -y;
    if (y >= H - 1 - p && y <= H - 2) s[k++] = 2 * (H - 1) - y;
    return k;
}

// value of the (possibly padded-grid) gradient at interior pixel (y, x), channel chunk ch
template <typename T>
__device__ __forceinline__ void fold_load(Chunk<T>& out, const T* __restrict__ d, int n, int y, int x, int ch, int H,
                                          int W, int p, int ld) {
    if (p == 0) {
        out.load(d + (((size_t)n * H + y) * W + x) * ld + ch, ld);
        return;
    }
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    int ys[3], xs[3];
    const int ny = fold_srcs(y, H, p, ys), nx = fold_srcs(x, W, p, xs);
    out.zero();
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            Chunk<T> t;
            t.load(d + (((size_t)n * Hp + ys[a] + p) * Wp + xs[b] + p) * ld + ch, ld);
#pragma unroll
            for (int e = 0; e < Chunk<T>::N; ++e) out.v[e] += t.v[e];
        }
}

// ---------------------------------------------------------------------------
// per-(n, c) two-moment partial reductions.  MODE 0: (x, x^2).  MODE 1: IN backward
// (g, g*xhat) with g = fold(dout) * act'(xhat).  MODE 2: (g, -) bias gradient.
// grid = (nslabs, B); block 256 = (256/CPP pixel lanes) x (CPP channel chunks).
// ---------------------------------------------------------------------------
// MODE 3: MODE 2 through an activation: x = the activation's saved OUTPUT y, gm = fold(dout) * act'(y) is WRITTEN to `gout`
// (the gradient the conv's backward passes consume) and summed -- the LeakyReLU backward and the bias gradient of a
// conv + bias + LeakyReLU layer (trainer/layers.py:97-104) in one pass instead of two.
template <typename T, int MODE, typename TX = T>
__global__ __launch_bounds__(256) void moments_partial_kernel(const TX* __restrict__ x, int x_ld,
                                                              const T* __restrict__ dout, int d_ld, int pad,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int act, int H, int W,
                                                              int C, float* __restrict__ part, T* __restrict__ gout = nullptr,
                                                              int go_ld = 0) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC;                      // chunks per pixel (power of two, <= 256)
    const int PL = 256 / CPP;                     // pixel lanes
    const int tid = threadIdx.x;
    const int cc = tid % CPP, pl = tid / CPP;
    const int n = blockIdx.y, slab = blockIdx.x, nslabs = gridDim.x;
    const int HW = H * W;
    const int per = (HW + nslabs - 1) / nslabs;
    const int pbeg = slab * per, pend = min(pbeg + per, HW);
    const int ch = cc * EPC;
    float s1[EPC], s2[EPC], mu[EPC], rs[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        s1[e] = 0.f; s2[e] = 0.f;
        if (MODE == 1) { mu[e] = mean[n * C + ch + e]; rs[e] = rstd[n * C + ch + e]; }
    }
    // 4 pixels per trip: the loads of a trip are independent, so 4-8 16-byte requests per lane are in flight
    constexpr int UNR = 4;
    for (int pb = pbeg + pl; pb < pend; pb += PL * UNR) {
        Chunk<TX> v[UNR];
        Chunk<T> g[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = pb + u * PL;
            ok[u] = p < pend;
            const int pc = ok[u] ? p : pbeg;
            if (MODE == 0 || MODE == 1 || MODE == 3) v[u].load(x + ((size_t)n * HW + pc) * x_ld + ch, x_ld);
            if (MODE == 1 || MODE == 2 || MODE == 3) {
                if (pad == 0) {
                    g[u].load(dout + ((size_t)n * HW + pc) * d_ld + ch, d_ld);
                } else {
                    const int y = pc / W;
                    fold_load<T>(g[u], dout, n, y, pc - y * W, ch, H, W, pad, d_ld);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (!ok[u]) continue;
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) { s1[e] += v[u].v[e]; s2[e] += v[u].v[e] * v[u].v[e]; }
            } else if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float xh = (v[u].v[e] - mu[e]) * rs[e];
                    float gg = g[u].v[e];
                    if (act == ACT_RELU) gg = xh > 0.f ? gg : 0.f;
                    else if (act == ACT_LRELU) gg = xh > 0.f ? gg : LRELU_SLOPE * gg;
                    s1[e] += gg; s2[e] += gg * xh;
                }
            } else if (MODE == 3) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    o.v[e] = g[u].v[e] * act_grad_from_out(v[u].v[e], act);
                    s1[e] += o.v[e];
                }
                o.store(gout + ((size_t)n * HW + pb + u * PL) * go_ld + ch, go_ld);
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) s1[e] += g[u].v[e];
            }
        }
    }
    // reduce over the PL pixel lanes through LDS, one moment at a time
    float* out = part + (((size_t)n * nslabs + slab) * C) * 2;
    __shared__ float stage[256 * 8];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) stage[pl * C + ch + e] = which ? s2[e] : s1[e];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float s = 0.f;
            for (int q = 0; q < PL; ++q) s += stage[q * C + c];
            out[c * 2 + which] = s;
        }
    }
}


// (mean, rstd) from any number of partial moments: one 64-lane wave per (n, c), fixed lane-strided order
__global__ __launch_bounds__(64) void moments_finalize_wave_kernel(const float* __restrict__ part, int nslabs, int C,
                                                                   float invHW, int mode, float* __restrict__ mean,
                                                                   float* __restrict__ rstd) {
    const int i = blockIdx.x;
    const int n = i / C, c = i - n * C;
    double a = 0.0, b = 0.0;
    for (int s = threadIdx.x; s < nslabs; s += 64) {
        const float* p = part + (((size_t)n * nslabs + s) * C + c) * 2;
        a += (double)p[0];
        b += (double)p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (threadIdx.x == 0) {
        if (mode == 0) {
            const double m = a * invHW;
            double var = b * invHW - m * m;
            var = var < 0.0 ? 0.0 : var;
            mean[i] = (float)m;
            rstd[i] = (float)(1.0 / sqrt(var + (double)IN_EPS));
        } else {   // plain means (the two sums of the InstanceNorm backward)
            mean[i] = (float)(a * invHW);
            rstd[i] = (float)(b * invHW);
        }
    }
}

// bias gradient: db[c] (+)= sum over samples and slabs of the MODE-2 partials; one 64-lane wave per channel
__global__ __launch_bounds__(64) void bias_finalize_kernel(const float* __restrict__ part, int B, int nslabs, int C,
                                                           int Creal, float* __restrict__ db, int accumulate) {
    const int c = blockIdx.x;
    if (c >= Creal) return;
    const int total = B * nslabs;
    float a = 0.f;
    for (int i = threadIdx.x; i < total; i += 64) a += part[((size_t)i * C + c) * 2];
    a = wave_sum(a);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + a : a;
}

// Elementwise InstanceNorm kernels share one decomposition: grid = (pixel blocks, B); a thread owns ONE
// 16-byte channel chunk (its per-(n,c) parameters stay in registers) and strides over the pixels of sample
// blockIdx.y, so consecutive lanes still touch consecutive addresses and no per-element parameter gathers or
// index divisions remain.
template <typename T> struct ChanParams {
    static constexpr int EPC = Chunk<T>::N;
    float v[EPC];
    __device__ __forceinline__ void load(const float* __restrict__ p) {
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
    }
};

// out = act((x - mean) * rstd) [+ res]
// (x, res and out carry no __restrict__: a forward that keeps nothing normalises in place, out == x -- engine.inorm_forward)
template <typename T>
__global__ __launch_bounds__(256) void in_apply_kernel(const T* x, int x_ld,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int act,
                                                       const T* res, int r_ld, T* out,
                                                       int o_ld, int HW, int C) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC, PL = 256 / CPP;
    const int cc = threadIdx.x % CPP, pl = threadIdx.x / CPP;
    const int n = blockIdx.y, ch = cc * EPC;
    ChanParams<T> mu, rs;
    mu.load(mean + n * C + ch);
    rs.load(rstd + n * C + ch);
    const size_t base = (size_t)n * HW;
    for (int p = blockIdx.x * PL + pl; p < HW; p += gridDim.x * PL) {
        Chunk<T> v, o;
        v.load(x + (base + p) * x_ld + ch, x_ld);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] = act_apply((v.v[e] - mu.v[e]) * rs.v[e], act);
        if (res != nullptr) {
            Chunk<T> r;
            r.load(res + (base + p) * r_ld + ch, r_ld);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] += r.v[e];
        }
        o.store(out + (base + p) * o_ld + ch, o_ld);
    }
}

// dx = rstd * (g - s1 - xhat * s2),  g = fold(dout) * act'(xhat)
template <typename T, typename TX = T>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const TX* __restrict__ x, int x_ld,
                                                           const T* __restrict__ dout, int d_ld, int pad,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ s1,
                                                           const float* __restrict__ s2, int act,
                                                           T* __restrict__ dx, int dx_ld, int H, int W, int C) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC, PL = 256 / CPP;
    const int cc = threadIdx.x % CPP, pl = threadIdx.x / CPP;
    const int n = blockIdx.y, ch = cc * EPC;
    const int HW = H * W;
    ChanParams<T> mu, rs, a1, a2;
    mu.load(mean + n * C + ch);
    rs.load(rstd + n * C + ch);
    a1.load(s1 + n * C + ch);
    a2.load(s2 + n * C + ch);
    const size_t base = (size_t)n * HW;
    for (int p = blockIdx.x * PL + pl; p < HW; p += gridDim.x * PL) {
        Chunk<TX> v;
        Chunk<T> g, o;
        v.load(x + (base + p) * x_ld + ch, x_ld);
        if (pad == 0) {
            g.load(dout + (base + p) * d_ld + ch, d_ld);
        } else {
            const int y = p / W;
            fold_load<T>(g, dout, n, y, p - y * W, ch, H, W, pad, d_ld);
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float xh = (v.v[e] - mu.v[e]) * rs.v[e];
            float gg = g.v[e];
            if (act == ACT_RELU) gg = xh > 0.f ? gg : 0.f;
            else if (act == ACT_LRELU) gg = xh > 0.f ? gg : LRELU_SLOPE * gg;
            o.v[e] = rs.v[e] * (gg - a1.v[e] - xh * a2.v[e]);
        }
        o.store(dx + (base + p) * dx_ld + ch, dx_ld);
    }
}

// ---------------------------------------------------------------------------
// Finalize fused into the elementwise pass (round 3): the two kernels below take the per-slab PARTIALS instead of finished
// per-(n, c) statistics.  A workgroup owns one sample, one group of CGC channel chunks (64 bf16 / 32 fp32 channels) and a
// strided set of pixels; its prologue sums the partials of ITS channels (nslabs x 2 CGE floats from L2, double
// accumulators, fixed order: slab stripes per thread, then stripes in order) -- the separate 6 us finalize launch per
// InstanceNorm and its launch gap disappear.  Worth it while the prologue stays small against the strip: the host side
// only takes this path for nslabs <= FUSED_MAX_SLABS.
// ---------------------------------------------------------------------------
#define FUSED_MAX_SLABS 128

// sums over slabs of part[n][s][c0 + ch][which] for ch < nch -> fa[ch], fb[ch] (LDS); MODE 0: (mean, rstd), MODE 1: plain means
template <int MODE>
__device__ __forceinline__ void wg_finalize(const float* __restrict__ part, int n, int nslabs, int C, int c0, int nch,
                                            float invHW, float* fa, float* fb, double* scratch) {
    const int tid = threadIdx.x;
    const int npairs = 2 * nch;                 // (channel, which) pairs of the group: 128 (bf16) or 64 (fp32)
    const int stripes = 256 / npairs;
    const int pair = tid % npairs, stripe = tid / npairs;
    double acc = 0.0;
    const float* src = part + ((size_t)n * nslabs * C + c0) * 2 + pair;
    for (int s = stripe; s < nslabs; s += stripes) acc += (double)src[(size_t)s * C * 2];
    scratch[stripe * npairs + pair] = acc;
    __syncthreads();
    if (tid < nch) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < stripes; ++q) { a += scratch[q * npairs + 2 * tid]; b += scratch[q * npairs + 2 * tid + 1]; }
        if (MODE == 0) {
            const double m = a * invHW;
            double var = b * invHW - m * m;
            var = var < 0.0 ? 0.0 : var;
            fa[tid] = (float)m;
            fb[tid] = (float)(1.0 / sqrt(var + (double)IN_EPS));
        } else {
            fa[tid] = (float)(a * invHW);
            fb[tid] = (float)(b * invHW);
        }
    }
    __syncthreads();
}

// out = act((x - mean) * rstd) [+ res] with (mean, rstd) finalized from `part` in the prologue; the workgroups with
// blockIdx.x == 0 also publish mean / rstd (the backward pass and the fused conv epilogues read them)
template <typename T>
__global__ __launch_bounds__(256) void in_apply_part_kernel(const T* x, int x_ld,
                                                            const float* __restrict__ part, int nslabs, float invHW,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int act,
                                                            const T* res, int r_ld, T* out,
                                                            int o_ld, int HW, int C, int CGC) {
    constexpr int EPC = Chunk<T>::N;
    __shared__ double scratch[256];
    __shared__ float fa[64], fb[64];
    const int PL = 256 / CGC;
    const int cc = threadIdx.x % CGC, pl = threadIdx.x / CGC;
    const int n = blockIdx.z, c0 = blockIdx.y * CGC * EPC, ch = c0 + cc * EPC;
    const size_t base = (size_t)n * HW;
    constexpr int UNR = 4;
    const int p0 = blockIdx.x * PL + pl, pstep = gridDim.x * PL;
    // the first trip's loads are in flight while the prologue runs
    Chunk<T> v[UNR], r[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int p = p0 + u * pstep;
        const int pc = p < HW ? p : 0;
        v[u].load(x + (base + pc) * x_ld + ch, x_ld);
        if (res != nullptr) r[u].load(res + (base + pc) * r_ld + ch, r_ld);
    }
    wg_finalize<0>(part, n, nslabs, C, c0, CGC * EPC, invHW, fa, fb, scratch);
    if (blockIdx.x == 0 && threadIdx.x < CGC * EPC) {
        mean[(size_t)n * C + c0 + threadIdx.x] = fa[threadIdx.x];
        rstd[(size_t)n * C + c0 + threadIdx.x] = fb[threadIdx.x];
    }
    float mu[EPC], rs[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { mu[e] = fa[cc * EPC + e]; rs[e] = fb[cc * EPC + e]; }
    for (int pb = p0; pb < HW; pb += UNR * pstep) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = pb + u * pstep;
            if (p < HW) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.v[e] = act_apply((v[u].v[e] - mu[e]) * rs[e], act);
                if (res != nullptr) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o.v[e] += r[u].v[e];
                }
                o.store(out + (base + p) * o_ld + ch, o_ld);
            }
        }
        const int pn = pb + UNR * pstep;
        if (pn < HW) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int p = pn + u * pstep;
                const int pc = p < HW ? p : 0;
                v[u].load(x + (base + pc) * x_ld + ch, x_ld);
                if (res != nullptr) r[u].load(res + (base + pc) * r_ld + ch, r_ld);
            }
        }
    }
}

// dx = rstd * (g - s1 - xhat * s2), g = fold(dout) * act'(xhat), with (s1, s2) finalized from `part` in the prologue
template <typename T, typename TX = T>
__global__ __launch_bounds__(256) void in_bwd_apply_part_kernel(const TX* __restrict__ x, int x_ld,
                                                                const T* __restrict__ dout, int d_ld, int pad,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ part, int nslabs, float invHW,
                                                                int act, T* __restrict__ dx, int dx_ld, int H, int W,
                                                                int C, int CGC) {
    constexpr int EPC = Chunk<T>::N;
    __shared__ double scratch[256];
    __shared__ float fa[64], fb[64];
    const int PL = 256 / CGC;
    const int cc = threadIdx.x % CGC, pl = threadIdx.x / CGC;
    const int n = blockIdx.z, c0 = blockIdx.y * CGC * EPC, ch = c0 + cc * EPC;
    const int HW = H * W;
    const size_t base = (size_t)n * HW;
    constexpr int UNR = 2;
    const int p0 = blockIdx.x * PL + pl, pstep = gridDim.x * PL;
    Chunk<TX> v[UNR];
    Chunk<T> g[UNR];
    auto fetch = [&](int pb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = pb + u * pstep;
            const int pc = p < HW ? p : 0;
            v[u].load(x + (base + pc) * x_ld + ch, x_ld);
            if (pad == 0) {
                g[u].load(dout + (base + pc) * d_ld + ch, d_ld);
            } else {
                const int y = pc / W;
                fold_load<T>(g[u], dout, n, y, pc - y * W, ch, H, W, pad, d_ld);
            }
        }
    };
    fetch(p0);
    wg_finalize<1>(part, n, nslabs, C, c0, CGC * EPC, invHW, fa, fb, scratch);
    ChanParams<T> mu, rs;
    mu.load(mean + (size_t)n * C + ch);
    rs.load(rstd + (size_t)n * C + ch);
    float a1[EPC], a2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { a1[e] = fa[cc * EPC + e]; a2[e] = fb[cc * EPC + e]; }
    for (int pb = p0; pb < HW; pb += UNR * pstep) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = pb + u * pstep;
            if (p < HW) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float xh = (v[u].v[e] - mu.v[e]) * rs.v[e];
                    float gg = g[u].v[e];
                    if (act == ACT_RELU) gg = xh > 0.f ? gg : 0.f;
                    else if (act == ACT_LRELU) gg = xh > 0.f ? gg : LRELU_SLOPE * gg;
                    o.v[e] = rs.v[e] * (gg - a1[e] - xh * a2[e]);
                }
                o.store(dx + (base + p) * dx_ld + ch, dx_ld);
            }
        }
        const int pn = pb + UNR * pstep;
        if (pn < HW) fetch(pn);
    }
}

// out = (a ? a : 0) + (b ? fold(b) : 0), then * act'(y) when y (the saved activation OUTPUT) is given
template <typename T>
__global__ void grad_combine_kernel(const T* __restrict__ a, int a_ld, const T* __restrict__ b, int b_ld, int pad,
                                    const T* __restrict__ yact, int y_ld, int act, T* __restrict__ out, int o_ld,
                                    int H, int W, int C, long items) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC;
    const int HW = H * W;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const unsigned pixu = (unsigned)it / (unsigned)CPP;   // 32-bit: items < 2^31 is checked by the host side
        const long pix = pixu;
        const int ch = (int)(it - pix * CPP) * EPC;
        Chunk<T> o;
        o.zero();
        if (a != nullptr) o.load(a + pix * a_ld + ch, a_ld);
        if (b != nullptr) {
            const int n = (int)(pixu / (unsigned)HW);
            const int p = (int)(pix - (long)n * HW);
            const int y = p / W, xx = p - y * W;
            Chunk<T> t;
            fold_load<T>(t, b, n, y, xx, ch, H, W, pad, b_ld);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] += t.v[e];
        }
        if (yact != nullptr) {
            Chunk<T> yv;
            yv.load(yact + pix * y_ld + ch, y_ld);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] *= act_grad_from_out(yv.v[e], act);
        }
        o.store(out + pix * o_ld + ch, o_ld);
    }
}

// scalar fold for tiny-channel fp32 gradients (d/d(image) of the reflection-padded 7x7 head conv, which the
// CycleGAN step needs: recovered_A = G_B2A(G_A2B(A)), trainer/CycTrainer.py:153)
__global__ void fold_f32_kernel(const float* __restrict__ dp, float* __restrict__ out, int B, int H, int W, int C,
                                int p) {
    const long total = (long)B * H * W * C;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % W);
        const int y = (int)((i / ((long)C * W)) % H);
        const int n = (int)(i / ((long)C * W * H));
        int ys[3], xs[3];
        const int ny = fold_srcs(y, H, p, ys), nx = fold_srcs(x, W, p, xs);
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) s += dp[(((size_t)n * Hp + ys[a] + p) * Wp + xs[b] + p) * C + c];
        out[i] = s;
    }
}

__global__ void act_bwd_f32_kernel(const float* __restrict__ g, const float* __restrict__ y, int act,
                                   float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = g[i] * act_grad_from_out(y[i], act);
}

static inline int ew_blocks(long items) {
    long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// (pixel blocks, B) grid for the per-sample elementwise kernels: ~16 pixels per thread, >= 1 block per sample
static inline dim3 pix_grid(int dtype, int B, int HW, int C) {
    const int cpp = C / (dtype == DT_F32 ? 4 : 8);
    const int pl = 256 / cpp;
    // 16 pixels per lane; fewer while that leaves the chip under ~2048 workgroups (small batches: a lane's trips are dependent
    // loads, and 128 workgroups of 16 trips each were latency-bound at 18 us for 17 MB)
    static const bool no_smallb = getenv("CTG_NO_SMALLB") != nullptr;      // A/B switch
    int per = 16;
    if (!no_smallb) while (per > 2 && (((long)HW + pl * per - 1) / (pl * per)) * B < 2048) per >>= 1;
    long bx = ((long)HW + pl * per - 1) / (pl * per);
    if (bx < 1) bx = 1;
    if (bx > 4096) bx = 4096;
    return dim3((unsigned)bx, (unsigned)B);
}

static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

#define DISPATCH_T(dtype, CALL)                   \
    if ((dtype) == DT_BF16) { typedef bf16_t T; CALL; } \
    else if ((dtype) == DT_F32) { typedef float T; CALL; } \
    else if ((dtype) == DT_PAIR) { typedef bfpair_t T; CALL; } \
    else return CTG_EINVAL;

// T = the gradient / result type, TX = the type of the saved forward activation `x` (DT_MIX: see common.h)
#define DISPATCH_TX(dtype, CALL)                  \
    if ((dtype) == DT_BF16) { typedef bf16_t T; typedef bf16_t TX; CALL; } \
    else if ((dtype) == DT_F32) { typedef float T; typedef float TX; CALL; } \
    else if ((dtype) == DT_PAIR) { typedef bfpair_t T; typedef bfpair_t TX; CALL; } \
    else if ((dtype) == DT_MIX) { typedef bf16_t T; typedef bfpair_t TX; CALL; } \
    else return CTG_EINVAL;

static int check_c(int dtype, int C) {
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (C % epc) return CTG_EINVAL;
    const int cpp = C / epc;
    return (pow2(cpp) && cpp <= 256) ? CTG_OK : CTG_EINVAL;
}

// part: B * nslabs * C * 2 floats (nslabs <= 64).  mean/rstd: B * C floats.
extern "C" int ctg_in_stats(int dtype, const void* x, int x_ld, int B, int H, int W, int C, int nslabs, float* part,
                            float* mean, float* rstd, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || nslabs < 1 || nslabs > MAX_SLABS) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((moments_partial_kernel<T, 0>), dim3(nslabs, B), dim3(256), 0, st,
                                         (const T*)x, x_ld, (const T*)nullptr, 0, 0, (const float*)nullptr,
                                         (const float*)nullptr, 0, H, W, C, part));
    if (mean != nullptr)   // NULL: partials only (ctg_in_apply_part finalizes them in its prologue)
        hipLaunchKernelGGL(moments_finalize_wave_kernel, dim3(B * C), dim3(64), 0, st, part, nslabs, C,
                           1.0f / (float)(H * W), 0, mean, rstd);
    return ctg_launch_status();
}

extern "C" int ctg_in_finalize(const float* part, int B, int C, int nslabs, int HW, int mode, float* mean, float* rstd,
                               void* stream) {
    CTG_ENTER();
    if (B < 1 || C < 1 || nslabs < 1 || HW < 1 || (mode != 0 && mode != 1)) return CTG_EINVAL;
    hipLaunchKernelGGL(moments_finalize_wave_kernel, dim3(B * C), dim3(64), 0, (hipStream_t)stream, part, nslabs, C,
                       1.0f / (float)HW, mode, mean, rstd);
    return ctg_launch_status();
}

extern "C" int ctg_in_apply(int dtype, const void* x, int x_ld, const float* mean, const float* rstd, int act,
                            const void* res, int r_ld, void* out, int o_ld, int B, int H, int W, int C, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || out == nullptr) return CTG_EINVAL;
    DISPATCH_T(dtype, hipLaunchKernelGGL((in_apply_kernel<T>), pix_grid(dtype, B, H * W, C), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, x_ld, mean, rstd, act, (const T*)res, r_ld,
                                         (T*)out, o_ld, H * W, C));
    return ctg_launch_status();
}

// (pixel blocks, channel groups, B) grid of the fused finalize + elementwise kernels: a workgroup's strip is >= 16 pixels
// per partial slab it sums in its prologue, unless that leaves the chip with fewer than ~1024 workgroups
static inline dim3 fused_grid(int dtype, int B, int HW, int C, int nslabs, int* cgc_out) {
    const int cpp = C / (dtype == DT_F32 ? 4 : 8);
    const int cgc = cpp < 8 ? cpp : 8;
    const int ncg = cpp / cgc, pl = 256 / cgc;
    long gx = (long)HW / (16L * nslabs);
    if (gx * ncg * B < 1024) gx = (1024 + (long)ncg * B - 1) / ((long)ncg * B);
    const long gmax = ((long)HW + pl * 4 - 1) / (pl * 4);
    if (gx > gmax) gx = gmax;
    if (gx < 1) gx = 1;
    *cgc_out = cgc;
    return dim3((unsigned)gx, (unsigned)ncg, (unsigned)B);
}

// out = act(IN(x)) [+ res] from the partial moments part[B][nslabs][C][2] (a conv epilogue's or ctg_in_stats'), finalized in
// the kernel's prologue; mean / rstd [B][C] are written for the backward pass.  nslabs <= 128.
extern "C" int ctg_in_apply_part(int dtype, const void* x, int x_ld, const float* part, int nslabs, float* mean,
                                 float* rstd, int act, const void* res, int r_ld, void* out, int o_ld, int B, int H,
                                 int W, int C, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || nslabs < 1 || nslabs > FUSED_MAX_SLABS || part == nullptr || mean == nullptr || rstd == nullptr)
        return CTG_EINVAL;
    int cgc;
    const dim3 grid = fused_grid(dtype, B, H * W, C, nslabs, &cgc);
    DISPATCH_T(dtype, hipLaunchKernelGGL((in_apply_part_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                         x_ld, part, nslabs, 1.0f / (float)(H * W), mean, rstd, act, (const T*)res, r_ld,
                                         (T*)out, o_ld, H * W, C, cgc));
    return ctg_launch_status();
}

// the statistics pass of the InstanceNorm backward alone: part[B][nslabs][C][2] = per-slab (sum g m, sum g m xhat)
extern "C" int ctg_in_bwd_partial(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad,
                                  const float* mean, const float* rstd, int act, int B, int H, int W, int C, int nslabs,
                                  float* part, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || nslabs < 1 || nslabs > MAX_SLABS || pad < 0 || pad >= H || pad >= W) return CTG_EINVAL;
    DISPATCH_TX(dtype, hipLaunchKernelGGL((moments_partial_kernel<T, 1, TX>), dim3(nslabs, B), dim3(256), 0,
                                          (hipStream_t)stream, (const TX*)x, x_ld, (const T*)dout, d_ld, pad, mean, rstd, act,
                                          H, W, C, part, (T*)nullptr, 0));
    return ctg_launch_status();
}

// IN backward = statistics pass + finalize + elementwise pass.  dout may live on a reflection-padded grid (pad > 0).
// s1/s2: B*C floats scratch.
extern "C" int ctg_in_bwd(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad, const float* mean,
                          const float* rstd, int act, void* dx, int dx_ld, int B, int H, int W, int C, int nslabs,
                          float* part, float* s1, float* s2, void* stream) {
    int st = ctg_in_bwd_partial(dtype, x, x_ld, dout, d_ld, pad, mean, rstd, act, B, H, W, C, nslabs, part, stream);
    if (st != CTG_OK) return st;
    st = ctg_in_finalize(part, B, C, nslabs, H * W, 1, s1, s2, stream);
    if (st != CTG_OK) return st;
    return ctg_in_bwd_apply(dtype, x, x_ld, dout, d_ld, pad, mean, rstd, s1, s2, act, dx, dx_ld, B, H, W, C, stream);
}

// the elementwise pass of the IN backward with finished sums s1 / s2 [B][C] (ctg_in_finalize mode 1)
extern "C" int ctg_in_bwd_apply(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad, const float* mean,
                                const float* rstd, const float* s1, const float* s2, int act, void* dx, int dx_ld, int B,
                                int H, int W, int C, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || pad < 0 || pad >= H || pad >= W || s1 == nullptr || s2 == nullptr) return CTG_EINVAL;
    DISPATCH_TX(dtype, hipLaunchKernelGGL((in_bwd_apply_kernel<T, TX>), pix_grid(dtype, B, H * W, C), dim3(256), 0,
                                          (hipStream_t)stream, (const TX*)x, x_ld, (const T*)dout, d_ld, pad, mean, rstd, s1, s2,
                                          act, (T*)dx, dx_ld, H, W, C));
    return ctg_launch_status();
}

// IN backward in ONE launch from partial sums part[B][nslabs <= 128][C][2] (ctg_in_bwd_partial's, or a fused conv epilogue's:
// ctg_conv_epilogue.bstats): the sums are finalized in the elementwise kernel's prologue.
extern "C" int ctg_in_bwd_stats(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad, const float* mean,
                                const float* rstd, int act, void* dx, int dx_ld, int B, int H, int W, int C, int nslabs,
                                const float* part, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || nslabs < 1 || nslabs > FUSED_MAX_SLABS || part == nullptr || pad < 0 || pad >= H || pad >= W)
        return CTG_EINVAL;
    int cgc;
    const dim3 grid = fused_grid(dtype, B, H * W, C, nslabs, &cgc);
    DISPATCH_TX(dtype, hipLaunchKernelGGL((in_bwd_apply_part_kernel<T, TX>), grid, dim3(256), 0, (hipStream_t)stream, (const TX*)x,
                                          x_ld, (const T*)dout, d_ld, pad, mean, rstd, part, nslabs, 1.0f / (float)(H * W), act,
                                          (T*)dx, dx_ld, H, W, C, cgc));
    return ctg_launch_status();
}

extern "C" int ctg_grad_combine(int dtype, const void* a, int a_ld, const void* b, int b_ld, int pad, const void* yact,
                                int y_ld, int act, void* out, int o_ld, int B, int H, int W, int C, void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (C % epc || (a == nullptr && b == nullptr) || pad < 0 || pad >= H || pad >= W) return CTG_EINVAL;
    const long items = (long)B * H * W * (C / epc);
    if (items >= (1L << 31)) return CTG_EINVAL;   // the kernels decode item indices in 32 bits
    DISPATCH_T(dtype, hipLaunchKernelGGL((grad_combine_kernel<T>), dim3(ew_blocks(items)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)a, a_ld, (const T*)b, b_ld, pad,
                                         (const T*)yact, y_ld, act, (T*)out, o_ld, H, W, C, items));
    return ctg_launch_status();
}

extern "C" int ctg_act_bwd_f32(const float* g, const float* y, int act, float* out, long n, void* stream) {
    CTG_ENTER();
    if (n < 1 || g == nullptr || y == nullptr || out == nullptr) return CTG_EINVAL;
    hipLaunchKernelGGL(act_bwd_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, y, act, out, n);
    return ctg_launch_status();
}

// db[c] (+)= sum_{n,y,x} fold(g)[n,y,x,c]   (bias gradient of a conv without a following InstanceNorm)
extern "C" int ctg_bias_grad(int dtype, const void* g, int g_ld, int pad, int B, int H, int W, int C, int Creal,
                             int nslabs, float* part, float* db, int accumulate, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || nslabs < 1 || nslabs > MAX_SLABS || Creal > C) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((moments_partial_kernel<T, 2>), dim3(nslabs, B), dim3(256), 0, st,
                                         (const T*)nullptr, 0, (const T*)g, g_ld, pad, (const float*)nullptr,
                                         (const float*)nullptr, 0, H, W, C, part));
    hipLaunchKernelGGL(bias_finalize_kernel, dim3(Creal), dim3(64), 0, st, part, B, nslabs, C, Creal, db,
                       accumulate);
    return ctg_launch_status();
}

// gm = fold(g) * act'(y) written to gout, db[c] (+)= sum gm: the activation backward and the bias gradient of a
// conv + bias + (Leaky)ReLU layer in one pass (ctg_grad_combine + ctg_bias_grad otherwise)
extern "C" int ctg_bias_grad_act(int dtype, const void* g, int g_ld, int pad, const void* yact, int y_ld, int act, void* gout,
                                 int go_ld, int B, int H, int W, int C, int Creal, int nslabs, float* part, float* db,
                                 int accumulate, void* stream) {
    CTG_ENTER();
    if (check_c(dtype, C) || nslabs < 1 || nslabs > MAX_SLABS || Creal > C || yact == nullptr || gout == nullptr) return CTG_EINVAL;
    if (act != ACT_RELU && act != ACT_LRELU) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((moments_partial_kernel<T, 3>), dim3(nslabs, B), dim3(256), 0, st,
                                         (const T*)yact, y_ld, (const T*)g, g_ld, pad, (const float*)nullptr,
                                         (const float*)nullptr, act, H, W, C, part, (T*)gout, go_ld));
    hipLaunchKernelGGL(bias_finalize_kernel, dim3(Creal), dim3(64), 0, st, part, B, nslabs, C, Creal, db, accumulate);
    return ctg_launch_status();
}

// out[B][H][W][C] = fold(dp[B][H+2p][W+2p][C]), fp32, any small C
extern "C" int ctg_fold_f32(const float* dp, float* out, int B, int H, int W, int C, int pad, void* stream) {
    CTG_ENTER();
    if (pad < 1 || pad >= H || pad >= W || C < 1) return CTG_EINVAL;
    hipLaunchKernelGGL(fold_f32_kernel, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dp,
                       out, B, H, W, C, pad);
    return ctg_launch_status();
}
