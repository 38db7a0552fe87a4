// Spatial transformer (trainer/transformer.py:11-31), smoothness loss
// (trainer/utils.py:165-173), L1 / masked-L1 (trainer/HdTrainer.py:721,726-735),
// global average pool of the PatchGAN map (Model/HdGan.py:145,279), weight
// packing and the multi-tensor Adam step (torch.optim.Adam(betas=(0.5,0.999)),
// trainer/HdTrainer.py:612-616).  fp32 throughout: these touch 1- and 2-channel
// 512x512 maps and the 16 M parameters, i.e. pure HBM streaming.
#include "common.h"

static inline int ew_blocks(long items) {
    long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// block-wide sum -> one partial per block (fixed order => deterministic)
__device__ __forceinline__ void block_partial(float v, float* __restrict__ part) {
    __shared__ float ws[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(256) void sum_finalize_kernel(const float* __restrict__ part, int n, float scale,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ double ws[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float r = (float)(((ws[0] + ws[1]) + (ws[2] + ws[3])) * (double)scale);
        out[0] = accumulate ? out[0] + r : r;
    }
}

// ------------------------------------------------------------------ grid sample (border, align_corners=True)
struct WarpGeom {
    float iy, ix, my, mx;  // clipped source coords and d(coord)/d(flow) multipliers
};

// Same float op sequence as the reference: grid + flow -> 2*(./(s-1) - 0.5) (transformer.py:22-25)
// -> ATen grid_sampler_unnormalize ((c+1)/2*(s-1)) -> clip_coordinates.  Near-integer coordinates
// (the flow starts at ~1e-5) round the same way, so the floor() cell and hence d/dflow agree.
__device__ __forceinline__ void warp_coord(float g, float f, int size, float& c, float& mult) {
    const float sm1 = (float)(size - 1);
    float loc = g + f;
    loc = 2.f * (loc / sm1 - 0.5f);
    float u = ((loc + 1.f) / 2.f) * sm1;
    float m = sm1 / 2.f;  // unnormalize grad
    if (u <= 0.f) { u = 0.f; m = 0.f; }
    else if (u >= sm1) { u = sm1; m = 0.f; }
    c = u;
    mult = (m * 2.f) / sm1;  // chain through 2*(x/(s-1) - 0.5)
}

__global__ void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow, long fs_n, long fs_c,
                                long fs_y, long fs_x, float* __restrict__ out, int B, int H, int W) {
    const long total = (long)B * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int n = (int)(i / ((long)W * H));
        const float* fl = flow + n * fs_n + y * fs_y + x * fs_x;
        float iy, ix, my, mx;
        warp_coord((float)y, fl[0], H, iy, my);
        warp_coord((float)x, fl[fs_c], W, ix, mx);
        const float fy = floorf(iy), fx = floorf(ix);
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
        const float nw = ((fx + 1.f) - ix) * ((fy + 1.f) - iy), ne = (ix - fx) * ((fy + 1.f) - iy);
        const float sw = ((fx + 1.f) - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
        const float* s = src + (size_t)n * H * W;
        float r = s[(size_t)y0 * W + x0] * nw;
        if (x1 < W) r += s[(size_t)y0 * W + x1] * ne;
        if (y1 < H) r += s[(size_t)y1 * W + x0] * sw;
        if (y1 < H && x1 < W) r += s[(size_t)y1 * W + x1] * se;
        out[i] = r;
    }
}

__global__ void zero_f32_kernel(float* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

// d_src must be zero-filled by the caller (scatter with float atomics: last-bit order dependence only);
// d_flow is written with the same strides as flow.
__global__ void warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ flow, long fs_n, long fs_c,
                                long fs_y, long fs_x, const float* __restrict__ gout, float* __restrict__ dsrc,
                                float* __restrict__ dflow, int B, int H, int W) {
    const long total = (long)B * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int n = (int)(i / ((long)W * H));
        const long fo = n * fs_n + y * fs_y + x * fs_x;
        float iy, ix, my, mx;
        warp_coord((float)y, flow[fo], H, iy, my);
        warp_coord((float)x, flow[fo + fs_c], W, ix, mx);
        const float fy = floorf(iy), fx = floorf(ix);
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
        const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
        const float g = gout[i];
        const float* s = src + (size_t)n * H * W;
        float* ds = dsrc ? dsrc + (size_t)n * H * W : nullptr;
        const bool bx = x1 < W, by = y1 < H;
        const float vnw = s[(size_t)y0 * W + x0];
        const float vne = bx ? s[(size_t)y0 * W + x1] : 0.f;
        const float vsw = by ? s[(size_t)y1 * W + x0] : 0.f;
        const float vse = (bx && by) ? s[(size_t)y1 * W + x1] : 0.f;
        if (ds) {
            atomicAdd(ds + (size_t)y0 * W + x0, wx0 * wy0 * g);
            if (bx) atomicAdd(ds + (size_t)y0 * W + x1, wx1 * wy0 * g);
            if (by) atomicAdd(ds + (size_t)y1 * W + x0, wx0 * wy1 * g);
            if (bx && by) atomicAdd(ds + (size_t)y1 * W + x1, wx1 * wy1 * g);
        }
        if (dflow) {
            const float gix = (-vnw * wy0 + vne * wy0 - vsw * wy1 + vse * wy1) * g;
            const float giy = (-vnw * wx0 - vne * wx1 + vsw * wx0 + vse * wx1) * g;
            dflow[fo] = my * giy;
            dflow[fo + fs_c] = mx * gix;
        }
    }
}

// ---- deterministic scatter (test / debug mode; SURVEY.md section 7 "reduction order").  Float atomics make d_src depend on
// the order in which the four bilinear contributions of 260 k pixels arrive; INTEGER addition is associative, so the same
// scatter in fixed point is bit-identical from run to run whatever the order: contributions are scaled by 2^(38 - e), e the
// exponent of max |gout| (an order-independent atomicMax on the float bits), rounded to int64 and added with 64-bit integer
// atomics; a last pass converts back.  Resolution 2^-38 of the largest incoming gradient (fp32 itself: 2^-24 of each value),
// headroom for 2^24 contributions per destination.
__global__ void warp_det_prep_kernel(const float* __restrict__ gout, long long* __restrict__ acc, unsigned* __restrict__ maxbits,
                                     long n) {
    unsigned m = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        acc[i] = 0;
        const unsigned b = __float_as_uint(gout[i]) & 0x7fffffffu;
        m = b > m ? b : m;       // |g| as bits: ordered like the values for finite inputs; inf / nan stay the largest
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(maxbits, m);
}

__device__ __forceinline__ float det_scale(unsigned maxbits, bool inverse) {
    int e = (int)(maxbits >> 23) - 127;                    // floor(log2 max|g|) (denormal / zero maxima: -127)
    e = e < -88 ? -88 : (e > 88 ? 88 : e);                 // 38 - e + 127 and e - 38 + 127 both stay inside the exponent field [1, 254]
    return __uint_as_float((unsigned)((inverse ? e - 38 : 38 - e) + 127) << 23);       // 2^(38 - e) or its inverse, exact
}

__global__ void warp_bwd_det_kernel(const float* __restrict__ flow, long fs_n, long fs_c, long fs_y, long fs_x,
                                    const float* __restrict__ gout, long long* __restrict__ acc,
                                    const unsigned* __restrict__ maxbits, int B, int H, int W) {
    const long total = (long)B * H * W;
    const float sc = det_scale(maxbits[0], false);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int n = (int)(i / ((long)W * H));
        const long fo = n * fs_n + y * fs_y + x * fs_x;
        float iy, ix, my, mx;
        warp_coord((float)y, flow[fo], H, iy, my);
        warp_coord((float)x, flow[fo + fs_c], W, ix, mx);
        const float fy = floorf(iy), fx = floorf(ix);
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
        const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
        const float g = gout[i] * sc;
        unsigned long long* ds = reinterpret_cast<unsigned long long*>(acc + (size_t)n * H * W);
        const bool bx = x1 < W, by = y1 < H;
        // (two's complement: adding the unsigned image of a negative value is the signed addition)
        atomicAdd(ds + (size_t)y0 * W + x0, (unsigned long long)__float2ll_rn(wx0 * wy0 * g));
        if (bx) atomicAdd(ds + (size_t)y0 * W + x1, (unsigned long long)__float2ll_rn(wx1 * wy0 * g));
        if (by) atomicAdd(ds + (size_t)y1 * W + x0, (unsigned long long)__float2ll_rn(wx0 * wy1 * g));
        if (bx && by) atomicAdd(ds + (size_t)y1 * W + x1, (unsigned long long)__float2ll_rn(wx1 * wy1 * g));
    }
}

__global__ void warp_det_finish_kernel(const long long* __restrict__ acc, const unsigned* __restrict__ maxbits,
                                       float* __restrict__ dsrc, long n) {
    const float inv = det_scale(maxbits[0], true);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dsrc[i] = (float)acc[i] * inv;
}

// ------------------------------------------------------------------ smoothness
__global__ void smooth_fwd_kernel(const float* __restrict__ f, long sn, long sc, long sy, long sx, int B, int C,
                                  int H, int W, float inv_nx, float inv_ny, float* __restrict__ part) {
    const long total = (long)B * C * H * W;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((long)W * H)) % C);
        const int n = (int)(i / ((long)W * H * C));
        const float* p = f + n * sn + c * sc + y * sy + x * sx;
        const float v = p[0];
        if (x + 1 < W) { const float d = p[sx] - v; acc += d * d * inv_nx; }
        if (y + 1 < H) { const float d = p[sy] - v; acc += d * d * inv_ny; }
    }
    block_partial(acc, part);
}

// df = gscale[0] * d(mean(dx^2) + mean(dy^2))/df, written with the same strides
__global__ void smooth_bwd_kernel(const float* __restrict__ f, long sn, long sc, long sy, long sx, int B, int C,
                                  int H, int W, float inv_nx, float inv_ny, const float* __restrict__ gscale,
                                  float* __restrict__ df, int accumulate) {
    const long total = (long)B * C * H * W;
    const float g = gscale[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)((i / ((long)W * H)) % C);
        const int n = (int)(i / ((long)W * H * C));
        const long o = n * sn + c * sc + y * sy + x * sx;
        const float v = f[o];
        float r = 0.f;
        if (x + 1 < W) r -= 2.f * (f[o + sx] - v) * inv_nx;
        if (x > 0) r += 2.f * (v - f[o - sx]) * inv_nx;
        if (y + 1 < H) r -= 2.f * (f[o + sy] - v) * inv_ny;
        if (y > 0) r += 2.f * (v - f[o - sy]) * inv_ny;
        df[o] = accumulate ? df[o] + g * r : g * r;
    }
}

// ------------------------------------------------------------------ L1 and the stage-2 masked L1
// mask == nullptr: plain mean |a - b|.  Otherwise (HdTrainer.py:726-735):
//   bb = (mask >= 0.3);  bm = b*bb, bm[bm == 0] = -1;  am = a*bb, am[am == 0] = -1;  mean |am - bm|
__device__ __forceinline__ float l1_term(float a, float b, const float* mask, long i, float& dsign) {
    if (mask == nullptr) {
        const float d = a - b;
        dsign = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        return fabsf(d);
    }
    const float bb = mask[i] < 0.3f ? 0.f : 1.f;
    float bm = b * bb;
    if (bm == 0.f) bm = -1.f;
    const float ab = a * bb;
    const float am = ab == 0.f ? -1.f : ab;
    const float d = am - bm;
    dsign = ab == 0.f ? 0.f : bb * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));  // constant where overwritten
    return fabsf(d);
}

__global__ void l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ mask, long n, float* __restrict__ part) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float ds;
        acc += l1_term(a[i], b[i], mask, i, ds);
    }
    block_partial(acc, part);
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ mask, long n, const float* __restrict__ gscale, float inv_n,
                              float* __restrict__ da, int accumulate) {
    const float g = gscale[0] * inv_n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float ds;
        l1_term(a[i], b[i], mask, i, ds);
        da[i] = accumulate ? da[i] + g * ds : g * ds;
    }
}

// ------------------------------------------------------------------ global average pool of a 1-channel map
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, int HW, float* __restrict__ out) {
    __shared__ float ws[4];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) acc += x[(size_t)n * HW + i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = ((ws[0] + ws[1]) + (ws[2] + ws[3])) / (float)HW;
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ gout, int HW, float* __restrict__ dx, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        dx[i] = gout[i / HW] / (float)HW;
}

// ------------------------------------------------------------------ LSGAN loss on the PatchGAN map, fused
// loss = sum_b s_b (mean_hw x[b] - t_b)^2 with (t_b, s_b) = (t0, s0) for b < nb, (t1, s1) otherwise: GANLoss
// (Model/HdGan.py:276-285: w_i * MSE(avg_pool(map), target)) with the loss weight folded in, and -- nb < B -- the D step's
// fake and real halves of one batched discriminator pass (HdTrainer.py:745-747) in one go.  pooled[B] is kept for the backward.
// mode 0: (p - t)^2 (nn.MSELoss, the LSGAN default); mode 1: -(t log p + (1 - t) log(1 - p)) with both logs clamped at -100
// (nn.BCELoss, GANLoss(use_lsgan=False) on a sigmoid discriminator)
__device__ __forceinline__ double gan_term(double p, double t, int mode) {
    if (mode == 0) return (p - t) * (p - t);
    const double lp = fmax(log(p), -100.0), lq = fmax(log(1.0 - p), -100.0);
    return -(t * lp + (1.0 - t) * lq);
}
__device__ __forceinline__ float gan_term_grad(float p, float t, int mode) {
    if (mode == 0) return 2.f * (p - t);
    return (p - t) / fmaxf(p * (1.f - p), 1e-12f);        // torch's binary_cross_entropy backward (EPSILON 1e-12)
}

__global__ __launch_bounds__(256) void lsgan_finalize_kernel(const float* __restrict__ pooled, int B, int nb, float t0,
                                                             float s0, float t1, float s1, int mode,
                                                             float* __restrict__ out) {
    __shared__ double ws[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += 256)
        s += (double)(b < nb ? s0 : s1) * gan_term((double)pooled[b], (double)(b < nb ? t0 : t1), mode);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((ws[0] + ws[1]) + (ws[2] + ws[3]));
}

__global__ void lsgan_bwd_kernel(const float* __restrict__ pooled, int HW, int nb, float t0, float s0, float t1, float s1,
                                 int mode, const float* __restrict__ gscale, float* __restrict__ dx, long total) {
    const float g = gscale[0] / (float)HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        dx[i] = g * (b < nb ? s0 : s1) * gan_term_grad(pooled[b], b < nb ? t0 : t1, mode);
    }
}

// out = sum of n (<= 8) device scalars, in argument order
struct ScalarList { const float* p[8]; int n; };
__global__ void sum_scalars_kernel(const ScalarList L, float* __restrict__ out) {
    float s = 0.f;
    for (int i = 0; i < L.n; ++i) s += L.p[i][0];
    out[0] = s;
}

// out[i] = g[i] * act'(y[i]) and, in the same pass, per-block partial sums of out (-> the bias gradient of a 1-channel conv)
__global__ void act_bwd_sum_kernel(const float* __restrict__ g, const float* __restrict__ y, int act,
                                   float* __restrict__ out, long n, float* __restrict__ part) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = g[i] * act_grad_from_out(y[i], act);
        out[i] = v;
        acc += v;
    }
    block_partial(acc, part);
}

// ------------------------------------------------------------------ weight pack
// dst[t][n][k] = src[n*sn + k*sk + t*st] (zero for n >= Nreal or k >= Kreal); dst is T, src fp32 master
template <typename T>
__global__ void weight_pack_kernel(const float* __restrict__ src, long sn, long sk, long stp, int Nreal, int Kreal,
                                   T* __restrict__ dst, int ntaps, int Npad, int Kpad) {
    const long total = (long)ntaps * Npad * Kpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const int n = (int)((i / Kpad) % Npad);
        const int t = (int)(i / ((long)Kpad * Npad));
        const float v = (n < Nreal && k < Kreal) ? src[n * sn + k * sk + t * stp] : 0.f;
        st1(dst + i, v);
    }
}

// many weight tensors in ONE launch (a network's packs after an optimiser step): job j owns blocks
// [first[j], first[j+1]) of the grid
#define PACK_MAX_T 24
struct PackList {
    const float* src[PACK_MAX_T];
    void* dst[PACK_MAX_T];
    long sn[PACK_MAX_T], sk[PACK_MAX_T], stp[PACK_MAX_T];
    int nreal[PACK_MAX_T], kreal[PACK_MAX_T], ntaps[PACK_MAX_T], npad[PACK_MAX_T], kpad[PACK_MAX_T];
    int first[PACK_MAX_T + 1];
    int count;
};
#define PACK_CHUNK 2048   // elements per block

template <typename T>
__global__ __launch_bounds__(256) void weight_pack_multi_kernel(const PackList L) {
    int j = 0;
    while (j + 1 < L.count && (int)blockIdx.x >= L.first[j + 1]) ++j;
    const int Kpad = L.kpad[j], Npad = L.npad[j], Nreal = L.nreal[j], Kreal = L.kreal[j];
    const long sn = L.sn[j], sk = L.sk[j], stp = L.stp[j];
    const float* __restrict__ src = L.src[j];
    T* __restrict__ dst = (T*)L.dst[j];
    const unsigned total = (unsigned)L.ntaps[j] * Npad * Kpad;
    const unsigned base = ((unsigned)blockIdx.x - L.first[j]) * PACK_CHUNK;
#pragma unroll 2
    for (unsigned e = threadIdx.x; e < PACK_CHUNK; e += 256) {
        const unsigned i = base + e;
        if (i >= total) break;
        const unsigned k = i % (unsigned)Kpad, r = i / (unsigned)Kpad;
        const unsigned n = r % (unsigned)Npad, t = r / (unsigned)Npad;
        const float v = ((int)n < Nreal && (int)k < Kreal) ? src[n * sn + k * sk + t * stp] : 0.f;
        st1(dst + i, v);
    }
}

// ------------------------------------------------------------------ Adam (multi-tensor, fp32)
#define ADAM_MAX_T 24
struct AdamList {
    float* p[ADAM_MAX_T];
    const float* g[ADAM_MAX_T];
    float* m[ADAM_MAX_T];
    float* v[ADAM_MAX_T];
    int n[ADAM_MAX_T];
    int count;
};
#define ADAM_CHUNK 4096

// torch.optim.Adam's update order: m.lerp_(g, 1-b1); v = b2*v + (1-b2) g^2;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// dev_bc != nullptr: bias corrections come from device memory (written by adam_tick_kernel), so the launch
// carries no per-step host constant and can be replayed from a captured hipGraph
__global__ __launch_bounds__(256) void adam_kernel(const AdamList L, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, const float* __restrict__ dev_bc) {
    if (dev_bc != nullptr) {
        bc1 = dev_bc[1];
        bc2_sqrt = dev_bc[2];
    }
    const int t = blockIdx.y;
    if (t >= L.count) return;
    const int n = L.n[t];
    const int beg = blockIdx.x * ADAM_CHUNK;
    if (beg >= n) return;
    const int end = min(beg + ADAM_CHUNK, n);
    float* __restrict__ p = L.p[t];
    const float* __restrict__ g = L.g[t];
    float* __restrict__ m = L.m[t];
    float* __restrict__ v = L.v[t];
    const float step = lr / bc1;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.f - b1);
        vi = vi * b2 + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

// state = {step, 1 - b1^step, sqrt(1 - b2^step)}: advance by one optimiser step on the device
__global__ void adam_tick_kernel(float* __restrict__ state, float b1, float b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double step = (double)state[0] + 1.0;
        state[0] = (float)step;
        state[1] = (float)(1.0 - pow((double)b1, step));
        state[2] = (float)sqrt(1.0 - pow((double)b2, step));
    }
}

// =================================================================== C ABI
extern "C" int ctg_warp_fwd(const float* src, const float* flow, long fs_n, long fs_c, long fs_y, long fs_x,
                            float* out, int B, int H, int W, void* stream) {
    CTG_ENTER();
    if (H < 2 || W < 2) return CTG_EINVAL;
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(ew_blocks((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                       flow, fs_n, fs_c, fs_y, fs_x, out, B, H, W);
    return ctg_launch_status();
}

extern "C" int ctg_warp_bwd(const float* src, const float* flow, long fs_n, long fs_c, long fs_y, long fs_x,
                            const float* gout, float* dsrc, float* dflow, int B, int H, int W, void* det_ws, void* stream) {
    CTG_ENTER();
    if (H < 2 || W < 2) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dsrc && det_ws) {
        // deterministic mode: d_src by fixed-point integer atomics (bit-identical from run to run), d_flow (a gather) as always
        if ((uintptr_t)det_ws & 7) return CTG_EINVAL;
        const long n = (long)B * H * W;
        long long* acc = (long long*)det_ws;
        unsigned* maxbits = (unsigned*)(acc + n);
        hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(64), 0, st, (float*)maxbits, 2L);
        hipLaunchKernelGGL(warp_det_prep_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, gout, acc, maxbits, n);
        hipLaunchKernelGGL(warp_bwd_det_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, flow, fs_n, fs_c, fs_y, fs_x, gout, acc,
                           maxbits, B, H, W);
        hipLaunchKernelGGL(warp_det_finish_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, acc, maxbits, dsrc, n);
        if (dflow)
            hipLaunchKernelGGL(warp_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, src, flow, fs_n, fs_c, fs_y, fs_x, gout,
                               (float*)nullptr, dflow, B, H, W);
        return ctg_launch_status();
    }
    if (dsrc) {
        // zero-fill by a KERNEL, not hipMemsetAsync: as a memset node of a captured hipGraph the fill was observed to race with
        // the scatter below on replay (garbage / inf gradients of the generator at the third replay, intermittently)
        hipLaunchKernelGGL(zero_f32_kernel, dim3(ew_blocks((long)B * H * W / 4 + 1)), dim3(256), 0, st, dsrc, (long)B * H * W);
    }
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(ew_blocks((long)B * H * W)), dim3(256), 0, st, src, flow, fs_n, fs_c,
                       fs_y, fs_x, gout, dsrc, dflow, B, H, W);
    return ctg_launch_status();
}

// part: >= 4096 floats of scratch; out: 1 float
extern "C" int ctg_smooth_fwd(const float* f, long sn, long sc, long sy, long sx, int B, int C, int H, int W,
                              float weight, float* part, float* out, void* stream) {
    CTG_ENTER();
    if (H < 2 || W < 2) return CTG_EINVAL;
    const long total = (long)B * C * H * W;
    const int nb = ew_blocks(total);
    const float inv_nx = 1.f / (float)((long)B * C * H * (W - 1)), inv_ny = 1.f / (float)((long)B * C * (H - 1) * W);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nb), dim3(256), 0, st, f, sn, sc, sy, sx, B, C, H, W, inv_nx, inv_ny,
                       part);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nb, weight, out, 0);
    return ctg_launch_status();
}

extern "C" int ctg_smooth_bwd(const float* f, long sn, long sc, long sy, long sx, int B, int C, int H, int W,
                              float weight, const float* gscale, float* df, int accumulate, void* stream) {
    CTG_ENTER();
    if (H < 2 || W < 2) return CTG_EINVAL;
    const long total = (long)B * C * H * W;
    const float inv_nx = weight / (float)((long)B * C * H * (W - 1)), inv_ny = weight / (float)((long)B * C * (H - 1) * W);
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, f, sn, sc, sy, sx,
                       B, C, H, W, inv_nx, inv_ny, gscale, df, accumulate);
    return ctg_launch_status();
}

extern "C" int ctg_l1_fwd(const float* a, const float* b, const float* mask, long n, float weight, float* part,
                          float* out, void* stream) {
    CTG_ENTER();
    if (n < 1) return CTG_EINVAL;
    const int nb = ew_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(nb), dim3(256), 0, st, a, b, mask, n, part);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nb, weight / (float)n, out, 0);
    return ctg_launch_status();
}

extern "C" int ctg_l1_bwd(const float* a, const float* b, const float* mask, long n, float weight, const float* gscale,
                          float* da, int accumulate, void* stream) {
    CTG_ENTER();
    if (n < 1) return CTG_EINVAL;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, mask, n, gscale,
                       weight / (float)n, da, accumulate);
    return ctg_launch_status();
}

extern "C" int ctg_avgpool_fwd(const float* x, int B, int HW, float* out, void* stream) {
    CTG_ENTER();
    if (B < 1 || HW < 1) return CTG_EINVAL;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, HW, out);
    return ctg_launch_status();
}

extern "C" int ctg_lsgan_fwd(const float* x, int B, int HW, int nb, float t0, float s0, float t1, float s1, int mode,
                             float* pooled, float* out, void* stream) {
    CTG_ENTER();
    if (B < 1 || HW < 1 || nb < 0 || nb > B || (mode != 0 && mode != 1)) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(B), dim3(256), 0, st, x, HW, pooled);
    hipLaunchKernelGGL(lsgan_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)pooled, B, nb, t0, s0, t1, s1, mode, out);
    return ctg_launch_status();
}

extern "C" int ctg_lsgan_bwd(const float* pooled, int B, int HW, int nb, float t0, float s0, float t1, float s1, int mode,
                             const float* gscale, float* dx, void* stream) {
    CTG_ENTER();
    if (B < 1 || HW < 1 || nb < 0 || nb > B || (mode != 0 && mode != 1)) return CTG_EINVAL;
    const long total = (long)B * HW;
    hipLaunchKernelGGL(lsgan_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, pooled, HW, nb, t0, s0,
                       t1, s1, mode, gscale, dx, total);
    return ctg_launch_status();
}

extern "C" int ctg_sum_scalars(int count, const void* const* scalars, float* out, void* stream) {
    CTG_ENTER();
    if (count < 1 || count > 8) return CTG_EINVAL;
    ScalarList L;
    for (int i = 0; i < 8; ++i) L.p[i] = i < count ? (const float*)scalars[i] : nullptr;
    L.n = count;
    hipLaunchKernelGGL(sum_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, L, out);
    return ctg_launch_status();
}

extern "C" int ctg_act_bwd_sum_f32(const float* g, const float* y, int act, float* out, long n, float* part, float* sum_out,
                                   int accumulate, void* stream) {
    CTG_ENTER();
    if (n < 1 || g == nullptr || y == nullptr || out == nullptr || part == nullptr || sum_out == nullptr) return CTG_EINVAL;
    const int nb = ew_blocks(n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(act_bwd_sum_kernel, dim3(nb), dim3(256), 0, st, g, y, act, out, n, part);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nb, 1.f, sum_out, accumulate);
    return ctg_launch_status();
}

extern "C" int ctg_avgpool_bwd(const float* gout, int B, int HW, float* dx, void* stream) {
    CTG_ENTER();
    if (B < 1 || HW < 1) return CTG_EINVAL;
    const long total = (long)B * HW;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, gout, HW, dx,
                       total);
    return ctg_launch_status();
}

extern "C" int ctg_weight_pack(int dtype, const float* src, long sn, long sk, long stp, int Nreal, int Kreal,
                               void* dst, int ntaps, int Npad, int Kpad, void* stream) {
    CTG_ENTER();
    if (Nreal > Npad || Kreal > Kpad || ntaps < 1) return CTG_EINVAL;
    const long total = (long)ntaps * Npad * Kpad;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL((weight_pack_kernel<bf16_t>), dim3(ew_blocks(total)), dim3(256), 0, st, src, sn, sk, stp,
                           Nreal, Kreal, (bf16_t*)dst, ntaps, Npad, Kpad);
    else if (dtype == DT_F32)
        hipLaunchKernelGGL((weight_pack_kernel<float>), dim3(ew_blocks(total)), dim3(256), 0, st, src, sn, sk, stp,
                           Nreal, Kreal, (float*)dst, ntaps, Npad, Kpad);
    else return CTG_EINVAL;
    return ctg_launch_status();
}

// ctg_weight_pack for `count` tensors given as parallel host arrays (one launch per 24 tensors)
extern "C" int ctg_weight_pack_multi(int dtype, int count, const void* const* src, void* const* dst, const long* sn,
                                     const long* sk, const long* stp, const int* nreal, const int* kreal,
                                     const int* ntaps, const int* npad, const int* kpad, void* stream) {
    CTG_ENTER();
    if (count < 0 || (dtype != DT_BF16 && dtype != DT_F32)) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < count; base += PACK_MAX_T) {
        PackList L;
        L.count = count - base < PACK_MAX_T ? count - base : PACK_MAX_T;
        int blocks = 0;
        for (int i = 0; i < L.count; ++i) {
            const int q = base + i;
            if (nreal[q] > npad[q] || kreal[q] > kpad[q] || ntaps[q] < 1 || src[q] == nullptr || dst[q] == nullptr) return CTG_EINVAL;
            const long total = (long)ntaps[q] * npad[q] * kpad[q];
            if (total >= (1L << 31)) return CTG_EINVAL;
            L.src[i] = (const float*)src[q]; L.dst[i] = dst[q];
            L.sn[i] = sn[q]; L.sk[i] = sk[q]; L.stp[i] = stp[q];
            L.nreal[i] = nreal[q]; L.kreal[i] = kreal[q]; L.ntaps[i] = ntaps[q]; L.npad[i] = npad[q]; L.kpad[i] = kpad[q];
            L.first[i] = blocks;
            blocks += (int)((total + PACK_CHUNK - 1) / PACK_CHUNK);
        }
        L.first[L.count] = blocks;
        if (blocks == 0) continue;
        if (dtype == DT_BF16) hipLaunchKernelGGL((weight_pack_multi_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, L);
        else hipLaunchKernelGGL((weight_pack_multi_kernel<float>), dim3(blocks), dim3(256), 0, st, L);
    }
    return ctg_launch_status();
}

// One Adam step over `count` fp32 tensors given as parallel host arrays of device pointers and sizes.
// `step` is the 1-based step index (bias corrections are computed here, on the host, in double).
extern "C" int ctg_adam_tick(float* state3, float beta1, float beta2, void* stream) {
    CTG_ENTER();
    if (state3 == nullptr) return CTG_EINVAL;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state3, beta1, beta2);
    return ctg_launch_status();
}

extern "C" int ctg_adam_step(int count, void* const* params, const void* const* grads, void* const* exp_avg,
                             void* const* exp_avg_sq, const long* numel, float lr, float beta1, float beta2, float eps,
                             int step, const float* dev_state3, void* stream) {
    CTG_ENTER();
    if (count < 0 || (step < 1 && dev_state3 == nullptr)) return CTG_EINVAL;
    if (step < 1) step = 1;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < count; base += ADAM_MAX_T) {
        AdamList L;
        L.count = count - base < ADAM_MAX_T ? count - base : ADAM_MAX_T;
        long maxn = 0;
        for (int i = 0; i < L.count; ++i) {
            if (numel[base + i] > 0x7fffffffL) return CTG_EINVAL;
            L.p[i] = (float*)params[base + i];
            L.g[i] = (const float*)grads[base + i];
            L.m[i] = (float*)exp_avg[base + i];
            L.v[i] = (float*)exp_avg_sq[base + i];
            L.n[i] = (int)numel[base + i];
            if (numel[base + i] > maxn) maxn = numel[base + i];
        }
        if (maxn == 0) continue;
        dim3 grid((unsigned)((maxn + ADAM_CHUNK - 1) / ADAM_CHUNK), L.count);
        hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, st, L, lr, beta1, beta2, eps, (float)bc1,
                           (float)sqrt(bc2), dev_state3);
    }
    return ctg_launch_status();
}
