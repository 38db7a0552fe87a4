// The PatchGAN's last layer: Conv2d(512, 1, 4, padding=1) -- Model/HdGan.py:136-137 (every scale of Discriminator_m),
// Model/HdGan.py:217-219 (NLayerDiscriminator), Model/CycleGan.py:104 (Discriminator) -- forward, input gradient and weight gradient.
//
// On the matrix-core kernels this layer is a GEMM with ONE useful column: the forward rode conv_halo_kernel<BN = 16> (15 of 16
// output columns zero), the backward zero-padded the 1-channel gradient to 32 channels for a 32 -> 512 conv and a 32 x 512 weight
// gradient: 95 + 68 + 113 us per call in bf16 and 200 + 120 + 340 us on split-pair operands (rounds 1-4; 0.08-0.23 of their
// bounds, profiles/r05_conv_roofline_*.md) for a layer whose whole input is 67 / 134 MB.  It is a matrix-VECTOR product per pixel;
// these kernels do it on the vector ALUs in fp32, reading the wide tensor once:
//   * a lane owns 8 of the 512 channels (one 16-byte chunk per pixel and plane), a wave a run of 16 pixels of one row;
//   * forward: 16 output accumulators per lane (its 8-channel share of each output pixel of the run), every input pixel of the
//     4 x 19 window loaded once and used for up to 4 outputs; ONE cross-lane reduction per 16 outputs;
//   * input gradient: the 4 x 19 patch of the scalar gradient map broadcast to every lane, dx[pixel][8 channels] = sum over the 16
//     taps of g x w: 128 FMAs per pixel and lane, stored as one chunk per plane;
//   * weight gradient: the same patch, dw[tap][8 channels] += g x x over the wave's pixels (128 accumulators per lane), the four
//     waves of a workgroup summed through LDS in a fixed order, one fp32 partial per workgroup for ctg_wgrad_reduce.
// Weights stay fp32 (the master copy: no rounding at all); split-pair operands enter as hi + lo (exact in fp32), so in the
// "bf16x3" mode these three launches are closer to the fp32 reference than the three-product MFMA form they replace.
#include "common.h"

#define C1_CIN 512
#define C1_RUN 16

template <typename T, int KS>
__global__ __launch_bounds__(256, 2) void cout1_fwd_kernel(const T* __restrict__ x, int x_ld, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int act,
                                                            int Hi, int Wi, int pad, int Ho, int Wo, int ntask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntask) return;      // (no barrier in this kernel)
    const int segs = (Wo + C1_RUN - 1) / C1_RUN;
    const int seg = task % segs, oy = (task / segs) % Ho, b = task / (segs * Ho);
    const int ox0 = seg * C1_RUN;
    float wr[KS * KS][8];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + t * C1_CIN + 8 * lane);
        const f32x4 c = *reinterpret_cast<const f32x4*>(w + t * C1_CIN + 8 * lane + 4);
        wr[t][0] = a[0]; wr[t][1] = a[1]; wr[t][2] = a[2]; wr[t][3] = a[3];
        wr[t][4] = c[0]; wr[t][5] = c[1]; wr[t][6] = c[2]; wr[t][7] = c[3];
    }
    float acc[C1_RUN];
#pragma unroll
    for (int o = 0; o < C1_RUN; ++o) acc[o] = 0.f;
    const T* __restrict__ X = x + (size_t)b * Hi * Wi * x_ld + 8 * lane;
    static_for<KS>([&](auto kyc) {
        constexpr int ky = decltype(kyc)::value;
        const int iy = oy + ky - pad;
        if ((unsigned)iy < (unsigned)Hi) {      // wave-uniform
            static_for<C1_RUN + KS - 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int ix = ox0 + j - pad;
                if ((unsigned)ix < (unsigned)Wi) {      // wave-uniform
                    Chunk<T> c;
                    c.load(X + (size_t)(iy * Wi + ix) * x_ld, x_ld);
                    static_for<KS>([&](auto kxc) {
                        constexpr int kx = decltype(kxc)::value;
                        constexpr int o = j - kx;       // ox - ox0 of the output this input pixel feeds through tap (ky, kx)
                        if constexpr (o >= 0 && o < C1_RUN) {
                            float s = acc[o];
#pragma unroll
                            for (int q = 0; q < 8; ++q) s = fmaf(c.v[q], wr[ky * KS + kx][q], s);
                            acc[o] = s;
                        }
                    });
                }
            });
        }
    });
    float mine = 0.f;
#pragma unroll
    for (int o = 0; o < C1_RUN; ++o) {
        const float s = wave_sum(acc[o]);
        if (lane == o) mine = s;
    }
    if (lane < C1_RUN && ox0 + lane < Wo) {
        const float v = mine + (bias != nullptr ? bias[0] : 0.f);
        y[((size_t)b * Ho + oy) * Wo + ox0 + lane] = act_apply(v, act);
    }
}

// the KS x (16 + KS - 1) patch of the scalar map g around a run of 16 input pixels (row iy, columns ix0 ..): entry [ky][c] =
// g[iy - ky + pad][ix0 + c - (KS - 1) + pad] (0 outside the map) -- input pixel ix0 + i meets it through tap (ky, kx) at c = i - kx + KS - 1
template <int KS>
__device__ __forceinline__ void cout1_patch(const float* __restrict__ G, int Ho, int Wo, int iy, int ix0, int pad, int lane,
                                            float* __restrict__ sp, float (&pr)[KS][C1_RUN + KS - 1]) {
    constexpr int PW = C1_RUN + KS - 1;
    for (int i = lane; i < KS * PW; i += 64) {
        const int ky = i / PW, c = i - ky * PW;
        const int oy = iy - ky + pad, ox = ix0 + c - (KS - 1) + pad;
        sp[i] = ((unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo) ? G[(size_t)oy * Wo + ox] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int c = 0; c < PW; ++c) pr[ky][c] = sp[ky * PW + c];      // same address in every lane: a broadcast read
}

template <typename T, int KS>
__global__ __launch_bounds__(256, 2) void cout1_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                            T* __restrict__ dx, int dx_ld, int Hi, int Wi, int pad, int Ho,
                                                            int Wo, int ntask) {
    constexpr int PW = C1_RUN + KS - 1;
    __shared__ float spatch[4][KS * PW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wave;
    const bool live = task < ntask;         // (every wave reaches the barrier in cout1_patch)
    const int segs = (Wi + C1_RUN - 1) / C1_RUN;
    const int t2 = live ? task : 0;
    const int seg = t2 % segs, iy = (t2 / segs) % Hi, b = t2 / (segs * Hi);
    const int ix0 = seg * C1_RUN;
    float pr[KS][PW];
    cout1_patch<KS>(g + (size_t)b * Ho * Wo, Ho, Wo, iy, ix0, pad, lane, spatch[wave], pr);
    if (!live) return;
    float wr[KS * KS][8];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + t * C1_CIN + 8 * lane);
        const f32x4 c = *reinterpret_cast<const f32x4*>(w + t * C1_CIN + 8 * lane + 4);
        wr[t][0] = a[0]; wr[t][1] = a[1]; wr[t][2] = a[2]; wr[t][3] = a[3];
        wr[t][4] = c[0]; wr[t][5] = c[1]; wr[t][6] = c[2]; wr[t][7] = c[3];
    }
    T* __restrict__ D = dx + ((size_t)(b * Hi + iy) * Wi) * dx_ld + 8 * lane;
    static_for<C1_RUN>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (ix0 + i < Wi) {     // wave-uniform
            Chunk<T> c;
            c.zero();
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float gv = pr[ky][i - kx + KS - 1];
#pragma unroll
                    for (int q = 0; q < 8; ++q) c.v[q] = fmaf(gv, wr[ky * KS + kx][q], c.v[q]);
                }
            c.store(D + (size_t)(ix0 + i) * dx_ld, dx_ld);
        }
    });
}

template <typename T, int KS>
__global__ __launch_bounds__(256, 2) void cout1_wgrad_kernel(const float* __restrict__ g, const T* __restrict__ x, int x_ld,
                                                              float* __restrict__ part, int Hi, int Wi, int pad, int Ho, int Wo,
                                                              int ntask) {
    constexpr int PW = C1_RUN + KS - 1;
    __shared__ float spatch[4][KS * PW];
    __shared__ float red[KS * KS * C1_CIN];     // 32 KB for 4x4 taps
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int segs = (Wi + C1_RUN - 1) / C1_RUN;
    float acc[KS * KS][8];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[t][q] = 0.f;
    const int rounds = (ntask + gridDim.x * 4 - 1) / (gridDim.x * 4);      // the same for every wave: uniform barrier counts
    for (int r = 0; r < rounds; ++r) {
        const int task = (r * gridDim.x + blockIdx.x) * 4 + wave;
        const bool live = task < ntask;
        const int t2 = live ? task : 0;
        const int seg = t2 % segs, iy = (t2 / segs) % Hi, b = t2 / (segs * Hi);
        const int ix0 = seg * C1_RUN;
        float pr[KS][PW];
        __syncthreads();        // the previous round's broadcast reads are done before the patch is overwritten
        cout1_patch<KS>(g + (size_t)b * Ho * Wo, Ho, Wo, iy, ix0, pad, lane, spatch[wave], pr);
        if (live) {
            const T* __restrict__ X = x + ((size_t)(b * Hi + iy) * Wi) * x_ld + 8 * lane;
            static_for<C1_RUN>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (ix0 + i < Wi) {     // wave-uniform
                    Chunk<T> c;
                    c.load(X + (size_t)(ix0 + i) * x_ld, x_ld);
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            const float gv = pr[ky][i - kx + KS - 1];
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc[ky * KS + kx][q] = fmaf(gv, c.v[q], acc[ky * KS + kx][q]);
                        }
                }
            });
        }
    }
    // the four waves' sums in a fixed order (wave 0 stores, waves 1..3 add one after the other)
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < KS * KS; ++t)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float* p = red + t * C1_CIN + 8 * lane + q;
                    *p = wv == 0 ? acc[t][q] : *p + acc[t][q];
                }
        }
    }
    __syncthreads();
    float* __restrict__ dst = part + (size_t)blockIdx.x * KS * KS * C1_CIN;
    for (int i = threadIdx.x; i < KS * KS * C1_CIN; i += 256) dst[i] = red[i];
}

static bool cout1_shape_ok(int dtype, int B, int Hi, int Wi, int Cin, int k, int pad, int Ho, int Wo) {
    return (dtype == DT_BF16 || dtype == DT_PAIR) && Cin == C1_CIN && k == 4 && pad >= 0 && pad < k && B >= 1 && Hi >= 1 && Wi >= 1 &&
           Ho == Hi + 2 * pad - k + 1 && Wo == Wi + 2 * pad - k + 1 && Ho >= 1 && Wo >= 1 &&
           (long)B * Hi * Wi * 2 * Cin < (1L << 31);
}

extern "C" int ctg_conv_cout1_fwd(int dtype, const void* x, int x_ld, const float* w, const float* bias, float* y, int act, int B,
                                  int Hi, int Wi, int Cin, int k, int pad, int Ho, int Wo, void* stream) {
    CTG_ENTER();
    if (x == nullptr || w == nullptr || y == nullptr || !cout1_shape_ok(dtype, B, Hi, Wi, Cin, k, pad, Ho, Wo)) return CTG_EINVAL;
    if (x_ld % 8 || x_ld < (dtype == DT_PAIR ? 2 : 1) * Cin || (dtype == DT_PAIR && x_ld % 16) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15))
        return CTG_EINVAL;
    const long ntask = (long)B * Ho * ((Wo + C1_RUN - 1) / C1_RUN);
    if (ntask >= (1L << 30)) return CTG_EINVAL;
    const dim3 grid((unsigned)((ntask + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_PAIR)
        hipLaunchKernelGGL((cout1_fwd_kernel<bfpair_t, 4>), grid, dim3(256), 0, st, (const bfpair_t*)x, x_ld, w, bias, y, act, Hi, Wi, pad,
                           Ho, Wo, (int)ntask);
    else
        hipLaunchKernelGGL((cout1_fwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, x_ld, w, bias, y, act, Hi, Wi, pad, Ho,
                           Wo, (int)ntask);
    return ctg_launch_status();
}

extern "C" int ctg_conv_cout1_bwd(int dtype, const float* g, const float* w, void* dx, int dx_ld, int B, int Hi, int Wi, int Cin,
                                  int k, int pad, int Ho, int Wo, void* stream) {
    CTG_ENTER();
    if (g == nullptr || w == nullptr || dx == nullptr || !cout1_shape_ok(dtype, B, Hi, Wi, Cin, k, pad, Ho, Wo)) return CTG_EINVAL;
    if (dx_ld % 8 || dx_ld < (dtype == DT_PAIR ? 2 : 1) * Cin || (dtype == DT_PAIR && dx_ld % 16) || ((uintptr_t)dx & 15) || ((uintptr_t)w & 15))
        return CTG_EINVAL;
    const long ntask = (long)B * Hi * ((Wi + C1_RUN - 1) / C1_RUN);
    if (ntask >= (1L << 30)) return CTG_EINVAL;
    const dim3 grid((unsigned)((ntask + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_PAIR)
        hipLaunchKernelGGL((cout1_bwd_kernel<bfpair_t, 4>), grid, dim3(256), 0, st, g, w, (bfpair_t*)dx, dx_ld, Hi, Wi, pad, Ho, Wo,
                           (int)ntask);
    else
        hipLaunchKernelGGL((cout1_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, g, w, (bf16_t*)dx, dx_ld, Hi, Wi, pad, Ho, Wo, (int)ntask);
    return ctg_launch_status();
}

extern "C" int ctg_conv_cout1_wgrad(int dtype, const float* g, const void* x, int x_ld, float* part, int Z, int B, int Hi, int Wi,
                                    int Cin, int k, int pad, int Ho, int Wo, void* stream) {
    CTG_ENTER();
    if (g == nullptr || x == nullptr || part == nullptr || Z < 1 || Z > 65535 || !cout1_shape_ok(dtype, B, Hi, Wi, Cin, k, pad, Ho, Wo))
        return CTG_EINVAL;
    if (x_ld % 8 || x_ld < (dtype == DT_PAIR ? 2 : 1) * Cin || (dtype == DT_PAIR && x_ld % 16) || ((uintptr_t)x & 15)) return CTG_EINVAL;
    const long ntask = (long)B * Hi * ((Wi + C1_RUN - 1) / C1_RUN);
    if (ntask >= (1L << 30)) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_PAIR)
        hipLaunchKernelGGL((cout1_wgrad_kernel<bfpair_t, 4>), dim3(Z), dim3(256), 0, st, g, (const bfpair_t*)x, x_ld, part, Hi, Wi, pad, Ho,
                           Wo, (int)ntask);
    else
        hipLaunchKernelGGL((cout1_wgrad_kernel<bf16_t, 4>), dim3(Z), dim3(256), 0, st, g, (const bf16_t*)x, x_ld, part, Hi, Wi, pad, Ho, Wo,
                           (int)ntask);
    return ctg_launch_status();
}
