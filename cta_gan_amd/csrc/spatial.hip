// Spatial resampling and packing kernels of the registration U-Net path
// (trainer/layers.py:172 MaxPool2d(2); trainer/reg.py:93 F.interpolate bilinear,
// align_corners=False; torch.cat at reg.py:77,94) and the tiny-channel packers
// that let 1- and 2-channel tensors ride the MFMA implicit-GEMM kernels.
// NHWC, 16-byte chunks, HBM-bound, no atomics (backward passes are gathers).
#include "common.h"

static inline int ew_blocks(long items) {
    long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

#define DISPATCH_T(dtype, CALL)                   \
    if ((dtype) == DT_BF16) { typedef bf16_t T; CALL; } \
    else if ((dtype) == DT_F32) { typedef float T; CALL; } \
    else if ((dtype) == DT_PAIR) { typedef bfpair_t T; CALL; } \
    else return CTG_EINVAL;

// ------------------------------------------------------------------ max pool
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ out, int o_ld, int H, int W,
                                    int C, long items) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC, Ho = H / 2, Wo = W / 2;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const unsigned pixu = (unsigned)it / (unsigned)CPP;   // 32-bit: items < 2^31 is checked by the host side
        const long pix = pixu;
        const int ch = (int)(it - pix * CPP) * EPC;
        const int ox = (int)(pixu % (unsigned)Wo);
        const int oy = (int)((pixu / (unsigned)Wo) % (unsigned)Ho);
        const int n = (int)(pixu / ((unsigned)Wo * (unsigned)Ho));
        const T* base = x + (((size_t)n * H + 2 * oy) * W + 2 * ox) * x_ld + ch;
        Chunk<T> a, b, c, d, o;
        a.load(base, x_ld); b.load(base + x_ld, x_ld); c.load(base + (size_t)W * x_ld, x_ld); d.load(base + (size_t)(W + 1) * x_ld, x_ld);
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] = fmaxf(fmaxf(a.v[e], b.v[e]), fmaxf(c.v[e], d.v[e]));
        o.store(out + pix * o_ld + ch, o_ld);
    }
}

// dx[y,x] = dout[y/2,x/2] if (y,x) is the FIRST maximum of its window in scan order (ATen's tie rule), else 0;
// optionally accumulated onto dx's previous content (skip tensors receive a second gradient from the decoder).
template <typename T, typename TX = T>
__global__ void maxpool2_bwd_kernel(const TX* __restrict__ x, int x_ld, const T* __restrict__ dout, int d_ld,
                                    T* __restrict__ dx, int dx_ld, int accumulate, int H, int W, int C, long items) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC, Ho = H / 2, Wo = W / 2;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const unsigned pixu = (unsigned)it / (unsigned)CPP;   // 32-bit: items < 2^31 is checked by the host side
        const long pix = pixu;
        const int ch = (int)(it - pix * CPP) * EPC;
        const int xx = (int)(pixu % (unsigned)W);
        const int y = (int)((pixu / (unsigned)W) % (unsigned)H);
        const int n = (int)(pixu / ((unsigned)W * (unsigned)H));
        const int oy = y >> 1, ox = xx >> 1;
        Chunk<T> o;
        o.zero();
        if (oy < Ho && ox < Wo) {
            const TX* base = x + (((size_t)n * H + 2 * oy) * W + 2 * ox) * x_ld + ch;
            Chunk<TX> w[4];
            Chunk<T> g;
            w[0].load(base, x_ld); w[1].load(base + x_ld, x_ld); w[2].load(base + (size_t)W * x_ld, x_ld);
            w[3].load(base + (size_t)(W + 1) * x_ld, x_ld);
            g.load(dout + (((size_t)n * Ho + oy) * Wo + ox) * d_ld + ch, d_ld);
            const int me = (y & 1) * 2 + (xx & 1);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                int arg = 0;
                float m = w[0].v[e];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (w[k].v[e] > m) { m = w[k].v[e]; arg = k; }
                o.v[e] = arg == me ? g.v[e] : 0.f;
            }
        }
        T* dst = dx + pix * dx_ld + ch;
        if (accumulate) {
            Chunk<T> prev;
            prev.load(dst, dx_ld);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] += prev.v[e];
        }
        o.store(dst, dx_ld);
    }
}

// ------------------------------------------------------------------ bilinear x2
__device__ __forceinline__ void bil_src(int o, float scale, int Hi, int& i0, int& i1, float& l) {
    float s = scale * ((float)o + 0.5f) - 0.5f;  // area_pixel_compute_source_index, align_corners=False
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < Hi - 1 ? 1 : 0);
    l = s - (float)i0;
}

template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ out, int o_ld, int Hi, int Wi,
                                    int Ho, int Wo, int C, long items) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC;
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const unsigned pixu = (unsigned)it / (unsigned)CPP;   // 32-bit: items < 2^31 is checked by the host side
        const long pix = pixu;
        const int ch = (int)(it - pix * CPP) * EPC;
        const int ox = (int)(pixu % (unsigned)Wo);
        const int oy = (int)((pixu / (unsigned)Wo) % (unsigned)Ho);
        const int n = (int)(pixu / ((unsigned)Wo * (unsigned)Ho));
        int y0, y1, x0, x1;
        float ly, lx;
        bil_src(oy, sh, Hi, y0, y1, ly);
        bil_src(ox, sw, Wi, x0, x1, lx);
        const T* b = x + (size_t)n * Hi * Wi * x_ld + ch;
        Chunk<T> v00, v01, v10, v11, o;
        v00.load(b + ((size_t)y0 * Wi + x0) * x_ld, x_ld); v01.load(b + ((size_t)y0 * Wi + x1) * x_ld, x_ld);
        v10.load(b + ((size_t)y1 * Wi + x0) * x_ld, x_ld); v11.load(b + ((size_t)y1 * Wi + x1) * x_ld, x_ld);
        const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            o.v[e] = hy * (hx * v00.v[e] + lx * v01.v[e]) + ly * (hx * v10.v[e] + lx * v11.v[e]);
        o.store(out + pix * o_ld + ch, o_ld);
    }
}

// gather form of the transpose: every output row oy whose y0 or y1 equals iy lies in [2iy-2, 2iy+2] for Ho = 2*Hi
template <typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dout, int d_ld, T* __restrict__ dx, int dx_ld, int Hi,
                                    int Wi, int Ho, int Wo, int C, long items) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC;
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const unsigned pixu = (unsigned)it / (unsigned)CPP;   // 32-bit: items < 2^31 is checked by the host side
        const long pix = pixu;
        const int ch = (int)(it - pix * CPP) * EPC;
        const int ix = (int)(pixu % (unsigned)Wi);
        const int iy = (int)((pixu / (unsigned)Wi) % (unsigned)Hi);
        const int n = (int)(pixu / ((unsigned)Wi * (unsigned)Hi));
        float wy[5], wx[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int oy = 2 * iy - 2 + k, ox = 2 * ix - 2 + k;
            wy[k] = 0.f; wx[k] = 0.f;
            if (oy >= 0 && oy < Ho) {
                int a0, a1; float l;
                bil_src(oy, sh, Hi, a0, a1, l);
                wy[k] = (a0 == iy ? 1.f - l : 0.f) + (a1 == iy ? l : 0.f);
            }
            if (ox >= 0 && ox < Wo) {
                int a0, a1; float l;
                bil_src(ox, sw, Wi, a0, a1, l);
                wx[k] = (a0 == ix ? 1.f - l : 0.f) + (a1 == ix ? l : 0.f);
            }
        }
        Chunk<T> o;
        o.zero();
        const T* b = dout + (size_t)n * Ho * Wo * d_ld + ch;
        for (int a = 0; a < 5; ++a) {
            if (wy[a] == 0.f) continue;
            const int oy = 2 * iy - 2 + a;
            for (int c = 0; c < 5; ++c) {
                if (wx[c] == 0.f) continue;
                const int ox = 2 * ix - 2 + c;
                Chunk<T> g;
                g.load(b + ((size_t)oy * Wo + ox) * d_ld, d_ld);
                const float w = wy[a] * wx[c];
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.v[e] += w * g.v[e];
            }
        }
        o.store(dx + pix * dx_ld + ch, dx_ld);
    }
}

// ------------------------------------------------------------------ packers
// fp32 [P][Cs] (Cs <= 4) -> T [P][Cpad], zero padded: lets a 1- or 2-channel gradient be an MFMA operand
template <typename T>
__global__ void chan_pad_kernel(const float* __restrict__ src, int Cs, T* __restrict__ dst, int Cpad, long P) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = Cpad / EPC;
    // one pixel per thread: its Cs floats are one contiguous 4-16-byte read, its Cpad outputs CPP adjacent 16-byte stores
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < P; pix += (long)gridDim.x * blockDim.x) {
        Chunk<T> o;
        o.zero();
        if (Cs == 2 && ((uintptr_t)src & 7) == 0) {
            const float2 v = *reinterpret_cast<const float2*>(src + pix * 2);
            o.v[0] = v.x; o.v[1] = v.y;
        } else {
            for (int c = 0; c < Cs; ++c) o.v[c] = src[pix * Cs + c];
        }
        constexpr int PL = PlanesOf<T>::N;      // (split-pair: a dense row is [hi Cpad | lo Cpad])
        T* d = dst + pix * Cpad * PL;
        o.store(d, Cpad * PL);
        o.zero();
        for (int cc = 1; cc < CPP; ++cc) o.store(d + cc * EPC, Cpad * PL);
    }
}

// im2col of up to two 1-channel fp32 images into [B][Ho][Wo][Kpad], k = (c*kh + ky)*kw + kx (the order of
// weight.view(Cout, Cin*kh*kw)), zero beyond Cin*kh*kw: the Cin in {1, 2} first-layer convs
// (Model/HdGan.py:70,120; trainer/reg.py:77) then run as 1x1 implicit GEMMs on the matrix cores.
template <typename T>
__global__ void im2col_pack_kernel(const float* __restrict__ s0, const float* __restrict__ s1, int Cin, int Hi, int Wi,
                                   int kh, int kw, int stride, int pad, int pad_mode, T* __restrict__ dst, int Ho,
                                   int Wo, int Kpad, long items) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = Kpad / EPC;
    const int K = Cin * kh * kw;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const unsigned pixu = (unsigned)it / (unsigned)CPP;   // 32-bit: items < 2^31 is checked by the host side
        const long pix = pixu;
        const int cc = (int)(it - pix * CPP);
        const int ox = (int)(pixu % (unsigned)Wo);
        const int oy = (int)((pixu / (unsigned)Wo) % (unsigned)Ho);
        const int n = (int)(pixu / ((unsigned)Wo * (unsigned)Ho));
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int k = cc * EPC + e;
            float v = 0.f;
            if (k < K) {
                const int c = k / (kh * kw);
                const int r = k - c * kh * kw;
                const int ky = r / kw, kx = r - ky * kw;
                int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                bool ok = true;
                if (pad_mode == PAD_REFLECT) { iy = reflect_idx(iy, Hi); ix = reflect_idx(ix, Wi); }
                else ok = (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
                if (ok) v = (c == 0 ? s0 : s1)[((size_t)n * Hi + iy) * Wi + ix];
            }
            o.v[e] = v;
        }
        o.store(dst + pix * Kpad * PlanesOf<T>::N + cc * EPC, Kpad * PlanesOf<T>::N);
    }
}

// generic strided-channel copy / convert: dst[p][0..C) = src[p][0..C)  (T -> T)
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int s_ld, T* __restrict__ dst, int d_ld, int C,
                                     long items) {
    constexpr int EPC = Chunk<T>::N;
    const int CPP = C / EPC;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const unsigned pixu = (unsigned)it / (unsigned)CPP;   // 32-bit: items < 2^31 is checked by the host side
        const long pix = pixu;
        const int ch = (int)(it - pix * CPP) * EPC;
        *reinterpret_cast<u32x4*>(dst + pix * d_ld + ch) = *reinterpret_cast<const u32x4*>(src + pix * s_ld + ch);
        if constexpr (PlanesOf<T>::N == 2)     // the lo plane of a split-pair row
            *reinterpret_cast<u32x4*>(dst + pix * d_ld + (d_ld >> 1) + ch) = *reinterpret_cast<const u32x4*>(src + pix * s_ld + (s_ld >> 1) + ch);
    }
}

extern "C" int ctg_maxpool2_fwd(int dtype, const void* x, int x_ld, void* out, int o_ld, int B, int H, int W, int C,
                                void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (C % epc || H < 2 || W < 2) return CTG_EINVAL;
    const long items = (long)B * (H / 2) * (W / 2) * (C / epc);
    if (items >= (1L << 31)) return CTG_EINVAL;   // the kernels decode item indices in 32 bits
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_fwd_kernel<T>), dim3(ew_blocks(items)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, x_ld, (T*)out, o_ld, H, W, C, items));
    return ctg_launch_status();
}

extern "C" int ctg_maxpool2_bwd(int dtype, const void* x, int x_ld, const void* dout, int d_ld, void* dx, int dx_ld,
                                int accumulate, int B, int H, int W, int C, void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (C % epc || H < 2 || W < 2) return CTG_EINVAL;
    const long items = (long)B * H * W * (C / epc);
    if (items >= (1L << 31)) return CTG_EINVAL;   // the kernels decode item indices in 32 bits
    // (DT_MIX: the saved input x is a split pair -- the argmax is taken on hi + lo, as the forward took it; on the hi plane alone two
    //  window elements tie in ~1 % of the windows and the gradient goes to the wrong pixel)
    if (dtype == DT_MIX) {
        hipLaunchKernelGGL((maxpool2_bwd_kernel<bf16_t, bfpair_t>), dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream,
                           (const bfpair_t*)x, x_ld, (const bf16_t*)dout, d_ld, (bf16_t*)dx, dx_ld, accumulate, H, W, C, items);
        return ctg_launch_status();
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_bwd_kernel<T>), dim3(ew_blocks(items)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, x_ld, (const T*)dout, d_ld, (T*)dx, dx_ld,
                                         accumulate, H, W, C, items));
    return ctg_launch_status();
}

extern "C" int ctg_bilinear_fwd(int dtype, const void* x, int x_ld, void* out, int o_ld, int B, int Hi, int Wi, int Ho,
                                int Wo, int C, void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (C % epc) return CTG_EINVAL;
    const long items = (long)B * Ho * Wo * (C / epc);
    if (items >= (1L << 31)) return CTG_EINVAL;   // the kernels decode item indices in 32 bits
    DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T>), dim3(ew_blocks(items)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, x_ld, (T*)out, o_ld, Hi, Wi, Ho, Wo, C,
                                         items));
    return ctg_launch_status();
}

extern "C" int ctg_bilinear_bwd(int dtype, const void* dout, int d_ld, void* dx, int dx_ld, int B, int Hi, int Wi,
                                int Ho, int Wo, int C, void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (C % epc || Ho != 2 * Hi || Wo != 2 * Wi) return CTG_EINVAL;  // the U-Net only ever doubles
    const long items = (long)B * Hi * Wi * (C / epc);
    if (items >= (1L << 31)) return CTG_EINVAL;   // the kernels decode item indices in 32 bits
    DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_bwd_kernel<T>), dim3(ew_blocks(items)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)dout, d_ld, (T*)dx, dx_ld, Hi, Wi, Ho, Wo, C,
                                         items));
    return ctg_launch_status();
}

// fp32 [P][x_ld] (C channels used) -> bf16 [P][2C]: packed WEIGHTS as the operand of the split-bf16 ("bf16x3") convolutions.
// w = hi + lo + O(2^-17 w) with hi = bf16(w), lo = bf16(w - hi); per 32 channels the row holds [hi 32 | lo 32] -- one K step
// of a conv on a split-pair input (ConvArgs::pair_lo), which contracts x_hi.w_hi + x_hi.w_lo + x_lo.w_hi from it.
__global__ void split_weights_kernel(const float* __restrict__ x, long x_ld, bf16_t* __restrict__ out, int C, long P) {
    const int cpp = C / 8;
    const long items = P * cpp;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const long p = it / cpp;
        const int c = (int)(it - p * cpp) * 8;
        const float* src = x + p * x_ld + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (bf16_t)a[e];
            lo[e] = (bf16_t)(a[e] - (float)hi[e]);
            hi[4 + e] = (bf16_t)b[e];
            lo[4 + e] = (bf16_t)(b[e] - (float)hi[4 + e]);
        }
        bf16_t* row = out + p * (2L * C);
        const int j = c >> 5, i = c & 31;
        *reinterpret_cast<bf16x8*>(row + 64 * j + i) = hi;
        *reinterpret_cast<bf16x8*>(row + 64 * j + 32 + i) = lo;
    }
}

extern "C" int ctg_split_weights(const float* x, long x_ld, void* out, int C, long P, void* stream) {
    CTG_ENTER();
    if (C < 32 || C % 32 || x_ld < C || x_ld % 4 || P < 1 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return CTG_EINVAL;
    hipLaunchKernelGGL(split_weights_kernel, dim3(ew_blocks(P * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, x_ld,
                       (bf16_t*)out, C, P);
    return ctg_launch_status();
}

// ctg_split_weights for `count` packs in ONE launch (a network's packs after an optimiser step: ~80 launches of 5-30 us per
// step otherwise): job j owns blocks [first[j], first[j+1]) of the grid, 256 items (8 channels each) per block
#define SPLIT_MAX_T 32
struct SplitList {
    const float* src[SPLIT_MAX_T];
    bf16_t* dst[SPLIT_MAX_T];
    int C[SPLIT_MAX_T];
    long P[SPLIT_MAX_T];
    int first[SPLIT_MAX_T + 1];
    int count;
};
__global__ __launch_bounds__(256) void split_weights_multi_kernel(const SplitList L) {
    int j = 0;
    while (j + 1 < L.count && (int)blockIdx.x >= L.first[j + 1]) ++j;
    const int C = L.C[j], cpp = C / 8;
    const long items = L.P[j] * cpp;
    const long it = ((long)blockIdx.x - L.first[j]) * 256 + threadIdx.x;
    if (it >= items) return;
    const long p = it / cpp;
    const int c = (int)(it - p * cpp) * 8;
    const float* src = L.src[j] + p * C + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16_t)a[e];
        lo[e] = (bf16_t)(a[e] - (float)hi[e]);
        hi[4 + e] = (bf16_t)b[e];
        lo[4 + e] = (bf16_t)(b[e] - (float)hi[4 + e]);
    }
    bf16_t* row = L.dst[j] + p * (2L * C);
    const int jj = c >> 5, i = c & 31;
    *reinterpret_cast<bf16x8*>(row + 64 * jj + i) = hi;
    *reinterpret_cast<bf16x8*>(row + 64 * jj + 32 + i) = lo;
}

extern "C" int ctg_split_weights_multi(int count, const void* const* x, void* const* out, const int* C, const long* P, void* stream) {
    CTG_ENTER();
    if (count < 0) return CTG_EINVAL;
    for (int base = 0; base < count; base += SPLIT_MAX_T) {
        SplitList L;
        L.count = count - base < SPLIT_MAX_T ? count - base : SPLIT_MAX_T;
        long blocks = 0;
        for (int i = 0; i < L.count; ++i) {
            const int q = base + i;
            if (C[q] < 32 || C[q] % 32 || P[q] < 1 || x[q] == nullptr || out[q] == nullptr || ((uintptr_t)x[q] & 15) || ((uintptr_t)out[q] & 15))
                return CTG_EINVAL;
            L.src[i] = (const float*)x[q]; L.dst[i] = (bf16_t*)out[q]; L.C[i] = C[q]; L.P[i] = P[q];
            L.first[i] = (int)blocks;
            blocks += (P[q] * (C[q] / 8) + 255) / 256;
            if (blocks >= (1L << 31)) return CTG_EINVAL;
        }
        L.first[L.count] = (int)blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(split_weights_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, L);
    }
    return ctg_launch_status();
}

// fp32 [P][s_ld] <-> split-pair [P][d_ld] (C channels of each row): the two places where "bf16x3" tensors meet fp32 ones -- wide
// network inputs / outputs at the Python boundary (a stand-alone ResidualBlock, the feature maps Discriminator_m returns)
__global__ void pair_from_f32_kernel(const float* __restrict__ src, long s_ld, bfpair_t* __restrict__ dst, int d_ld, int C,
                                     long items) {
    const int cpp = C / 8;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const long p = it / cpp;
        const int c = (int)(it - p * cpp) * 8;
        Chunk<float> a, b;
        a.load(src + p * s_ld + c);
        b.load(src + p * s_ld + c + 4);
        Chunk<bfpair_t> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o.v[e] = a.v[e]; o.v[4 + e] = b.v[e]; }
        o.store(dst + p * d_ld + c, d_ld);
    }
}
__global__ void pair_to_f32_kernel(const bfpair_t* __restrict__ src, int s_ld, float* __restrict__ dst, long d_ld, int C,
                                   long items) {
    const int cpp = C / 8;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const long p = it / cpp;
        const int c = (int)(it - p * cpp) * 8;
        Chunk<bfpair_t> v;
        v.load(src + p * s_ld + c, s_ld);
        Chunk<float> a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a.v[e] = v.v[e]; b.v[e] = v.v[4 + e]; }
        a.store(dst + p * d_ld + c);
        b.store(dst + p * d_ld + c + 4);
    }
}

// dir 0: fp32 rows (pitch s_ld floats) -> split-pair rows (pitch d_ld bf16 elements, lo plane at + d_ld / 2); dir 1: the reverse
// (s_ld in bf16 elements, d_ld in floats).  C % 8 == 0, 16-byte aligned rows.
extern "C" int ctg_pair_convert(int dir, const void* src, long s_ld, void* dst, long d_ld, int C, long P, void* stream) {
    CTG_ENTER();
    if (C < 8 || C % 8 || P < 1 || (dir != 0 && dir != 1) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return CTG_EINVAL;
    const long pl = dir == 0 ? d_ld : s_ld, fl = dir == 0 ? s_ld : d_ld;
    if (pl < 2 * C || pl % 16 || pl >= (1L << 31) || fl < C || fl % 4) return CTG_EINVAL;
    const long items = P * (C / 8);
    if (dir == 0)
        hipLaunchKernelGGL(pair_from_f32_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream, (const float*)src,
                           s_ld, (bfpair_t*)dst, (int)d_ld, C, items);
    else
        hipLaunchKernelGGL(pair_to_f32_kernel, dim3(ew_blocks(items)), dim3(256), 0, (hipStream_t)stream,
                           (const bfpair_t*)src, (int)s_ld, (float*)dst, d_ld, C, items);
    return ctg_launch_status();
}

extern "C" int ctg_chan_pad(int dtype, const float* src, int Cs, void* dst, int Cpad, long P, void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (Cs < 1 || Cs > 4 || Cpad % epc || P * (Cpad / epc) >= (1L << 31)) return CTG_EINVAL;
    DISPATCH_T(dtype, hipLaunchKernelGGL((chan_pad_kernel<T>), dim3(ew_blocks(P)), dim3(256), 0,
                                         (hipStream_t)stream, src, Cs, (T*)dst, Cpad, P));
    return ctg_launch_status();
}

extern "C" int ctg_im2col_pack(int dtype, const float* s0, const float* s1, int Cin, int B, int Hi, int Wi, int kh,
                               int kw, int stride, int pad, int pad_mode, void* dst, int Ho, int Wo, int Kpad,
                               void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (Cin < 1 || Cin > 2 || (Cin == 2 && s1 == nullptr) || Kpad % epc || Kpad < Cin * kh * kw) return CTG_EINVAL;
    if (pad_mode == PAD_REFLECT && (pad >= Hi || pad >= Wi)) return CTG_EINVAL;
    if (Ho != (Hi + 2 * pad - kh) / stride + 1 || Wo != (Wi + 2 * pad - kw) / stride + 1) return CTG_EINVAL;
    const long items = (long)B * Ho * Wo * (Kpad / epc);
    if (items >= (1L << 31)) return CTG_EINVAL;   // the kernels decode item indices in 32 bits
    DISPATCH_T(dtype, hipLaunchKernelGGL((im2col_pack_kernel<T>), dim3(ew_blocks(items)), dim3(256), 0,
                                         (hipStream_t)stream, s0, s1, Cin, Hi, Wi, kh, kw, stride, pad, pad_mode,
                                         (T*)dst, Ho, Wo, Kpad, items));
    return ctg_launch_status();
}

extern "C" int ctg_copy_channels(int dtype, const void* src, int s_ld, void* dst, int d_ld, int C, long P,
                                 void* stream) {
    CTG_ENTER();
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (C % epc || s_ld % epc || d_ld % epc) return CTG_EINVAL;
    const long items = P * (C / epc);
    if (items >= (1L << 31)) return CTG_EINVAL;   // the kernels decode item indices in 32 bits
    DISPATCH_T(dtype, hipLaunchKernelGGL((copy_channels_kernel<T>), dim3(ew_blocks(items)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)src, s_ld, (T*)dst, d_ld, C, items));
    return ctg_launch_status();
}

extern "C" int ctg_abi_version(void) { return CTG_ABI_VERSION; }
