// Implicit-GEMM convolution for gfx950 (MI355X): forward, backward-data and
// transposed convolution of every conv on the CTA-GAN hot path, as ONE
// gather-GEMM kernel family.
//
//   Y[n, oy, ox, co] = act( bias[co] + sum_t sum_ci  X[n, iy_t, ix_t, ci] * W[t][co][ci] )
//   (oy, ox) = (j*os + oy0, i*os + ox0),  (iy_t, ix_t) = pad(j*is + dy_t, i*is + dx_t)
//
// The host passes a tap list (dy_t, dx_t, weight-slice index), so the same
// kernel is the stride-1/2 forward conv (is = stride), the four parity classes
// of a stride-2 transposed conv (os = 2, one launch per class), and the
// backward-data pass of both (it is the dual conv: W slices transposed by the
// weight packer).  Reflection padding is done in the gather, so
// nn.ReflectionPad2d (Model/HdGan.py:53,57,69,100) never materialises.
//
// GEMM mapping: M = output pixels of ONE sample (tiles never straddle samples),
// N = Cout, K = taps x Cin.  NHWC activations make every A row a contiguous
// 16-byte-chunked run of channels; weights are pre-packed [tap][Cout][Cin].
// 256 threads = 4 waves; tile BM x BN; per K-step KCH 16-byte chunks per row
// are register-staged (global -> VGPR -> LDS, issued before and written after
// the MFMA cluster), LDS is double-buffered with one barrier per step, rows are
// XOR-swizzled so ds_read_b128 fragment reads are conflict-free.
//   bf16: v_mfma_f32_16x16x32_bf16  (8 bf16 per lane = one 16-byte chunk)
//   fp32: v_mfma_f32_16x16x4_f32 x4 (the 4 floats of a chunk feed 4 MFMAs; the
//         K order inside a step is permuted identically for A and B)
#include "common.h"

struct ConvArgs {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    int B, Hi, Wi, Cin, x_ld;
    int Ho, Wo, Cout, y_ld;
    int Hs, Ws, oy0, ox0, os, is;
    int pad_mode, act;
    int w_tap_stride;  // elements between weight slices (= Npad * Cin)
    int ntaps;
    int taps[64];      // (dy+64) | (dx+64) << 8 | widx << 16
};

template <int KCH> __device__ __forceinline__ int swz(int row, int c) {
    if constexpr (KCH == 4) return c ^ ((-(row >> 2)) & 3);
    else return c ^ ((row >> 1) & 7);
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int KCH>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int EPC = VecOf<T>::N;
    constexpr int BKE = KCH * EPC;
    constexpr int TM = BM / (WM * 16), TN = BN / (WN * 16);
    constexpr int A_CH = BM * KCH, B_CH = BN * KCH;
    constexpr int A_IT = (A_CH + 255) / 256, B_IT = (B_CH + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + 2 * A_CH * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n = blockIdx.z;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int Ms = a.Hs * a.Ws;
    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ W = (const T*)a.w;

    int aj[A_IT], ai[A_IT];
    bool aok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int c = tid + 256 * it;
        const int m = m0 + c / KCH;
        aok[it] = (c < A_CH) && (m < Ms);
        const int j = m / a.Ws;
        aj[it] = j * a.is;
        ai[it] = (m - j * a.Ws) * a.is;
    }
    const int kPerTap = a.Cin / BKE;
    const int S = a.ntaps * kPerTap;

    u32x4 ra[A_IT], rb[B_IT];
    // branch-free gather: out-of-image / out-of-tile rows read a valid dummy address and are zeroed after
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int tap = s / kPerTap;
        const int kc0 = (s - tap * kPerTap) * BKE;
        const int tw = a.taps[tap];
        const int dy = (tw & 0xff) - 64, dx = ((tw >> 8) & 0xff) - 64, wi = tw >> 16;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int c = tid + 256 * it;
            const int kc = c % KCH;
            int iy = aj[it] + dy, ix = ai[it] + dx;
            bool ok = aok[it];
            if (a.pad_mode == PAD_REFLECT) {
                iy = reflect_idx(iy, a.Hi);
                ix = reflect_idx(ix, a.Wi);
            } else {
                ok = ok && ((unsigned)iy < (unsigned)a.Hi) && ((unsigned)ix < (unsigned)a.Wi);
            }
            const size_t off = ok ? ((((size_t)n * a.Hi + iy) * a.Wi + ix) * a.x_ld + kc0 + kc * EPC) : (size_t)0;
            u32x4 v = *reinterpret_cast<const u32x4*>(X + off);
            if (!ok) v = u32x4{0u,0u,0u,0u};
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            int c = tid + 256 * it;
            if (B_CH % 256 != 0) c = c < B_CH ? c : B_CH - 1;
            const int row = c / KCH, kc = c % KCH;
            rb[it] = *reinterpret_cast<const u32x4*>(W + (size_t)wi * a.w_tap_stride + (size_t)(n0 + row) * a.Cin + kc0 + kc * EPC);
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int c = tid + 256 * it;
            if (A_CH % 256 == 0 || c < A_CH) {
                const int row = c / KCH, kc = c % KCH;
                *reinterpret_cast<u32x4*>(sA + (buf * A_CH + row * KCH + swz<KCH>(row, kc)) * 16) = ra[it];
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int c = tid + 256 * it;
            if (B_CH % 256 == 0 || c < B_CH) {
                const int row = c / KCH, kc = c % KCH;
                *reinterpret_cast<u32x4*>(sB + (buf * B_CH + row * KCH + swz<KCH>(row, kc)) * 16) = rb[it];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* pa = sA + buf * A_CH * 16;
        const char* pb = sB + buf * B_CH * 16;
#pragma unroll
        for (int ks = 0; ks < KCH / 4; ++ks) {
            u32x4 fa[TM], fb[TN];
            const int kc = ks * 4 + (lane >> 4);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int row = (wm * TM + mt) * 16 + (lane & 15);
                fa[mt] = *reinterpret_cast<const u32x4*>(pa + (row * KCH + swz<KCH>(row, kc)) * 16);
            }
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int row = (wn * TN + nt) * 16 + (lane & 15);
                fb[nt] = *reinterpret_cast<const u32x4*>(pb + (row * KCH + swz<KCH>(row, kc)) * 16);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    if constexpr (sizeof(T) == 2) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[mt]), __builtin_bit_cast(bf16x8, fb[nt]), acc[mt][nt], 0, 0, 0);
                    } else {
                        const f32x4 va = __builtin_bit_cast(f32x4, fa[mt]);
                        const f32x4 vb = __builtin_bit_cast(f32x4, fb[nt]);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[q], vb[q], acc[mt][nt], 0, 0, 0);
                    }
                }
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int cur = s & 1;
        gload(s + 1 < S ? s + 1 : s);
        compute(cur);
        lstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: D layout col = lane & 15, row = (lane >> 4) * 4 + r
    OutT* __restrict__ Y = (OutT*)a.y;
    float bv[TN];
    int colv[TN];
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
        colv[nt] = n0 + (wn * TN + nt) * 16 + (lane & 15);
        bv[nt] = (a.bias != nullptr && colv[nt] < a.Cout) ? a.bias[colv[nt]] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (wm * TM + mt) * 16 + (lane >> 4) * 4 + r;
            if (m < Ms) {
                const int j = m / a.Ws, i = m - j * a.Ws;
                OutT* yp = Y + (((size_t)n * a.Ho + (j * a.os + a.oy0)) * a.Wo + (i * a.os + a.ox0)) * a.y_ld;
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    if (colv[nt] < a.Cout) st1(yp + colv[nt], act_apply(acc[mt][nt][r] + bv[nt], a.act));
            }
        }
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int KCH>
static int launch_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * KCH * 16;
    dim3 grid((a.Hs * a.Ws + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.B);
    hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, BM, BN, WM, WN, KCH>), grid, dim3(256), smem, st, a);
    return ctg_launch_status();
}

template <typename T, int KCH>
static int launch_t(const ConvArgs& a, int out_f32, hipStream_t st) {
    if (a.Cout > 64) return out_f32 ? CTG_EINVAL : launch_cfg<T, T, 128, 128, 2, 2, KCH>(a, st);
    if (a.Cout > 32) return out_f32 ? CTG_EINVAL : launch_cfg<T, T, 128, 64, 4, 1, KCH>(a, st);
    if (a.Cout > 16) return out_f32 ? CTG_EINVAL : launch_cfg<T, T, 128, 32, 4, 1, KCH>(a, st);
    if (out_f32) return launch_cfg<T, float, 128, 16, 4, 1, KCH>(a, st);
    return launch_cfg<T, T, 128, 16, 4, 1, KCH>(a, st);
}

// ---------------------------------------------------------------------------
// C ABI.  Replaces the ATen conv2d / conv_transpose2d (+ReflectionPad2d, +bias,
// +LeakyReLU/Tanh) forward and backward-data dispatches made by nn.Conv2d /
// nn.ConvTranspose2d at Model/HdGan.py:54,58,70,78,93,101,120-136 and
// trainer/layers.py:85,282,295.  Returns 0, CTG_EINVAL or 1000+hipError_t.
// ---------------------------------------------------------------------------
extern "C" int ctg_conv_igemm(int dtype, int out_f32, const void* x, const void* w, void* y, const float* bias,
                              int B, int Hi, int Wi, int Cin, int x_ld, int Ho, int Wo, int Cout, int y_ld,
                              int Hs, int Ws, int oy0, int ox0, int os, int is, int pad_mode, int act,
                              int w_npad, int ntaps, const int* taps_host, void* stream) {
    if (ntaps < 1 || ntaps > 64 || B < 1 || Hs < 1 || Ws < 1) return CTG_EINVAL;
    const int epc = dtype == DT_BF16 ? 8 : 4;
    if (dtype != DT_F32 && dtype != DT_BF16) return CTG_EINVAL;
    if (Cin % (4 * epc) != 0 || x_ld % epc != 0 || x_ld < Cin || y_ld < Cout) return CTG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return CTG_EINVAL;
    // the weight slab must cover every N tile the grid touches
    const int bn = Cout > 64 ? 128 : Cout > 32 ? 64 : Cout > 16 ? 32 : 16;
    if (w_npad < ((Cout + bn - 1) / bn) * bn) return CTG_EINVAL;
    if ((Hs - 1) * os + oy0 >= Ho || (Ws - 1) * os + ox0 >= Wo) return CTG_EINVAL;
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.x_ld = x_ld;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.y_ld = y_ld;
    a.Hs = Hs; a.Ws = Ws; a.oy0 = oy0; a.ox0 = ox0; a.os = os; a.is = is;
    a.pad_mode = pad_mode; a.act = act; a.w_tap_stride = w_npad * Cin; a.ntaps = ntaps;
    for (int t = 0; t < ntaps; ++t) {
        const int tw = taps_host[t];
        const int dy = (tw & 0xff) - 64, dx = ((tw >> 8) & 0xff) - 64;
        if (pad_mode == PAD_REFLECT) {
            // a single reflection must land inside the image for every output pixel
            const int ymin = dy, ymax = (Hs - 1) * is + dy, xmin = dx, xmax = (Ws - 1) * is + dx;
            if (-ymin >= Hi || ymax - (Hi - 1) >= Hi || -xmin >= Wi || xmax - (Wi - 1) >= Wi) return CTG_EINVAL;
        }
        a.taps[t] = tw;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool k8 = (Cin % (8 * epc)) == 0;
    if (dtype == DT_BF16) return k8 ? launch_t<bf16_t, 8>(a, out_f32, st) : launch_t<bf16_t, 4>(a, out_f32, st);
    return k8 ? launch_t<float, 8>(a, out_f32, st) : launch_t<float, 4>(a, out_f32, st);
}
