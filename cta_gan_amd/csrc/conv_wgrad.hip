// Weight-gradient of every conv on the CTA-GAN hot path, gfx950 (MI355X).
//
//   dW[t][m][c] = sum_n sum_(j,i)  G[n, j, i, m] * X[n, pad(j*is + dy_t), pad(i*is + dx_t), c]
//
// G is the tensor that lives on the conv's OUTPUT grid (dL/dy for a normal
// conv; the layer INPUT for a transposed conv, whose roles swap), X the tensor
// that is read at the tapped positions.  GEMM per tap: M = Mc, N = Nc, K = all
// pixels of the batch -- a tiny output and an enormous K, so K is split into
// slabs across workgroups; each writes an fp32 partial [z][tap][Mc][Nc] and
// ctg_wgrad_reduce sums the slabs in a fixed order (deterministic, no float
// atomics) straight into the caller's (Cout,Cin,kh,kw)-strided gradient.
//
// Both operands are pixel-major in memory (NHWC), i.e. K-strided for MFMA.
// Tiles are staged [pixel][channel] in LDS (row pitch padded so that the
// transposing reads are bank-conflict-free) and fragments are fetched with
//   bf16: ds_read_b64_tr_b16 (hardware transpose), 2 per 16x16x32 operand; the
//         K order inside a 32-pixel step is permuted identically for A and B
//   fp32: ds_read_b32 (one float per lane is exactly the 16x16x4 operand)
#include "common.h"

struct WgradArgs {
    const void* g;
    const void* x;
    float* part;
    int B, Hs, Ws, Mc, g_ld;
    int Hi, Wi, Nc, x_ld;
    int is, pad_mode;
    int slab, sps;
    int ntaps;
    int taps[64];
};

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> { static constexpr int PK = 64, PK_MT = 32, PADB = 32; };
template <> struct WgCfg<float> { static constexpr int PK = 32, PK_MT = 16, PADB = 64; };

// NT = taps handled by one workgroup.  NT > 1 (narrow layers: the 32-channel 512x512 levels of the U-Net, the
// 1-channel generator tail) loads the G tile ONCE per pixel step and sweeps NT shifted X tiles against it:
// these layers are HBM-bound and the per-tap variant re-read both operands ntaps times.
template <typename T, int BM, int BN, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int EPC = VecOf<T>::N;
    constexpr int PK = NT > 1 ? WgCfg<T>::PK_MT : WgCfg<T>::PK;
    constexpr int RSM = BM * (int)sizeof(T) + WgCfg<T>::PADB;  // LDS row pitch (bytes) of the G tile
    constexpr int RSN = BN * (int)sizeof(T) + WgCfg<T>::PADB;
    constexpr int CPM = BM / EPC, CPN = BN / EPC;              // 16-byte chunks per pixel row
    constexpr int G_CH = PK * CPM, X_CH = PK * CPN;
    constexpr int G_IT = (G_CH + 255) / 256, X_IT = (X_CH + 255) / 256;
    constexpr int TM = BM / 32, TN = BN / 32;                  // 2x2 waves
    constexpr int TILE_G = PK * RSM, TILE_X = PK * RSN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sG = smem;
    char* sX = smem + 2 * TILE_G;   // [2][NT][TILE_X]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = a.Nc / BN;
    const int m0 = (blockIdx.x / tilesN) * BM, n0 = (blockIdx.x % tilesN) * BN;
    const int tap0 = blockIdx.y * NT;
    const int z = blockIdx.z;
    const int n = z / a.sps;
    const int HW = a.Hs * a.Ws;
    const int p0 = (z - n * a.sps) * a.slab;
    const int pend = min(p0 + a.slab, HW);
    int tdy[NT], tdx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tw = a.taps[tap0 + t];
        tdy[t] = (tw & 0xff) - 64;
        tdx[t] = ((tw >> 8) & 0xff) - 64;
    }
    const T* __restrict__ G = (const T*)a.g;
    const T* __restrict__ X = (const T*)a.x;

    u32x4 rg[G_IT], rx[NT][X_IT];
    auto gload = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            int c = tid + 256 * it;
            if (G_CH % 256 != 0) c = c < G_CH ? c : G_CH - 1;
            const int pix = c / CPM, ch = c % CPM;
            const int P = p + pix;
            const bool ok = P < pend;
            const size_t off = ok ? (((size_t)n * HW + P) * a.g_ld + m0 + ch * EPC) : (size_t)0;
            u32x4 v = *reinterpret_cast<const u32x4*>(G + off);
            if (!ok) v = u32x4{0u, 0u, 0u, 0u};
            rg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            int c = tid + 256 * it;
            if (X_CH % 256 != 0) c = c < X_CH ? c : X_CH - 1;
            const int pix = c / CPN, ch = c % CPN;
            const int P = p + pix;
            const bool pok = P < pend;
            const int j = P / a.Ws, i = P - j * a.Ws;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int iy = j * a.is + tdy[t], ix = i * a.is + tdx[t];
                bool ok = pok;
                if (a.pad_mode == PAD_REFLECT) {
                    iy = reflect_idx(iy, a.Hi);
                    ix = reflect_idx(ix, a.Wi);
                } else {
                    ok = ok && ((unsigned)iy < (unsigned)a.Hi) && ((unsigned)ix < (unsigned)a.Wi);
                }
                const size_t off = ok ? ((((size_t)n * a.Hi + iy) * a.Wi + ix) * a.x_ld + n0 + ch * EPC) : (size_t)0;
                u32x4 v = *reinterpret_cast<const u32x4*>(X + off);
                if (!ok) v = u32x4{0u, 0u, 0u, 0u};
                rx[t][it] = v;
            }
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int c = tid + 256 * it;
            if (G_CH % 256 == 0 || c < G_CH)
                *reinterpret_cast<u32x4*>(sG + buf * TILE_G + (c / CPM) * RSM + (c % CPM) * 16) = rg[it];
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int c = tid + 256 * it;
            if (X_CH % 256 == 0 || c < X_CH) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<u32x4*>(sX + (buf * NT + t) * TILE_X + (c / CPN) * RSN + (c % CPN) * 16) = rx[t][it];
            }
        }
    };

    f32x4 acc[NT][TM][TN];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* pg = sG + buf * TILE_G;
        if constexpr (sizeof(T) == 2) {
            typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
            // lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block
            const int rsel = 4 * (lane >> 4) + ((lane >> 2) & 3);
            const int csel = 4 * (lane & 3);
#pragma unroll
            for (int kb = 0; kb < PK; kb += 32) {
                bf16x8 fa[TM];
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const char* p = pg + (kb + rsel) * RSM + ((wm * TM + mt) * 16 + csel) * 2;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 16 * RSM));
                    fa[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const char* px = sX + (buf * NT + t) * TILE_X;
                    bf16x8 fb[TN];
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt) {
                        const char* p = px + (kb + rsel) * RSN + ((wn * TN + nt) * 16 + csel) * 2;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + 16 * RSN));
                        fb[nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < TN; ++nt)
                            acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[t][mt][nt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < PK / 4; ++q) {
                float fa[TM];
                const int row = q * 4 + (lane >> 4);
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    fa[mt] = *reinterpret_cast<const float*>(pg + row * RSM + ((wm * TM + mt) * 16 + (lane & 15)) * 4);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const char* px = sX + (buf * NT + t) * TILE_X;
                    float fb[TN];
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
                        fb[nt] = *reinterpret_cast<const float*>(px + row * RSN + ((wn * TN + nt) * 16 + (lane & 15)) * 4);
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                        for (int nt = 0; nt < TN; ++nt)
                            acc[t][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt], fb[nt], acc[t][mt][nt], 0, 0, 0);
                }
            }
        }
    };

    if (p0 < pend) {
        const int S = (pend - p0 + PK - 1) / PK;
        gload(p0);
        lstore(0);
        __syncthreads();
        for (int s = 0; s < S; ++s) {
            const int cur = s & 1;
            gload(p0 + (s + 1 < S ? s + 1 : s) * PK);
            compute(cur);
            lstore(cur ^ 1);
            __syncthreads();
        }
    }

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* __restrict__ out = a.part + ((size_t)z * a.ntaps + tap0 + t) * a.Mc * a.Nc;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + mt) * 16 + (lane >> 4) * 4 + r;
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int c = n0 + (wn * TN + nt) * 16 + (lane & 15);
                    out[(size_t)m * a.Nc + c] = acc[t][mt][nt][r];
                }
            }
    }
}

template <typename T, int BM, int BN, int NT>
static int launch_wg(const WgradArgs& a, hipStream_t st) {
    constexpr int PK = NT > 1 ? WgCfg<T>::PK_MT : WgCfg<T>::PK;
    constexpr int smem = 2 * PK * ((BM * (int)sizeof(T) + WgCfg<T>::PADB) + NT * (BN * (int)sizeof(T) + WgCfg<T>::PADB));
    static_assert(smem <= 160 * 1024, "LDS");
    static bool attr_done = false;
    if (!attr_done) {
        if (smem > 65536) {
            hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_kernel<T, BM, BN, NT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return 1000 + (int)e;
        }
        attr_done = true;
    }
    dim3 grid((a.Mc / BM) * (a.Nc / BN), a.ntaps / NT, a.B * a.sps);
    hipLaunchKernelGGL((conv_wgrad_kernel<T, BM, BN, NT>), grid, dim3(256), smem, st, a);
    return ctg_launch_status();
}

template <typename T>
static int launch_wg_t(const WgradArgs& a, hipStream_t st) {
    const int bm = a.Mc % 128 == 0 ? 128 : a.Mc % 64 == 0 ? 64 : 32;
    const int bn = a.Nc % 128 == 0 ? 128 : a.Nc % 64 == 0 ? 64 : 32;
    // narrow, many-tap layers: sweep the taps inside the workgroup
    if (bm == 32 && bn == 32 && a.ntaps == 9) return launch_wg<T, 32, 32, 9>(a, st);
    if (bm == 64 && bn == 32 && a.ntaps == 9) return launch_wg<T, 64, 32, 9>(a, st);
    if (bm == 32 && bn == 64 && a.ntaps == 9) return launch_wg<T, 32, 64, 9>(a, st);
    if (bm == 32 && bn == 64 && a.ntaps == 49) return launch_wg<T, 32, 64, 7>(a, st);
#define WG_CASE(M_, N_) if (bm == M_ && bn == N_) return launch_wg<T, M_, N_, 1>(a, st);
    WG_CASE(128, 128) WG_CASE(128, 64) WG_CASE(128, 32)
    WG_CASE(64, 128) WG_CASE(64, 64) WG_CASE(64, 32)
    WG_CASE(32, 128) WG_CASE(32, 64) WG_CASE(32, 32)
#undef WG_CASE
    return CTG_EINVAL;
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int Z, int ntaps, int Mc, int Nc,
                                    float* __restrict__ dst, int Mreal, int Nreal, long sm, long sn, long stp,
                                    int accumulate) {
    const long E = (long)ntaps * Mc * Nc;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < E; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Nc);
        const int m = (int)((idx / Nc) % Mc);
        const int t = (int)(idx / ((long)Nc * Mc));
        if (m >= Mreal || c >= Nreal) continue;
        // fixed summation order (deterministic); 4 independent chains keep 4+ loads in flight per lane
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int zz = 0;
        for (; zz + 4 <= Z; zz += 4) {
            s0 += part[(long)zz * E + idx];
            s1 += part[(long)(zz + 1) * E + idx];
            s2 += part[(long)(zz + 2) * E + idx];
            s3 += part[(long)(zz + 3) * E + idx];
        }
        for (; zz < Z; ++zz) s0 += part[(long)zz * E + idx];
        const float s = (s0 + s1) + (s2 + s3);
        float* d = dst + m * sm + c * sn + t * stp;
        *d = accumulate ? *d + s : s;
    }
}

// C ABI.  Replaces the weight-gradient half of ATen's convolution_backward for
// nn.Conv2d / nn.ConvTranspose2d (same call sites as ctg_conv_igemm).
//   part: caller workspace of (B*sps) * ntaps * Mc * Nc floats, sps = ceil(Hs*Ws / slab).
extern "C" int ctg_conv_wgrad(int dtype, const void* g, const void* x, float* part, int B, int Hs, int Ws, int Mc,
                              int g_ld, int Hi, int Wi, int Nc, int x_ld, int is, int pad_mode, int slab, int ntaps,
                              const int* taps_host, void* stream) {
    if (dtype != DT_F32 && dtype != DT_BF16) return CTG_EINVAL;
    const int epc = dtype == DT_BF16 ? 8 : 4;
    if (ntaps < 1 || ntaps > 64 || B < 1 || slab < 1 || Mc % 32 || Nc % 32) return CTG_EINVAL;
    if (g_ld % epc || x_ld % epc || g_ld < Mc || x_ld < Nc) return CTG_EINVAL;
    if (((uintptr_t)g & 15) || ((uintptr_t)x & 15)) return CTG_EINVAL;
    WgradArgs a;
    a.g = g; a.x = x; a.part = part;
    a.B = B; a.Hs = Hs; a.Ws = Ws; a.Mc = Mc; a.g_ld = g_ld;
    a.Hi = Hi; a.Wi = Wi; a.Nc = Nc; a.x_ld = x_ld;
    a.is = is; a.pad_mode = pad_mode; a.slab = slab; a.sps = (Hs * Ws + slab - 1) / slab; a.ntaps = ntaps;
    for (int t = 0; t < ntaps; ++t) {
        const int tw = taps_host[t];
        const int dy = (tw & 0xff) - 64, dx = ((tw >> 8) & 0xff) - 64;
        if (pad_mode == PAD_REFLECT) {
            const int ymax = (Hs - 1) * is + dy, xmax = (Ws - 1) * is + dx;
            if (-dy >= Hi || ymax - (Hi - 1) >= Hi || -dx >= Wi || xmax - (Wi - 1) >= Wi) return CTG_EINVAL;
        }
        a.taps[t] = tw;
    }
    hipStream_t st = (hipStream_t)stream;
    return dtype == DT_BF16 ? launch_wg_t<bf16_t>(a, st) : launch_wg_t<float>(a, st);
}

extern "C" int ctg_wgrad_reduce(const float* part, int Z, int ntaps, int Mc, int Nc, float* dst, int Mreal, int Nreal,
                                long sm, long sn, long stp, int accumulate, void* stream) {
    if (Z < 1 || ntaps < 1 || Mreal > Mc || Nreal > Nc) return CTG_EINVAL;
    const long E = (long)ntaps * Mc * Nc;
    const int blocks = (int)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, part, Z, ntaps, Mc, Nc,
                       dst, Mreal, Nreal, sm, sn, stp, accumulate);
    return ctg_launch_status();
}
