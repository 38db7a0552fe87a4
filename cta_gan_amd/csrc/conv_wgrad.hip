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
#include <stdlib.h>
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
    static unsigned long long attr_mask = 0;       // per device
    if (smem > 65536) {
        const int rc = ctg_lds_attr_once((const void*)conv_wgrad_kernel<T, BM, BN, NT>, smem, &attr_mask);
        if (rc != CTG_OK) return rc;
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

// ===========================================================================
// Halo-resident weight gradient for stride-1 convs.
//
// The per-tap kernel above re-reads G and X once per tap.  Here a workgroup owns a (co-tile, ci-tile) pair and
// a run of 8x16-pixel spatial tiles of one sample; per tile it loads the G tile (128 px) and the X HALO tile
// ((8+khb-1) x (16+kw-1) px) ONCE by LDS-DMA and accumulates all NT taps of its tap-row group from shifted
// transposing reads of the same halo: L2->LDS bytes per FLOP drop ~NT-fold, narrow layers become HBM-bound
// and the 256-channel layers MFMA/LDS-bound.  LDS rows are unpadded 64/128-byte channel runs (LDS-DMA writes
// lane-linearly); ds_read_b64_tr_b16 stays conflict-free through an XOR swizzle of the 32-byte column pairs
// keyed on the pixel row (inverse applied on the DMA source address).  bf16 only; partial layout and the
// deterministic reduce are shared with the per-tap kernel.
// ===========================================================================
#define WGH_TH 8
#define WGH_TW 16

template <int CPR> __device__ __forceinline__ int wg_swz(int row, int c) {
    if constexpr (CPR == 8) return c ^ (((row >> 1) & 3) << 1);
    else if constexpr (CPR == 4) return c ^ (((row >> 2) & 1) << 1);
    else return c;   // 32-byte rows: 8 consecutive rows already tile one 256-byte bank row
}

// Transposing LDS read as inline asm.  Through the builtin the compiler cannot tell that the fragment reads never touch
// the LDS buffer an in-flight global_load_lds (the next tile's prefetch) is filling and puts `s_waitcnt vmcnt(0)` in front
// of the first read of every k-step -- the prefetch then overlaps nothing.  The asm form is invisible to that analysis;
// its results are only valid after lds_tr_wait() (s_waitcnt lgkmcnt(0)) and the per-fragment lds_tr_use() that follow.
__device__ __forceinline__ bf16x4 lds_read_tr16_b64(const char* p) {
    bf16x4 r;
    const unsigned addr = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
// the same read with the constant part of the address in the instruction's 16-bit offset field: no VALU add per fragment
template <int OFF> __device__ __forceinline__ bf16x4 lds_read_tr16_b64_o(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void lds_tr_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// wait until at most N of the LDS reads issued so far are still in flight (they return in order)
template <int N> __device__ __forceinline__ void lds_tr_wait_le() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt field");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_tr_use(bf16x8& f) { asm volatile("" : "+v"(f)); }   // orders consumers behind lds_tr_wait()

struct WgHaloArgs {
    const void* g;
    const void* x;
    float* part;
    int B, Hs, Ws, Mc, g_ld;
    int Hi, Wi, Nc, x_ld;
    int pad_mode, sps, ntaps;
    int kw, khb, dy0, dx0;   // tap window: kw columns, khb rows per workgroup; origin of the full window
    int prefetch;            // 1: next tile's loads ride behind this tile's MFMAs (A/B switch CTG_WG_NOPREFETCH)
    int xcd;                 // 1: XCD-contiguous workgroup order (A/B switch CTG_WG_NOXCD)
    // input stride 2 (weight gradient of a stride-2 conv / of a transposed conv with the roles swapped): one launch per
    // polyphase component (py, px) of X.  Output pixel q reads X[2 (q + d') + (py, px)] for the taps of that phase, which
    // form a small stride-1 window in d' -- so the phase is a stride-1 problem on the sub-sampled image and the X halo holds
    // every second pixel.  gtaps = tap slots of the whole conv in `part`; tmap[t] = slot of this launch's tap t.
    int is, py, px, gtaps;
    int tmap[64];
    // split-pair operands ("bf16x3" mode): phases == 3, g / x are split pairs whose lo planes lie g_lo / x_lo elements behind.
    // Every pixel tile is swept three times -- (g_hi, x_hi), (g_hi, x_lo), (g_lo, x_hi) -- into the SAME accumulators (the
    // contraction runs over pixels, so the three products are one longer K): one partial, one launch.  phases == 1: plain bf16.
    // phase_split (small grids): the three sweeps run in three workgroups instead, each writing its own partial (slab index
    // 3 z + phase; the reduce sums them) -- a launch of < ~1.5 workgroups per CU is bound by one workgroup's serial chain
    int phases, g_lo, x_lo, phase_split;
    int no_reuse;            // 1: the three sweeps re-fetch every plane they read (A/B switch CTG_WG_NO_REUSE)
};

typedef const __attribute__((address_space(1))) void* wg_gptr_t;
typedef __attribute__((address_space(3))) void* wg_lptr_t;
__device__ __attribute__((aligned(16))) unsigned g_wg_zero_chunk[4];

// WN4 = 1: the four waves split the ci-tile 1 x 4 (each wave: all BM co x BN/4 ci), so every X fragment a wave
// reads feeds BM/16 MFMAs instead of BM/32: fewer LDS bytes per MFMA at the same accumulator budget.
template <int BM, int BN, int NT, int WN4, int KW>
__global__ __launch_bounds__(256, (NT > 16 ? 1 : 2)) void conv_wgrad_halo_kernel(const WgHaloArgs a) {
    typedef bf16_t T;
    constexpr int CPM = BM / 8, CPN = BN / 8;          // 16-byte chunks per pixel row
    constexpr int TM = WN4 ? BM / 16 : BM / 32, TN = WN4 ? BN / 64 : BN / 32;
    constexpr int G_CH = WGH_TH * WGH_TW * CPM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN4 ? 0 : wave >> 1, wn = WN4 ? wave : wave & 1;
    const int tilesN = a.Nc / BN;
    // XCD-aware order: the (co-tile, ci-tile) workgroups of ONE pixel slab re-read each other's G / X channel slices (every
    // G slice Nc/BN times, every X slice Mc/BM times); dealt round-robin over the 8 XCDs they would miss in 8 different L2s.
    // Each XCD gets a contiguous run of (slab, tap group, tile pair) ids instead, so the slab's slices are fetched once per XCD.
    int bx = blockIdx.x, tg = blockIdx.y, z = blockIdx.z;
    if (a.xcd) {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int L = xcd_contiguous(lin, gridDim.x * gridDim.y * gridDim.z);
        bx = L % gridDim.x;
        tg = (L / gridDim.x) % gridDim.y;           // tap-row group
        z = L / (gridDim.x * gridDim.y);
    }
    const int m0 = (bx / tilesN) * BM, n0 = (bx % tilesN) * BN;
    const int z_out = z;                               // partial slab this workgroup writes
    int ph0 = 0;
    if (a.phase_split) { ph0 = z % 3; z /= 3; }
    const int n = z / a.sps, part_i = z - n * a.sps;
    const int HPW = WGH_TW + a.kw - 1, HPH = WGH_TH + a.khb - 1;
    const int X_CH = HPH * HPW * CPN;
    const int X_CH64 = (X_CH + 63) & ~63;
    const int tx_n = (a.Ws + WGH_TW - 1) / WGH_TW, ty_n = (a.Hs + WGH_TH - 1) / WGH_TH;
    const int ntile = tx_n * ty_n;
    const int per = (ntile + a.sps - 1) / a.sps;
    const int t_beg = part_i * per, t_end = min(t_beg + per, ntile);
    const T* __restrict__ G0 = (const T*)a.g + (size_t)n * a.Hs * a.Ws * a.g_ld + m0;
    const T* __restrict__ X0 = (const T*)a.x + (size_t)n * a.Hi * a.Wi * a.x_ld + n0;
    const unsigned hpw_magic = (unsigned)((0x100000000ULL + HPW - 1) / HPW);
    const int Hs = a.Hs, Ws = a.Ws, Hi = a.Hi, Wi = a.Wi, g_ld = a.g_ld, x_ld = a.x_ld, pad_mode = a.pad_mode;
    const int dy_g = a.dy0 + tg * a.khb, dx_g = a.dx0;
    const int IS = a.is;
    const int x_it = (X_CH64 + 255) / 256;

    f32x4 acc[NT][TM][TN];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // tap t of this group sits at (t / KW, t % KW) of the group's window (the host verified row-major order), so
    // its LDS row delta is a compile-time combination of HPW: no per-tap table in registers
    static_assert(NT % KW == 0, "row-major window");
    typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
    const int rsel = 4 * (lane >> 4) + ((lane >> 2) & 3);   // pixel row inside a 16-row run supplied by this lane
    const int psel = lane & 3;                               // 4-column group inside the 16-column block
    constexpr int HPWC = WGH_TW + KW - 1;                    // == HPW (the host dispatches NT by the window width)
    constexpr int ROWB = HPWC * CPN * 16;                    // bytes per halo row
    int goff[TM];                                            // G fragment of co-tile mt: rows rsel (+16), loop invariant
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int cidx = (wm * TM + mt) * 2 + (psel >> 1);
        goff[mt] = (rsel * CPM + wg_swz<CPM>(rsel, cidx)) * 16 + 8 * (psel & 1);
    }
    int xoff[KW][TN];
#pragma unroll
    for (int tx = 0; tx < KW; ++tx)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int hx = tx + rsel, cidx = (wn * TN + nt) * 2 + (psel >> 1);
            xoff[tx][nt] = (hx * CPN + wg_swz<CPN>(hx, cidx)) * 16 + 8 * (psel & 1);
        }

    // LDS holds TWO (G tile, X halo) pairs: the loads of tile t+1 are issued before the MFMAs of tile t and are
    // drained by the single __syncthreads() that ends the tile (one barrier per tile, loads fully overlapped).
    const int pair_bytes = (G_CH + X_CH64) * 16;
    const int NPH = a.phase_split ? 1 : a.phases;
    auto issue_G = [&](int tile, char* bG, bool lo) __attribute__((always_inline)) {
        const int y0 = (tile / tx_n) * WGH_TH, x0 = (tile % tx_n) * WGH_TW;
        const T* __restrict__ G = G0 + (lo ? a.g_lo : 0);
        // ---- G tile: slot s -> (pixel p, chunk)
#pragma unroll
        for (int it = 0; it < G_CH / 256; ++it) {
            const int sl = tid + 256 * it;
            const int p = sl / CPM;
            const int kc = wg_swz<CPM>(p, sl % CPM);
            const int oy = y0 + p / WGH_TW, ox = x0 + p % WGH_TW;
            const bool ok = oy < Hs && ox < Ws;
            const T* src = ok ? G + ((size_t)(oy * Ws + ox) * g_ld + kc * 8) : (const T*)g_wg_zero_chunk;
            __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(bG + (256 * it + 64 * wave) * 16), 16, 0, 0);
        }
    };
    auto issue_X = [&](int tile, char* bX, bool lo) __attribute__((always_inline)) {
        const int y0 = (tile / tx_n) * WGH_TH, x0 = (tile % tx_n) * WGH_TW;
        const T* __restrict__ X = X0 + (lo ? a.x_lo : 0);
        // ---- X halo tile
        for (int it = 0; it < x_it; ++it) {
            if (256 * it + 64 * wave < X_CH64) {
                const int sl = tid + 256 * it;
                const int hrow = sl / CPN;
                const int hy = (int)__umulhi((unsigned)hrow, hpw_magic), hx = hrow - hy * HPW;
                const int kc = wg_swz<CPN>(hx, sl % CPN);   // swizzled by the halo COLUMN: see the fragment reads
                int iy = (y0 + dy_g + hy) * IS + a.py, ix = (x0 + dx_g + hx) * IS + a.px;
                if (pad_mode == PAD_REFLECT) {
                    iy = reflect_idx(iy, Hi);
                    ix = reflect_idx(ix, Wi);
                }
                const bool ok = sl < X_CH && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
                const T* src = ok ? X + ((size_t)(iy * Wi + ix) * x_ld + kc * 8) : (const T*)g_wg_zero_chunk;
                __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(bX + (256 * it + 64 * wave) * 16), 16, 0, 0);
            }
        }
    };
    auto issue_tile = [&](int tile, int buf, int ph) __attribute__((always_inline)) {
        char* bG = smem + buf * pair_bytes;
        issue_G(tile, bG, ph + ph0 == 2);
        issue_X(tile, bG + G_CH * 16, ph + ph0 == 1);
    };
    // Three sweeps in one workgroup (split pair, prefetching): each of the four planes has ITS buffer -- G0 <- g_lo, X0 <- x_hi,
    // G1 <- g_hi, X1 <- x_lo -- and the sweeps run (g_lo, x_hi), (g_hi, x_hi), (g_hi, x_lo): x_hi and g_hi are fetched ONCE per
    // pixel tile instead of twice (four tile loads instead of six; the loads of a step land in a buffer no sweep is reading).
    const bool reuse = NPH == 3 && a.prefetch && !a.no_reuse;
    char* const G0b = smem, * const X0b = smem + G_CH * 16, * const G1b = smem + pair_bytes, * const X1b = G1b + G_CH * 16;
    if (t_beg < t_end) {
        if (reuse) { issue_G(t_beg, G0b, true); issue_X(t_beg, X0b, false); }
        else issue_tile(t_beg, 0, 0);
    }
    __syncthreads();
    int tile = t_beg, ph = 0;
    for (int u = 0; tile < t_end; ++u) {
        int tile_n = tile, ph_n = ph + 1;          // the step after this one
        if (ph_n == NPH) { ph_n = 0; ++tile_n; }
        const int cur = a.prefetch ? u & 1 : 0;
        const char* sG = smem + cur * pair_bytes;
        const char* sX = sG + G_CH * 16;
        if (reuse) {
            if (ph == 0) issue_G(tile, G1b, false);
            else if (ph == 1) issue_X(tile, X1b, true);
            else if (tile_n < t_end) { issue_G(tile_n, G0b, true); issue_X(tile_n, X0b, false); }
            sG = ph == 0 ? G0b : G1b;
            sX = ph == 2 ? X1b : X0b;
        } else if (a.prefetch && tile_n < t_end) {
            issue_tile(tile_n, cur ^ 1, ph_n);
        }
        // ---- 4 k-steps of 32 pixels (= two 16-pixel tile rows); the K order inside a step is the same for A and B.
        // Fully unrolled: every fragment address is a per-tile base register (G: one per co-tile; X: one per tap COLUMN
        // and ci-tile -- the halo is swizzled by its column) plus an instruction immediate (k-step, tap row), so the loop
        // carries no address arithmetic.  The taps of a k-step go in groups of one kernel row (KW taps): the fragments of
        // row g+1 (and, at the end of a k-step, the next k-step's G fragments and first row) are requested BEFORE the MFMAs
        // of row g and retired with a counted lgkmcnt, so a wave's LDS latency hides behind its own matrix work instead of
        // only behind the other wave of the SIMD (r01: all 26 reads of a k-step, one lgkmcnt(0), then 36 MFMAs; 30 address
        // adds per k-step).
        unsigned gbase[TM], xbase[KW][TN];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) gbase[mt] = lds_addr(sG) + goff[mt];
#pragma unroll
        for (int tx = 0; tx < KW; ++tx)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) xbase[tx][nt] = lds_addr(sX) + xoff[tx][nt];
        constexpr int NG = NT / KW;                         // tap rows of this workgroup's window
        constexpr int KSTEPS = WGH_TH / 2;
        constexpr int ROW_RD = KW * TN * 2, FA_RD = TM * 2;   // LDS reads of one tap row / of the G fragments
        constexpr bool PIPE = ROW_RD <= 15;                   // a row's reads fit the lgkmcnt field: request it one row ahead
        constexpr bool PIPE_K = PIPE && FA_RD + ROW_RD <= 15; // ... and the next k-step's G fragments + first row as well
        bf16x8 fa[2][TM];                                    // G fragments of this and the next k-step
        bf16x8 fb[2][KW][TN];                                // two tap rows in flight (ping-pong over the (k-step, row) sequence)
        auto issue_fa = [&](auto kbc) __attribute__((always_inline)) {
            constexpr int kb = decltype(kbc)::value;
            constexpr int GK = kb * (32 * CPM * 16);         // 32 pixel rows per k-step; the swizzle has period 8 rows
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const bf16x4 lo = lds_read_tr16_b64_o<GK>(gbase[mt]);
                const bf16x4 hi = lds_read_tr16_b64_o<GK + 16 * CPM * 16>(gbase[mt]);
                fa[kb & 1][mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        auto issue_row = [&](auto sc) __attribute__((always_inline)) {
            constexpr int s_ = decltype(sc)::value, kb = s_ / NG, g = s_ % NG;
            constexpr int XO = kb * (2 * ROWB) + g * ROWB;   // tile row 2 kb + tap row g; second half of the k-step: + ROWB
#pragma unroll
            for (int tx = 0; tx < KW; ++tx)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const bf16x4 lo = lds_read_tr16_b64_o<XO>(xbase[tx][nt]);
                    const bf16x4 hi = lds_read_tr16_b64_o<XO + ROWB>(xbase[tx][nt]);
                    fb[s_ & 1][tx][nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
        };
        issue_fa(std::integral_constant<int, 0>{});
        issue_row(std::integral_constant<int, 0>{});
        static_for<KSTEPS * NG>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s_ = decltype(sc)::value, kb = s_ / NG, g = s_ % NG;
            constexpr bool last = s_ + 1 == KSTEPS * NG;
            constexpr bool new_k = g + 1 == NG;              // the next step opens a k-step: it needs its G fragments too
            // request what the NEXT step consumes, then wait for everything older than that request
            constexpr bool ahead = !last && (new_k ? PIPE_K : PIPE);
            if constexpr (ahead) {
                if constexpr (new_k) issue_fa(std::integral_constant<int, kb + 1>{});
                issue_row(std::integral_constant<int, s_ + 1>{});
                lds_tr_wait_le<(new_k ? FA_RD : 0) + ROW_RD>();
            } else {
                lds_tr_wait();
            }
            if constexpr (g == 0) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) lds_tr_use(fa[kb & 1][mt]);
            }
#pragma unroll
            for (int tx = 0; tx < KW; ++tx)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) lds_tr_use(fb[s_ & 1][tx][nt]);
#pragma unroll
            for (int tx = 0; tx < KW; ++tx)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
                        acc[g * KW + tx][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            fa[kb & 1][mt], fb[s_ & 1][tx][nt], acc[g * KW + tx][mt][nt], 0, 0, 0);
            if constexpr (!ahead && !last) {
                if constexpr (new_k) issue_fa(std::integral_constant<int, kb + 1>{});
                issue_row(std::integral_constant<int, s_ + 1>{});
            }
        });
        __syncthreads();   // next tile landed (vmcnt(0)) and every wave is done with this one
        if (!a.prefetch && tile_n < t_end) {
            issue_tile(tile_n, 0, ph_n);
            __syncthreads();
        }
        tile = tile_n;
        ph = ph_n;
    }

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* __restrict__ out = a.part + ((size_t)z_out * a.gtaps + a.tmap[tg * NT + t]) * a.Mc * a.Nc;
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

template <int BM, int BN, int NT, int WN4, int KW>
static int launch_wgh(const WgHaloArgs& a, hipStream_t st) {
    const int hp = (WGH_TH + a.khb - 1) * (WGH_TW + a.kw - 1);
    const int pair = (WGH_TH * WGH_TW * (BM / 8) + ((hp * (BN / 8) + 63) & ~63)) * 16;
    if (pair > 80 * 1024 || hp >= 65536) return -1;
    WgHaloArgs b = a;
    if (2 * pair > 80 * 1024) b.prefetch = 0;       // one (G, X-halo) pair fits, two do not: no prefetch
    const int smem = 2 * pair > 80 * 1024 ? pair + 16 : 2 * pair;
    static unsigned long long attr_mask = 0;       // per device
    if (smem > 65536) {
        const int rc = ctg_lds_attr_once((const void*)conv_wgrad_halo_kernel<BM, BN, NT, WN4, KW>, 80 * 1024, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    dim3 grid((a.Mc / BM) * (a.Nc / BN), a.ntaps / NT, a.B * a.sps * (a.phase_split ? 3 : 1));
    hipLaunchKernelGGL((conv_wgrad_halo_kernel<BM, BN, NT, WN4, KW>), grid, dim3(256), smem, st, b);
    return ctg_launch_status();
}

// -1: shape not served here
static int launch_wgh_any(const WgHaloArgs& a, hipStream_t st) {
    const int bm = a.Mc % 64 == 0 ? 64 : a.Mc % 32 == 0 ? 32 : 16, bn = a.Nc % 64 == 0 ? 64 : 32;
    const int nt = a.khb * a.kw;
#define WGH_CASE(M_, N_, T_, W_, K_) if (bm == M_ && bn == N_ && nt == T_ && a.kw == K_) return launch_wgh<M_, N_, T_, W_, K_>(a, st);
    WGH_CASE(64, 64, 9, 1, 3) WGH_CASE(64, 32, 9, 0, 3) WGH_CASE(32, 64, 9, 1, 3) WGH_CASE(32, 32, 9, 0, 3)
    WGH_CASE(32, 64, 7, 1, 7) WGH_CASE(32, 32, 7, 0, 7)
    WGH_CASE(16, 64, 49, 1, 7)   // the generator's 64 -> 1 channel 7x7 tail: one real gradient channel, all 49 taps at once
    WGH_CASE(64, 64, 4, 1, 4) WGH_CASE(32, 64, 4, 1, 4)
    // polyphase components of stride-2 convs (3x3: 2x2, 2x1, 1x2, 1x1 taps; 4x4: 2x2 each)
    WGH_CASE(64, 64, 4, 1, 2) WGH_CASE(64, 64, 2, 1, 2) WGH_CASE(64, 64, 2, 1, 1) WGH_CASE(64, 64, 1, 1, 1)
    WGH_CASE(64, 32, 4, 0, 2) WGH_CASE(64, 32, 2, 0, 2) WGH_CASE(64, 32, 2, 0, 1) WGH_CASE(64, 32, 1, 0, 1)
    WGH_CASE(32, 32, 1, 0, 1)
    WGH_CASE(32, 64, 4, 1, 2) WGH_CASE(32, 64, 2, 1, 2) WGH_CASE(32, 64, 2, 1, 1) WGH_CASE(32, 64, 1, 1, 1)
#undef WGH_CASE
    return -1;
}

// ===========================================================================
// Merged polyphase weight gradient of a STRIDE-2 3x3 conv (Model/HdGan.py:78-80: the generator's down-sampling layers; :93-95:
// its transposed convs, whose weight gradient is the same contraction with the roles of the two tensors swapped).
//
// As four launches of conv_wgrad_halo_kernel -- one per polyphase component (py, px) of X, windows 1x1, 1x2, 2x1, 2x2 -- every
// launch re-reads the whole G tensor: 4 x G + X bytes, and a 1-tap launch takes as long as the 4-tap one (split pair, 128 x 64
// channels at 256^2: 143 / 170 / 167 / 177 us for 537 MB of G each time).  Here ONE workgroup owns a (co-tile, ci-tile) pair and a
// run of 8 x 16-pixel tiles, keeps the G tile in LDS and walks the four phases over it: G is fetched once per tile.
//   * the nine tap accumulators live in registers at once (64 x 64 channels: 144 VGPRs, the budget of the stride-1 3x3 launch);
//   * every phase uses ONE halo geometry -- the (8 + 1) x (16 + 1) pixels of the phase image behind window origin (-1, -1) -- and
//     its taps are the sub-window at rows {G0..1} x columns {T0..1} of it (phase (0,0): {1}x{1}; (0,1): {1}x{0,1}; (1,0): {0,1}x{1};
//     (1,1): all four), so the fragment machinery of conv_wgrad_halo_kernel is reused with (KH, KW, G0, T0) as template arguments;
//   * buffers: G (one per plane) and two X buffers; the loads of a step land in buffers no step is reading:
//       bf16:        step p computes (G, X[p & 1]) and fetches phase p + 1 into X[(p + 1) & 1] (p = 3: the next tile's phase 0 and,
//                    into the other G buffer, its G);
//       split pair:  twelve steps per tile, phase-major, sweeps (g_lo, x_hi), (g_hi, x_hi), (g_hi, x_lo) as in the stride-1 kernel:
//                    sweep 0 fetches x_lo(p) (and, in the tile's first step, g_hi -- free since the last step of the tile before),
//                    sweep 2 fetches x_hi(p + 1) (last phase: the next tile's x_hi(0)); the next tile's g_lo follows its last use;
//     72 KB of LDS, two workgroups per CU, one barrier per step.
// Partials: [z][9][Mc][Nc] like every weight-gradient kernel (slot = position in the caller's tap list: row-major ky, kx).
// ===========================================================================
template <int BM, int BN, int KH, int KW, int G0, int T0, int PY, int PX>
__device__ __forceinline__ void wg_s2m_phase(const char* sG, const char* sX, const int (&goff)[BM / 16], const int (&xoffc)[2][BN / 64],
                                             f32x4 (&acc)[9][BM / 16][BN / 64]) {
    constexpr int CPM = BM / 8, CPN = BN / 8, TM = BM / 16, TN = BN / 64;
    constexpr int HPWC = WGH_TW + 1, ROWB = HPWC * CPN * 16;
    unsigned gbase[TM], xbase[KW][TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) gbase[mt] = lds_addr(sG) + goff[mt];
#pragma unroll
    for (int tx = 0; tx < KW; ++tx)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) xbase[tx][nt] = lds_addr(sX) + xoffc[T0 + tx][nt];
    constexpr int NG = KH, KSTEPS = WGH_TH / 2;
    constexpr int ROW_RD = KW * TN * 2, FA_RD = TM * 2;
    static_assert(FA_RD + ROW_RD <= 15, "a k-step's G fragments and a tap row fit the lgkmcnt field");
    bf16x8 fa[2][TM];
    bf16x8 fb[2][KW][TN];
    auto issue_fa = [&](auto kbc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int GK = kb * (32 * CPM * 16);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const bf16x4 lo = lds_read_tr16_b64_o<GK>(gbase[mt]);
            const bf16x4 hi = lds_read_tr16_b64_o<GK + 16 * CPM * 16>(gbase[mt]);
            fa[kb & 1][mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    auto issue_row = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s_ = decltype(sc)::value, kb = s_ / NG, g = s_ % NG;
        constexpr int XO = kb * (2 * ROWB) + (g + G0) * ROWB;      // tile row 2 kb + window row g + G0; second half of the k-step: + ROWB
#pragma unroll
        for (int tx = 0; tx < KW; ++tx)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const bf16x4 lo = lds_read_tr16_b64_o<XO>(xbase[tx][nt]);
                const bf16x4 hi = lds_read_tr16_b64_o<XO + ROWB>(xbase[tx][nt]);
                fb[s_ & 1][tx][nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
    };
    issue_fa(std::integral_constant<int, 0>{});
    issue_row(std::integral_constant<int, 0>{});
    static_for<KSTEPS * NG>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s_ = decltype(sc)::value, kb = s_ / NG, g = s_ % NG;
        constexpr bool last = s_ + 1 == KSTEPS * NG;
        constexpr bool new_k = g + 1 == NG;
        if constexpr (!last) {
            if constexpr (new_k) issue_fa(std::integral_constant<int, kb + 1>{});
            issue_row(std::integral_constant<int, s_ + 1>{});
            lds_tr_wait_le<(new_k ? FA_RD : 0) + ROW_RD>();
        } else {
            lds_tr_wait();
        }
        if constexpr (g == 0) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) lds_tr_use(fa[kb & 1][mt]);
        }
#pragma unroll
        for (int tx = 0; tx < KW; ++tx)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) lds_tr_use(fb[s_ & 1][tx][nt]);
#pragma unroll
        for (int tx = 0; tx < KW; ++tx) {
            // window (row g + G0, column tx + T0) of the phase image at origin (-1, -1): dy = 2 (g + G0 - 1) + py, ky = dy + 1
            constexpr int ti0 = (2 * (g + G0) - 1 + PY) * 3 + (2 * T0 - 1 + PX);
            const int ti = ti0 + 2 * tx;      // (tx is an unrolled loop index: a constant)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[ti][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kb & 1][mt], fb[s_ & 1][tx][nt], acc[ti][mt][nt], 0, 0, 0);
        }
    });
}

template <int BM, int BN, bool PAIR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_s2m_kernel(const WgHaloArgs a) {
    typedef bf16_t T;
    constexpr int CPM = BM / 8, CPN = BN / 8;
    constexpr int TM = BM / 16, TN = BN / 64;             // the four waves split the ci-tile (WN4 of conv_wgrad_halo_kernel)
    constexpr int G_CH = WGH_TH * WGH_TW * CPM;
    constexpr int HPW = WGH_TW + 1, HPH = WGH_TH + 1;
    constexpr int X_CH = HPH * HPW * CPN, X_CH64 = (X_CH + 63) & ~63;
    constexpr int X_IT = (X_CH64 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave;
    const int tilesN = a.Nc / BN;
    int bx = blockIdx.x, z = blockIdx.z;
    if (a.xcd) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.z;
        const int L = xcd_contiguous(lin, gridDim.x * gridDim.z);
        bx = L % gridDim.x;
        z = L / gridDim.x;
    }
    const int m0 = (bx / tilesN) * BM, n0 = (bx % tilesN) * BN;
    const int n = z / a.sps, part_i = z - n * a.sps;
    const int tx_n = (a.Ws + WGH_TW - 1) / WGH_TW, ty_n = (a.Hs + WGH_TH - 1) / WGH_TH;
    const int ntile = tx_n * ty_n;
    const int per = (ntile + a.sps - 1) / a.sps;
    const int t_beg = part_i * per, t_end = min(t_beg + per, ntile);
    const T* __restrict__ G0p = (const T*)a.g + (size_t)n * a.Hs * a.Ws * a.g_ld + m0;
    const T* __restrict__ X0p = (const T*)a.x + (size_t)n * a.Hi * a.Wi * a.x_ld + n0;
    constexpr unsigned hpw_magic = (unsigned)((0x100000000ULL + HPW - 1) / HPW);
    const int Hs = a.Hs, Ws = a.Ws, Hi = a.Hi, Wi = a.Wi, g_ld = a.g_ld, x_ld = a.x_ld;

    f32x4 acc[9][TM][TN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int rsel = 4 * (lane >> 4) + ((lane >> 2) & 3);
    const int psel = lane & 3;
    int goff[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int cidx = mt * 2 + (psel >> 1);
        goff[mt] = (rsel * CPM + wg_swz<CPM>(rsel, cidx)) * 16 + 8 * (psel & 1);
    }
    int xoffc[2][TN];                                      // halo column offsets 0 and 1 of a window column
#pragma unroll
    for (int tx = 0; tx < 2; ++tx)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int hx = tx + rsel, cidx = (wn * TN + nt) * 2 + (psel >> 1);
            xoffc[tx][nt] = (hx * CPN + wg_swz<CPN>(hx, cidx)) * 16 + 8 * (psel & 1);
        }

    auto issue_G = [&](int tile, char* bG, bool lo) __attribute__((always_inline)) {
        const int y0 = (tile / tx_n) * WGH_TH, x0 = (tile % tx_n) * WGH_TW;
        const T* __restrict__ G = G0p + (lo ? a.g_lo : 0);
#pragma unroll 1
        for (int it = 0; it < G_CH / 256; ++it) {
            const int sl = tid + 256 * it;
            const int p = sl / CPM;
            const int kc = wg_swz<CPM>(p, sl % CPM);
            const int oy = y0 + p / WGH_TW, ox = x0 + p % WGH_TW;
            const bool ok = oy < Hs && ox < Ws;
            const T* src = ok ? G + ((size_t)(oy * Ws + ox) * g_ld + kc * 8) : (const T*)g_wg_zero_chunk;
            __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(bG + (256 * it + 64 * wave) * 16), 16, 0, 0);
        }
    };
    // the (8 + 1) x (16 + 1) pixels of polyphase component (py, px) of X behind window origin (-1, -1)
    auto issue_X = [&](int tile, char* bX, bool lo, int py, int px) __attribute__((always_inline)) {
        const int y0 = (tile / tx_n) * WGH_TH, x0 = (tile % tx_n) * WGH_TW;
        const T* __restrict__ X = X0p + (lo ? a.x_lo : 0);
#pragma unroll 1
        for (int it = 0; it < X_IT; ++it) {
            if (256 * it + 64 * wave < X_CH64) {
                const int sl = tid + 256 * it;
                const int hrow = sl / CPN;
                const int hy = (int)__umulhi((unsigned)hrow, hpw_magic), hx = hrow - hy * HPW;
                const int kc = wg_swz<CPN>(hx, sl % CPN);
                const int iy = (y0 - 1 + hy) * 2 + py, ix = (x0 - 1 + hx) * 2 + px;
                const bool ok = sl < X_CH && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
                const T* src = ok ? X + ((size_t)(iy * Wi + ix) * x_ld + kc * 8) : (const T*)g_wg_zero_chunk;
                __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(bX + (256 * it + 64 * wave) * 16), 16, 0, 0);
            }
        }
    };
    char* const GA = smem, * const GB = smem + G_CH * 16, * const XA = smem + 2 * G_CH * 16, * const XB = XA + X_CH64 * 16;
    // phase p = 2 py + px, its sub-window inside the uniform halo
#define S2M_PHASE(P, SG, SX)                                                                                                      \
    {                                                                                                                            \
        if constexpr ((P) == 0) wg_s2m_phase<BM, BN, 1, 1, 1, 1, 0, 0>(SG, SX, goff, xoffc, acc);                                \
        else if constexpr ((P) == 1) wg_s2m_phase<BM, BN, 1, 2, 1, 0, 0, 1>(SG, SX, goff, xoffc, acc);                           \
        else if constexpr ((P) == 2) wg_s2m_phase<BM, BN, 2, 1, 0, 1, 1, 0>(SG, SX, goff, xoffc, acc);                           \
        else wg_s2m_phase<BM, BN, 2, 2, 0, 0, 1, 1>(SG, SX, goff, xoffc, acc);                                                   \
    }
    if constexpr (!PAIR) {
        // ---- bf16: GA / GB alternate per tile, XA / XB per phase
        if (t_beg < t_end) { issue_G(t_beg, GA, false); issue_X(t_beg, XA, false, 0, 0); }
        __syncthreads();
        for (int tile = t_beg, u = 0; tile < t_end; ++tile, ++u) {
            const char* sG = (u & 1) ? GB : GA;
            char* nG = (u & 1) ? GA : GB;
            static_for<4>([&](auto pc) __attribute__((always_inline)) {
                constexpr int P = decltype(pc)::value;
                if constexpr (P < 3) issue_X(tile, (P & 1) ? XA : XB, false, (P + 1) >> 1, (P + 1) & 1);
                else if (tile + 1 < t_end) { issue_G(tile + 1, nG, false); issue_X(tile + 1, XA, false, 0, 0); }
                S2M_PHASE(P, sG, (P & 1) ? XB : XA)
                __syncthreads();   // the step's loads landed (vmcnt(0)) and every wave is done with the buffers it read
            });
        }
    } else {
        // ---- split pair: GA = g_lo, GB = g_hi, XA = x_hi(p), XB = x_lo(p); sweeps (g_lo, x_hi), (g_hi, x_hi), (g_hi, x_lo)
        if (t_beg < t_end) { issue_G(t_beg, GA, true); issue_X(t_beg, XA, false, 0, 0); }
        __syncthreads();
        for (int tile = t_beg; tile < t_end; ++tile) {
            static_for<4>([&](auto pc) __attribute__((always_inline)) {
                constexpr int P = decltype(pc)::value;
                constexpr int py = P >> 1, px = P & 1;
                // sweep 0: (g_lo, x_hi); fetch x_lo(p) -- and, in the tile's first step, g_hi (GB is free since the tile before)
                issue_X(tile, XB, true, py, px);
                if constexpr (P == 0) issue_G(tile, GB, false);
                S2M_PHASE(P, GA, XA)
                __syncthreads();
                // sweep 1: (g_hi, x_hi); behind the last use of g_lo (phase 3, sweep 0) the next tile's g_lo
                if constexpr (P == 3) { if (tile + 1 < t_end) issue_G(tile + 1, GA, true); }
                S2M_PHASE(P, GB, XA)
                __syncthreads();
                // sweep 2: (g_hi, x_lo); fetch x_hi of the next phase (of the next tile's phase 0)
                if constexpr (P < 3) issue_X(tile, XA, false, (P + 1) >> 1, (P + 1) & 1);
                else if (tile + 1 < t_end) issue_X(tile + 1, XA, false, 0, 0);
                S2M_PHASE(P, GB, XB)
                __syncthreads();
            });
        }
    }
#undef S2M_PHASE

#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* __restrict__ out = a.part + ((size_t)z * 9 + t) * a.Mc * a.Nc;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mt * 16 + (lane >> 4) * 4 + r;
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int c = n0 + (wn * TN + nt) * 16 + (lane & 15);
                    out[(size_t)m * a.Nc + c] = acc[t][mt][nt][r];
                }
            }
    }
}

// -1: shape not served (the caller launches the four polyphase components one by one)
static int launch_wg_s2m(const WgHaloArgs& a, hipStream_t st) {
    static const bool off = getenv("CTG_NO_WG_S2M") != nullptr;      // A/B switch
    if (off || a.Mc % 64 || a.Nc % 64 || a.phase_split) return -1;
    constexpr int BM = 64, BN = 64;
    constexpr int smem = (2 * WGH_TH * WGH_TW * (BM / 8) + 2 * ((((WGH_TH + 1) * (WGH_TW + 1) * (BN / 8)) + 63) & ~63)) * 16;
    static_assert(smem <= 80 * 1024, "two workgroups per CU");
    static unsigned long long attr_mask = 0;       // per device
    static unsigned long long attr_mask_p = 0;
    dim3 grid((a.Mc / BM) * (a.Nc / BN), 1, a.B * a.sps);
    if (a.phases == 3) {
        const int rc = ctg_lds_attr_once((const void*)conv_wgrad_s2m_kernel<BM, BN, true>, 80 * 1024, &attr_mask_p);
        if (rc != CTG_OK) return rc;
        hipLaunchKernelGGL((conv_wgrad_s2m_kernel<BM, BN, true>), grid, dim3(256), smem, st, a);
    } else {
        const int rc = ctg_lds_attr_once((const void*)conv_wgrad_s2m_kernel<BM, BN, false>, 80 * 1024, &attr_mask);
        if (rc != CTG_OK) return rc;
        hipLaunchKernelGGL((conv_wgrad_s2m_kernel<BM, BN, false>), grid, dim3(256), smem, st, a);
    }
    return ctg_launch_status();
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

// Few elements, many slabs (the narrow full-resolution layers: 32x32x9 weights, 256+ slabs): the slab loop is a
// latency chain, so ZP lanes share it and combine through LDS in a fixed order (still deterministic).
template <int ZP>
__global__ __launch_bounds__(256) void wgrad_reduce_zp_kernel(const float* __restrict__ part, int Z, int ntaps, int Mc,
                                                              int Nc, float* __restrict__ dst, int Mreal, int Nreal,
                                                              long sm, long sn, long stp, int accumulate) {
    constexpr int IW = 256 / ZP;
    __shared__ float red[ZP][IW];
    const long E = (long)ntaps * Mc * Nc;
    const int il = threadIdx.x % IW, zq = threadIdx.x / IW;
    const long idx = (long)blockIdx.x * IW + il;
    float s0 = 0.f, s1 = 0.f;
    if (idx < E) {
        int zz = zq;
        for (; zz + ZP < Z; zz += 2 * ZP) {
            s0 += part[(long)zz * E + idx];
            s1 += part[(long)(zz + ZP) * E + idx];
        }
        if (zz < Z) s0 += part[(long)zz * E + idx];
    }
    red[zq][il] = s0 + s1;
    __syncthreads();
    if (zq == 0 && idx < E) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < ZP; ++q) s += red[q][il];
        const int c = (int)(idx % Nc);
        const int m = (int)((idx / Nc) % Mc);
        const int t = (int)(idx / ((long)Nc * Mc));
        if (m < Mreal && c < Nreal) {
            float* d = dst + m * sm + c * sn + t * stp;
            *d = accumulate ? *d + s : s;
        }
    }
}

// C ABI.  Replaces the weight-gradient half of ATen's convolution_backward for
// nn.Conv2d / nn.ConvTranspose2d (same call sites as ctg_conv_igemm).
//   part: caller workspace of (B*sps) * ntaps * Mc * Nc floats, sps = ceil(Hs*Ws / slab).
extern "C" int ctg_conv_wgrad(int dtype, const void* g, const void* x, float* part, int B, int Hs, int Ws, int Mc,
                              int g_ld, int Hi, int Wi, int Nc, int x_ld, int is, int pad_mode, int slab, int ntaps,
                              const int* taps_host, void* stream) {
    CTG_ENTER();
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_PAIR) return CTG_EINVAL;
    // DT_PAIR ("bf16x3"): g and x are split pairs; the halo-resident kernel sweeps every pixel tile three times -- (g_hi, x_hi),
    // (g_hi, x_lo), (g_lo, x_hi) -- into one partial (return 0), or -- small grids -- runs the sweeps in three workgroups that write
    // three partials (return 3: `part` must hold 3 x the slabs).  Returns 2 when the shape is not served that way: the caller then
    // makes the three bf16 calls on the plane views (3 x the partials as well).
    const bool pair = dtype == DT_PAIR;
    if (pair) {
        if (g_ld % 16 || x_ld % 16 || g_ld < 2 * Mc || x_ld < 2 * Nc) return CTG_EINVAL;
        dtype = DT_BF16;
    }
    const int epc = dtype == DT_BF16 ? 8 : 4;
    if (ntaps < 1 || ntaps > 64 || B < 1 || slab < 1 || Mc % 16 || Nc % 32) return CTG_EINVAL;
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
    // split-pair operands on a small grid: one workgroup per sweep (3 x the partials: return value 3 tells the caller)
    const int bm_ = Mc % 64 == 0 ? 64 : Mc % 32 == 0 ? 32 : 16, bn_ = Nc % 64 == 0 ? 64 : 32;
    static const bool nosplit = getenv("CTG_NO_WG_PHASE_SPLIT") != nullptr;      // A/B switch
    const int split = (pair && !nosplit && (long)(Mc / bm_) * (Nc / bn_) * B * a.sps < 384) ? 1 : 0;
    // ---- bf16, stride 1, full kh x kw tap window in row-major order: halo-resident kernel
    // (a single tap -- the 1x1 convs of the registration U-Net -- is a 1x1 "window": same kernel, no halo overlap)
    if (dtype == DT_BF16 && is == 1 && Hs >= WGH_TH && Ws >= WGH_TW && getenv("CTG_NO_HALO") == nullptr &&
        (ntaps > 1 || getenv("CTG_NO_WG_1TAP") == nullptr) &&
        (long)Hi * Wi * x_ld < (1L << 31) && (long)Hs * Ws * g_ld < (1L << 31)) {
        int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
        for (int t = 0; t < ntaps; ++t) {
            const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
            dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax;
            dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
        }
        const int kh = dymax - dymin + 1, kw = dxmax - dxmin + 1;
        bool rowmajor = ntaps == kh * kw;
        for (int t = 0; rowmajor && t < ntaps; ++t) {
            const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
            rowmajor = (dy == dymin + t / kw) && (dx == dxmin + t % kw);
        }
        if (rowmajor) {
            WgHaloArgs h;
            h.g = g; h.x = x; h.part = part;
            h.B = B; h.Hs = Hs; h.Ws = Ws; h.Mc = Mc; h.g_ld = g_ld;
            h.Hi = Hi; h.Wi = Wi; h.Nc = Nc; h.x_ld = x_ld;
            h.pad_mode = pad_mode; h.sps = a.sps; h.ntaps = ntaps;
            h.kw = kw; h.khb = (kh * kw <= 9 || Mc == 16) ? kh : 1; h.dy0 = dymin; h.dx0 = dxmin;
            h.prefetch = getenv("CTG_WG_NOPREFETCH") == nullptr;
            h.xcd = getenv("CTG_WG_NOXCD") == nullptr;
            h.is = 1; h.py = 0; h.px = 0; h.gtaps = ntaps;
            h.phases = pair ? 3 : 1; h.g_lo = g_ld / 2; h.x_lo = x_ld / 2; h.phase_split = split;
            { static const int nr = getenv("CTG_WG_NO_REUSE") != nullptr; h.no_reuse = nr; }
            for (int t = 0; t < ntaps; ++t) h.tmap[t] = t;
            const int rc = launch_wgh_any(h, st);
            if (rc != -1) return (rc == 0 && split) ? 3 : rc;
        }
    }
    // ---- bf16, input stride 2: one halo launch per polyphase component of X (each a small stride-1 window)
    if (dtype == DT_BF16 && is == 2 && Hs >= WGH_TH && Ws >= WGH_TW && getenv("CTG_NO_HALO") == nullptr &&
        getenv("CTG_NO_WG_S2") == nullptr && (long)Hi * Wi * x_ld < (1L << 31) && (long)Hs * Ws * g_ld < (1L << 31) &&
        pad_mode == PAD_ZERO) {
        WgHaloArgs ph[4];
        bool ok = true;
        int nph = 0, covered = 0;
        for (int p = 0; p < 4 && ok; ++p) {
            const int py = p >> 1, px = p & 1;
            WgHaloArgs& h = ph[nph];
            int cnt = 0, dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
            for (int t = 0; t < ntaps; ++t) {
                const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
                if (((dy & 1) != py) || ((dx & 1) != px)) continue;
                const int qy = (dy - py) / 2, qx = (dx - px) / 2;       // exact: dy - py is even
                dymin = qy < dymin ? qy : dymin; dymax = qy > dymax ? qy : dymax;
                dxmin = qx < dxmin ? qx : dxmin; dxmax = qx > dxmax ? qx : dxmax;
                h.tmap[cnt++] = t;
            }
            if (cnt == 0) continue;
            const int kh = dymax - dymin + 1, kw = dxmax - dxmin + 1;
            ok = cnt == kh * kw;
            for (int i = 0; ok && i < cnt; ++i) {   // row-major full window in phase coordinates
                const int tw = a.taps[h.tmap[i]];
                const int qy = (((tw & 0xff) - 64) - py) / 2, qx = ((((tw >> 8) & 0xff) - 64) - px) / 2;
                ok = (qy == dymin + i / kw) && (qx == dxmin + i % kw);
            }
            h.g = g; h.x = x; h.part = part;
            h.B = B; h.Hs = Hs; h.Ws = Ws; h.Mc = Mc; h.g_ld = g_ld;
            h.Hi = Hi; h.Wi = Wi; h.Nc = Nc; h.x_ld = x_ld;
            h.pad_mode = pad_mode; h.sps = a.sps; h.ntaps = cnt; h.gtaps = ntaps;
            h.kw = kw; h.khb = kh; h.dy0 = dymin; h.dx0 = dxmin;
            h.is = 2; h.py = py; h.px = px;
            h.phases = pair ? 3 : 1; h.g_lo = g_ld / 2; h.x_lo = x_ld / 2; h.phase_split = split;
            { static const int nr = getenv("CTG_WG_NO_REUSE") != nullptr; h.no_reuse = nr; }
            h.prefetch = getenv("CTG_WG_NOPREFETCH") == nullptr;
            h.xcd = getenv("CTG_WG_NOXCD") == nullptr;
            // every configuration this phase needs must exist before anything is launched
            const int bm = h.Mc % 64 == 0 ? 64 : h.Mc % 32 == 0 ? 32 : 16, bn = h.Nc % 64 == 0 ? 64 : 32;
            ok = ok && bm >= 32 && !(bm == 32 && bn == 32) && (cnt == 1 || cnt == 2 || cnt == 4) && kw <= 2 && kh <= 2;
            covered += cnt;
            ++nph;
        }
        if (ok && covered == ntaps && ntaps == 9 && nph == 4 && !split) {
            // a stride-2 3x3 window in row-major order: all four polyphase components in ONE launch (G fetched once per tile)
            bool std33 = true;
            for (int t = 0; t < 9; ++t) {
                const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
                std33 = std33 && dy == t / 3 - 1 && dx == t % 3 - 1;
            }
            if (std33) {
                WgHaloArgs h = ph[0];
                h.ntaps = 9; h.gtaps = 9; h.kw = 2; h.khb = 2; h.dy0 = -1; h.dx0 = -1; h.py = 0; h.px = 0;
                const int rc = launch_wg_s2m(h, st);
                if (rc != -1) return rc;
            }
        }
        if (ok && covered == ntaps) {
            for (int p = 0; p < nph; ++p) {
                const int rc = launch_wgh_any(ph[p], st);
                if (rc != 0) return rc == -1 ? CTG_EINVAL : rc;
            }
            return split ? 3 : 0;
        }
    }
    if (pair) return 2;               // (the per-tap kernel has no three-phase sweep)
    if (Mc % 32) return CTG_EINVAL;   // the per-tap kernel tiles M by 32
    return dtype == DT_BF16 ? launch_wg_t<bf16_t>(a, st) : launch_wg_t<float>(a, st);
}

// ---- all weight-gradient reductions of a network's backward in one launch -------------------------------------
#define RED_MAX_T 24
struct ReduceList {
    const float* part[RED_MAX_T];
    float* dst[RED_MAX_T];
    long sm[RED_MAX_T], sn[RED_MAX_T], stp[RED_MAX_T];
    int Z[RED_MAX_T], ntaps[RED_MAX_T], Mc[RED_MAX_T], Nc[RED_MAX_T], Mreal[RED_MAX_T], Nreal[RED_MAX_T];
    int acc[RED_MAX_T], shift[RED_MAX_T];   // shift: log2 of the lanes that share one element's slab loop (0, 4, 5)
    int first[RED_MAX_T + 1];
    int count;
};

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const ReduceList L) {
    __shared__ float red[256];
    int j = 0;
    while (j + 1 < L.count && (int)blockIdx.x >= L.first[j + 1]) ++j;
    const int shift = L.shift[j], ZP = 1 << shift, IW = 256 >> shift;
    const int il = threadIdx.x & (IW - 1), zq = threadIdx.x >> (8 - shift);
    const int Z = L.Z[j], Mc = L.Mc[j], Nc = L.Nc[j];
    const long E = (long)L.ntaps[j] * Mc * Nc;
    const long idx = (long)((int)blockIdx.x - L.first[j]) * IW + il;
    const float* __restrict__ part = L.part[j];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < E) {
        int zz = zq;
        for (; zz + 3 * ZP < Z; zz += 4 * ZP) {      // fixed order, four loads in flight
            s0 += part[(long)zz * E + idx];
            s1 += part[(long)(zz + ZP) * E + idx];
            s2 += part[(long)(zz + 2 * ZP) * E + idx];
            s3 += part[(long)(zz + 3 * ZP) * E + idx];
        }
        for (; zz < Z; zz += ZP) s0 += part[(long)zz * E + idx];
    }
    float s = (s0 + s1) + (s2 + s3);
    if (shift) {                                      // block-uniform
        red[zq * IW + il] = s;
        __syncthreads();
        s = 0.f;
        if (zq == 0)
            for (int q = 0; q < ZP; ++q) s += red[q * IW + il];
    }
    if (zq == 0 && idx < E) {
        const int c = (int)(idx % Nc);
        const int m = (int)((idx / Nc) % Mc);
        const int t = (int)(idx / ((long)Nc * Mc));
        if (m < L.Mreal[j] && c < L.Nreal[j]) {
            float* d = L.dst[j] + m * L.sm[j] + c * L.sn[j] + t * L.stp[j];
            *d = L.acc[j] ? *d + s : s;
        }
    }
}

extern "C" int ctg_wgrad_reduce_multi(int count, const void* const* part, void* const* dst, const int* Z,
                                      const int* ntaps, const int* Mc, const int* Nc, const int* Mreal, const int* Nreal,
                                      const long* sm, const long* sn, const long* stp, const int* accumulate,
                                      void* stream) {
    CTG_ENTER();
    if (count < 0) return CTG_EINVAL;
    for (int base = 0; base < count; base += RED_MAX_T) {
        ReduceList L;
        L.count = count - base < RED_MAX_T ? count - base : RED_MAX_T;
        long blocks = 0;
        for (int i = 0; i < L.count; ++i) {
            const int q = base + i;
            if (Z[q] < 1 || ntaps[q] < 1 || Mreal[q] > Mc[q] || Nreal[q] > Nc[q] || part[q] == nullptr || dst[q] == nullptr)
                return CTG_EINVAL;
            const long E = (long)ntaps[q] * Mc[q] * Nc[q];
            const int shift = (Z[q] >= 64 && E <= 65536) ? (E <= 16384 ? 5 : 4) : 0;
            L.part[i] = (const float*)part[q]; L.dst[i] = (float*)dst[q];
            L.sm[i] = sm[q]; L.sn[i] = sn[q]; L.stp[i] = stp[q];
            L.Z[i] = Z[q]; L.ntaps[i] = ntaps[q]; L.Mc[i] = Mc[q]; L.Nc[i] = Nc[q]; L.Mreal[i] = Mreal[q]; L.Nreal[i] = Nreal[q];
            L.acc[i] = accumulate[q]; L.shift[i] = shift;
            L.first[i] = (int)blocks;
            const int iw = 256 >> shift;
            blocks += (E + iw - 1) / iw;
            if (blocks >= (1L << 30)) return CTG_EINVAL;
        }
        L.first[L.count] = (int)blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, L);
    }
    return ctg_launch_status();
}

extern "C" int ctg_wgrad_reduce(const float* part, int Z, int ntaps, int Mc, int Nc, float* dst, int Mreal, int Nreal,
                                long sm, long sn, long stp, int accumulate, void* stream) {
    CTG_ENTER();
    if (Z < 1 || ntaps < 1 || Mreal > Mc || Nreal > Nc) return CTG_EINVAL;
    const long E = (long)ntaps * Mc * Nc;
    if (Z >= 64 && E <= 65536) {
        // E <= 64K elements: 16 (or 32) lanes per element share the slab loop
        if (E <= 16384) {
            hipLaunchKernelGGL(wgrad_reduce_zp_kernel<32>, dim3((unsigned)((E + 7) / 8)), dim3(256), 0,
                               (hipStream_t)stream, part, Z, ntaps, Mc, Nc, dst, Mreal, Nreal, sm, sn, stp, accumulate);
        } else {
            hipLaunchKernelGGL(wgrad_reduce_zp_kernel<16>, dim3((unsigned)((E + 15) / 16)), dim3(256), 0,
                               (hipStream_t)stream, part, Z, ntaps, Mc, Nc, dst, Mreal, Nreal, sm, sn, stp, accumulate);
        }
        return ctg_launch_status();
    }
    const int blocks = (int)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, part, Z, ntaps, Mc, Nc,
                       dst, Mreal, Nreal, sm, sn, stp, accumulate);
    return ctg_launch_status();
}
