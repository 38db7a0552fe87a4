// Halo-resident direct convolution for stride-1 convs (gfx950).
//
// The gather-GEMM in conv_igemm.hip re-loads every input pixel once per tap (9x for 3x3, 49x for 7x7):
// measured, it runs at the L2 -> LDS gather rate (~13 TB/s chip-wide), not at the MFMA rate.  A convolution
// offers reuse a GEMM does not: here a workgroup owns a 16x16 tile of output pixels, keeps the (16+kh-1) x
// (16+kw-1) input HALO tile of the current 64-channel (128-byte) slice resident in LDS and sweeps all taps over
// it with shifted ds_read_b128 fragment reads; only the weights stream (double buffered, one tile per tap).
// L2 -> LDS bytes per FLOP drop ~2-4x for the 256-channel layers and ~9-49x for the narrow / 7x7 layers.
//
//   * halo and weight tiles arrive by LDS-DMA (global_load_lds_dwordx4); reflection / zero padding and the
//     inverse LDS XOR swizzle live in the per-lane source address, exactly as in conv_igemm.hip;
//   * the halo of channel slice c+1 is fetched in pieces behind the tap steps of slice c;
//   * MFMA: weights = A operand, pixels = B operand (a lane owns 4 consecutive channels of one pixel);
//     one MFMA "pixel fragment" = one 16-pixel row of the tile, so a tap shift is a constant LDS row offset;
//   * bf16 results are staged through LDS and leave as 16-byte channel chunks; fp32 as float4.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"

struct ConvArgs {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    int B, Hi, Wi, Cin, x_ld;
    int Ho, Wo, Cout, y_ld;
    int Hs, Ws, oy0, ox0, os, is;
    int frame;             // gather kernel only: 1 = compute just the 1-pixel frame of the Hs x Ws grid
    int pad_mode, act;
    int w_tap_stride;  // elements between weight slices (= Npad * Cin)
    int ntaps;
    int taps[64];      // (dy+64) | (dx+64) << 8 | widx << 16
    // halo kernel only: window extent and origin of the tap rectangle
    int kh, kw, dy0, dx0;
    // halo kernel only: when non-null, per-(sample, spatial tile, channel) partial sums (sum, sum of squares) of the
    // fp32 accumulators, [B][tiles][Cout][2]: InstanceNorm statistics without re-reading the output
    float* stats;
    // halo kernel only, os == 1 and the output grid == the tile grid (Ho == Hs, Wo == Ws, oy0 == ox0 == 0), OutT-typed:
    // res  [B][Hs][Ws][res_ld]: added to the result (the skip gradient of a residual block);
    // fold [B][Hs+2][Ws+2][fold_ld]: a gradient on the 1-pixel reflection-padded grid of which only the FRAME is read: the
    //      frame pixels are added to the interior pixels they mirror (rows 1 / Hs-2, columns 1 / Ws-2), so the launch
    //      emits the folded, unpadded gradient.  Both are applied to the ROUNDED result, like a separate combine pass.
    const void* res;
    const void* fold;
    int res_ld, fold_ld;
    // FUSE launches, bf16 only: the result g is the gradient of out = act(IN(z)) [+ skip]; when bstats is non-null the two
    // sums of that InstanceNorm's backward, (sum g m, sum g m xhat) with xhat = (z - mean) rstd and m = act'(xhat), are
    // accumulated over the tile and written per (sample, tile, channel) like `stats` -- the separate statistics pass over
    // g and z is not needed.  bz [B][Hs][Ws][bz_ld] (dtype), bmean / brstd [B][Cout].
    const void* bz;
    const float* bmean;
    const float* brstd;
    float* bstats;
    int bz_ld, bact;
    // halo kernel, MC instantiations (ncls == 4): the four parity classes of a stride-2 transposed conv / stride-2 backward-data
    // pass in ONE launch.  Class q of spatial tile sp is workgroup id (sp * 4 + q) * ntn + n-tile, so the four workgroups that
    // read (almost) the same input halo are dispatched back to back to the same XCD and share it through L2 instead of
    // fetching it from HBM once per class launch.  Per class: its taps are taps[c_tap0[q] .. + c_ntaps[q]), its window and
    // output phase below; Hs x Ws is the common class grid.  Partials (stats) are indexed by (sp * 4 + q).
    int ncls;
    // S2D instantiations (s2d != 0): a STRIDE-2 conv as a stride-1 halo conv over the four polyphase components of its input.
    // Input row 2 j + dy = 2 (j + a) + p (a = floor(dy / 2), p in {0, 1}): the taps of phase (p, q) form a small stride-1 window in
    // (a, b) over the sub-sampled image x[2 y + p][2 x + q], so the K loop walks (phase, channel slice) pairs -- each with its own
    // halo of every second pixel -- and sweeps that phase's taps over it: a 4x4 stride-2 conv is four 2x2 stride-1 convs summed in
    // the accumulators, with the halo kernel's reuse instead of one gather per tap.  ncls = phases, c_ntaps / c_tap0 = the phase's
    // taps in `taps` (dy, dx there are (a, b)), c_oy0 / c_ox0 = (p, q); kh / kw / dy0 / dx0 = the window that covers every phase.
    int s2d;
    // split-pair input ("bf16x3" mode; PK instantiations, T = bf16, KCH = 8): x rows are [hi | lo] planes, the lo plane
    // pair_lo elements behind the hi plane.  One K step covers 32 channels: the LDS row of a pixel is [hi 32 | lo 32] (chunks
    // 0-3 from the hi plane, 4-7 from the lo plane) and a weight row [w_hi 32 | w_lo 32] (ctg_split_weights), and the step
    // contracts hi.w_hi + hi.w_lo + lo.w_hi -- three MFMAs per pair of fragment reads, all from one halo and one weight tile.
    // Cin = 2 x the channel count (the K length of a weight row).
    int pair_lo;
    // NIE instantiations (nie_sync != null; the no-grad forward of a residual block, where nobody reads the conv result z itself):
    // y = act(InstanceNorm(conv(x))) [+ res] from ONE launch.  A workgroup publishes the moments of its tile (stats) and counts
    // itself in at nie_sync[1 + g] (g = sample * n-tiles + n-tile: the workgroups whose moments make up the statistics of its
    // channels); when the count reaches the next multiple of the group size it sums the group's partials in a fixed order,
    // normalises its accumulators in registers and stores only the activated result.  Deadlock-free because workgroups are
    // dispatched in launch order (x fastest, then the sample) and every workgroup of ONE sample -- its n-tile groups are interleaved
    // in dispatch order: tiles x n-tiles workgroups -- fits the share of the chip's slots this launch may count on (host-checked
    // against the kernel's real occupancy: launch_halo_cfg); the poll is bounded all the same (nie_budget polls; nie_sync[0] = 1 and
    // a NaN result when it runs out).  The assumptions are spelled out next to ctg_conv_epilogue in include/ctagan_hip.h.
    unsigned long long* nie_sync;
    int nie_act;
    int nie_budget;
    int c_ntaps[4], c_tap0[4], c_oy0[4], c_ox0[4], c_kh[4], c_kw[4], c_dy0[4], c_dx0[4];
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

extern __device__ __attribute__((aligned(16))) unsigned g_zero_chunk[4];

__device__ __forceinline__ void unpack_bf16x8(const u32x4 t, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(t[i] << 16);
        f[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void add_bf16x8(float (&f)[8], const bf16_t* p) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] += __uint_as_float(t[i] << 16);
        f[2 * i + 1] += __uint_as_float(t[i] & 0xffff0000u);
    }
}

#define HALO_W 16  // output tile is TH x HALO_W pixels (one MFMA pixel fragment = one 16-pixel tile row)

// ABUF = halo buffers: 2 prefetches the next channel slice's halo behind the tap steps (one workgroup per CU);
// 1 reloads it at the slice boundary and halves the LDS footprint, so two workgroups share a CU and cover
// each other's barrier and load waits.
template <typename T, typename OutT, int BN, int WM, int WN, int KCH, int ABUF, int TH, bool FUSE, int KWC, bool MC = false, bool PK = false,
          bool S2D = false, bool NIE = false>
__global__ __launch_bounds__(WM * WN * 64, ((TH / WM) * (BN / (WN * 16)) > 16 ? 2 : 4))
void conv_halo_kernel(const ConvArgs a) {
    static_assert(!NIE || (!MC && !S2D && sizeof(T) == 2 && !std::is_same<OutT, float>::value),
                  "InstanceNorm in the epilogue: plain bf16 / split-pair launches");
    static_assert(!S2D || (!MC && !FUSE && KWC == 0 && ABUF == 1), "polyphase K walk: plain single-buffer instantiations");
    static_assert(!PK || (KCH == 8 && sizeof(T) == 2), "split-pair K steps: [hi 32 | lo 32] rows of bf16");
    // second launch bound = waves per SIMD: <= 128 VGPRs keeps two 8-wave (or four 4-wave) workgroups on a CU;
    // the 32-MFMA-tile-per-wave configurations (128 accumulator registers) run two 4-wave workgroups per CU
    constexpr int NTH = WM * WN * 64;
    constexpr int EPC = VecOf<T>::N;
    constexpr int BKE = KCH * EPC;
    constexpr int BM = TH * HALO_W;
    constexpr int TM = TH / WM;             // pixel fragments (tile rows) per wave
    static_assert(TH % WM == 0, "tile rows per wave");
    constexpr int TN = BN / (WN * 16);
    constexpr int B_CH = BN * KCH;
    constexpr int B_IT = (B_CH + NTH - 1) / NTH;
    static_assert(B_CH % 64 == 0, "weight tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SGPR
    const int wm = wave / WN, wn = wave % WN;
    const int n = blockIdx.y;
    // ---- tile map: N tiles fastest, then (MC: the four parity classes of a spatial tile, then) an XCD-contiguous run of
    // spatial tiles
    const int ntn = (a.Cout + BN - 1) / BN;
    const int id = xcd_contiguous(blockIdx.x, gridDim.x);
    const int n0 = (id % ntn) * BN;
    const int spc = id / ntn;                       // (spatial tile, class) index: also the slot of the moment partials
    int sp = spc, kh_ = a.kh, kw_ = a.kw, dy0 = a.dy0, dx0 = a.dx0, ntaps = a.ntaps, tap0 = 0, oy0_ = a.oy0, ox0_ = a.ox0;
    if constexpr (MC) {
        const int q = spc & 3;
        sp = spc >> 2;
        kh_ = a.c_kh[q]; kw_ = a.c_kw[q]; dy0 = a.c_dy0[q]; dx0 = a.c_dx0[q];
        ntaps = a.c_ntaps[q]; tap0 = a.c_tap0[q]; oy0_ = a.c_oy0[q]; ox0_ = a.c_ox0[q];
    }
    // KWC != 0: the window width is a compile-time constant (3x3 convs), so the halo row pitch is one too and the tile-row
    // offsets of the pixel fragments become instruction immediates
    const int HPW = KWC ? HALO_W + KWC - 1 : HALO_W + kw_ - 1, HPH = TH + kh_ - 1;
    const int HPC = HPH * HPW * KCH;                 // halo slots (16-byte chunks)
    const int HPC64 = (HPC + 63) & ~63;
    const int nchunk = a.Cin / BKE;
    const int nbufA = (nchunk > 1 && ABUF == 2) ? 2 : 1;
    char* sA = smem;
    char* sB = smem + nbufA * HPC64 * 16;

    const int tx_n = (a.Ws + HALO_W - 1) / HALO_W;
    const int y0 = (sp / tx_n) * TH, x0 = (sp % tx_n) * HALO_W;
    const T* __restrict__ X = (const T*)a.x + (size_t)n * a.Hi * a.Wi * a.x_ld;
    const T* __restrict__ W = (const T*)a.w;

    // ---- halo gather: slot s -> source address, computed on the fly (one or two slots per thread and step;
    // keeping per-slot offsets in registers cost 16 VGPRs and, worse, pushed the kernel arguments out of SGPRs)
    const int h_it = (HPC64 + NTH - 1) / NTH;
    const unsigned hpw_magic = (unsigned)((0x100000000ULL + HPW - 1) / HPW);   // hrow / HPW for hrow < 2^16
    const int Hi = a.Hi, Wi = a.Wi, x_ld = a.x_ld, pad_mode = a.pad_mode;
    const int iy00 = y0 + dy0, ix00 = x0 + dx0;
    // 3x3 windows (KWC == 3, kh <= 3: every launch of the residual blocks): a slot's source offset inside the sample does not
    // depend on the channel slice, so it is computed ONCE per tile (H_PRE registers; -1 = zero page) and a slice's halo
    // fetch is an add and a select per slot.  Measured by SQ counters on this kernel (profiles/r03_*): 3046 VALU
    // instructions per wave and tile around 1152 MFMAs -- the vector issue port, which an MFMA holds for 8 of its 16 cycles,
    // is oversubscribed -- of which the per-slice address generation (reflection, bounds, swizzle, 64-bit address) was a third.
    constexpr bool PRE = (KWC >= 3) && !MC;
    constexpr int H_PRE = PRE ? ((((TH + KWC - 1) * (HALO_W + KWC - 1) * KCH + 63) & ~63) + NTH - 1) / NTH : 1;
    int hoff[H_PRE];
    if constexpr (PRE) {
#pragma unroll
        for (int it = 0; it < H_PRE; ++it) {
            const int sl = tid + NTH * it;
            const int hrow = sl / KCH;
            const int hy = (int)__umulhi((unsigned)hrow, hpw_magic), hx = hrow - hy * HPW;
            const int kc = swz<KCH>(hx, sl % KCH);
            int iy = iy00 + hy, ix = ix00 + hx;
            if (pad_mode == PAD_REFLECT) {
                iy = reflect_idx(iy, Hi);
                ix = reflect_idx(ix, Wi);
            }
            const bool ok = (sl < HPC) && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
            hoff[it] = ok ? (iy * Wi + ix) * x_ld + (PK ? (kc & 3) * EPC + (kc >> 2) * a.pair_lo : kc * EPC) : -1;
        }
    }
    int phy = S2D ? a.c_oy0[0] : 0, phx = S2D ? a.c_ox0[0] : 0;      // S2D: pixel phase of the slice being fetched
    auto issue_halo = [&](int it, int buf, int kc0) __attribute__((always_inline)) {
        // wave-uniform skip of 64-slot groups that lie wholly beyond the halo
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < H_PRE; ++j) {
                if (j == it && NTH * j + 64 * wave < HPC64) {
                    const T* src = hoff[j] >= 0 ? X + (hoff[j] + kc0) : (const T*)g_zero_chunk;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + (buf * HPC64 + NTH * j + 64 * wave) * 16), 16, 0, 0);
                }
            }
        } else
        if (NTH * it + 64 * wave < HPC64) {
            const int sl = tid + NTH * it;
            const int hrow = sl / KCH;
            const int hy = (int)__umulhi((unsigned)hrow, hpw_magic), hx = hrow - hy * HPW;
            const int kc = swz<KCH>(hx, sl % KCH);   // swizzled by the halo COLUMN (see compute())
            int iy = iy00 + hy, ix = ix00 + hx;
            if constexpr (S2D) { iy = 2 * iy + phy; ix = 2 * ix + phx; }      // every second pixel of the current phase
            if (pad_mode == PAD_REFLECT) {
                // halo rows of a tile that hangs over the grid may reflect out of range: they only feed
                // masked outputs, so they read the zero page like any other out-of-image pixel
                iy = reflect_idx(iy, Hi);
                ix = reflect_idx(ix, Wi);
            }
            const bool ok = (sl < HPC) && ((unsigned)iy < (unsigned)Hi) && ((unsigned)ix < (unsigned)Wi);
            const T* src = ok ? X + ((iy * Wi + ix) * x_ld + (PK ? (kc & 3) * EPC + (kc >> 2) * a.pair_lo : kc * EPC) + kc0)
                              : (const T*)g_zero_chunk;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + (buf * HPC64 + NTH * it + 64 * wave) * 16), 16, 0, 0);
        }
    };
    int boff[B_IT];
    const bool b_active = (B_CH % NTH == 0) || (tid < B_CH);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int s = (tid + NTH * it) % B_CH;
        const int row = s / KCH;
        boff[it] = (n0 + row) * a.Cin + swz<KCH>(row, s % KCH) * EPC;
    }
    const int w_tap_stride = a.w_tap_stride;
    auto issue_w = [&](int buf, int tw, int kc0) __attribute__((always_inline)) {
        if (b_active) {
            const int wbase = (tw >> 16) * w_tap_stride + kc0;
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                if (B_CH % NTH == 0 || NTH * it + 64 * wave < B_CH)
                    __builtin_amdgcn_global_load_lds((gptr_t)(W + wbase + boff[it]),
                                                     (lptr_t)(sB + (buf * B_CH + NTH * it + 64 * wave) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight fragment of MFMA tile nt: rows (wn TN + nt) 16 + (lane & 15) of the tap's [BN][KCH] tile; its swizzle (swz<>:
    // row & 7 resp. (row >> 1) & 3) is invariant under + 16 rows, so it only depends on lane & 15: one loop-invariant offset
    const int wo0 = (((wn * TN) * 16 + (lane & 15)) * KCH + swz<KCH>(lane & 15, lane >> 4)) * 16;
    auto compute = [&](int abuf, int bbuf, int tw) __attribute__((always_inline)) {
        const char* pa = sA + abuf * HPC64 * 16;
        const char* pb = sB + bbuf * B_CH * 16;
        const int ky = (tw & 0xff) - 64 - dy0, kx = ((tw >> 8) & 0xff) - 64 - dx0;
        // the halo is swizzled by its column, so a pixel fragment's address is a per-lane column offset (one per k-step
        // of the tap) plus wave-uniform row offsets: one vector add per fragment read instead of a swizzle computation
        const int hx = kx + (lane & 15);
        const char* prow = pa + (ky + wm * TM) * (HPW * KCH * 16);
        const int xo0 = (hx * KCH + swz<KCH>(hx, lane >> 4)) * 16;
        if constexpr (PK) {
            // chunks 0-3 of a row are the hi halves (x_hi / w_hi), 4-7 the lo halves (offset ^ 64): hi.w_hi, hi.w_lo, lo.w_hi
            u32x4 fa[TM], fb[TN], fl[TN];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) fa[mt] = *reinterpret_cast<const u32x4*>(prow + xo0 + mt * (HPW * KCH * 16));
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) fb[nt] = *reinterpret_cast<const u32x4*>(pb + wo0 + nt * (16 * KCH * 16));
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fb[nt]), __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) fl[nt] = *reinterpret_cast<const u32x4*>(pb + (wo0 ^ 64) + nt * (16 * KCH * 16));
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fl[nt]), __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) fa[mt] = *reinterpret_cast<const u32x4*>(prow + (xo0 ^ 64) + mt * (HPW * KCH * 16));
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fb[nt]), __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
        } else
#pragma unroll
        for (int ks = 0; ks < KCH / 4; ++ks) {
            u32x4 fa[TM], fb[TN];
            // k-step ks reads chunk 4 ks + (lane >> 4): with KCH = 8 that flips bit 2 of the (XOR-swizzled) chunk = bit 6
            // of the byte offset, for the pixel and the weight fragments alike
            const int xo = xo0 ^ (ks * 64), wo = wo0 ^ (ks * 64);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
                fa[mt] = *reinterpret_cast<const u32x4*>(prow + xo + mt * (HPW * KCH * 16));
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
                fb[nt] = *reinterpret_cast<const u32x4*>(pb + wo + nt * (16 * KCH * 16));
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    if constexpr (sizeof(T) == 2) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fb[nt]), __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
                    } else {
                        const f32x4 va = __builtin_bit_cast(f32x4, fa[mt]);
                        const f32x4 vb = __builtin_bit_cast(f32x4, fb[nt]);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[q], va[q], acc[mt][nt], 0, 0, 0);
                    }
                }
        }
    };

    // ragged bottom tiles (e.g. the 130-row padded grid of a backward-data pass): waves whose pixel rows all lie
    // below the grid skip the MFMA work (they still take part in loads and barriers)
    const bool wave_rows_valid = y0 + wm * TM < a.Hs;
    // ---- main loop over (channel slice c, tap t); __syncthreads() drains the LDS-DMA of the step.
    // The only scalar-memory read of a step (the next tap word) is issued before the LDS fragment reads, so the
    // compiler can use counted lgkmcnt waits inside the MFMA cluster.
    const int pps = (h_it + ntaps - 1) / ntaps;      // halo pieces fetched behind each tap step
    // the whole halo of one channel slice
    auto issue_halo_all = [&](int buf, int kc0) __attribute__((always_inline)) {
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < H_PRE; ++j)
                if (NTH * j + 64 * wave < HPC64) {      // (wave-uniform; also false for j >= h_it)
                    const T* src = hoff[j] >= 0 ? X + (hoff[j] + kc0) : (const T*)g_zero_chunk;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + (buf * HPC64 + NTH * j + 64 * wave) * 16), 16, 0, 0);
                }
        } else {
            for (int it = 0; it < h_it; ++it) issue_halo(it, buf, kc0);
        }
    };
    // element offset of the input channels of K step c (PK: 32 channels per step, both planes)
    auto xslice = [&](int cc) __attribute__((always_inline)) { return PK ? cc * (BKE / 2) : cc * BKE; };
    if constexpr (S2D) { ntaps = a.c_ntaps[0]; tap0 = a.c_tap0[0]; }
    issue_halo_all(0, 0);
    int tw_cur = a.taps[tap0];
    issue_w(0, tw_cur, 0);
    __syncthreads();
    // S2D: the slices are (phase, channel slice) pairs, phase-major; a phase's taps are taps[c_tap0 .. + c_ntaps)
    int S = nchunk * ntaps;
    if constexpr (S2D) {
        S = 0;
        for (int q = 0; q < a.ncls; ++q) S += nchunk * a.c_ntaps[q];
    }
    const int nslices = S2D ? nchunk * a.ncls : nchunk;
    int c = 0, t = 0;
    for (int s = 0; s < S; ++s) {
        int tn = t + 1, cn = c;
        if (tn == ntaps) { tn = 0; cn = c + 1; }
        int tw_next = 0;
        int ntaps_n = ntaps, tap0_n = tap0;
        if constexpr (S2D) {
            if (cn != c && cn < nslices) { const int q = cn / nchunk; ntaps_n = a.c_ntaps[q]; tap0_n = a.c_tap0[q]; }
        }
        if (s + 1 < S) {
            tw_next = a.taps[tap0_n + tn];
            issue_w((s + 1) & 1, tw_next, (S2D ? cn % nchunk : cn) * BKE);
        }
        if (ABUF == 2 && c + 1 < nchunk) {
            for (int q = 0; q < pps; ++q) {
                const int it = t * pps + q;
                if (it < h_it) issue_halo(it, (c + 1) & 1, xslice(c + 1));
            }
        }
        if (wave_rows_valid) compute(nbufA == 2 ? (c & 1) : 0, s & 1, tw_cur);
        __syncthreads();
        if (ABUF == 1 && cn != c && cn < nslices) {
            // single halo buffer: every wave is past its last read of slice c (barrier above); refill for c+1
            if constexpr (S2D) { const int q = cn / nchunk; phy = a.c_oy0[q]; phx = a.c_ox0[q]; }
            issue_halo_all(0, xslice(S2D ? cn % nchunk : cn));
            __syncthreads();
        }
        t = tn;
        c = cn;
        ntaps = ntaps_n;
        tap0 = tap0_n;
        tw_cur = tw_next;
    }

    // ---- epilogue.  acc[mt][nt][r]: pixel (row wm*TM+mt, col lane&15), co = (wn*TN+nt)*16 + (lane>>4)*4 + r
    const int co_l = (lane >> 4) * 4;
    if (a.stats != nullptr) {
        // InstanceNorm moments fused here: sum and sum of squares of the (bias-free) fp32 results over the valid
        // pixels of the tile, reduced lane -> 16-lane row (DPP shuffles) -> waves (LDS) in a fixed order.
        float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]; the ring buffers are free after the last barrier
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {   // one 16-channel group at a time: 8 live sums, not 8*TN
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const bool valid = (y0 + wm * TM + mt < a.Hs) && (x0 + (lane & 15) < a.Ws);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = valid ? acc[mt][nt][r] : 0.f;
                    s1[r] += v;
                    s2[r] += v * v;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row16_sum_to_lane15(s1[r]);
                s2[r] = row16_sum_to_lane15(s2[r]);
                if ((lane & 15) == 15) {
                    const int cl = (wn * TN + nt) * 16 + co_l + r;
                    red[(wm * BN + cl) * 2] = s1[r];
                    red[(wm * BN + cl) * 2 + 1] = s2[r];
                }
            }
        }
        __syncthreads();
        const int ntile = gridDim.x / ntn;
        for (int cl = tid; cl < BN; cl += NTH) {
            if (n0 + cl < a.Cout) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + cl) * 2]; t2 += red[(w * BN + cl) * 2 + 1]; }
                float* dst = a.stats + (((size_t)n * ntile + spc) * a.Cout + n0 + cl) * 2;
                if constexpr (NIE) {     // read by other workgroups of this launch, possibly behind another L2: a device-coherent store
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst),
                                       (unsigned long long)__float_as_uint(t1) | ((unsigned long long)__float_as_uint(t2) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    dst[0] = t1;
                    dst[1] = t2;
                }
            }
        }
        __syncthreads();
    }
    // ---- NIE: (mean, rstd) of this lane's channels once the whole group has published its moments
    float nmr[NIE ? TN : 1][4], nrr[NIE ? TN : 1][4];
    if constexpr (NIE) {
        int* s_ok = reinterpret_cast<int*>(smem);
        float* smr = reinterpret_cast<float*>(smem + 64);                        // [BN][2]
        double* dred = reinterpret_cast<double*>(smem + 64 + BN * 8);            // [NTH / BN][BN][2]
        const int ntile = gridDim.x / ntn;
        unsigned long long* cnt = a.nie_sync + 1 + (n * ntn + n0 / BN);
        // Every word the workgroups of a group exchange (tile moments, arrival counter) is accessed with device-scope atomics --
        // write-through / L2-bypassing accesses on this multi-L2 part -- and ordered by waiting for the stores' acknowledgement
        // (vmcnt) before the arrival that announces them.  Device-scope FENCES are avoided on purpose: a release / acquire pair
        // writes back and invalidates the whole L2 of the XCD, i.e. the halos and weights of every other workgroup on it
        // (measured: 1 ms per launch instead of 0.25).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's moments have reached memory
        __syncthreads();
        if (tid == 0) {
            // The counter only ever grows, by ntile per launch and group (a buffer serves ONE ntile): our group is complete at the
            // next multiple of ntile above the value we found.
            const unsigned long long old = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long target = (old / (unsigned)ntile + 1ull) * (unsigned)ntile;
            int budget = a.nie_budget;     // x ~0.3 us (default 2^22: a second); never reached unless the dispatch-order assumption breaks
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && --budget > 0)
                __builtin_amdgcn_s_sleep(8);
            *s_ok = budget > 0;
            if (budget <= 0) __hip_atomic_store(a.nie_sync, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const bool ok = *s_ok != 0;
        {
            // every workgroup sums the group's tile moments itself, in one fixed order (so all of them get the same bits): no
            // second hand-over for (mean, rstd).  Eight independent 8-byte loads in flight per thread -- they go to memory.
            static_assert(NTH % BN == 0, "finalize: whole channel rounds");
            constexpr int QS = NTH / BN;
            const int q = tid / BN, cl = tid % BN;
            double t1 = 0.0, t2 = 0.0;
            for (int s0 = q; s0 < ntile; s0 += 8 * QS) {
                unsigned long long v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int sl = s0 + j * QS;
                    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(
                        a.stats + (((size_t)n * ntile + (sl < ntile ? sl : q)) * a.Cout + n0 + cl) * 2);
                    v[j] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j * QS < ntile) {
                        t1 += (double)__uint_as_float((unsigned)(v[j] & 0xffffffffull));
                        t2 += (double)__uint_as_float((unsigned)(v[j] >> 32));
                    }
            }
            dred[(q * BN + cl) * 2] = t1;
            dred[(q * BN + cl) * 2 + 1] = t2;
            __syncthreads();
            if (tid < BN) {
                double u1 = 0.0, u2 = 0.0;
#pragma unroll
                for (int qq = 0; qq < QS; ++qq) { u1 += dred[(qq * BN + tid) * 2]; u2 += dred[(qq * BN + tid) * 2 + 1]; }
                const double inv = 1.0 / ((double)a.Hs * (double)a.Ws);
                const double m = u1 * inv;
                double var = u2 * inv - m * m;
                var = var < 0.0 ? 0.0 : var;
                const float nan = __uint_as_float(0x7fc00000u);
                smr[2 * tid] = ok ? (float)m : nan;
                smr[2 * tid + 1] = ok ? (float)(1.0 / sqrt(var + (double)IN_EPS)) : nan;
            }
            __syncthreads();
        }
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nmr[nt][r] = smr[2 * ((wn * TN + nt) * 16 + co_l + r)];
                nrr[nt][r] = smr[2 * ((wn * TN + nt) * 16 + co_l + r) + 1];
            }
        __syncthreads();                 // smr is read: the staging tiles below reuse the memory
    }
    if constexpr (std::is_same<OutT, bfpair_t>::value) {
        // split-pair result: the UNROUNDED fp32 accumulators are staged through LDS in NR rounds (64 channels per round at most)
        // and leave as whole 16-byte chunks of the hi and of the lo plane; residual / frame fold / InstanceNorm-backward sums
        // (FUSE) are applied to the fp32 value in the store loop
        constexpr int NR = TN >= 4 ? 2 : 1;       // rounds: two for the 64-accumulator waves (BN = 128, BN = 64), else one
        static_assert(WN == 1 || TN >= 4, "pair epilogue: the host sizes the staging tile for BN / WN channels");
        constexpr int TNR = TN / NR;              // channel tiles a wave stages per round
        constexpr int CR = WN * TNR * 16;         // channels per round
        constexpr int RS = CR * 4 + 16;           // fp32 row + 16 bytes: the 16 pixel rows of a fragment hit 16 distinct bank groups
        constexpr int CPR = CR / 8;
        constexpr int NIT = BM * CPR / NTH;
        static_assert(BM * CPR % NTH == 0 && NTH % CPR == 0, "pair epilogue: whole trips, a thread keeps its channel chunk");
        char* st = smem;
        bf16_t* __restrict__ Y = (bf16_t*)a.y;
        const bf16_t* __restrict__ R = (const bf16_t*)a.res;
        const bf16_t* __restrict__ F = (const bf16_t*)a.fold;
        const bf16_t* __restrict__ Z = (const bf16_t*)a.bz;
        const bool bst = FUSE && a.bstats != nullptr;
        const int y_lo = a.y_ld >> 1, r_lo = a.res_ld >> 1, f_lo = a.fold_ld >> 1, z_lo = a.bz_ld >> 1;
        // A round stages TN / WN of EVERY wave's channel tiles (not all tiles of the waves with wn == rd): each wave's live
        // accumulators halve with every round, which is what keeps the fused store loop (residual, fold, InstanceNorm input,
        // 32 running sums per thread) inside the 128-register budget -- staging by wave left 64 accumulators live through
        // round 0 and 272 bytes per lane of scratch.  Column block b of the staging tile = channel tile (b / TNR) TN + rd TNR + b % TNR.
        auto chan_of = [&](int col, int rd) __attribute__((always_inline)) {
            const int blk = col >> 4;
            return ((blk / TNR) * TN + rd * TNR + blk % TNR) * 16 + (col & 15);
        };
#pragma unroll
        for (int rd = 0; rd < NR; ++rd) {
            if (rd) __syncthreads();              // the previous round's readers are done with the staging tile
            {
#pragma unroll
                for (int ntl = 0; ntl < TNR; ++ntl) {
                    const int nt = rd * TNR + ntl;
                    const int co = (wn * TN + nt) * 16 + co_l;
                    float bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        bv[r] = (a.bias != nullptr && n0 + co + r < a.Cout) ? a.bias[n0 + co + r] : 0.f;
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
                        const int prow = (wm * TM + mt) * HALO_W + (lane & 15);
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (NIE) o[r] = act_apply((acc[mt][nt][r] - nmr[nt][r]) * nrr[nt][r], a.nie_act);
                            else o[r] = act_apply(acc[mt][nt][r] + bv[r], a.act);
                        }
                        *reinterpret_cast<f32x4*>(st + prow * RS + ((wn * TNR + ntl) * 16 + co_l) * 4) = o;
                    }
                }
            }
            __syncthreads();
            const int chl = chan_of((tid % CPR) * 8, rd);      // the thread's channel chunk inside the N tile, every trip
            float bs1[8], bs2[8], bmu[8], brs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; bmu[e] = 0.f; brs[e] = 0.f; }
            if constexpr (FUSE) {
                if (bst && n0 + chl < a.Cout) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bmu[e] = a.bmean[(size_t)n * a.Cout + n0 + chl + e];
                        brs[e] = a.brstd[(size_t)n * a.Cout + n0 + chl + e];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cidx = tid + NTH * it;
                const int prow = cidx / CPR;
                const int oy = y0 + prow / HALO_W, ox = x0 + prow % HALO_W;
                if (oy < a.Hs && ox < a.Ws && n0 + chl < a.Cout) {
                    float f[8];
                    {
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + prow * RS + (tid % CPR) * 32);
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + prow * RS + (tid % CPR) * 32 + 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { f[e] = v0[e]; f[4 + e] = v1[e]; }
                    }
                    if constexpr (FUSE) {
                        if (R != nullptr) {
                            const bf16_t* rp = R + (((size_t)n * a.Hs + oy) * a.Ws + ox) * a.res_ld + n0 + chl;
                            add_bf16x8(f, rp);
                            add_bf16x8(f, rp + r_lo);
                        }
                        if (F != nullptr) {
                            const int ey = oy == 1 ? 0 : (oy == a.Hs - 2 ? a.Hs + 1 : -1);
                            const int ex = ox == 1 ? 0 : (ox == a.Ws - 2 ? a.Ws + 1 : -1);
                            const bf16_t* Fn = F + (size_t)n * (a.Hs + 2) * (a.Ws + 2) * a.fold_ld + n0 + chl;
                            if (ey >= 0) {
                                const bf16_t* q = Fn + ((size_t)ey * (a.Ws + 2) + ox + 1) * a.fold_ld;
                                add_bf16x8(f, q); add_bf16x8(f, q + f_lo);
                            }
                            if (ex >= 0) {
                                const bf16_t* q = Fn + ((size_t)(oy + 1) * (a.Ws + 2) + ex) * a.fold_ld;
                                add_bf16x8(f, q); add_bf16x8(f, q + f_lo);
                            }
                            if (ey >= 0 && ex >= 0) {
                                const bf16_t* q = Fn + ((size_t)ey * (a.Ws + 2) + ex) * a.fold_ld;
                                add_bf16x8(f, q); add_bf16x8(f, q + f_lo);
                            }
                        }
                    }
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        hi[e] = (bf16_t)f[e];
                        lo[e] = (bf16_t)(f[e] - (float)hi[e]);
                    }
                    bf16_t* yp = Y + (((size_t)n * a.Ho + (oy * a.os + oy0_)) * a.Wo + (ox * a.os + ox0_)) * a.y_ld + n0 + chl;
                    *reinterpret_cast<bf16x8*>(yp) = hi;
                    *reinterpret_cast<bf16x8*>(yp + y_lo) = lo;
                    if constexpr (FUSE) {
                        if (bst) {   // InstanceNorm-backward sums of the STORED gradient (hi + lo)
                            const bf16_t* zp = Z + (((size_t)n * a.Hs + oy) * a.Ws + ox) * a.bz_ld + n0 + chl;
                            float zf[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) zf[e] = 0.f;
                            add_bf16x8(zf, zp);
                            add_bf16x8(zf, zp + z_lo);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float xh = (zf[e] - bmu[e]) * brs[e];
                                float gg = (float)hi[e] + (float)lo[e];
                                if (a.bact == ACT_RELU) gg = xh > 0.f ? gg : 0.f;
                                else if (a.bact == ACT_LRELU) gg = xh > 0.f ? gg : LRELU_SLOPE * gg;
                                bs1[e] += gg;
                                bs2[e] += gg * xh;
                            }
                        }
                    }
                }
            }
            if constexpr (FUSE) {
                if (bst) {
                    __syncthreads();   // every thread is done reading the staged tile
                    float* red = reinterpret_cast<float*>(smem);   // [NTH][16]
#pragma unroll
                    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = bs1[e]; red[tid * 16 + 8 + e] = bs2[e]; }
                    __syncthreads();
                    const int ntile = gridDim.x / ntn;
                    for (int cl = tid; cl < CR; cl += NTH) {
                        const int ch = chan_of(cl, rd);
                        if (n0 + ch < a.Cout) {
                            float t1 = 0.f, t2 = 0.f;
                            for (int q = 0; q < NTH / CPR; ++q) {
                                const float* r = red + (q * CPR + (cl >> 3)) * 16 + (cl & 7);
                                t1 += r[0];
                                t2 += r[8];
                            }
                            float* dst = a.bstats + (((size_t)n * ntile + spc) * a.Cout + n0 + ch) * 2;
                            dst[0] = t1;
                            dst[1] = t2;
                        }
                    }
                }
            }
        }
    } else if constexpr (sizeof(OutT) == 2) {
        constexpr int RS = BN * 2 + 16;
        char* st = smem;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int co = (wn * TN + nt) * 16 + co_l;
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bv[r] = (a.bias != nullptr && n0 + co + r < a.Cout) ? a.bias[n0 + co + r] : 0.f;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int prow = (wm * TM + mt) * HALO_W + (lane & 15);
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (NIE) o[r] = (bf16_t)act_apply((acc[mt][nt][r] - nmr[nt][r]) * nrr[nt][r], a.nie_act);
                    else o[r] = (bf16_t)act_apply(acc[mt][nt][r] + bv[r], a.act);
                }
                *reinterpret_cast<bf16x4*>(st + prow * RS + co * 2) = o;
            }
        }
        OutT* __restrict__ Y = (OutT*)a.y;
        const bf16_t* __restrict__ R = (const bf16_t*)a.res;
        const bf16_t* __restrict__ F = (const bf16_t*)a.fold;
        constexpr int CPR = BN / 8;
        constexpr int NIT = BM * CPR / NTH;
        u32x4 rv[FUSE ? NIT : 1], zv[FUSE ? NIT : 1];   // residual / InstanceNorm-input chunks of the thread's pixels
        if constexpr (FUSE) {   // the residual chunks are requested before the staging barrier and land behind it
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cidx = tid + NTH * it;
                const int prow = cidx / CPR, ch = (cidx % CPR) * 8;
                const int oy = y0 + prow / HALO_W, ox = x0 + prow % HALO_W;
                const bool ok = R != nullptr && oy < a.Hs && ox < a.Ws && n0 + ch < a.Cout;
                rv[it] = *reinterpret_cast<const u32x4*>(
                    ok ? R + (((size_t)n * a.Hs + oy) * a.Ws + ox) * a.res_ld + n0 + ch : (const bf16_t*)g_zero_chunk);
                const bool okz = a.bstats != nullptr && oy < a.Hs && ox < a.Ws && n0 + ch < a.Cout;
                zv[it] = *reinterpret_cast<const u32x4*>(
                    okz ? (const bf16_t*)a.bz + (((size_t)n * a.Hs + oy) * a.Ws + ox) * a.bz_ld + n0 + ch : (const bf16_t*)g_zero_chunk);
            }
        }
        const bool bst = FUSE && a.bstats != nullptr;
        float bs1[8], bs2[8], bmu[8], brs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; bmu[e] = 0.f; brs[e] = 0.f; }
        if constexpr (FUSE) {
            static_assert(NTH % CPR == 0, "a thread keeps its channel chunk over the store loop");
            const int chl = (tid % CPR) * 8;
            if (bst && n0 + chl < a.Cout) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    bmu[e] = a.bmean[(size_t)n * a.Cout + n0 + chl + e];
                    brs[e] = a.brstd[(size_t)n * a.Cout + n0 + chl + e];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cidx = tid + NTH * it;
            const int prow = cidx / CPR, ch = (cidx % CPR) * 8;
            const int oy = y0 + prow / HALO_W, ox = x0 + prow % HALO_W;
            if (oy < a.Hs && ox < a.Ws && n0 + ch < a.Cout) {
                OutT* yp = Y + (((size_t)n * a.Ho + (oy * a.os + oy0_)) * a.Wo + (ox * a.os + ox0_)) * a.y_ld + n0 + ch;
                u32x4 v = *reinterpret_cast<const u32x4*>(st + prow * RS + ch * 2);
                if constexpr (FUSE) {   // own instantiation: launches without res / fold run the plain store loop
                    float f[8], g[8];
                    unpack_bf16x8(v, f);
                    unpack_bf16x8(rv[it], g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                    if (F != nullptr) {
                        const int ey = oy == 1 ? 0 : (oy == a.Hs - 2 ? a.Hs + 1 : -1);
                        const int ex = ox == 1 ? 0 : (ox == a.Ws - 2 ? a.Ws + 1 : -1);
                        const OutT* Fn = F + (size_t)n * (a.Hs + 2) * (a.Ws + 2) * a.fold_ld + n0 + ch;
                        if (ey >= 0) add_bf16x8(f, Fn + ((size_t)ey * (a.Ws + 2) + ox + 1) * a.fold_ld);
                        if (ex >= 0) add_bf16x8(f, Fn + ((size_t)(oy + 1) * (a.Ws + 2) + ex) * a.fold_ld);
                        if (ey >= 0 && ex >= 0) add_bf16x8(f, Fn + ((size_t)ey * (a.Ws + 2) + ex) * a.fold_ld);
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)f[e];
                    v = __builtin_bit_cast(u32x4, o);
                    if (bst) {   // InstanceNorm-backward sums of the STORED (rounded) gradient; ch is the same in every trip
                        float zf[8];
                        unpack_bf16x8(zv[it], zf);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xh = (zf[e] - bmu[e]) * brs[e];
                            float gg = (float)o[e];
                            if (a.bact == ACT_RELU) gg = xh > 0.f ? gg : 0.f;
                            else if (a.bact == ACT_LRELU) gg = xh > 0.f ? gg : LRELU_SLOPE * gg;
                            bs1[e] += gg;
                            bs2[e] += gg * xh;
                        }
                    }
                }
                *reinterpret_cast<u32x4*>(yp) = v;
            }
        }
        if constexpr (FUSE) {
            if (bst) {
                // per-thread sums (8 channels x its NIT pixels) -> LDS -> one thread per channel adds the NTH / CPR
                // threads that share its chunk, in a fixed order; layout of the partials as for `stats`
                __syncthreads();   // every thread is done reading the staged tile
                float* red = reinterpret_cast<float*>(smem);   // [NTH][16]
#pragma unroll
                for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = bs1[e]; red[tid * 16 + 8 + e] = bs2[e]; }
                __syncthreads();
                const int ntile = gridDim.x / ntn;
                for (int cl = tid; cl < BN; cl += NTH) {
                    if (n0 + cl < a.Cout) {
                        float t1 = 0.f, t2 = 0.f;
                        for (int q = 0; q < NTH / CPR; ++q) {
                            const float* r = red + (q * CPR + (cl >> 3)) * 16 + (cl & 7);
                            t1 += r[0];
                            t2 += r[8];
                        }
                        float* dst = a.bstats + (((size_t)n * ntile + spc) * a.Cout + n0 + cl) * 2;
                        dst[0] = t1;
                        dst[1] = t2;
                    }
                }
            }
        }
    } else {
        OutT* __restrict__ Y = (OutT*)a.y;
        const bool vec_ok = ((a.Cout & 3) == 0) && ((a.y_ld & 3) == 0);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int oy = y0 + wm * TM + mt, ox = x0 + (lane & 15);
            if (oy >= a.Hs || ox >= a.Ws) continue;
            OutT* yp = Y + (((size_t)n * a.Ho + (oy * a.os + oy0_)) * a.Wo + (ox * a.os + ox0_)) * a.y_ld;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int co = n0 + (wn * TN + nt) * 16 + co_l;
                if (co >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float b = (a.bias != nullptr && co + r < a.Cout) ? a.bias[co + r] : 0.f;
                    v[r] = act_apply(acc[mt][nt][r] + b, a.act);
                }
                if (FUSE && a.res != nullptr) {
                    const float* rp = (const float*)a.res + (((size_t)n * a.Hs + oy) * a.Ws + ox) * a.res_ld + co;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < a.Cout) v[r] += rp[r];
                }
                if (FUSE && a.fold != nullptr) {
                    const int ey = oy == 1 ? 0 : (oy == a.Hs - 2 ? a.Hs + 1 : -1);
                    const int ex = ox == 1 ? 0 : (ox == a.Ws - 2 ? a.Ws + 1 : -1);
                    const float* Fn = (const float*)a.fold + (size_t)n * (a.Hs + 2) * (a.Ws + 2) * a.fold_ld + co;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (co + r >= a.Cout) continue;
                        if (ey >= 0) v[r] += Fn[((size_t)ey * (a.Ws + 2) + ox + 1) * a.fold_ld + r];
                        if (ex >= 0) v[r] += Fn[((size_t)(oy + 1) * (a.Ws + 2) + ex) * a.fold_ld + r];
                        if (ey >= 0 && ex >= 0) v[r] += Fn[((size_t)ey * (a.Ws + 2) + ex) * a.fold_ld + r];
                    }
                }
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(yp + co) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < a.Cout) yp[co + r] = v[r];
                }
            }
        }
    }
}

template <typename T, typename OutT, int BN, int WM, int WN, int KCH, int ABUF, int TH = 16, bool FUSE = false, int KWC = 0, bool MC = false, bool PK = false,
          bool S2D = false, bool NIE = false>
static int launch_halo_cfg(const ConvArgs& a, hipStream_t st, int* tiles_out = nullptr) {
    if constexpr (!NIE && KWC == 3 && (TH == 16 || (TH == 8 && !PK)) && BN == 128 && WM == 4 && WN == 2 && KCH == 8 && !MC && !S2D &&
                  sizeof(T) == 2 && ((std::is_same<OutT, bf16_t>::value && !PK) || (std::is_same<OutT, bfpair_t>::value && PK))) {
        // InstanceNorm in the epilogue (ConvArgs::nie_sync): the residual blocks' 3x3 launches, plain or with the skip added
        if (a.nie_sync != nullptr) {
            if ((FUSE && (a.fold != nullptr || a.bstats != nullptr)) || a.stats == nullptr || a.Cout % BN) return CTG_EINVAL;
            return launch_halo_cfg<T, OutT, BN, WM, WN, KCH, ABUF, TH, FUSE, KWC, false, PK, false, true>(a, st, tiles_out);
        }
    }
    if constexpr (!S2D && !FUSE && !MC && KWC == 0 && TH == 16 && sizeof(T) == 2 && KCH == 8 && BN >= 64 && !std::is_same<OutT, float>::value) {
        if (a.s2d) {      // stride-2 conv as polyphase stride-1 slices (plain launches only)
            if (a.res != nullptr || a.fold != nullptr) return -1;
            return launch_halo_cfg<T, OutT, BN, WM, WN, KCH, ABUF, TH, false, 0, false, PK, true>(a, st, tiles_out);
        }
    }
    if (a.s2d && !S2D) return -1;
    if constexpr (!FUSE && !MC && !S2D && !NIE) {   // launches with an epilogue residual / frame fold are their own kernel
        if (a.res != nullptr || a.fold != nullptr) return launch_halo_cfg<T, OutT, BN, WM, WN, KCH, ABUF, TH, true, KWC, false, PK>(a, st, tiles_out);
    }
    if constexpr (!MC && !S2D && !FUSE && KWC == 0 && (TH == 16 || (TH == 8 && BN == 64)) && sizeof(T) == 2 &&
                  ((std::is_same<OutT, bf16_t>::value && !PK) || (std::is_same<OutT, bfpair_t>::value && PK))) {   // four parity classes, one launch
        if (a.ncls == 4) return launch_halo_cfg<T, OutT, BN, WM, WN, KCH, ABUF, TH, false, 0, true, PK>(a, st, tiles_out);
    }
    if constexpr (KWC == 0 && !MC && !S2D && sizeof(T) == 2) {   // bf16 3x3 windows: compile-time halo pitch
        if (a.kw == 3 && a.kh <= 3 && a.ncls <= 1) return launch_halo_cfg<T, OutT, BN, WM, WN, KCH, ABUF, TH, FUSE, 3, false, PK>(a, st, tiles_out);
        // (a compile-time 4x4 window for the PatchGAN's 256 -> 512 stride-1 layers measured +-0 in both modes: not instantiated)
    }
    if (a.ncls > 1 && !MC && !S2D) return -1;   // not served by this configuration
    if (a.nie_sync != nullptr && !NIE) return CTG_EINVAL;      // (ctg_conv_igemm only asks for shapes the branch above serves)
    constexpr int NTH = WM * WN * 64;
    const int hpw = HALO_W + a.kw - 1, hph = TH + a.kh - 1;   // (MC: the host put the largest class window into kw / kh)
    const int hpc = hph * hpw * KCH;
    const int hpc64 = (hpc + 63) & ~63;
    const int epc = VecOf<T>::N;
    const int nchunk = a.Cin / (KCH * epc);
    const int main_lds = (((nchunk > 1 && ABUF == 2) ? 2 : 1) * hpc64 + 2 * BN * KCH) * 16;
    const int epi_lds = std::is_same<OutT, bfpair_t>::value ? TH * HALO_W * ((BN / WN) * 4 + 16)
                        : sizeof(OutT) == 2 ? TH * HALO_W * (BN * 2 + 16) : 0;
    if (FUSE && a.bstats != nullptr && WM * WN * 64 * 64 > (main_lds > epi_lds ? main_lds : epi_lds)) return -1;   // [NTH][16] floats
    const int smem = main_lds > epi_lds ? main_lds : epi_lds;
    if (smem > 160 * 1024 || hph * hpw >= 65536) return -1;   // -> gather-GEMM
    static unsigned long long attr_mask = 0;       // per device
    {
        const int rc = ctg_lds_attr_once((const void*)conv_halo_kernel<T, OutT, BN, WM, WN, KCH, ABUF, TH, FUSE, KWC, MC, PK, S2D, NIE>,
                                         160 * 1024, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const int tiles = ((a.Hs + TH - 1) / TH) * ((a.Ws + HALO_W - 1) / HALO_W) * (MC ? 4 : 1);
    if (tiles_out != nullptr) *tiles_out = tiles;
    const int ntn = (a.Cout + BN - 1) / BN;
    if constexpr (NIE) {
        // residency (include/ctagan_hip.h, ctg_conv_epilogue): every workgroup of one sample -- its ntn statistics groups are
        // interleaved in dispatch order -- must fit this launch's share of the chip's workgroup slots
        static int occ_dev[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        dev &= 63;
        if (occ_dev[dev] == 0) {
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                    &occ, (const void*)conv_halo_kernel<T, OutT, BN, WM, WN, KCH, ABUF, TH, FUSE, KWC, MC, PK, S2D, NIE>, NTH, smem) != hipSuccess ||
                occ < 1)
                occ = 1;
            occ_dev[dev] = occ;
        }
        static const int share = getenv("CTG_NIE_SHARE") && atoi(getenv("CTG_NIE_SHARE")) > 0 ? atoi(getenv("CTG_NIE_SHARE")) : 2;
        if ((long)tiles * ntn * share > (long)occ_dev[dev] * ctg_cu_count()) return 2;      // not served: nothing launched
    }
    dim3 grid(tiles * ntn, a.B);
    hipLaunchKernelGGL((conv_halo_kernel<T, OutT, BN, WM, WN, KCH, ABUF, TH, FUSE, KWC, MC, PK, S2D, NIE>), grid, dim3(NTH), smem, st, a);
    return ctg_launch_status();
}

// returns -1 when the shape is not served by the halo kernel.
// Shipped configurations only.  Measured and rejected on MI355X (interleaved A/B, numbers in DESIGN.md section 5): larger
// wave tiles with two 4-wave workgroups per CU, one 8- or 16-wave BN=256 workgroup per CU, 8x16-pixel tiles, an
// all-taps-resident mode for the narrow layers, a "column stage" loop sharing pixel fragments between the taps of a kernel
// column, and a phase-split halo for stride-2 inputs.
static int launch_strip32(const ConvArgs& a, hipStream_t st, int* tiles_out, bool pair);     // conv_strip.h

template <typename T, int KCH>
static int launch_halo_t(const ConvArgs& a, int out_f32, hipStream_t st, int* tiles_out) {
    if (a.is != 1 && !a.s2d) return -1;
    if (out_f32 >= 2) {      // split-pair input ("bf16x3" mode): 2 = split-pair result (Cout % 8 == 0), 3 = fp32 result
        if constexpr (sizeof(T) == 2 && KCH == 8) {
            if (out_f32 == 2) {
                if (a.Cout > 64) return launch_halo_cfg<T, bfpair_t, 128, 4, 2, 8, 1, 16, false, 0, false, true>(a, st, tiles_out);
                if (a.Cout > 32) {
                    // merged parity classes of the 64-channel layers (u2 forward, d1 backward-data: 65536 short-lived workgroups):
                    // 8-row tiles -- 35.6 KB of LDS and 88 registers, so FOUR workgroups share a CU instead of three; what hides a
                    // short workgroup's serial chain is the number of workgroups in flight (887 -> 782 us; bf16 / 128-channel
                    // tiles, where the 8-row form stays at two workgroups per CU: 256 -> 274 us, not used)
                    static const bool th8c_off = getenv("CTG_NO_MC_TH8") != nullptr;      // A/B switch
                    if (a.ncls == 4 && !th8c_off) return launch_halo_cfg<T, bfpair_t, 64, 4, 1, 8, 1, 8, false, 0, false, true>(a, st, tiles_out);
                    static const bool th8_64_off = getenv("CTG_NO_TH8_64") != nullptr;      // A/B switch (both modes)
                    if (!th8_64_off && a.ncls <= 1 && a.Hs >= 16)      // (as for bf16 below: bf16x3 +0.45 %)
                        return launch_halo_cfg<T, bfpair_t, 64, 4, 1, 8, 1, 8, false, 0, false, true>(a, st, tiles_out);
                    return launch_halo_cfg<T, bfpair_t, 64, 4, 1, 8, 1, 16, false, 0, false, true>(a, st, tiles_out);
                }
                if (a.Cout > 16) {
                    // 32 -> 32 channel 3x3 layers on large maps (Reg's full-resolution level): sliding-window kernel
                    const int rc = launch_strip32(a, st, tiles_out, true);
                    if (rc != -1) return rc;
                    return launch_halo_cfg<T, bfpair_t, 32, 4, 1, 8, 1, 16, false, 0, false, true>(a, st, tiles_out);
                }
                return -1;
            }
            if (a.Cout > 64) return launch_halo_cfg<T, float, 128, 4, 2, 8, 1, 16, false, 0, false, true>(a, st, tiles_out);
            if (a.Cout > 32) return launch_halo_cfg<T, float, 64, 4, 1, 8, 1, 16, false, 0, false, true>(a, st, tiles_out);
            if (a.Cout > 16) return launch_halo_cfg<T, float, 32, 4, 1, 8, 1, 16, false, 0, false, true>(a, st, tiles_out);
            return launch_halo_cfg<T, float, 16, 4, 1, 8, 1, 16, false, 0, false, true>(a, st, tiles_out);
        }
        return -1;
    }
    // bf16 operands with an fp32 result wider than 16 channels: the split-bf16 ("bf16x3") mode, whose K dimension carries
    // [hi | hi | lo] x [hi | lo | hi] and whose accumulators leave unrounded
    if (a.Cout > 64) {
        if (out_f32) {
            if constexpr (sizeof(T) == 2) return launch_halo_cfg<T, float, 128, 4, 2, KCH, 1>(a, st, tiles_out);
            return -1;
        }
        // small batches (the reference ships batchSize 1): 16x16-pixel tiles leave most of the 512 workgroup slots of the
        // chip empty (128^2 x B=1 = 128 workgroups); 8x16-pixel tiles double the workgroups at the same bytes per FLOP
        if constexpr (sizeof(T) == 2) {
            static const bool th8_off = getenv("CTG_NO_TH8") != nullptr;
            static const long th8_wgs = getenv("CTG_TH8_WGS") ? atol(getenv("CTG_TH8_WGS")) : 384;   // A/B knob
            // (the merged parity-class launch has four workgroups per spatial tile and no 8-row instantiation: it keeps 16x16)
            const long wgs = (long)((a.Hs + 15) / 16) * ((a.Ws + HALO_W - 1) / HALO_W) * ((a.Cout + 127) / 128) * a.B *
                             (a.ncls == 4 ? 4 : 1);
            if (!th8_off && a.ncls != 4 && wgs < th8_wgs && a.Hs >= 16)     // (ctg_conv_igemm's nie_tiles check mirrors this choice)
                return launch_halo_cfg<T, T, 128, 4, 2, KCH, 1, 8>(a, st, tiles_out);
        }
        return launch_halo_cfg<T, T, 128, 4, 2, KCH, 1>(a, st, tiles_out);
    }
    if (a.Cout > 32) {
        if (out_f32) {
            if constexpr (sizeof(T) == 2) return launch_halo_cfg<T, float, 64, 4, 1, KCH, 1>(a, st, tiles_out);
            return -1;
        }
        if constexpr (sizeof(T) == 2 && KCH == 8) {
            // 64-channel tiles (Reg's 64 -> 64 layers at 256^2 and below: thousands of workgroups of 18 tap steps each) on 8-row
            // tiles: 39 KB of LDS instead of 57.5, so four workgroups share a CU instead of two -- what hides a short workgroup's
            // serial chain is the number of workgroups in flight (bf16 step +0.9 %)
            static const bool th8_64_off = getenv("CTG_NO_TH8_64") != nullptr;      // A/B switch
            // (also the merged parity classes of the PatchGAN's 128 -> 64 backward-data: 132 -> 108 us)
            if (!th8_64_off && (a.ncls <= 1 || a.ncls == 4) && a.Hs >= 16) return launch_halo_cfg<T, T, 64, 4, 1, KCH, 1, 8>(a, st, tiles_out);
        }
        return launch_halo_cfg<T, T, 64, 4, 1, KCH, 1>(a, st, tiles_out);
    }
    if (a.Cout > 16) {
        if (out_f32) {
            if constexpr (sizeof(T) == 2) return launch_halo_cfg<T, float, 32, 4, 1, KCH, 1>(a, st, tiles_out);
            return -1;
        }
        if constexpr (sizeof(T) == 2 && KCH == 4) {
            // 32 -> 32 channel 3x3 layers on large maps (Reg's full-resolution level): wave-autonomous sliding-window kernel
            const int rc = launch_strip32(a, st, tiles_out, false);
            if (rc != -1) return rc;
        }
        return launch_halo_cfg<T, T, 32, 4, 1, KCH, 1>(a, st, tiles_out);      // (8-row tiles: +-0 for these)
    }
    if (out_f32 || sizeof(T) == 4) return launch_halo_cfg<T, float, 16, 4, 1, KCH, 1>(a, st, tiles_out);
    return -1;
}
