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
// backward-data pass of both (the dual conv: W slices transposed by the weight
// packer).  Reflection padding is done in the gather, so nn.ReflectionPad2d
// (Model/HdGan.py:53,57,69,100) never materialises.
//
// GEMM mapping: M = output pixels of ONE sample (tiles never straddle samples),
// N = Cout, K = taps x Cin.  NHWC activations make every A row a run of
// contiguous channels; weights are pre-packed [tap][Cout][Cin].
//
// Structure (256 threads = 4 waves, tile BM x BN, KCH 16-byte chunks of K per step):
//  * both tiles go global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no ds_write);
//    the per-lane SOURCE address does the gather (tap shift, reflection, zero padding via a 16-byte
//    zero page) and carries the inverse of the LDS XOR swizzle, the LDS image itself is lane-linear;
//  * LDS double buffered, one barrier per K-step, next step's loads issued before the MFMA cluster;
//  * fragments by ds_read_b128 from the XOR-swizzled image (conflict-free);
//  * MFMA with the WEIGHTS as the A operand and the pixels as B, so a lane ends up holding 4
//    consecutive output channels of one pixel: bf16 results are staged through LDS and leave as whole
//    16-byte channel chunks (coalesced NHWC rows), fp32 results as float4;
//      bf16: v_mfma_f32_16x16x32_bf16      fp32: 4 x v_mfma_f32_16x16x4_f32 per chunk (exact f32; the
//      K order inside a step is permuted identically for both operands);
//  * workgroup -> tile map keeps the N-tiles of one M-tile adjacent and gives each XCD (blockIdx % 8) a
//    contiguous run of M-tiles, so halo rows and the shared A rows hit the same L2.
#include <stdlib.h>
#include "conv_halo.h"   // ConvArgs, swz<>, LDS-DMA pointer types, and the halo-resident stride-1 kernel
#include "conv_strip.h"  // the wave-autonomous sliding-window kernel for the 32 -> 32 channel layers on large maps
#include "conv_stript.h" // the cooperative sliding window for the 128 -> 64 channel stride-2 transposed conv (four parity classes per step)
#include "conv_pair_strips.h" // ... their split-pair ("bf16x3") forms live in their own translation unit (conv_pair_strips.hip)
#include "conv_strips2.h" // and for the 64 -> 128 channel stride-2 conv
#include "conv_strips2w.h" // ... and the 128 -> 256 channel one (eight waves, half of the output channels per workgroup)

__device__ __attribute__((aligned(16))) unsigned g_zero_chunk[4];  // source of zero padding for the LDS-DMA gathers

#ifndef CTG_BIG_TILE
#define CTG_BIG_TILE 1
#endif

// sub-grid pixel of linear index m.  frame mode enumerates only the 1-pixel frame of the grid -- the part of a padded-grid
// backward-data pass that the 16x16-tiled halo kernel would serve with 17 ragged tiles of 81 -- EDGE BY EDGE, each edge padded to a
// multiple of 128 slots (top row, bottom row, left column, right column; a pad slot repeats its edge's last pixel: the same operands
// in the same order give the same bits, so its store rewrites that pixel with the value it already has).  A 128-pixel tile then lies
// on ONE edge, and only the 3 taps that point inwards from that edge are walked (the kernel's live-tap scan): round 6, 38.7 -> 29 us
// with the scan alone (a tile that spans two edges walks 5-6 taps), edge-aligned tiles on top.
#define FRAME_EDGE_PAD 128
__device__ __forceinline__ void grid_pixel(const ConvArgs& a, int m, int& j, int& i) {
    if (a.frame) {
        const int l1 = a.Hs - 2;
        const int p0 = (a.Ws + FRAME_EDGE_PAD - 1) & ~(FRAME_EDGE_PAD - 1), p1 = (l1 + FRAME_EDGE_PAD - 1) & ~(FRAME_EDGE_PAD - 1);
        if (m < p0) { j = 0; i = m < a.Ws ? m : a.Ws - 1; }
        else if (m < 2 * p0) { j = a.Hs - 1; i = m - p0 < a.Ws ? m - p0 : a.Ws - 1; }
        else {
            const int r = m - 2 * p0;
            if (r < p1) { j = (r < l1 ? r : l1 - 1) + 1; i = 0; }
            else { j = (r - p1 < l1 ? r - p1 : l1 - 1) + 1; i = a.Ws - 1; }
        }
    } else {
        j = m / a.Ws;
        i = m - j * a.Ws;
    }
}
__host__ __device__ __forceinline__ int grid_pixels(const ConvArgs& a) {
    if (a.frame)
        return 2 * ((a.Ws + FRAME_EDGE_PAD - 1) & ~(FRAME_EDGE_PAD - 1)) + 2 * ((a.Hs - 2 + FRAME_EDGE_PAD - 1) & ~(FRAME_EDGE_PAD - 1));
    return a.Hs * a.Ws;
}

// NST = LDS ring depth.  2: one __syncthreads() per step (it also drains the LDS-DMA).  3: the loads of step
// s+2 are issued before the MFMAs of step s and stay in flight ACROSS the barrier; each wave retires exactly the
// stage it is about to read with a counted s_waitcnt vmcnt(loads per stage) in front of a raw s_barrier.
template <typename T, typename OutT, int BM, int BN, int WM, int WN, int KCH, int NST, bool PK = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(const ConvArgs a) {
    static_assert(!PK || (KCH == 8 && sizeof(T) == 2), "split-pair K steps (see ConvArgs::pair_lo)");
    constexpr int NTH = WM * WN * 64;
    constexpr int EPC = VecOf<T>::N;
    constexpr int BKE = KCH * EPC;
    constexpr int TM = BM / (WM * 16), TN = BN / (WN * 16);
    constexpr int A_CH = BM * KCH, B_CH = BN * KCH;
    constexpr int A_IT = A_CH / NTH, B_IT = (B_CH + NTH - 1) / NTH;
    static_assert(A_CH % NTH == 0 && (B_CH % NTH == 0 || B_CH < NTH) && B_CH % 64 == 0, "tile");
    static_assert(NST == 2 || (NST == 3 && B_CH % NTH == 0), "3-stage ring needs every wave to issue the same loads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + NST * A_CH * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n = blockIdx.y;
    // XCD-aware tile map over gridDim.x = mtiles * ntiles
    const int ntn = (a.Cout + BN - 1) / BN;
    int id = blockIdx.x;
    id = xcd_contiguous(id, gridDim.x);
    const int m0 = (id / ntn) * BM, n0 = (id % ntn) * BN;
    const int Ms = grid_pixels(a);
    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ W = (const T*)a.w;

    // ---- per-thread gather bookkeeping: LDS slot s = tid + 256*it  <->  (row, swizzled chunk)
    int aj[A_IT], ai[A_IT], akc[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int s = tid + NTH * it;
        const int row = s / KCH;
        const int kcs = swz<KCH>(row, s % KCH);       // source chunk that belongs in this slot
        akc[it] = PK ? (kcs & 3) * EPC + (kcs >> 2) * a.pair_lo : kcs * EPC;      // (PK: chunks 0-3 hi plane, 4-7 lo plane)
        int m = m0 + row;
        m = m < Ms ? m : Ms - 1;                      // tail rows gather a valid pixel; masked at the store
        int j, i;
        grid_pixel(a, m, j, i);
        aj[it] = j * a.is;
        ai[it] = i * a.is;
    }
    int boff[B_IT];
    const bool b_active = (B_CH % NTH == 0) || (tid < B_CH);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int s = (tid + NTH * it) % B_CH;
        const int row = s / KCH;
        boff[it] = (n0 + row) * a.Cin + swz<KCH>(row, s % KCH) * EPC;
    }
    const int kPerTap = a.Cin / BKE;
    // Taps that can reach a non-zero input pixel from THIS tile.  Everywhere but in frame mode that is every tap.  The 1-pixel frame of
    // a padded-grid backward-data pass is different: its source (the gradient on the unpadded grid, zero outside) is only within
    // reach of the taps that point inwards -- 3 of 9 on an edge, 1 in a corner -- and a tile of 128 consecutive frame pixels lies on
    // one edge (grid_pixel pads every edge to whole tiles), so two thirds of the K loop of these ~20-workgroup launches gathered zeros:
    // their serial chain was ~39 us whatever the batch (18 launches per generator backward).
    unsigned long long live = a.ntaps >= 64 ? ~0ull : ((1ull << a.ntaps) - 1ull);
    if (a.frame && a.pad_mode != PAD_REFLECT) {      // (uniform)
        unsigned long long mine = 0ull;
        for (int t = 0; t < a.ntaps; ++t) {
            const int tw = a.taps[t];
            const int dy = (tw & 0xff) - 64, dx = ((tw >> 8) & 0xff) - 64;
            bool any = false;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                any = any || (((unsigned)(aj[it] + dy) < (unsigned)a.Hi) && ((unsigned)(ai[it] + dx) < (unsigned)a.Wi));
            if (any) mine |= 1ull << t;
        }
        __shared__ unsigned long long live_s;
        if (tid == 0) live_s = 0ull;
        __syncthreads();
        if (mine) atomicOr(&live_s, mine);
        __syncthreads();
        live = live_s;
        if (live == 0ull) live = 1ull;      // (a tile out of everyone's reach still walks one tap of zeros: the loop structure stays)
    }
    const int S = __popcll(live) * kPerTap;
    auto next_tap = [&]() __attribute__((always_inline)) {
        const int t = __ffsll((long long)live) - 1;
        live &= live - 1ull;
        return t;
    };

    long aoff[A_IT];   // element offset of (pixel, channel 0) for the current tap, -1 = zero padding
    int wbase = 0;
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const int tw = a.taps[tap];
        const int dy = (tw & 0xff) - 64, dx = ((tw >> 8) & 0xff) - 64;
        wbase = (tw >> 16) * a.w_tap_stride;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int iy = aj[it] + dy, ix = ai[it] + dx;
            bool ok = true;
            if (a.pad_mode == PAD_REFLECT) {
                iy = reflect_idx(iy, a.Hi);
                ix = reflect_idx(ix, a.Wi);
            } else {
                ok = ((unsigned)iy < (unsigned)a.Hi) && ((unsigned)ix < (unsigned)a.Wi);
            }
            aoff[it] = ok ? ((((long)n * a.Hi + iy) * a.Wi + ix) * a.x_ld + akc[it]) : -1L;
        }
    };
    auto issue = [&](int buf, int kc0) __attribute__((always_inline)) {
        const int xk = PK ? kc0 >> 1 : kc0;       // (PK: a K step is 32 input channels of both planes, 64 weight elements)
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const T* src = aoff[it] >= 0 ? X + aoff[it] + xk : (const T*)g_zero_chunk;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + (buf * A_CH + NTH * it + 64 * wave) * 16), 16, 0, 0);
        }
        if (b_active) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const T* src = W + wbase + boff[it] + kc0;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sB + (buf * B_CH + NTH * it + 64 * wave) * 16), 16, 0, 0);
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
        if constexpr (PK) {
            // chunks 0-3 of a row: hi halves, 4-7: lo halves -- hi.w_hi, hi.w_lo, lo.w_hi (conv_halo.h)
            u32x4 fa[TM], fb[TN], fl[TN];
            const int kc = lane >> 4;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int row = (wm * TM + mt) * 16 + (lane & 15);
                fa[mt] = *reinterpret_cast<const u32x4*>(pa + (row * KCH + swz<KCH>(row, kc)) * 16);
            }
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int row = (wn * TN + nt) * 16 + (lane & 15);
                fb[nt] = *reinterpret_cast<const u32x4*>(pb + (row * KCH + swz<KCH>(row, kc)) * 16);
                fl[nt] = *reinterpret_cast<const u32x4*>(pb + (row * KCH + swz<KCH>(row, kc + 4)) * 16);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fb[nt]), __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fl[nt]), __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
                }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int row = (wm * TM + mt) * 16 + (lane & 15);
                fa[mt] = *reinterpret_cast<const u32x4*>(pa + (row * KCH + swz<KCH>(row, kc + 4)) * 16);
            }
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
                    // weights as A, pixels as B:  D[row = co][col = pixel]
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

    int kk = 0;
    auto advance = [&]() __attribute__((always_inline)) {
        if (++kk == kPerTap) {
            kk = 0;
            set_tap(next_tap());
        }
    };
    set_tap(next_tap());
    if constexpr (NST == 2) {
        // ---- one barrier per K-step; __syncthreads() also drains the LDS-DMA (vmcnt(0))
        issue(0, 0);
        __syncthreads();
        for (int s = 0; s < S; ++s) {
            const int cur = s & 1;
            if (s + 1 < S) {
                advance();
                issue(cur ^ 1, kk * BKE);
            }
            compute(cur);
            __syncthreads();
        }
    } else {
        // ---- 3-deep ring, prefetch distance 2, counted vmcnt + raw barrier (no vmcnt(0) inside the loop)
        constexpr int LOADS = A_IT + B_IT;
        issue(0, 0);
        if (S > 1) {
            advance();
            issue(1, kk * BKE);
        }
        int st = 0;  // stage holding step s
        for (int s = 0; s < S; ++s) {
            if (s + 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + 2 < S) {
                advance();
                issue(st == 0 ? 2 : st - 1, kk * BKE);   // stage (s+2) % 3, last read in step s-1
            }
            compute(st);
            st = st == 2 ? 0 : st + 1;
        }
        __syncthreads();
    }

    // ---- epilogue.  acc[mt][nt][r]: pixel = (wm*TM+mt)*16 + (lane & 15), co = (wn*TN+nt)*16 + (lane >> 4)*4 + r
    const int co_l = (lane >> 4) * 4;
    if (a.stats != nullptr) {
        // InstanceNorm moments of this M tile (as in conv_halo.h): per-(sample, m-tile, channel) partial (sum, sum of
        // squares) of the bias-free fp32 results, reduced lane -> 16-lane row (DPP) -> waves (LDS), fixed order
        float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]; the ring buffers are free after the last barrier
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const bool valid = m0 + (wm * TM + mt) * 16 + (lane & 15) < Ms;
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
        const int mtiles = gridDim.x / ntn;
        for (int cl = tid; cl < BN; cl += NTH) {
            if (n0 + cl < a.Cout) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + cl) * 2]; t2 += red[(w * BN + cl) * 2 + 1]; }
                float* dst = a.stats + (((size_t)n * mtiles + id / ntn) * a.Cout + n0 + cl) * 2;
                dst[0] = t1;
                dst[1] = t2;
            }
        }
        __syncthreads();
    }
    if constexpr (std::is_same<OutT, bfpair_t>::value) {
        // split-pair result: unrounded fp32 accumulators staged through LDS, one round per N half of the workgroup, leaving as
        // 16-byte chunks of the hi and of the lo plane (as in conv_halo.h)
        constexpr int CR = TN * 16, RS = CR * 4 + 16, CPR = CR / 8, NIT = BM * CPR / NTH;
        static_assert(BM * CPR % NTH == 0, "pair epilogue: whole trips");
        char* st = smem;
        bf16_t* __restrict__ Y = (bf16_t*)a.y;
        const int y_lo = a.y_ld >> 1;
#pragma unroll
        for (int rd = 0; rd < WN; ++rd) {
            if (rd) __syncthreads();
            if (wn == rd) {
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int co = (wn * TN + nt) * 16 + co_l;
                    float bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        bv[r] = (a.bias != nullptr && n0 + co + r < a.Cout) ? a.bias[n0 + co + r] : 0.f;
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
                        const int prow = (wm * TM + mt) * 16 + (lane & 15);
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = act_apply(acc[mt][nt][r] + bv[r], a.act);
                        *reinterpret_cast<f32x4*>(st + prow * RS + (nt * 16 + co_l) * 4) = o;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = tid + NTH * it;
                const int prow = c / CPR, q = c % CPR, ch = rd * CR + q * 8;
                const int m = m0 + prow;
                if (m < Ms && n0 + ch < a.Cout) {
                    int j, i;
                    grid_pixel(a, m, j, i);
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + prow * RS + q * 32);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + prow * RS + q * 32 + 16);
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (bf16_t)v0[e];
                        lo[e] = (bf16_t)(v0[e] - (float)hi[e]);
                        hi[4 + e] = (bf16_t)v1[e];
                        lo[4 + e] = (bf16_t)(v1[e] - (float)hi[4 + e]);
                    }
                    bf16_t* yp = Y + (((size_t)n * a.Ho + (j * a.os + a.oy0)) * a.Wo + (i * a.os + a.ox0)) * a.y_ld + n0 + ch;
                    *reinterpret_cast<bf16x8*>(yp) = hi;
                    *reinterpret_cast<bf16x8*>(yp + y_lo) = lo;
                }
            }
        }
    } else if constexpr (sizeof(OutT) == 2) {
        // bf16: stage the tile [pixel][co] in LDS (row pitch padded by 16 B), then whole 16-byte chunks leave
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
                const int prow = (wm * TM + mt) * 16 + (lane & 15);
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)act_apply(acc[mt][nt][r] + bv[r], a.act);
                *reinterpret_cast<bf16x4*>(st + prow * RS + co * 2) = o;
            }
        }
        __syncthreads();
        OutT* __restrict__ Y = (OutT*)a.y;
        constexpr int CPR = BN / 8;  // 16-byte chunks per pixel row of the tile
#pragma unroll
        for (int it = 0; it < BM * CPR / NTH; ++it) {
            const int c = tid + NTH * it;
            const int prow = c / CPR, ch = (c % CPR) * 8;
            const int m = m0 + prow;
            if (m < Ms && n0 + ch < a.Cout) {
                int j, i;
                grid_pixel(a, m, j, i);
                OutT* yp = Y + (((size_t)n * a.Ho + (j * a.os + a.oy0)) * a.Wo + (i * a.os + a.ox0)) * a.y_ld + n0 + ch;
                *reinterpret_cast<u32x4*>(yp) = *reinterpret_cast<const u32x4*>(st + prow * RS + ch * 2);
            }
        }
    } else {
        OutT* __restrict__ Y = (OutT*)a.y;
        const bool vec_ok = ((a.Cout & 3) == 0) && ((a.y_ld & 3) == 0);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m0 + (wm * TM + mt) * 16 + (lane & 15);
            if (m >= Ms) continue;
            int j, i;
                grid_pixel(a, m, j, i);
            OutT* yp = Y + (((size_t)n * a.Ho + (j * a.os + a.oy0)) * a.Wo + (i * a.os + a.ox0)) * a.y_ld;
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

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int KCH, int NST, bool PK = false>
static int launch_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int main_lds = NST * (BM + BN) * KCH * 16;
    constexpr int epi_lds = std::is_same<OutT, bfpair_t>::value ? BM * ((BN / WN) * 4 + 16) : sizeof(OutT) == 2 ? BM * (BN * 2 + 16) : 0;
    constexpr int smem = main_lds > epi_lds ? main_lds : epi_lds;
    static_assert(smem <= 160 * 1024, "LDS");
    static unsigned long long attr_mask = 0;       // per device
    if (smem > 65536) {
        const int rc = ctg_lds_attr_once((const void*)conv_igemm_kernel<T, OutT, BM, BN, WM, WN, KCH, NST, PK>, smem, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const int mt = (grid_pixels(a) + BM - 1) / BM, nt = (a.Cout + BN - 1) / BN;
    dim3 grid(mt * nt, a.B);
    hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, BM, BN, WM, WN, KCH, NST, PK>), grid, dim3(WM * WN * 64), smem, st, a);
    return ctg_launch_status();
}

template <typename T, int KCH>
static int launch_t(const ConvArgs& a, int out_f32, hipStream_t st) {
    if (out_f32 >= 2) {      // split-pair input ("bf16x3" mode): 2 = split-pair result, 3 = fp32 result
        if constexpr (sizeof(T) == 2 && KCH == 8) {
            if (out_f32 == 2) {
                // (the 1-pixel frame of a padded grid: few pixels, long K -- narrower N tiles double the workgroups in flight)
                // (launches of a few workgroups: the 3-stage ring, as on the bf16 path below)
                static const bool ring_off_p = getenv("CTG_NO_SMALL_RING") != nullptr;
                const bool small_p = !ring_off_p && (long)((grid_pixels(a) + 127) / 128) * ((a.Cout + 63) / 64) * a.B <= 1024;
                if (a.Cout > 64 && a.frame)
                    return small_p ? launch_cfg<T, bfpair_t, 128, 64, 4, 1, 8, 3, true>(a, st) : launch_cfg<T, bfpair_t, 128, 64, 4, 1, 8, 2, true>(a, st);
                // (the 256x128 three-stage ring tile of the bf16 path measured no faster here: 119.0 vs 118.5 ms per step)
                if (a.Cout > 64)
                    return small_p ? launch_cfg<T, bfpair_t, 128, 128, 2, 2, 8, 3, true>(a, st) : launch_cfg<T, bfpair_t, 128, 128, 2, 2, 8, 2, true>(a, st);
                if (a.Cout > 32)
                    return small_p ? launch_cfg<T, bfpair_t, 128, 64, 4, 1, 8, 3, true>(a, st) : launch_cfg<T, bfpair_t, 128, 64, 4, 1, 8, 2, true>(a, st);
                if (a.Cout > 16) return launch_cfg<T, bfpair_t, 128, 32, 4, 1, 8, 2, true>(a, st);
                return CTG_EINVAL;
            }
            if (a.Cout > 64) return launch_cfg<T, float, 128, 128, 2, 2, 8, 2, true>(a, st);
            if (a.Cout > 32) return launch_cfg<T, float, 128, 64, 4, 1, 8, 2, true>(a, st);
            if (a.Cout > 16) return launch_cfg<T, float, 128, 32, 4, 1, 8, 2, true>(a, st);
            return launch_cfg<T, float, 128, 16, 4, 1, 8, 2, true>(a, st);
        }
        return CTG_EINVAL;
    }
    if (a.Cout > 64) {
        if (out_f32) {   // split-bf16 mode: bf16 operands, unrounded fp32 result
            if constexpr (sizeof(T) == 2) return launch_cfg<T, float, 128, 128, 2, 2, KCH, 2>(a, st);
            return CTG_EINVAL;
        }
        // wide layers at scale (the residual-block convs): 256x128 tile, 8 waves, 3-stage LDS-DMA ring
        static const bool big_off = getenv("CTG_NO_BIG_TILE") != nullptr;
        if (sizeof(T) == 2 && KCH == 8 && grid_pixels(a) >= 4096 && CTG_BIG_TILE && !big_off)
            return launch_cfg<T, T, 256, 128, 4, 2, 8, 3>(a, st);
        // the 1-pixel frame of a padded grid: few pixels, long K -- narrower N tiles double the workgroups in flight
        static const bool frame64 = getenv("CTG_FRAME_BN128") == nullptr;
        // A launch of a few workgroups is a serial chain of K steps: with the 2-stage form every step pays a whole global -> LDS
        // latency (its __syncthreads drains the loads it has just issued); the 3-stage ring keeps two steps of loads in flight
        // across the barrier (round 6: the frame launches and the U-Net's 4^2 ... 16^2 levels)
        static const bool ring_off = getenv("CTG_NO_SMALL_RING") != nullptr;      // A/B switch
        const bool small = sizeof(T) == 2 && !ring_off &&
                           (long)((grid_pixels(a) + 127) / 128) * ((a.Cout + 63) / 64) * a.B <= 1024;
        if (a.frame && frame64) {
            if constexpr (sizeof(T) == 2) if (small) return launch_cfg<T, T, 128, 64, 4, 1, KCH, 3>(a, st);
            return launch_cfg<T, T, 128, 64, 4, 1, KCH, 2>(a, st);
        }
        if constexpr (sizeof(T) == 2) if (small) return launch_cfg<T, T, 128, 128, 2, 2, KCH, 3>(a, st);
        return launch_cfg<T, T, 128, 128, 2, 2, KCH, 2>(a, st);
    }
    if (a.Cout > 32) {
        if (out_f32) {
            if constexpr (sizeof(T) == 2) return launch_cfg<T, float, 128, 64, 4, 1, KCH, 2>(a, st);
            return CTG_EINVAL;
        }
        if constexpr (sizeof(T) == 2) {
            static const bool ring_off64 = getenv("CTG_NO_SMALL_RING") != nullptr;
            if (!ring_off64 && (long)((grid_pixels(a) + 127) / 128) * a.B <= 512) return launch_cfg<T, T, 128, 64, 4, 1, KCH, 3>(a, st);
        }
        return launch_cfg<T, T, 128, 64, 4, 1, KCH, 2>(a, st);
    }
    if (a.Cout > 16) {
        if (out_f32) {
            if constexpr (sizeof(T) == 2) return launch_cfg<T, float, 128, 32, 4, 1, KCH, 2>(a, st);
            return CTG_EINVAL;
        }
        return launch_cfg<T, T, 128, 32, 4, 1, KCH, 2>(a, st);
    }
    if (out_f32 || sizeof(T) == 4) return launch_cfg<T, float, 128, 16, 4, 1, KCH, 2>(a, st);
    return CTG_EINVAL;  // bf16 output narrower than 17 channels does not occur on this path
}

// ---------------------------------------------------------------------------
// C ABI.  Replaces the ATen conv2d / conv_transpose2d (+ReflectionPad2d, +bias,
// +LeakyReLU/Tanh) forward and backward-data dispatches made by nn.Conv2d /
// nn.ConvTranspose2d at Model/HdGan.py:54,58,70,78,93,101,120-136 and
// trainer/layers.py:85,282,295.  Returns 0, CTG_EINVAL or 1000+hipError_t.
// ---------------------------------------------------------------------------
extern "C" int ctg_conv_igemm(int dtype, int out_f32, const void* x, const void* w, void* y, const float* bias,
                              int B, int Hi, int Wi, int Cin, int x_ld, int Ho, int Wo, int Cout, int y_ld,
                              int Hs, int Ws, int oy0, int ox0, int os, int is, int frame, int pad_mode, int act,
                              int w_npad, int ntaps, const int* taps_host, float* stats_part, int* stats_slabs_out,
                              const ctg_conv_epilogue* epi, void* stream) {
    CTG_ENTER();
    const void* res = epi ? epi->res : nullptr;
    const void* fold = epi ? epi->fold : nullptr;
    const int res_ld = epi ? epi->res_ld : 0, fold_ld = epi ? epi->fold_ld : 0;
    if (ntaps < 1 || ntaps > 64 || B < 1 || Hs < 1 || Ws < 1 || Cout < 1) return CTG_EINVAL;
    if (frame != 0 && (frame != 1 || Hs < 3 || Ws < 3)) return CTG_EINVAL;
    if (frame != 0 && stats_part != nullptr) return CTG_EINVAL;      // (frame tiles hold pad slots that repeat a pixel: see grid_pixel)
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_PAIR) return CTG_EINVAL;
    // DT_PAIR ("bf16x3" mode): x is a split-pair tensor ([hi | lo] planes per pixel row, lo at + x_ld / 2), Cin its channel count,
    // w the packed weights split along K as [w_hi 32 | w_lo 32] per 32 channels (ctg_split_weights: 2 Cin elements per row);
    // out_f32 == 0: split-pair result (y_ld its row pitch, res / fold / bz likewise), 1: fp32 result.  bf16 MFMA inside.
    const bool pair = dtype == DT_PAIR;
    if (pair) {
        if (x_ld % 16 != 0 || x_ld < 2 * Cin || Cin % 32 != 0 || (out_f32 != 0 && out_f32 != 1)) return CTG_EINVAL;
        if (!out_f32 && (y_ld % 16 != 0 || y_ld < 2 * Cout || Cout % 8 != 0 || ((uintptr_t)y & 15))) return CTG_EINVAL;
        dtype = DT_BF16;
    } else if (out_f32 != 0 && out_f32 != 1) return CTG_EINVAL;
    const int omode = pair ? (out_f32 ? 3 : 2) : out_f32;      // 0: T, 1: fp32; split-pair input: 2: split-pair, 3: fp32
    const int epc = dtype == DT_BF16 ? 8 : 4;
    if (Cin % (4 * epc) != 0 || x_ld % epc != 0 || x_ld < Cin || y_ld < Cout) return CTG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return CTG_EINVAL;
    const bool narrow_out = out_f32 || dtype == DT_F32;
    if (!narrow_out && (Cout % 8 != 0 || y_ld % 8 != 0 || ((uintptr_t)y & 15))) return CTG_EINVAL;  // 16-byte chunks
    // the weight slab must cover every N tile the grid touches
    const int bn = Cout > 64 ? 128 : Cout > 32 ? 64 : Cout > 16 ? 32 : 16;
    if (w_npad < ((Cout + bn - 1) / bn) * bn) return CTG_EINVAL;
    if ((Hs - 1) * os + oy0 >= Ho || (Ws - 1) * os + ox0 >= Wo) return CTG_EINVAL;
    if (stats_slabs_out != nullptr) *stats_slabs_out = 0;
    const bool fused = res != nullptr || fold != nullptr;
    if (fused) {
        // epilogue-fused residual / frame fold: unit-stride launches that cover the whole (unpadded) output, halo kernel only
        if (os != 1 || is != 1 || frame || oy0 || ox0 || Ho != Hs || Wo != Ws || Hs < 4 || Ws < 4 || (out_f32 && Cout <= 16)) return CTG_EINVAL;
        if (res != nullptr && (res_ld < Cout || res_ld % epc || ((uintptr_t)res & 15))) return CTG_EINVAL;
        if (fold != nullptr && (fold_ld < Cout || fold_ld % epc || ((uintptr_t)fold & 15))) return CTG_EINVAL;
    }
    ConvArgs a;
    a.stats = nullptr;
    a.ncls = 1;
    a.nie_sync = nullptr; a.nie_act = ACT_NONE; a.nie_budget = 0;
    a.res = res; a.fold = fold; a.res_ld = res_ld; a.fold_ld = fold_ld;
    a.bz = nullptr; a.bmean = nullptr; a.brstd = nullptr; a.bstats = nullptr; a.bz_ld = 0; a.bact = ACT_NONE;
    if (epi != nullptr && epi->bstats != nullptr) {
        // fused InstanceNorm-backward sums: bf16 launches with a fused fold / residual epilogue only
        if (!fused || dtype != DT_BF16 || (pair && out_f32) || epi->bz == nullptr || epi->bmean == nullptr || epi->brstd == nullptr ||
            epi->bz_ld < Cout || epi->bz_ld % 8 || ((uintptr_t)epi->bz & 15) || Cout <= 16) return CTG_EINVAL;
        if (epi->bact != ACT_NONE && epi->bact != ACT_RELU && epi->bact != ACT_LRELU) return CTG_EINVAL;
        a.bz = epi->bz; a.bmean = epi->bmean; a.brstd = epi->brstd; a.bstats = epi->bstats; a.bz_ld = epi->bz_ld;
        a.bact = epi->bact;
    }
    a.x = x; a.w = w; a.y = y; a.bias = bias;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.x_ld = x_ld;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.y_ld = y_ld;
    a.Hs = Hs; a.Ws = Ws; a.oy0 = oy0; a.ox0 = ox0; a.os = os; a.is = is; a.frame = frame;
    a.pad_mode = pad_mode; a.act = act; a.w_tap_stride = w_npad * Cin; a.ntaps = ntaps;
    a.pair_lo = 0;
    const bool k8 = pair || (Cin % (8 * epc)) == 0;
    if (pair) {
        a.pair_lo = x_ld / 2;
        a.Cin = 2 * Cin;               // the K length of a weight row: [w_hi 32 | w_lo 32] per 32 channels
        a.w_tap_stride = w_npad * 2 * Cin;
        if (fused && ((res != nullptr && (res_ld % 16 || res_ld < 2 * Cout)) || (fold != nullptr && (fold_ld % 16 || fold_ld < 2 * Cout))))
            return CTG_EINVAL;
        if (a.bstats != nullptr && (a.bz_ld % 16 || a.bz_ld < 2 * Cout)) return CTG_EINVAL;
    }
    for (int t = 0; t < ntaps; ++t) {
        const int tw = taps_host[t];
        const int dy = (tw & 0xff) - 64, dx = ((tw >> 8) & 0xff) - 64;
        if (pad_mode == PAD_REFLECT) {
            // a single reflection must land inside the image for every output pixel
            const int ymax = (Hs - 1) * is + dy, xmax = (Ws - 1) * is + dx;
            if (-dy >= Hi || ymax - (Hi - 1) >= Hi || -dx >= Wi || xmax - (Wi - 1) >= Wi) return CTG_EINVAL;
        }
        a.taps[t] = tw;
    }
    hipStream_t st = (hipStream_t)stream;
    a.s2d = 0;
    if (epi != nullptr && epi->nie_sync != nullptr) {
        // InstanceNorm (+ activation, + skip) in the epilogue -- ConvArgs::nie_sync.  Served: a full 3x3 (or narrower) unit-stride window
        // on the 128-channel-tile halo kernel, bf16 or split pair, whole output, no bias / activation of its own, and a sample
        // whose workgroups (spatial tiles x channel tiles: the groups of a sample are interleaved in dispatch order) are resident at
        // once whatever else runs -- checked against the kernel's real occupancy in launch_halo_cfg (CTG_NIE_SHARE launches, default
        // two -- streams, processes on one card -- can wait at the same time without starving each other).  Anything else: 2.
        int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
        for (int t = 0; t < ntaps; ++t) {
            const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
            dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax;
            dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
        }
        const int kh = dymax - dymin + 1, kw = dxmax - dxmin + 1;
        long tiles = (long)((Hs + 15) / 16) * ((Ws + 15) / 16);
        {   // the 8-row tile variant launch_halo_t picks for small bf16 grids: twice the tiles per sample
            static const bool th8_off = getenv("CTG_NO_TH8") != nullptr;
            static const long th8_wgs = getenv("CTG_TH8_WGS") ? atol(getenv("CTG_TH8_WGS")) : 384;
            if (!pair && !th8_off && tiles * ((Cout + 127) / 128) * B < th8_wgs) tiles = (long)((Hs + 7) / 8) * ((Ws + 15) / 16);
        }
        static const bool nie_off = getenv("CTG_NO_NIE") != nullptr;
        if (nie_off || dtype != DT_BF16 || out_f32 || !k8 || Cout % 128 || bias != nullptr || act != ACT_NONE || fold != nullptr ||
            a.bstats != nullptr || frame || os != 1 || is != 1 || oy0 || ox0 || Ho != Hs || Wo != Ws || Hs < 16 || Ws < 16 ||
            ntaps != kh * kw || ntaps < 2 || kw != 3 || kh > 3 ||
            stats_part == nullptr || stats_slabs_out == nullptr ||
            (long)Hi * Wi * x_ld >= (1L << 31) || getenv("CTG_NO_HALO") != nullptr)
            return 2;
        if (epi->nie_act != ACT_NONE && epi->nie_act != ACT_RELU && epi->nie_act != ACT_LRELU) return CTG_EINVAL;
        if (epi->nie_tiles != (int)tiles || ((uintptr_t)epi->nie_sync & 7)) return CTG_EINVAL;
        if ((long)B * (Cout / 128) > (long)epi->nie_groups || epi->nie_budget < 0) return CTG_EINVAL;      // the counters the kernel indexes
        a.nie_sync = (unsigned long long*)epi->nie_sync; a.nie_act = epi->nie_act;
        a.nie_budget = epi->nie_budget > 0 ? epi->nie_budget : (1 << 22);
    }
    // ---- split-pair mode: the 64 -> 128 channel stride-2 3x3 conv on large maps has its own sliding-window kernel (conv_strips2p.h)
    if (pair && !out_f32 && !fused && epi == nullptr) {
        const bool ws = stats_part != nullptr && stats_slabs_out != nullptr;
        int slabs = 0;
        a.ncls = 1;
        const int rc = pairstrip_launch_s2(a, ws ? stats_part : nullptr, st, &slabs);
        if (rc != -1) {
            if (rc == 0 && ws) *stats_slabs_out = slabs;
            return rc;
        }
    }
    // ---- stride-2 convs as polyphase stride-1 slices on the halo-resident kernel (ConvArgs::s2d): the PatchGAN's 4x4 stride-2
    // layers (Model/HdGan.py:124-131) in bf16 -- the generator's 3x3 ones have their sliding-window kernels below -- and every
    // stride-2 conv of the split-pair mode, which has none
    {
        static const bool s2d_off = getenv("CTG_NO_S2D") != nullptr;      // A/B switch
        const int cslice = pair ? 32 : 64;
        if (!s2d_off && dtype == DT_BF16 && is == 2 && os == 1 && !frame && !fused && pad_mode == PAD_ZERO && oy0 == 0 && ox0 == 0 &&
            Ho == Hs && Wo == Ws && Hs >= 16 && Ws >= 16 && Cin % cslice == 0 && Cout > 32 && (pair ? !out_f32 : (ntaps == 16 && !out_f32)) &&
            (long)Hi * Wi * x_ld < (1L << 31)) {
            ConvArgs b = a;
            int cnt[4] = {0, 0, 0, 0}, amin = 127, amax = -128, bmin = 127, bmax = -128;
            int ph_of[64], ay[64], ax[64];
            for (int t = 0; t < ntaps; ++t) {
                const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
                const int fy = dy >= 0 ? dy / 2 : -((1 - dy) / 2), fx = dx >= 0 ? dx / 2 : -((1 - dx) / 2);      // floor(d / 2)
                ay[t] = fy; ax[t] = fx;
                ph_of[t] = (dy - 2 * fy) * 2 + (dx - 2 * fx);
                ++cnt[ph_of[t]];
                amin = fy < amin ? fy : amin; amax = fy > amax ? fy : amax;
                bmin = fx < bmin ? fx : bmin; bmax = fx > bmax ? fx : bmax;
            }
            int nph = 0, t0 = 0;
            for (int q = 0; q < 4; ++q) {
                if (cnt[q] == 0) continue;
                b.c_ntaps[nph] = cnt[q]; b.c_tap0[nph] = t0; b.c_oy0[nph] = q >> 1; b.c_ox0[nph] = q & 1;
                for (int t = 0; t < ntaps; ++t)
                    if (ph_of[t] == q) b.taps[t0++] = (ay[t] + 64) | ((ax[t] + 64) << 8) | (a.taps[t] & 0xffff0000);
                ++nph;
            }
            b.ncls = nph; b.s2d = 1;
            b.kh = amax - amin + 1; b.kw = bmax - bmin + 1; b.dy0 = amin; b.dx0 = bmin;
            int ntile = 0;
            const bool want_stats = stats_part != nullptr && stats_slabs_out != nullptr && bias == nullptr && act == ACT_NONE;
            b.stats = want_stats ? stats_part : nullptr;
            const int rc = launch_halo_t<bf16_t, 8>(b, omode, st, &ntile);
            if (rc != -1) {
                if (want_stats && rc == 0) *stats_slabs_out = ntile;
                return rc;
            }
        }
    }
    // ---- stride-1 convs whose taps form a full kh x kw window: halo-resident kernel (conv_halo.h)
    {
        int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
        for (int t = 0; t < ntaps; ++t) {
            const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
            dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax;
            dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
        }
        a.kh = dymax - dymin + 1; a.kw = dxmax - dxmin + 1; a.dy0 = dymin; a.dx0 = dxmin;
        static const bool halo_off = getenv("CTG_NO_HALO") != nullptr;
        // full window; a single tap only as one parity class of a transposed conv (os == 2), where the other
        // classes run here too
        const bool window = ntaps == a.kh * a.kw && (ntaps > 1 || os == 2);
        if (!halo_off && !frame && window && is == 1 && (os == 1 || os == 2) && Hs >= 16 && Ws >= 16 &&
            (long)Hi * Wi * x_ld < (1L << 31)) {
            // fused InstanceNorm moments: only meaningful without bias/activation and for one N-partition layout
            int ntile = 0;
            const bool want_stats = stats_part != nullptr && stats_slabs_out != nullptr && bias == nullptr &&
                                    act == ACT_NONE && Cout > 16 &&
                                    ((os == 1 && Hs == Ho && Ws == Wo) || os == 2);   // os == 2: one parity class of the output;
                                    // the caller concatenates the partials of its classes
            a.stats = want_stats ? stats_part : nullptr;
            int rc = -1;
            if (dtype == DT_BF16) rc = k8 ? launch_halo_t<bf16_t, 8>(a, omode, st, &ntile) : launch_halo_t<bf16_t, 4>(a, omode, st, &ntile);
            else rc = k8 ? launch_halo_t<float, 8>(a, omode, st, &ntile) : launch_halo_t<float, 4>(a, omode, st, &ntile);
            if (rc != -1) {
                if ((want_stats || a.bstats != nullptr) && rc == 0 && stats_slabs_out != nullptr) *stats_slabs_out = ntile;
                return rc;
            }
            a.stats = nullptr;
        }
    }
    if (fused) return CTG_EINVAL;   // only the halo kernel's epilogue implements res / fold
    // ---- the 64 -> 128 and 128 -> 256 channel stride-2 3x3 convs on large maps: sliding-window kernels (conv_strips2.h, conv_strips2w.h)
    if (dtype == DT_BF16 && !out_f32 && !pair) {
        const bool ws = stats_part != nullptr && stats_slabs_out != nullptr;      // (its shape check covers bias / act / offsets)
        int slabs = 0;
        int rc = launch_strips2(a, ws ? stats_part : nullptr, st, &slabs);
        if (rc == -1) rc = launch_strips2w(a, ws ? stats_part : nullptr, st, &slabs);
        if (rc != -1) {
            if (rc == 0 && ws) *stats_slabs_out = slabs;
            return rc;
        }
    }
    // gather kernel: moments per M tile (whole output in this launch, no bias / activation, dense bf16 / fp32 store)
    int mtiles = 0;
    if (stats_part != nullptr && stats_slabs_out != nullptr && bias == nullptr && act == ACT_NONE &&
        Cout > 16 && os == 1 && !frame && Hs == Ho && Ws == Wo && oy0 == 0 && ox0 == 0) {
        const int bm = pair ? 128
                     : (Cout > 64 && dtype == DT_BF16 && !out_f32 && k8 && (long)Hs * Ws >= 4096 && CTG_BIG_TILE &&
                        getenv("CTG_NO_BIG_TILE") == nullptr) ? 256 : 128;   // mirrors the tile launch_t picks
        mtiles = (Hs * Ws + bm - 1) / bm;
        const long bound = (long)((Hs + 7) / 8) * ((Ws + 15) / 16);      // what the caller sized the buffer for
        if (mtiles <= bound) a.stats = stats_part;
        else mtiles = 0;
    }
    int rc;
    if (dtype == DT_BF16) rc = k8 ? launch_t<bf16_t, 8>(a, omode, st) : launch_t<bf16_t, 4>(a, omode, st);
    else rc = k8 ? launch_t<float, 8>(a, omode, st) : launch_t<float, 4>(a, omode, st);
    if (rc == 0 && a.stats != nullptr) *stats_slabs_out = mtiles;
    return rc;
}


// ---------------------------------------------------------------------------
// The four parity classes of a stride-2 transposed conv (nn.ConvTranspose2d(k=3, s=2, p=1, output_padding=1),
// Model/HdGan.py:93-95) or of the backward-data pass of a stride-2 conv (Model/HdGan.py:78-80, 124-131) in ONE launch of the
// halo-resident kernel: out[2j + oy0_q, 2i + ox0_q] = sum over class q's taps.  The four workgroups of a spatial tile run back
// to back on one XCD and share the input halo through L2 (a launch per class fetches the input from HBM four times).
// bf16 in / bf16 out.  Returns 0, CTG_EINVAL, 1000+hipError_t, or 2 = "shape not served here": the caller then launches the
// classes one by one through ctg_conv_igemm (same results up to the summation order of the InstanceNorm partials).
// stats_part (optional): B * 4 * ceil(Hs/8) * ceil(Ws/16) * Cout * 2 floats, *stats_slabs_out = partials per sample.
// ---------------------------------------------------------------------------
extern "C" int ctg_conv_igemm_classes(int dtype, const void* x, const void* w, void* y, const float* bias, int B, int Hi,
                                      int Wi, int Cin, int x_ld, int Ho, int Wo, int Cout, int y_ld, int Hs, int Ws,
                                      int pad_mode, int act, int w_npad, const int* cls_ntaps, const int* cls_oy0,
                                      const int* cls_ox0, const int* taps_host, float* stats_part, int* stats_slabs_out,
                                      void* stream) {
    CTG_ENTER();
    if (stats_slabs_out != nullptr) *stats_slabs_out = 0;
    if (dtype != DT_BF16 && dtype != DT_PAIR) return 2;
    const bool pair = dtype == DT_PAIR;      // split-pair in and out, w split by ctg_split_weights (as for ctg_conv_igemm)
    if (B < 1 || Hs < 1 || Ws < 1 || Cout < 1 || cls_ntaps == nullptr || cls_oy0 == nullptr || cls_ox0 == nullptr) return CTG_EINVAL;
    if (Cin % 32 != 0 || x_ld % 8 != 0 || x_ld < Cin || y_ld < Cout || Cout % 8 != 0 || y_ld % 8 != 0) return CTG_EINVAL;
    if (pair && (x_ld % 16 || x_ld < 2 * Cin || y_ld % 16 || y_ld < 2 * Cout)) return CTG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return CTG_EINVAL;
    const int bn = Cout > 64 ? 128 : Cout > 32 ? 64 : Cout > 16 ? 32 : 16;
    if (w_npad < ((Cout + bn - 1) / bn) * bn) return CTG_EINVAL;
    static const bool off = getenv("CTG_NO_HALO") != nullptr || getenv("CTG_NO_CLASS_MERGE") != nullptr;
    if (off || Hs < 16 || Ws < 16 || Cout <= 16 || (long)Hi * Wi * x_ld >= (1L << 31)) return 2;
    ConvArgs a;
    a.stats = nullptr;
    a.nie_sync = nullptr; a.nie_act = ACT_NONE; a.nie_budget = 0;
    a.res = nullptr; a.fold = nullptr; a.res_ld = 0; a.fold_ld = 0;
    a.bz = nullptr; a.bmean = nullptr; a.brstd = nullptr; a.bstats = nullptr; a.bz_ld = 0; a.bact = ACT_NONE;
    a.x = x; a.w = w; a.y = y; a.bias = bias;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.x_ld = x_ld;
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.y_ld = y_ld;
    a.Hs = Hs; a.Ws = Ws; a.oy0 = 0; a.ox0 = 0; a.os = 2; a.is = 1; a.frame = 0;
    a.pad_mode = pad_mode; a.act = act; a.w_tap_stride = w_npad * Cin;
    a.ncls = 4;
    a.s2d = 0;
    a.pair_lo = 0;
    if (pair) {
        a.pair_lo = x_ld / 2;
        a.Cin = 2 * Cin;
        a.w_tap_stride = w_npad * 2 * Cin;
    }
    int t0 = 0, kh_max = 1, kw_max = 1;
    for (int q = 0; q < 4; ++q) {
        const int nt = cls_ntaps[q];
        if (nt < 1 || t0 + nt > 64) return nt < 1 ? 2 : CTG_EINVAL;
        if ((Hs - 1) * 2 + cls_oy0[q] >= Ho || (Ws - 1) * 2 + cls_ox0[q] >= Wo || cls_oy0[q] < 0 || cls_ox0[q] < 0) return CTG_EINVAL;
        int dymin = 127, dymax = -128, dxmin = 127, dxmax = -128;
        for (int t = 0; t < nt; ++t) {
            const int tw = taps_host[t0 + t];
            const int dy = (tw & 0xff) - 64, dx = ((tw >> 8) & 0xff) - 64;
            if (pad_mode == PAD_REFLECT) {
                const int ymax = (Hs - 1) + dy, xmax = (Ws - 1) + dx;
                if (-dy >= Hi || ymax - (Hi - 1) >= Hi || -dx >= Wi || xmax - (Wi - 1) >= Wi) return CTG_EINVAL;
            }
            dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax;
            dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
            a.taps[t0 + t] = tw;
        }
        const int kh = dymax - dymin + 1, kw = dxmax - dxmin + 1;
        if (nt != kh * kw) return 2;                 // not a full window: the gather kernel's business
        a.c_ntaps[q] = nt; a.c_tap0[q] = t0; a.c_oy0[q] = cls_oy0[q]; a.c_ox0[q] = cls_ox0[q];
        a.c_kh[q] = kh; a.c_kw[q] = kw; a.c_dy0[q] = dymin; a.c_dx0[q] = dxmin;
        kh_max = kh > kh_max ? kh : kh_max;
        kw_max = kw > kw_max ? kw : kw_max;
        t0 += nt;
    }
    a.ntaps = a.c_ntaps[0];
    a.kh = kh_max; a.kw = kw_max; a.dy0 = a.c_dy0[0]; a.dx0 = a.c_dx0[0];   // kh / kw size the LDS for the largest class halo
    const bool want_stats = stats_part != nullptr && stats_slabs_out != nullptr && bias == nullptr && act == ACT_NONE;
    a.stats = want_stats ? stats_part : nullptr;
    int ntile = 0;
    const bool k8 = (Cin % 64) == 0;
    int rc = pair ? pairstrip_launch_t(a, (hipStream_t)stream, &ntile) : launch_stript(a, (hipStream_t)stream, &ntile);
    if (rc == -1 && pair) rc = launch_halo_t<bf16_t, 8>(a, 2, (hipStream_t)stream, &ntile);
    if (rc == -1 && !pair)
        rc = k8 ? launch_halo_t<bf16_t, 8>(a, 0, (hipStream_t)stream, &ntile) : launch_halo_t<bf16_t, 4>(a, 0, (hipStream_t)stream, &ntile);
    if (rc == -1) return 2;
    if (rc == 0 && want_stats) *stats_slabs_out = ntile;
    return rc;
}
