// First-layer convolutions (Cin in {1, 2}) with the im2col matrix built in LDS (gfx950).
//
// Replaces, for nn.Conv2d(1|2, C, k) first layers -- Generator head 7x7 reflect (Model/HdGan.py:70,
// Model/CycleGan.py:28), Discriminator 4x4 stride 2 (HdGan.py:120,158), Reg's first 3x3 (trainer/reg.py:77 via
// trainer/layers.py:71-104) -- the pair "im2col_pack to HBM + 1x1 implicit GEMM": a 512x512 one-channel image is
// 1 MB, its 7x7 im2col matrix 32 MB, so the packed detour moved 64x the bytes the layer needs.  Here a workgroup
// owns a 16x16 tile of output pixels:
//   1. the (15*stride + kh) x (15*stride + kw) fp32 input patch of every plane goes to LDS (reflection / zero
//      padding resolved once per patch element);
//   2. every thread assembles 16-byte K-chunks of the [256 px][Kpad] im2col tile from it (its chunk column, hence
//      its tap offsets, is fixed -> registers) into the XOR-swizzled layout of conv_halo.h;
//   3. weights = A operand, pixels = B operand of v_mfma_f32_16x16x32_bf16 (fp32: 16x16x4), K = Kpad in one go;
//   4. epilogue as in conv_halo.h: optional InstanceNorm moments, bias + activation, bf16 staged through LDS.
// Workgroups are persistent (a run of tiles each): the weights stay in LDS and the next tile's patch is fetched
// into registers while the current tile is assembled, multiplied and stored.
// The same kernel serves the backward-data pass of a 1-channel OUTPUT layer (Generator tail, HdGan.py:110): its
// input gradient is "a 7x7 conv of the 1-channel output gradient with the flipped weights" on the padded grid.
#include <stdlib.h>
#include "common.h"

struct SmallArgs {
    const float* s0;
    const float* s1;
    const void* w;        // [Npad][Kpad] packed weights, k = (c*kh + ky)*kw + kx
    const float* bias;
    void* y;
    float* stats;         // [B][tiles][Cout][2] or null
    int B, Hi, Wi, Cin, kh, kw, stride, pad, pad_mode;
    int Ho, Wo, Cout, y_ld, act, Kreal, ntiles;
};

#define SMALL_PREF 5   // patch elements per thread: up to 5*256 = 1280 (4x4 stride 2: 34x34 = 1156)

// PO (T = float): the result leaves as a split pair ([hi | lo] bf16 planes, y_ld = its pitch in bf16 elements) -- the first
// layers of the "bf16x3" mode: exact-f32 MFMA on the fp32 image planes, fp32 accumulators staged through LDS and split there.
// X3 (T = bf16, PO): the im2col tile is built as bf16 hi and lo halves, a sub-tile row = [hi 32 k | lo 32 k], the weights are the
// ctg_split_weights operand [w_hi 32 | w_lo 32], and a sub-tile contracts hi.w_hi + hi.w_lo + lo.w_hi on the bf16 matrix cores.
// KXW (one fp32 plane, stride 1, kh, kw <= 8, KPAD = 64): the K axis is laid out k = 8 ky + kx ("kx window": chunk ky of an
// im2col row is 8 consecutive pixels of patch row y + ky), so a thread that owns kernel row ky and 8 consecutive pixels of a
// tile row reads kw + 7 patch values ONCE and forms its 8 chunks from a register window -- 14 LDS reads and no per-element
// select for 8 chunks instead of 64 reads: the assembly of the generic path (k = ky kw + kx: every chunk straddles kernel
// rows) issued 16.6 k VALU instructions per wave around 420 MFMAs (SQ counters, round 4).  Weights packed accordingly.
template <typename T, int KPAD, int BN, bool PO = false, bool X3 = false, bool KXW = false>
__global__ __launch_bounds__(256) void conv_small_kernel(const SmallArgs a) {
    static_assert(!KXW || (KPAD == 64 && sizeof(T) == 2), "kx-window layout: 8 kernel rows x 8 columns of bf16");
    static_assert(!PO || sizeof(T) == 4 || X3, "split-pair output: the fp32 or the split-bf16 instantiation");
    static_assert(!X3 || (sizeof(T) == 2 && PO), "split-bf16 MFMA: bf16 tiles, split-pair result");
    constexpr int EPC = VecOf<T>::N;
    constexpr int CPR = KPAD / EPC;                 // 16-byte chunks per im2col row (X3: k-chunks a row is assembled from)
    constexpr int KCH = X3 ? 8 : (CPR < 8 ? CPR : 8);   // chunks per row of one swizzled sub-tile (X3: 4 hi + 4 lo)
    constexpr int NSUB = X3 ? KPAD / 32 : CPR / KCH;
    constexpr int CPRW = X3 ? 2 * CPR : CPR;        // chunks per weight row
    constexpr int TM = 4, TN = BN / 16;
    constexpr int PPI = 256 / CPR;                  // pixels covered per build iteration
    constexpr int A_BYTES = 256 * KPAD * (int)sizeof(T) * (X3 ? 2 : 1);
    constexpr int ST_BYTES = PO ? 256 * (BN * 4 + 16) : sizeof(T) == 2 ? 256 * (BN * 2 + 16) : 4 * BN * 2 * 4;
    constexpr int A_REGION = A_BYTES > ST_BYTES ? A_BYTES : ST_BYTES;   // im2col tile, later the epilogue staging
    typedef T OutT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int tx_n = (a.Wo + 15) >> 4;
    const int PH = 15 * a.stride + a.kh, PW = 15 * a.stride + a.kw;
    const int plane = PH * PW;
    const int patch_elems = a.Cin * plane;

    char* sA = smem;                                       // NSUB x [256][KCH] chunks
    char* sW = smem + A_REGION;                            // NSUB x [BN][KCH] chunks, resident for all tiles
    float* patch = reinterpret_cast<float*>(sW + BN * CPRW * 16);

    // ---- persistent workgroup: a contiguous run of this sample's tiles (XCD-contiguous workgroup order)
    const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const int q = a.ntiles / (int)gridDim.x, rem = a.ntiles % (int)gridDim.x;
    const int t_begin = wg * q + (wg < rem ? wg : rem);
    const int t_end = t_begin + q + (wg < rem ? 1 : 0);
    if (t_begin >= t_end) return;

    // ---- weights -> LDS once
    {
        const T* W = (const T*)a.w;
        for (int i = tid; i < BN * CPRW; i += 256) {
            const int row = i / CPRW, kc = i - row * CPRW;
            const int sub = kc / KCH, kcl = kc - sub * KCH;
            *reinterpret_cast<u32x4*>(sW + ((sub * BN + row) * KCH + swz<KCH>(row, kcl)) * 16) =
                *reinterpret_cast<const u32x4*>(W + (size_t)row * (CPRW * EPC) + kc * EPC);
        }
    }
    // ---- this thread's patch elements (tile independent): plane and (row, col) inside the patch
    int pe_yx[SMALL_PREF];
#pragma unroll
    for (int j = 0; j < SMALL_PREF; ++j) {
        const int i = tid + 256 * j;
        const int c = i / plane, r = i - c * plane;
        const int py = r / PW, px = r - py * PW;
        pe_yx[j] = i < patch_elems ? (c << 30) | (py << 15) | px : -1;
    }
    float pre[SMALL_PREF];
    unsigned pre_ok = 0;
    auto fetch_patch = [&](int sp) __attribute__((always_inline)) {
        const int iy0 = (sp / tx_n) * 16 * a.stride - a.pad, ix0 = (sp % tx_n) * 16 * a.stride - a.pad;
#pragma unroll
        for (int j = 0; j < SMALL_PREF; ++j) {
            const int e = pe_yx[j];
            int iy = iy0 + ((e >> 15) & 0x7fff), ix = ix0 + (e & 0x7fff);
            if (a.pad_mode == PAD_REFLECT) { iy = reflect_idx(iy, a.Hi); ix = reflect_idx(ix, a.Wi); }
            // (a tile hanging over the grid may still fall outside after one reflection: it only feeds masked outputs)
            const bool ok = e >= 0 && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            // branch-free: out-of-image elements read pixel (0, 0) of plane 0 and are zeroed when they reach LDS
            const float* src = (ok && ((e >> 30) & 1)) ? a.s1 : a.s0;
            const int off = ok ? iy * a.Wi + ix : 0;
            pre[j] = src[(size_t)n * a.Hi * a.Wi + off];
            pre_ok = ok ? (pre_ok | (1u << j)) : (pre_ok & ~(1u << j));
        }
    };
    // ---- this thread's chunk column: tap offsets into the patch (tile independent)
    const int kc = tid % CPR;
    int koff[EPC];
    unsigned kvalid = 0;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int k = kc * EPC + e;
        const int kk = a.kh * a.kw;
        const int c = k / kk, r = k - c * kk;
        const int ky = r / a.kw, kx = r - ky * a.kw;
        koff[e] = k < a.Kreal ? (c * PH + ky) * PW + kx : 0;
        kvalid |= (k < a.Kreal ? 1u : 0u) << e;
    }
    const int co_l = (lane >> 4) * 4;
    OutT* __restrict__ Y = (OutT*)a.y;
    // none / ReLU / LeakyReLU as one negative-side slope (1, 0, 0.2): no per-element branch
    const float nslope = a.act == ACT_RELU ? 0.f : a.act == ACT_LRELU ? LRELU_SLOPE : 1.f;
    const bool raw = a.bias == nullptr && a.act == ACT_NONE;

    // bias of this lane's channels, read ONCE: a global load inside the tile loop makes the compiler drain vmcnt there,
    // i.e. wait for the previous tile's output stores
    float bv[TN][4];
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = nt * 16 + co_l + r;
            bv[nt][r] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
        }

    auto stash_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SMALL_PREF; ++j)
            if (pe_yx[j] >= 0) patch[tid + 256 * j] = ((pre_ok >> j) & 1u) ? pre[j] : 0.f;
    };
    fetch_patch(t_begin);
    stash_patch();
    for (int sp = t_begin; sp < t_end; ++sp) {
        const int y0 = (sp / tx_n) * 16, x0 = (sp % tx_n) * 16;
        // ---- 1. this tile's patch is in LDS (stashed during the previous tile); start fetching the next one.
        // The barrier also says: every thread is past its staging reads of the previous tile.
        lds_barrier();
        if (sp + 1 < t_end) fetch_patch(sp + 1);
        // ---- 2. im2col tile
        if constexpr (KXW) {
            const int ky = tid & 7, grp = tid >> 3;                 // kernel row = chunk column; 32 groups of 8 pixels
            const int r = grp >> 1, xh = (grp & 1) * 8;             // tile row, first column of the group
            const bool row_ok = ky < a.kh;
            const float* prow = patch + (r + ky) * PW + xh;
            float win[16];
#pragma unroll
            for (int j = 0; j < 15; ++j) win[j] = (row_ok && j < a.kw + 7) ? prow[j] : 0.f;
            win[15] = 0.f;
            const int sub = X3 ? ky >> 2 : 0, kcl = X3 ? ky & 3 : ky;
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const int p = r * 16 + xh + pp;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = e < a.kw ? win[pp + e] : 0.f;
                if constexpr (X3) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        hi[e] = (bf16_t)v[e];
                        lo[e] = (bf16_t)(v[e] - (float)hi[e]);
                    }
                    *reinterpret_cast<bf16x8*>(sA + ((sub * 256 + p) * KCH + swz<KCH>(p, kcl)) * 16) = hi;
                    *reinterpret_cast<bf16x8*>(sA + ((sub * 256 + p) * KCH + swz<KCH>(p, kcl + 4)) * 16) = lo;
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                    *reinterpret_cast<bf16x8*>(sA + ((sub * 256 + p) * KCH + swz<KCH>(p, kcl)) * 16) = o;
                }
            }
        } else {
            const int sub = X3 ? kc / 4 : kc / KCH, kcl = X3 ? kc & 3 : kc - sub * KCH;
#pragma unroll 2
            for (int it = 0; it < CPR; ++it) {
                const int p = tid / CPR + it * PPI;
                const int base = ((p >> 4) * a.stride) * PW + (p & 15) * a.stride;
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float v = patch[base + koff[e]];
                    o.v[e] = ((kvalid >> e) & 1u) ? v : 0.f;
                }
                if constexpr (X3) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        hi[e] = (bf16_t)o.v[e];
                        lo[e] = (bf16_t)(o.v[e] - (float)hi[e]);
                    }
                    *reinterpret_cast<bf16x8*>(sA + ((sub * 256 + p) * KCH + swz<KCH>(p, kcl)) * 16) = hi;
                    *reinterpret_cast<bf16x8*>(sA + ((sub * 256 + p) * KCH + swz<KCH>(p, kcl + 4)) * 16) = lo;
                } else {
                    o.store(reinterpret_cast<T*>(sA + ((sub * 256 + p) * KCH + swz<KCH>(p, kcl)) * 16));
                }
            }
        }
        lds_barrier();

        // ---- 3. MFMA: wave = 4 tile rows (64 pixels) x BN channels, K = KPAD
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            if constexpr (X3) {
                const int c4 = lane >> 4;
                u32x4 fa[TM], fb[TN], fl[TN];
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const int row = (wave * TM + mt) * 16 + (lane & 15);
                    fa[mt] = *reinterpret_cast<const u32x4*>(sA + ((sub * 256 + row) * KCH + swz<KCH>(row, c4)) * 16);
                }
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int row = nt * 16 + (lane & 15);
                    fb[nt] = *reinterpret_cast<const u32x4*>(sW + ((sub * BN + row) * KCH + swz<KCH>(row, c4)) * 16);
                    fl[nt] = *reinterpret_cast<const u32x4*>(sW + ((sub * BN + row) * KCH + swz<KCH>(row, c4 + 4)) * 16);
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
                    const int row = (wave * TM + mt) * 16 + (lane & 15);
                    fa[mt] = *reinterpret_cast<const u32x4*>(sA + ((sub * 256 + row) * KCH + swz<KCH>(row, c4 + 4)) * 16);
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
                const int c4 = ks * 4 + (lane >> 4);
                u32x4 fa[TM], fb[TN];
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const int row = (wave * TM + mt) * 16 + (lane & 15);
                    fa[mt] = *reinterpret_cast<const u32x4*>(sA + ((sub * 256 + row) * KCH + swz<KCH>(row, c4)) * 16);
                }
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int row = nt * 16 + (lane & 15);
                    fb[nt] = *reinterpret_cast<const u32x4*>(sW + ((sub * BN + row) * KCH + swz<KCH>(row, c4)) * 16);
                }
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
                            for (int qq = 0; qq < 4; ++qq)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[qq], va[qq], acc[mt][nt], 0, 0, 0);
                        }
                    }
            }
        }
        // the patch is dead since the barrier after step 2: park the next tile's patch there now.  Its loads have had
        // the im2col + MFMA phases to land, and the only older memory operations are the previous tile's stores --
        // waiting here never waits for THIS tile's stores (a wait at the loop top would).
        if (sp + 1 < t_end) stash_patch();
        lds_barrier();   // every wave is done with sA: the epilogue reuses the memory

        // ---- 4. epilogue.  acc[mt][nt][r]: pixel (row wave*4+mt, col lane&15), co = nt*16 + (lane>>4)*4 + r
        if (a.stats != nullptr) {
            float* red = reinterpret_cast<float*>(smem);   // [4 waves][BN][2]
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const bool valid = (y0 + wave * TM + mt < a.Ho) && (x0 + (lane & 15) < a.Wo);
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
                        const int cl = nt * 16 + co_l + r;
                        red[(wave * BN + cl) * 2] = s1[r];
                        red[(wave * BN + cl) * 2 + 1] = s2[r];
                    }
                }
            }
            lds_barrier();
            if (tid < BN && tid < a.Cout) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
                float* dst = a.stats + (((size_t)n * a.ntiles + sp) * a.Cout + tid) * 2;
                dst[0] = t1;
                dst[1] = t2;
            }
            lds_barrier();
        }
        if constexpr (sizeof(OutT) == 2 && !PO) {
            constexpr int RS = BN * 2 + 16;
            char* st = smem;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int co = nt * 16 + co_l;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const int prow = (wave * TM + mt) * 16 + (lane & 15);
                    bf16x4 o;
                    if (raw) {      // (wave-uniform) a conv that feeds an InstanceNorm: no bias, no activation -- 3 VALU per value less
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[mt][nt][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[mt][nt][r] + bv[nt][r];
                            o[r] = (bf16_t)(v > 0.f ? v : v * nslope);
                        }
                    }
                    *reinterpret_cast<bf16x4*>(st + prow * RS + co * 2) = o;
                }
            }
            lds_barrier();
            constexpr int CPO = BN / 8;
#pragma unroll
            for (int it = 0; it < CPO; ++it) {
                const int cidx = tid + 256 * it;
                const int prow = cidx / CPO, ch = (cidx % CPO) * 8;
                const int oy = y0 + (prow >> 4), ox = x0 + (prow & 15);
                if (oy < a.Ho && ox < a.Wo && ch < a.Cout)
                    *reinterpret_cast<u32x4*>(Y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ld + ch) =
                        *reinterpret_cast<const u32x4*>(st + prow * RS + ch * 2);
            }
        } else if constexpr (PO) {
            constexpr int RS = BN * 4 + 16;
            char* st = smem;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int co = nt * 16 + co_l;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const int prow = (wave * TM + mt) * 16 + (lane & 15);
                    f32x4 o;
                    if (raw) {
                        o = acc[mt][nt];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[mt][nt][r] + bv[nt][r];
                            o[r] = v > 0.f ? v : v * nslope;
                        }
                    }
                    *reinterpret_cast<f32x4*>(st + prow * RS + co * 4) = o;
                }
            }
            lds_barrier();
            bf16_t* __restrict__ Yp = (bf16_t*)a.y;
            const int y_lo = a.y_ld >> 1;
            constexpr int CPO = BN / 8;
#pragma unroll
            for (int it = 0; it < CPO; ++it) {
                const int cidx = tid + 256 * it;
                const int prow = cidx / CPO, ch = (cidx % CPO) * 8;
                const int oy = y0 + (prow >> 4), ox = x0 + (prow & 15);
                if (oy < a.Ho && ox < a.Wo && ch < a.Cout) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + prow * RS + ch * 4);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + prow * RS + ch * 4 + 16);
                    bf16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (bf16_t)v0[e];
                        lo[e] = (bf16_t)(v0[e] - (float)hi[e]);
                        hi[4 + e] = (bf16_t)v1[e];
                        lo[4 + e] = (bf16_t)(v1[e] - (float)hi[4 + e]);
                    }
                    bf16_t* yp = Yp + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ld + ch;
                    *reinterpret_cast<bf16x8*>(yp) = hi;
                    *reinterpret_cast<bf16x8*>(yp + y_lo) = lo;
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int oy = y0 + wave * TM + mt, ox = x0 + (lane & 15);
                if (oy >= a.Ho || ox >= a.Wo) continue;
                OutT* yp = Y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ld;
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int co = nt * 16 + co_l;
                    if (co >= a.Cout) continue;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = acc[mt][nt][r] + bv[nt][r];
                        v[r] = t > 0.f ? t : t * nslope;
                    }
                    *reinterpret_cast<f32x4*>(yp + co) = v;
                }
            }
        }
    }
}

template <typename T, int KPAD, int BN, bool PO = false, bool X3 = false, bool KXW = false>
static int launch_small(SmallArgs& a, hipStream_t st, int* tiles_out) {
    const int PH = 15 * a.stride + a.kh, PW = 15 * a.stride + a.kw;
    if (a.Cin * PH * PW > SMALL_PREF * 256 || PH >= 32768 || PW >= 32768) return CTG_EINVAL;
    const int a_bytes = 256 * KPAD * (int)sizeof(T) * (X3 ? 2 : 1);
    const int st_bytes = PO ? 256 * (BN * 4 + 16) : sizeof(T) == 2 ? 256 * (BN * 2 + 16) : 4 * BN * 2 * 4;
    const int smem = (a_bytes > st_bytes ? a_bytes : st_bytes) + BN * KPAD * (int)sizeof(T) * (X3 ? 2 : 1) + (a.Cin * PH * PW + 4) * 4;
    if (smem > 160 * 1024) return CTG_EINVAL;
    static unsigned long long attr_mask = 0;       // per device
    if (smem > 64 * 1024) {
        const int rc = ctg_lds_attr_once((const void*)conv_small_kernel<T, KPAD, BN, PO, X3, KXW>, 160 * 1024, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const int tiles = ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);
    a.ntiles = tiles;
    if (tiles_out != nullptr) *tiles_out = tiles;
    // persistent workgroups: about three per CU over the whole batch, each sweeping a run of tiles of one sample
    static const int wg_total = getenv("CTG_SMALL_WGS") ? atoi(getenv("CTG_SMALL_WGS")) : 768;
    int gx = (wg_total + a.B - 1) / a.B;
    if (gx > tiles) gx = tiles;
    hipLaunchKernelGGL((conv_small_kernel<T, KPAD, BN, PO, X3, KXW>), dim3(gx, a.B), dim3(256), smem, st, a);
    return ctg_launch_status();
}

// C ABI.  y[B][Ho][Wo][y_ld] (dtype) = act(conv(planes) + bias) for Cin in {1, 2} fp32 image planes; `w` is the
// [w_npad][Kpad] matrix ctg_weight_pack makes of weight.view(Cout, Cin*kh*kw).  Cout <= 64 and a multiple of 8 (4 in
// fp32), Kpad in {32, 64}.  stats_part as in ctg_conv_igemm (one slab per 16x16 tile).
extern "C" int ctg_conv_smallcin(int dtype, const float* s0, const float* s1, int Cin, int B, int Hi, int Wi, int kh,
                                 int kw, int stride, int pad, int pad_mode, const void* w, int w_npad, int Kpad, int w_layout,
                                 const float* bias, int act, void* y, int y_ld, int Ho, int Wo, int Cout,
                                 float* stats_part, int* stats_slabs_out, void* stream) {
    CTG_ENTER();
    // w_layout 1 ("kx window"; one plane, stride 1, kh, kw <= 8, Kpad 64, Cout > 32, bf16 / split pair): w[n][8 ky + kx]
    if (w_layout != 0 && (w_layout != 1 || Cin != 1 || stride != 1 || kh > 8 || kw > 8 || Kpad != 64 || Cout <= 32 ||
                          dtype == DT_F32)) return CTG_EINVAL;
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_PAIR) return CTG_EINVAL;
    // DT_PAIR ("bf16x3"): w = the pack split by ctg_split_weights ([w_npad][2 Kpad] bf16), split-bf16 MFMA on an im2col tile
    // built as bf16 hi / lo halves, y a split-pair tensor (y_ld its pitch in bf16 elements)
    const bool pair = dtype == DT_PAIR;
    if (pair && (Cout % 8 || y_ld % 16 || y_ld < 2 * Cout)) return CTG_EINVAL;
    const int epc = dtype == DT_BF16 ? 8 : 4;
    if (Cin < 1 || Cin > 2 || (Cin == 2 && s1 == nullptr) || s0 == nullptr || w == nullptr || y == nullptr) return CTG_EINVAL;
    if (B < 1 || kh < 1 || kw < 1 || stride < 1 || stride > 2 || pad < 0) return CTG_EINVAL;
    if (act != ACT_NONE && act != ACT_RELU && act != ACT_LRELU) return CTG_EINVAL;
    if ((w_layout == 0 && Cin * kh * kw > Kpad) || (Kpad != 32 && Kpad != 64)) return CTG_EINVAL;
    if (Cout < 1 || Cout > 64 || Cout % epc || y_ld % epc || y_ld < Cout) return CTG_EINVAL;
    if (pad_mode == PAD_REFLECT && (pad >= Hi || pad >= Wi)) return CTG_EINVAL;
    if (Ho != (Hi + 2 * pad - kh) / stride + 1 || Wo != (Wi + 2 * pad - kw) / stride + 1 || Ho < 1 || Wo < 1) return CTG_EINVAL;
    const int bn = Cout > 32 ? 64 : 32;
    if (w_npad < bn) return CTG_EINVAL;
    if (((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return CTG_EINVAL;
    SmallArgs a;
    a.s0 = s0; a.s1 = s1; a.w = w; a.bias = bias; a.y = y; a.stats = stats_part;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
    a.pad_mode = pad_mode; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.y_ld = y_ld; a.act = act; a.Kreal = Cin * kh * kw;
    hipStream_t st = (hipStream_t)stream;
    int tiles = 0, rc;
    if (w_layout == 1) {
        rc = pair ? launch_small<bf16_t, 64, 64, true, true, true>(a, st, &tiles) : launch_small<bf16_t, 64, 64, false, false, true>(a, st, &tiles);
    } else if (dtype == DT_BF16) {
        if (Kpad == 64) rc = bn == 64 ? launch_small<bf16_t, 64, 64>(a, st, &tiles) : launch_small<bf16_t, 64, 32>(a, st, &tiles);
        else rc = bn == 64 ? launch_small<bf16_t, 32, 64>(a, st, &tiles) : launch_small<bf16_t, 32, 32>(a, st, &tiles);
    } else if (pair) {
        if (Kpad == 64) rc = bn == 64 ? launch_small<bf16_t, 64, 64, true, true>(a, st, &tiles) : launch_small<bf16_t, 64, 32, true, true>(a, st, &tiles);
        else rc = bn == 64 ? launch_small<bf16_t, 32, 64, true, true>(a, st, &tiles) : launch_small<bf16_t, 32, 32, true, true>(a, st, &tiles);
    } else {
        if (Kpad == 64) rc = bn == 64 ? launch_small<float, 64, 64>(a, st, &tiles) : launch_small<float, 64, 32>(a, st, &tiles);
        else rc = bn == 64 ? launch_small<float, 32, 64>(a, st, &tiles) : launch_small<float, 32, 32>(a, st, &tiles);
    }
    if (stats_slabs_out != nullptr) *stats_slabs_out = stats_part != nullptr ? tiles : 0;
    return rc;
}
