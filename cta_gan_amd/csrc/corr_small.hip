// Weight gradient of the convolutions that touch a 1-/2-channel IMAGE (gfx950, bf16 compute):
//   * first layers  (Generator head 7x7 reflect, Model/HdGan.py:70 == CycleGan.py:28; Reg's first 3x3,
//     trainer/reg.py:77):       dW[co][c,ky,kx] = sum_p dY[p][co] * pad(I_c)[p + (ky,kx)]
//   * the 1-channel tail (HdGan.py:110):  dW[0][ci,ky,kx] = sum_p dY[p] * rpad(X)[p + (ky,kx)][ci]
//     = sum_q rpad(X)[q][ci] * zpad6(dY)[q + (6-ky, 6-kx)]    (q on the padded grid)
// Both are C[m][k] = sum_q G[q][m] * I[q - ipad + tap_k]: a wide NHWC tensor G (dY resp. the reflection-padded
// activation) correlated with the taps of an image I.  The packed detour (im2col matrix of I in HBM, then a GEMM)
// moved 64x the image; here a persistent workgroup sweeps 16x16-pixel tiles of one sample:
//   * the G tile [256 px][BM] arrives by LDS-DMA, double buffered; A fragments are transposing reads of it
//     (ds_read_b64_tr_b16: K = pixels), exactly the idiom of conv_wgrad_halo_kernel;
//   * the image patch is kept as FOUR column-shifted bf16 copies, so the B fragment of tap (ky,kx) -- four
//     consecutive pixels of a tile row, twice -- is two ALIGNED 8-byte reads (copy kx & 3);
//   * accumulators (BM x 64 taps) live in registers across all tiles; one fp32 partial per workgroup, summed by
//     ctg_wgrad_reduce (deterministic).
#include <stdlib.h>
#include "common.h"

struct CorrArgs {
    const void* g;          // bf16 [B][Gh][Gw][g_ld], channels [0, BM)
    const float* i0;        // fp32 [B][Ih][Iw]
    const float* i1;
    float* part;            // [B * gridDim.x][BM][64]
    int B, Gh, Gw, g_ld, gpad, g_pad_mode;
    int Ih, Iw, Cin, kh, kw, ipad, i_pad_mode;
    int Hs, Ws, Kreal, ntiles;
    int g_lo;               // PAIR instantiations: element offset of g's lo plane (its pitch / 2)
};

typedef const __attribute__((address_space(1))) void* cs_gptr_t;
typedef __attribute__((address_space(3))) void* cs_lptr_t;
__device__ __attribute__((aligned(16))) unsigned g_cs_zero_chunk[4];

template <int CPR> __device__ __forceinline__ int cs_swz(int row, int c) {   // = wg_swz of conv_wgrad.hip
    if constexpr (CPR == 8) return c ^ (((row >> 1) & 3) << 1);
    else return c ^ (((row >> 2) & 1) << 1);
}

#define CS_PREF 4     // patch elements per thread: Cin * (16+kh-1) * (16+kw-1) <= 1024
#define CS_RS 24      // elements per row of a shifted copy

// PAIR (split-pair "bf16x3" mode, round 5): g is a split pair and the image is split into bf16 hi / lo copies; every tile is swept
// three times -- (g_hi, I_hi), (g_hi, I_lo), (g_lo, I_hi) -- into the same accumulators, so ONE launch reads each plane of g once
// (rounds 3-4: three launches, g_hi read twice -- 1.6 GB instead of 1.07 GB for the generator head's gradient).  The two planes
// have ONE tile buffer each; g_hi of the next tile is fetched behind the second sweep and lands under the third, g_lo behind the
// third and lands under the next tile's first two.
template <int BM, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void corr_small_kernel(const CorrArgs a) {
    typedef bf16_t T;
    constexpr int CPM = BM / 8;                 // 16-byte chunks per G pixel row
    constexpr int MT = BM / 16;                 // M tiles; a wave owns one M tile and TNW of the 4 tap tiles
    constexpr int TNW = MT;                     // (4 waves / MT waves per M tile) -> 4 / (4 / MT)
    constexpr int G_BYTES = 256 * BM * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int tx_n = (a.Ws + 15) >> 4;
    const int PH = 16 + a.kh - 1, PW = 16 + a.kw - 1;
    const int plane = PH * PW;
    const int patch_elems = a.Cin * plane;
    const int copy_elems = PH * CS_RS;          // one shifted copy of one plane

    char* sG = smem;                                           // two G tiles (PAIR: tile 0 = g_hi, tile 1 = g_lo)
    T* sP = reinterpret_cast<T*>(smem + 2 * G_BYTES);          // [Cin][4 shifts][PH][CS_RS]  (PAIR: a second set, the lo halves, behind it)
    const int copies_bytes = a.Cin * 4 * copy_elems * 2;
    float* patch = reinterpret_cast<float*>(smem + 2 * G_BYTES + (PAIR ? 2 : 1) * copies_bytes);

    const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const int q = a.ntiles / (int)gridDim.x, rem = a.ntiles % (int)gridDim.x;
    const int t_begin = wg * q + (wg < rem ? wg : rem);
    const int t_end = t_begin + q + (wg < rem ? 1 : 0);

    const int mt_w = wave % MT, nt0 = (wave / MT) * TNW;
    f32x4 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (t_begin < t_end) {
        const T* __restrict__ G = (const T*)a.g + (size_t)n * a.Gh * a.Gw * a.g_ld;
        const int Gh = a.Gh, Gw = a.Gw, g_ld = a.g_ld, gpad = a.gpad, g_pad_mode = a.g_pad_mode, Hs = a.Hs, Ws = a.Ws;
        auto issue_g = [&](int sp, int buf) __attribute__((always_inline)) {      // PAIR: buf also selects the plane
            const int y0 = (sp / tx_n) * 16, x0 = (sp % tx_n) * 16;
            const T* __restrict__ Gp = G + ((PAIR && buf) ? a.g_lo : 0);
#pragma unroll
            for (int it = 0; it < CPM; ++it) {
                const int sl = tid + 256 * it;
                const int p = sl / CPM;
                const int kc = cs_swz<CPM>(p, sl % CPM);
                const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
                int gy = oy - gpad, gx = ox - gpad;
                if (g_pad_mode == PAD_REFLECT) { gy = reflect_idx(gy, Gh); gx = reflect_idx(gx, Gw); }
                const bool ok = oy < Hs && ox < Ws && (unsigned)gy < (unsigned)Gh && (unsigned)gx < (unsigned)Gw;
                const T* src = ok ? Gp + ((size_t)(gy * Gw + gx) * g_ld + kc * 8) : (const T*)g_cs_zero_chunk;
                __builtin_amdgcn_global_load_lds((cs_gptr_t)src, (cs_lptr_t)(sG + buf * G_BYTES + (256 * it + 64 * wave) * 16),
                                                 16, 0, 0);
            }
        };
        // ---- this thread's patch elements (tile independent) and their prefetch registers
        int pe_yx[CS_PREF];
#pragma unroll
        for (int j = 0; j < CS_PREF; ++j) {
            const int i = tid + 256 * j;
            const int c = i / plane, r = i - c * plane;
            const int py = r / PW, px = r - py * PW;
            pe_yx[j] = i < patch_elems ? (c << 30) | (py << 15) | px : -1;
        }
        float pre[CS_PREF];
        unsigned pre_ok = 0;
        auto fetch_patch = [&](int sp) __attribute__((always_inline)) {
            const int iy0 = (sp / tx_n) * 16 - a.ipad, ix0 = (sp % tx_n) * 16 - a.ipad;
#pragma unroll
            for (int j = 0; j < CS_PREF; ++j) {
                const int e = pe_yx[j];
                int iy = iy0 + ((e >> 15) & 0x7fff), ix = ix0 + (e & 0x7fff);
                if (a.i_pad_mode == PAD_REFLECT) { iy = reflect_idx(iy, a.Ih); ix = reflect_idx(ix, a.Iw); }
                const bool ok = e >= 0 && (unsigned)iy < (unsigned)a.Ih && (unsigned)ix < (unsigned)a.Iw;
                const float* src = (ok && ((e >> 30) & 1)) ? a.i1 : a.i0;
                const int off = ok ? iy * a.Iw + ix : 0;
                pre[j] = src[(size_t)n * a.Ih * a.Iw + off];
                pre_ok = ok ? (pre_ok | (1u << j)) : (pre_ok & ~(1u << j));
            }
        };
        auto stash_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < CS_PREF; ++j)
                if (pe_yx[j] >= 0) patch[tid + 256 * j] = ((pre_ok >> j) & 1u) ? pre[j] : 0.f;
        };
        // shifted copies: group u -> (plane c, shift s, row, 4-column group jj): P[c][s][row][4jj + e] = I[row][4jj + e + s]
        const int groups = a.Cin * 4 * PH * (CS_RS / 4);
        auto build_copies = [&]() __attribute__((always_inline)) {
            for (int u = tid; u < groups; u += 256) {
                const int jj = u % (CS_RS / 4);
                const int rest = u / (CS_RS / 4);
                const int row = rest % PH;
                const int cs = rest / PH;            // c * 4 + s
                const int s = cs & 3, c = cs >> 2;
                bf16x4 o, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = 4 * jj + e + s;
                    const float v = col < PW ? patch[c * plane + row * PW + col] : 0.f;
                    o[e] = (bf16_t)v;
                    ol[e] = (bf16_t)(v - (float)o[e]);      // (the remainder is exact in fp32)
                }
                *reinterpret_cast<bf16x4*>(sP + cs * copy_elems + row * CS_RS + 4 * jj) = o;
                if constexpr (PAIR) *reinterpret_cast<bf16x4*>(sP + (copies_bytes >> 1) + cs * copy_elems + row * CS_RS + 4 * jj) = ol;
            }
        };
        // ---- per-lane fragment addressing (tile independent)
        typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
        const int rsel = 4 * (lane >> 4) + ((lane >> 2) & 3);   // pixel row of a 16-row run this lane supplies to the transpose
        const int psel = lane & 3;
        int boff[TNW];                                          // element offset of (tap, tile row 0, column 4g) in sP
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            int k = (nt0 + j) * 16 + (lane & 15);
            if (k >= a.Kreal) k = 0;                            // padding taps: any finite data (cropped by the reduce)
            const int kk = a.kh * a.kw;
            const int c = k / kk, r = k - c * kk;
            const int ky = r / a.kw, kx = r - ky * a.kw;
            boff[j] = ((c * 4 + (kx & 3)) * PH + ky) * CS_RS + (kx & ~3) + 4 * (lane >> 4);
        }

        // one sweep of a tile: 8 k-steps of 32 pixels (two 16-pixel tile rows); K order inside a step: lane group g supplies
        // pixels (row 2kb, 4g..4g+3) then (row 2kb+1, 4g..4g+3), for A and B alike
        auto sweep = [&](const char* gt, const T* pc) __attribute__((always_inline)) {
#pragma unroll 2
            for (int kb = 0; kb < 8; ++kb) {
                const int cidx = mt_w * 2 + (psel >> 1);
                const int r0 = kb * 32 + rsel, r1 = r0 + 16;
                const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds_bf16x4_ptr)(gt + (r0 * CPM + cs_swz<CPM>(r0, cidx)) * 16 + 8 * (psel & 1)));
                const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds_bf16x4_ptr)(gt + (r1 * CPM + cs_swz<CPM>(r1, cidx)) * 16 + 8 * (psel & 1)));
                const bf16x8 fa = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    const T* p0 = pc + boff[j] + (2 * kb) * CS_RS;
                    const bf16x4 blo = *reinterpret_cast<const bf16x4*>(p0);
                    const bf16x4 bhi = *reinterpret_cast<const bf16x4*>(p0 + CS_RS);
                    const bf16x8 fb = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j], 0, 0, 0);
                }
            }
        };
        if constexpr (!PAIR) {
            issue_g(t_begin, 0);
            fetch_patch(t_begin);
            stash_patch();
            __syncthreads();
            build_copies();
            for (int sp = t_begin; sp < t_end; ++sp) {
                const int cur = (sp - t_begin) & 1;
                __syncthreads();            // G tile `cur` landed (vmcnt 0), shifted copies of this tile are visible
                if (sp + 1 < t_end) {
                    issue_g(sp + 1, cur ^ 1);
                    fetch_patch(sp + 1);
                }
                sweep(sG + cur * G_BYTES, sP);
                if (sp + 1 < t_end) {
                    __syncthreads();        // every wave is done with this tile's copies
                    stash_patch();
                    __syncthreads();
                    build_copies();
                }
            }
        } else {
            const T* sPl = sP + (copies_bytes >> 1);      // the lo halves of the shifted copies
            issue_g(t_begin, 0);
            issue_g(t_begin, 1);
            fetch_patch(t_begin);
            stash_patch();
            __syncthreads();
            build_copies();
            for (int sp = t_begin; sp < t_end; ++sp) {
                __syncthreads();            // both planes of this tile landed (vmcnt 0), its shifted copies are visible
                if (sp + 1 < t_end) fetch_patch(sp + 1);
                sweep(sG, sP);              // g_hi . I_hi
                sweep(sG, sPl);             // g_hi . I_lo
                __syncthreads();            // every wave is done with g_hi
                if (sp + 1 < t_end) issue_g(sp + 1, 0);
                sweep(sG + G_BYTES, sP);    // g_lo . I_hi   (the next tile's g_hi lands meanwhile)
                if (sp + 1 < t_end) {
                    __syncthreads();        // every wave is done with g_lo and with this tile's copies
                    issue_g(sp + 1, 1);     // lands under the next tile's first two sweeps ... (drained by its first barrier)
                    stash_patch();
                    __syncthreads();
                    build_copies();
                }
            }
        }
    }
    // ---- partial of this workgroup (zeros when it had no tile): acc[j][r] = C[m = 16 mt + 4 (lane>>4) + r][k = 16 nt + lane&15]
    float* __restrict__ out = a.part + ((size_t)n * gridDim.x + blockIdx.x) * BM * 64;
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mt_w * 16 + (lane >> 4) * 4 + r;
            out[m * 64 + (nt0 + j) * 16 + (lane & 15)] = acc[j][r];
        }
}

template <int BM, bool PAIR>
static int launch_corr(CorrArgs& a, int gx, hipStream_t st) {
    const int PH = 16 + a.kh - 1, PW = 16 + a.kw - 1;
    const int smem = 2 * 256 * BM * 2 + (PAIR ? 2 : 1) * a.Cin * 4 * PH * CS_RS * 2 + (a.Cin * PH * PW + 4) * 4;
    // two workgroups per CU up to 80 KB; the PAIR form's second set of shifted copies takes 2-plane images with 5x5 taps (and
    // wider: nothing in the reference's nets) past that -- those run one workgroup per CU
    constexpr int LDS_MAX = (PAIR ? 112 : 80) * 1024;
    if (smem > LDS_MAX) return CTG_EINVAL;
    static unsigned long long attr_mask = 0;       // per device
    if (smem > 64 * 1024) {
        const int rc = ctg_lds_attr_once((const void*)corr_small_kernel<BM, PAIR>, LDS_MAX, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    hipLaunchKernelGGL((corr_small_kernel<BM, PAIR>), dim3(gx, a.B), dim3(256), smem, st, a);
    return ctg_launch_status();
}

// C ABI.  part[(n * wgs + w)][m][k] (fp32, k < 64) = this workgroup's share of
//   C[m][k] = sum over grid pixels q in [0,Hs) x [0,Ws) of  Gpad[q][m] * Ipad[q + tap_k],   tap_k = (ky, kx) of channel c,
//   k = (c*kh + ky)*kw + kx,  Gpad[q] = g[pad_g(q - gpad)],  Ipad[j] = image[pad_i(j - ipad)],
// for bf16 g [B][Gh][Gw][g_ld] (channels [0, Mc), Mc in {32, 64}) and 1 or 2 fp32 image planes [B][Ih][Iw]; pad_* is
// reflection or zero.  `wgs` workgroups per sample (the caller sizes `part` as B*wgs*Mc*64 floats and finishes with
// ctg_wgrad_reduce(part, B*wgs, 1, Mc, 64, ...)).  Replaces the weight-gradient half of convolution_backward for
// nn.Conv2d(1|2, C, k) (Model/HdGan.py:70, trainer/reg.py:77) and nn.Conv2d(C, 1, 7) (HdGan.py:110).
// g_ld < 0 (ABI 8): g is a SPLIT PAIR of pitch -g_ld (lo plane -g_ld / 2 elements behind) and the launch accumulates
//   g_hi.I_hi + g_hi.I_lo + g_lo.I_hi (I_hi = bf16(I), I_lo = bf16(I - I_hi)): the "bf16x3" form, one pass over each plane of g.
extern "C" int ctg_corr_smallcin(const void* g, int Gh, int Gw, int g_ld, int Mc, int gpad, int g_pad_mode,
                                 const float* i0, const float* i1, int Cin, int Ih, int Iw, int kh, int kw, int ipad,
                                 int i_pad_mode, int B, int Hs, int Ws, float* part, int wgs, void* stream) {
    CTG_ENTER();
    if (g == nullptr || i0 == nullptr || part == nullptr || (Cin == 2 && i1 == nullptr)) return CTG_EINVAL;
    if (Cin < 1 || Cin > 2 || kh < 1 || kw < 1 || kw > 8 || Cin * kh * kw > 64) return CTG_EINVAL;
    const bool pair = g_ld < 0;
    if (pair) g_ld = -g_ld;
    if ((Mc != 32 && Mc != 64) || g_ld % 8 || g_ld < Mc || ((uintptr_t)g & 15)) return CTG_EINVAL;
    if (pair && (g_ld % 16 || g_ld < 2 * Mc)) return CTG_EINVAL;
    if (B < 1 || Hs < 1 || Ws < 1 || wgs < 1 || gpad < 0 || ipad < 0) return CTG_EINVAL;
    if (g_pad_mode == PAD_REFLECT && (gpad >= Gh || gpad >= Gw)) return CTG_EINVAL;
    if (i_pad_mode == PAD_REFLECT && (ipad >= Ih || ipad >= Iw)) return CTG_EINVAL;
    const int PH = 16 + kh - 1, PW = 16 + kw - 1;
    if (Cin * PH * PW > CS_PREF * 256) return CTG_EINVAL;
    CorrArgs a;
    a.g = g; a.i0 = i0; a.i1 = i1; a.part = part;
    a.B = B; a.Gh = Gh; a.Gw = Gw; a.g_ld = g_ld; a.gpad = gpad; a.g_pad_mode = g_pad_mode;
    a.Ih = Ih; a.Iw = Iw; a.Cin = Cin; a.kh = kh; a.kw = kw; a.ipad = ipad; a.i_pad_mode = i_pad_mode;
    a.Hs = Hs; a.Ws = Ws; a.Kreal = Cin * kh * kw;
    a.ntiles = ((Hs + 15) / 16) * ((Ws + 15) / 16);
    a.g_lo = g_ld / 2;
    hipStream_t st = (hipStream_t)stream;
    if (pair) return Mc == 64 ? launch_corr<64, true>(a, wgs, st) : launch_corr<32, true>(a, wgs, st);
    return Mc == 64 ? launch_corr<64, false>(a, wgs, st) : launch_corr<32, false>(a, wgs, st);
}
