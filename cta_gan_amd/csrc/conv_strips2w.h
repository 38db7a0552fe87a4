// The sliding-window stride-2 conv of conv_strips2.h for the next layer down: Conv2d(128, 256, 3, stride=2, padding=1) of the
// generator's second down-sampling layer (Model/HdGan.py:78-80, [B,256,256,128] -> [B,128,128,256]) and, the same operator, the
// backward-data pass of its first up-sampling layer ConvTranspose2d(256, 128, 3, stride=2, ...) (:93-95): three launches of 155
// GFLOP / 403 MB per step, 267 us each on the gather kernel (conv_igemm_kernel<256,128>, 580 TFLOP/s).
//
// The 590 KB of weights do not fit one workgroup's registers, half of them do: a workgroup of EIGHT waves owns a strip of 16 output
// columns x 128 of the 256 output channels (blockIdx picks the half; the two workgroups of a strip run next to each other on one
// XCD and share the input rows through L2).  Each wave holds the weights of ONE 16-channel MFMA tile over the whole K = 9 x 128:
// 36 A-fragments, 144 VGPRs, as in the narrower kernels.  Everything else is conv_strips2.h with 256-byte pixels: ring of input
// row pairs (2 x 33 px x 256 B, even input columns first, then odd; chunks XORed with 2 (slot & 7): conflict-free for kx = 0, 1, 2
// under ds_read_b128's lane groups), three DMA instructions per wave and step always, one barrier per step, a staging tile and
// whole-pixel (256-byte) stores, moments from the fp32 accumulators, the gather kernel's (tap, k-step) order: bit-identical results.
#pragma once
#include "conv_halo.h"

#define STRIPW_R 6              // ring slots (input row pairs) of the workgroup
#define STRIPW_D 4              // a pair is fetched this many steps before the step that reads it as "its" pair
#define STRIPW_ROWB 8448        // 33 px x 256 B
#define STRIPW_PAIRB (2 * STRIPW_ROWB)
#define STRIPW_STAGE 4096       // one step's output tile: 16 px x 128 channels, bf16
#define STRIPW_SMEM (STRIPW_R * STRIPW_PAIRB + 2 * STRIPW_STAGE)       // 109568 B: one workgroup (8 waves x ~230 VGPRs) per CU

struct StripS2WArgs {
    const bf16_t* x;            // [B][2 Ho][2 Wo][x_ld], 128 channels
    const bf16_t* w;            // packed [9][w_npad >= 256][128]
    bf16_t* y;                  // [B][Ho][Wo][y_ld], 256 channels
    float* stats;               // [B][slabs][256][2] or NULL
    int B, Ho, Wo, x_ld, y_ld, w_tap_stride;
    int band_rows, nbands, nstrips;
};

__global__ __launch_bounds__(512, 2) void conv_strips2_128_256_kernel(const StripS2WArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const int item = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int ng = item & 1;                              // which 128 of the 256 output channels
    const int t1 = item >> 1;
    const int strip = t1 % a.nstrips;
    const int t2 = t1 / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    char* ring = smem;
    char* stage = smem + STRIPW_R * STRIPW_PAIRB;
    const int ox0 = strip * 16, oyb = band * a.band_rows;
    const int nrows = min(a.band_rows, a.Ho - oyb);      // output rows (steps) of this band
    const int Hi = 2 * a.Ho, Wi = 2 * a.Wo, x_ld = a.x_ld;
    const int co0 = ng * 128 + wave * 16;                // this wave's 16 output channels

    // ---- weights of (tap t, k-step ks): rows co0 + (lane & 15), channels ks*32 + (lane >> 4)*8 ..
    u32x4 wf[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            wf[t][ks] = *reinterpret_cast<const u32x4*>(a.w + (size_t)t * a.w_tap_stride + (co0 + p) * 128 + ks * 32 + kg * 8);

    // ---- the ring starts as zeros: the slot of the column left of the image, and the pair above the image, are never written
    for (int i = tid; i < STRIPW_R * STRIPW_PAIRB / 16; i += 512) *reinterpret_cast<u32x4*>(ring + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- a pair's 1056 chunks as one linear array [row][slot][chunk]: DMA instruction i covers chunks 64 i .. 64 i + 63; wave w
    //      issues i = 2w and 2w + 1, and every wave the last 32 (eight copies of the same 512 bytes: the instruction count per wave
    //      stays uniform).  LDS chunk (row, slot, cs) holds source chunk cs ^ 2 (slot & 7) of the slot's column.
    unsigned voff[3];
    bool vok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int L = d < 2 ? 128 * wave + 64 * d + lane : 1024 + (lane & 31);
        const int row = L / 528, rem = L % 528, slot = rem >> 4, cs = rem & 15;
        const int q = slot < 17 ? 2 * slot : 2 * (slot - 17) + 1;
        const int cx = 2 * ox0 - 1 + q;
        vok[d] = cx >= 0 && cx < Wi;
        voff[d] = (unsigned)((row * Wi + q) * x_ld + (cs ^ (2 * (slot & 7))) * 8) * 2u;
    }
    const size_t ppitch = 2 * (size_t)Wi * x_ld * 2;                                                    // bytes per input row pair
    const char* __restrict__ X0 = reinterpret_cast<const char*>(a.x + (((size_t)n * Hi + 2 * oyb) * Wi + 2 * ox0 - 1) * (long)x_ld);
    auto issue = [&](int slot, const char* pp, bool pv) __attribute__((always_inline)) {   // pv (uniform): the pair exists
        const char* r = pv ? pp : reinterpret_cast<const char*>(g_zero_chunk);
        asm volatile("" : "+s"(r));             // opaque: keeps the addresses "uniform pair pointer + lane offset"
        char* dst = ring + slot * STRIPW_PAIRB;
        if (vok[0]) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[0] : 0u)), (lptr_t)(dst + 2048 * wave), 16, 0, 0);
        if (vok[1]) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[1] : 0u)), (lptr_t)(dst + 2048 * wave + 1024), 16, 0, 0);
        if (lane < 32) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[2] : 0u)), (lptr_t)(dst + 16384), 16, 0, 0);
    };
    // ---- fragment byte offsets inside a ring row for kx = 0, 1, 2: input column q = 2p + kx, logical chunk ks*4 + kg (ks XORs
    //      bits 6-7 of the byte offset)
    int loff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int q = 2 * p + kx;
        const int slot = (q & 1) ? 17 + (q >> 1) : (q >> 1);
        loff[kx] = slot * 256 + ((kg ^ (2 * (slot & 7))) * 16);
    }
    // ---- the staging tile [16 px][128 ch] (256-byte pixels, their 16-byte chunks XORed with the pixel): this lane writes its 4
    //      channels of pixel p; after the barrier wave w stores pixels 2w, 2w+1 whole (32 lanes: lane -> pixel 2w + lane/16, chunk
    //      lane%16)
    char* stage_w = stage + p * 256 + (((wave * 2 + (kg >> 1)) ^ p) * 16) + (kg & 1) * 8;
    const int spx = 2 * wave + ((lane >> 4) & 1), sch = lane & 15;
    const char* stage_r = stage + spx * 256 + ((sch ^ spx) * 16);
    const bool st_lane = lane < 32;
    bf16_t* __restrict__ yp = a.y + (((size_t)n * a.Ho + oyb) * a.Wo + ox0 + spx) * a.y_ld + ng * 128 + sch * 8;
    const size_t ystep = (size_t)a.Wo * a.y_ld;
    f32x2s_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    const bool want_stats = a.stats != nullptr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // the zeroed ring, before any DMA lands in it
    asm volatile("" ::: "memory");

    // Step j as in conv_strips2.h.  Newer than the DMA of pair j + 1 at the wait of step j: the DMA of pairs j+2 .. j+D (3 each),
    // the stores of the last min(j, D-1) steps (1 each).
    {
        const char* pp = X0;
        if (oyb > 0) issue(STRIPW_R - 1, X0 - ppitch, true);          // pair -1: its row B is input row 2 oyb - 1
#pragma unroll
        for (int k = 0; k < STRIPW_D; ++k) { issue(k, pp, k < nrows); pp += ppitch; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (STRIPW_D - 1)) : "memory");        // pairs -1 and 0 (and the weights)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#define STRIPW_TAP(T, ROWP, KX)                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                           \
        const u32x4 f = *reinterpret_cast<const u32x4*>((ROWP) + (loff[KX] ^ (ks * 64)));                                        \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][ks]), __builtin_bit_cast(bf16x8, f), acc, 0, 0, 0); \
    }
#define STRIPW_STEP(J, SLOT, PAR, NWAIT)                                                                                         \
    {                                                                                                                            \
        issue(((SLOT) + STRIPW_D) % STRIPW_R, pp, (J) + STRIPW_D < nrows);                                                       \
        pp += ppitch;                                                                                                            \
        const char* rm = ring + (((SLOT) + STRIPW_R - 1) % STRIPW_R) * STRIPW_PAIRB + STRIPW_ROWB;     /* input row 2j - 1 */    \
        const char* r0 = ring + (SLOT) * STRIPW_PAIRB;                                                 /* input row 2j     */    \
        const char* r1 = r0 + STRIPW_ROWB;                                                             /* input row 2j + 1 */    \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                                                        \
        STRIPW_TAP(0, rm, 0) STRIPW_TAP(1, rm, 1) STRIPW_TAP(2, rm, 2)                                                           \
        STRIPW_TAP(3, r0, 0) STRIPW_TAP(4, r0, 1) STRIPW_TAP(5, r0, 2)                                                           \
        STRIPW_TAP(6, r1, 0) STRIPW_TAP(7, r1, 1) STRIPW_TAP(8, r1, 2)                                                           \
        {                                                                                                                        \
            bf16x4 o;                                                                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[r];                                                 \
            *reinterpret_cast<bf16x4*>(stage_w + (PAR) * STRIPW_STAGE) = o;                                                      \
        }                                                                                                                        \
        if (want_stats) {                                                                                                        \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                      \
                const f32x2s_t v = {acc[2 * h], acc[2 * h + 1]};                                                                 \
                s1[h] += v;                                                                                                      \
                s2[h] = __builtin_elementwise_fma(v, v, s2[h]);                                                                  \
            }                                                                                                                    \
        }                                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");                                                  \
        __builtin_amdgcn_s_barrier();                                                                                            \
        asm volatile("" ::: "memory");                                                                                          \
        if (st_lane) {                                                                                                           \
            const u32x4 o0 = *reinterpret_cast<const u32x4*>(stage_r + (PAR) * STRIPW_STAGE);                                    \
            *reinterpret_cast<u32x4*>(yp) = o0;                                                                                  \
        }                                                                                                                        \
        yp += ystep;                                                                                                             \
    }
        int j = 0;
#pragma unroll
        for (int u = 0; u < STRIPW_D - 1; ++u) {             // the first D - 1 steps: fewer stores in flight
            if (u < nrows) STRIPW_STEP(u, u, u & 1, 3 * (STRIPW_D - 1) + u)
        }
        for (j = STRIPW_D - 1; j < nrows; j += STRIPW_R) {
#pragma unroll
            for (int u = 0; u < STRIPW_R; ++u) {
                if (j + u >= nrows) break;
                STRIPW_STEP(j + u, (STRIPW_D - 1 + u) % STRIPW_R, (STRIPW_D - 1 + u) & 1, 4 * (STRIPW_D - 1))
            }
        }
#undef STRIPW_STEP
#undef STRIPW_TAP
    }
    if (want_stats) {
        const int slab = band * a.nstrips + strip, slabs = a.nbands * a.nstrips;
        float* dst = a.stats + (((size_t)n * slabs + slab) * 256) * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t1 = row16_sum_to_lane15(s1[r >> 1][r & 1]), t2 = row16_sum_to_lane15(s2[r >> 1][r & 1]);
            if (p == 15) {
                const int ch = co0 + kg * 4 + r;
                dst[ch * 2] = t1;
                dst[ch * 2 + 1] = t2;
            }
        }
    }
}

// returns -1 when the launch is not this kernel's shape: a = the ConvArgs ctg_conv_igemm built (one stride-2 3x3 window)
static int launch_strips2w(const ConvArgs& a, float* stats, hipStream_t st, int* slabs_out) {
    static const bool off = getenv("CTG_NO_STRIPS2W") != nullptr;    // A/B switch (scripts/ab.sh)
    if (off || a.ncls != 1 || a.Cin != 128 || a.Cout != 256 || a.os != 1 || a.is != 2 || a.oy0 || a.ox0 || a.frame ||
        a.bias != nullptr || a.act != ACT_NONE || a.pad_mode != PAD_ZERO || a.res != nullptr || a.fold != nullptr ||
        a.Hs != a.Ho || a.Ws != a.Wo || a.Hi != 2 * a.Ho || a.Wi != 2 * a.Wo || (a.Wo & 15) || a.Ho < 8 || (a.x_ld & 7) || (a.y_ld & 7) ||
        a.ntaps != 9)
        return -1;
    if ((long)a.B * a.Ho * a.Wo < (1L << 17) || (long)a.Hi * a.Wi * a.x_ld >= (1L << 30)) return -1;
    for (int t = 0; t < 9; ++t) {    // Conv2d(k=3, s=2, p=1): tap t = (ky, kx) reads input (2 oy + ky - 1, 2 ox + kx - 1), weight t
        const int tw = a.taps[t];
        if ((tw & 0xff) - 64 != t / 3 - 1 || ((tw >> 8) & 0xff) - 64 != t % 3 - 1 || (tw >> 16) != t) return -1;
    }
    StripS2WArgs s;
    s.x = (const bf16_t*)a.x; s.w = (const bf16_t*)a.w; s.y = (bf16_t*)a.y; s.stats = stats;
    s.B = a.B; s.Ho = a.Ho; s.Wo = a.Wo; s.x_ld = a.x_ld; s.y_ld = a.y_ld; s.w_tap_stride = a.w_tap_stride;
    s.nstrips = a.Wo / 16;
    const int n_cu = ctg_cu_count();
    static const int band_env = getenv("CTG_STRIPS2W_BAND") ? atoi(getenv("CTG_STRIPS2W_BAND")) : 0;   // A/B knob
    // one workgroup per CU is resident: bands so that the grid (x 2 channel halves) fills the chip once
    long nb = (long)n_cu / (2L * a.B * s.nstrips);
    if (nb < 1) nb = 1;
    int band = (int)((a.Ho + nb - 1) / nb);
    if (band < 8) band = 8;
    if (band_env >= 8) band = band_env;      // (the caller sized the moments buffer for >= 8-row bands)
    s.band_rows = band;
    s.nbands = (a.Ho + band - 1) / band;
    // the caller sized the moments buffer for ceil(Ho / 8) x ceil(Wo / 16) slabs per sample
    if (stats != nullptr && slabs_out != nullptr) *slabs_out = s.nbands * s.nstrips;
    static unsigned long long attr_mask = 0;       // per device
    {
        const int rc = ctg_lds_attr_once((const void*)conv_strips2_128_256_kernel, STRIPW_SMEM, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const dim3 grid((unsigned)(2L * a.B * s.nbands * s.nstrips));
    hipLaunchKernelGGL(conv_strips2_128_256_kernel, grid, dim3(512), STRIPW_SMEM, st, s);
    return ctg_launch_status();
}
