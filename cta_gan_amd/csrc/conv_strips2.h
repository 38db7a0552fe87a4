// Sliding-window form of the STRIDE-2 3x3 convolution 64 -> 128 channels on large maps (gfx950, bf16): Conv2d(64, 128, 3,
// stride=2, padding=1) of the generator's first down-sampling layer (Model/HdGan.py:78-80, [B,512,512,64] -> [B,256,256,128]) and,
// the same operator, the backward-data pass of its second up-sampling layer ConvTranspose2d(128, 64, 3, stride=2, ...) (:93-95).
//
// On the gather kernel (conv_igemm_kernel<256,128>) these launches take 351 us for 155 GFLOP and 805 MB (441 TFLOP/s, 2.3 TB/s).
// The design is conv_stript.h's, mirrored:
//   * a workgroup owns a strip of 16 OUTPUT columns (33 input columns) of one sample and slides down a band of output rows; output
//     row j needs the input rows 2j-1, 2j, 2j+1: each step fetches one PAIR of input rows (2j, 2j+1) and keeps row 2j-1 from the
//     pair before;
//   * its four WAVES split the 128 output channels (two 16-channel MFMA tiles each): the wave's share of the weights -- 9 taps x
//     2 k-steps x 2 tiles = 36 A-fragments, 144 VGPRs -- stays in registers for the whole launch;
//   * the pairs go through one ring of 6 pair slots in LDS (2 x 33 px x 128 B each), fetched 4 steps ahead by LDS-DMA, three
//     instructions per wave and step, always (pairs past the band are fetched from a zero chunk: the counted s_waitcnt is the same
//     constant in every step).  A row is stored DE-INTERLEAVED -- even input columns, then odd ones -- so the 16 pixels 2p + kx of
//     a fragment read are 16 consecutive 128-byte slots, and their chunks are XORed with the slot index: conflict-free under
//     ds_read_b128's lane groups for kx = 0, 1, 2;
//   * per step and wave 18 pixel fragments, 36 MFMAs in the gather kernel's order (tap, then k-step): bit-identical results;
//   * ONE barrier per step: the waves put their 32 channels of the 16 pixels into a staging tile (double-buffered), and after the
//     barrier each wave stores four whole 256-byte pixels;
//   * InstanceNorm moments accumulate from the fp32 accumulators over the band.
#pragma once
#include "conv_halo.h"

#define STRIPS_R 6              // ring slots (input row pairs) of the workgroup
#define STRIPS_D 4              // a pair is fetched this many steps before the step that reads it as "its" pair
#define STRIPS_ROWB 4224        // 33 px x 128 B
#define STRIPS_PAIRB (2 * STRIPS_ROWB)
#define STRIPS_STAGE 4096       // one step's output tile: 16 px x 128 channels, bf16
#define STRIPS_SMEM (STRIPS_R * STRIPS_PAIRB + 2 * STRIPS_STAGE)       // 58880 B

struct StripS2Args {
    const bf16_t* x;            // [B][2 Ho][2 Wo][x_ld], 64 channels
    const bf16_t* w;            // packed [9][w_npad >= 128][64]
    bf16_t* y;                  // [B][Ho][Wo][y_ld], 128 channels
    float* stats;               // [B][slabs][128][2] or NULL
    int B, Ho, Wo, x_ld, y_ld, w_tap_stride;
    int band_rows, nbands, nstrips;
};

typedef float f32x2s_t __attribute__((ext_vector_type(2)));

// LDS slot of input column q (0 .. 32 <-> image column 2 ox0 - 1 + q) inside a ring row: even q first, then odd q
__device__ __forceinline__ int strips_pos(int q) { return (q & 1) ? 17 + (q >> 1) : (q >> 1); }

__global__ __launch_bounds__(256, 2) void conv_strips2_64_128_kernel(const StripS2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const int item = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int strip = item % a.nstrips;
    const int t2 = item / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    char* ring = smem;
    char* stage = smem + STRIPS_R * STRIPS_PAIRB;
    const int ox0 = strip * 16, oyb = band * a.band_rows;
    const int nrows = min(a.band_rows, a.Ho - oyb);      // output rows (steps) of this band
    const int Hi = 2 * a.Ho, Wi = 2 * a.Wo, x_ld = a.x_ld;

    // ---- weights of (tap t, k-step ks, tile nt): rows 32 wave + 16 nt + (lane & 15), channels ks*32 + (lane >> 4)*8 ..
    u32x4 wf[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[t][ks][nt] = *reinterpret_cast<const u32x4*>(a.w + (size_t)t * a.w_tap_stride + (32 * wave + 16 * nt + p) * 64 + ks * 32 + kg * 8);

    // ---- the ring starts as zeros: the slot of the column left of the image, and the pair above the image, are never written
    for (int i = tid; i < STRIPS_R * STRIPS_PAIRB / 16; i += 256) *reinterpret_cast<u32x4*>(ring + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- a pair's 528 chunks as one linear array [row][slot][chunk]: DMA instruction i covers chunks 64 i .. 64 i + 63; wave w
    //      issues i = 2w and 2w + 1, and every wave the last 16 (four copies of the same 256 bytes: the instruction count per wave
    //      stays uniform).  LDS chunk (row, slot, cs) holds source chunk cs ^ (slot & 7) of the slot's column.
    unsigned voff[3];
    bool vok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int L = d < 2 ? 128 * wave + 64 * d + lane : 512 + (lane & 15);
        const int row = L / 264, rem = L % 264, slot = rem >> 3, cs = rem & 7;
        const int q = slot < 17 ? 2 * slot : 2 * (slot - 17) + 1;
        const int cx = 2 * ox0 - 1 + q;
        vok[d] = cx >= 0 && cx < Wi;
        voff[d] = (unsigned)((row * Wi + q) * x_ld + (cs ^ (slot & 7)) * 8) * 2u;
    }
    const size_t ppitch = 2 * (size_t)Wi * x_ld * 2;                                                    // bytes per input row pair
    // pair 0 of the band: input row 2 oyb, column 2 ox0 - 1 (a pointer only; the column left of the image is never dereferenced)
    const char* __restrict__ X0 = reinterpret_cast<const char*>(a.x + (((size_t)n * Hi + 2 * oyb) * Wi + 2 * ox0 - 1) * (long)x_ld);
    auto issue = [&](int slot, const char* pp, bool pv) __attribute__((always_inline)) {   // pv (uniform): the pair exists
        const char* r = pv ? pp : reinterpret_cast<const char*>(g_zero_chunk);
        asm volatile("" : "+s"(r));             // opaque: keeps the addresses "uniform pair pointer + lane offset"
        char* dst = ring + slot * STRIPS_PAIRB;
        if (vok[0]) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[0] : 0u)), (lptr_t)(dst + 2048 * wave), 16, 0, 0);
        if (vok[1]) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[1] : 0u)), (lptr_t)(dst + 2048 * wave + 1024), 16, 0, 0);
        if (lane < 16) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[2] : 0u)), (lptr_t)(dst + 8192), 16, 0, 0);
    };
    // ---- fragment byte offsets inside a ring row for kx = 0, 1, 2: input column q = 2p + kx, logical chunk ks*4 + kg (ks flips
    //      bit 6 of the byte offset)
    int loff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int slot = strips_pos(2 * p + kx);
        loff[kx] = slot * 128 + ((kg ^ (slot & 7)) * 16);
    }
    // ---- the staging tile [16 px][128 ch] (256-byte pixels, their 16-byte chunks XORed with the pixel): this lane writes its
    //      2 x 4 channels of pixel p; after the barrier wave w stores pixels 4w .. 4w+3 whole: lane -> pixel 4w + lane/16, chunk
    //      lane%16
    char* stage_w[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) stage_w[nt] = stage + p * 256 + (((wave * 4 + 2 * nt + (kg >> 1)) ^ p) * 16) + (kg & 1) * 8;
    const int spx = 4 * wave + (lane >> 4), sch = lane & 15;
    const char* stage_r = stage + spx * 256 + ((sch ^ spx) * 16);
    bf16_t* __restrict__ yp = a.y + (((size_t)n * a.Ho + oyb) * a.Wo + ox0 + spx) * a.y_ld + sch * 8;
    const size_t ystep = (size_t)a.Wo * a.y_ld;
    f32x2s_t s1[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}}, s2[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
    const bool want_stats = a.stats != nullptr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // the zeroed ring, before any DMA lands in it
    asm volatile("" ::: "memory");

    // A step j of the workgroup: fetch pair j + D; multiply row B of pair j - 1 and rows A, B of pair j (visible since the barrier
    // of step j - 1) into the staging tile j & 1; wait for this wave's part of pair j + 1; barrier; store four whole pixels.
    // Between two barriers nobody reads a pair older than j - 1, so pair j + D may land in the slot of pair j + D - R = j - 2.
    // Newer than the DMA of pair j + 1 at the wait of step j: the DMA of pairs j+2 .. j+D (3 each), the stores of the last
    // min(j, D-1) steps (1 each).
    {
        const char* pp = X0;
        if (oyb > 0) issue(STRIPS_R - 1, X0 - ppitch, true);          // pair -1: its row B is input row 2 oyb - 1
#pragma unroll
        for (int k = 0; k < STRIPS_D; ++k) { issue(k, pp, k < nrows); pp += ppitch; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (STRIPS_D - 1)) : "memory");        // pairs -1 and 0 (and the weights)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#define STRIPS_TAP(T, ROWP, KX)                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                           \
        const u32x4 f = *reinterpret_cast<const u32x4*>((ROWP) + (loff[KX] ^ (ks * 64)));                                        \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                                         \
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][ks][nt]), __builtin_bit_cast(bf16x8, f),  \
                                                              acc[nt], 0, 0, 0);                                                \
    }
#define STRIPS_STEP(J, SLOT, PAR, NWAIT)                                                                                         \
    {                                                                                                                            \
        issue(((SLOT) + STRIPS_D) % STRIPS_R, pp, (J) + STRIPS_D < nrows);                                                       \
        pp += ppitch;                                                                                                            \
        const char* rm = ring + (((SLOT) + STRIPS_R - 1) % STRIPS_R) * STRIPS_PAIRB + STRIPS_ROWB;     /* input row 2j - 1 */    \
        const char* r0 = ring + (SLOT) * STRIPS_PAIRB;                                                 /* input row 2j     */    \
        const char* r1 = r0 + STRIPS_ROWB;                                                             /* input row 2j + 1 */    \
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};                                                             \
        STRIPS_TAP(0, rm, 0) STRIPS_TAP(1, rm, 1) STRIPS_TAP(2, rm, 2)                                                           \
        STRIPS_TAP(3, r0, 0) STRIPS_TAP(4, r0, 1) STRIPS_TAP(5, r0, 2)                                                           \
        STRIPS_TAP(6, r1, 0) STRIPS_TAP(7, r1, 1) STRIPS_TAP(8, r1, 2)                                                           \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                                       \
            bf16x4 o;                                                                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[nt][r];                                             \
            *reinterpret_cast<bf16x4*>(stage_w[nt] + (PAR) * STRIPS_STAGE) = o;                                                  \
        }                                                                                                                        \
        if (want_stats) {                                                                                                        \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                                     \
                _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                  \
                    const f32x2s_t v = {acc[nt][2 * h], acc[nt][2 * h + 1]};                                                     \
                    s1[nt][h] += v;                                                                                              \
                    s2[nt][h] = __builtin_elementwise_fma(v, v, s2[nt][h]);                                                      \
                }                                                                                                                \
        }                                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");                                                  \
        __builtin_amdgcn_s_barrier();                                                                                            \
        asm volatile("" ::: "memory");                                                                                          \
        const u32x4 o0 = *reinterpret_cast<const u32x4*>(stage_r + (PAR) * STRIPS_STAGE);                                        \
        *reinterpret_cast<u32x4*>(yp) = o0;                                                                                      \
        yp += ystep;                                                                                                             \
    }
        int j = 0;
#pragma unroll
        for (int u = 0; u < STRIPS_D - 1; ++u) {             // the first D - 1 steps: fewer stores in flight
            if (u < nrows) STRIPS_STEP(u, u, u & 1, 3 * (STRIPS_D - 1) + u)
        }
        for (j = STRIPS_D - 1; j < nrows; j += STRIPS_R) {
#pragma unroll
            for (int u = 0; u < STRIPS_R; ++u) {
                if (j + u >= nrows) break;
                STRIPS_STEP(j + u, (STRIPS_D - 1 + u) % STRIPS_R, (STRIPS_D - 1 + u) & 1, 4 * (STRIPS_D - 1))
            }
        }
#undef STRIPS_STEP
#undef STRIPS_TAP
    }
    if (want_stats) {
        const int slab = band * a.nstrips + strip, slabs = a.nbands * a.nstrips;
        float* dst = a.stats + (((size_t)n * slabs + slab) * 128) * 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t1 = row16_sum_to_lane15(s1[nt][r >> 1][r & 1]), t2 = row16_sum_to_lane15(s2[nt][r >> 1][r & 1]);
                if (p == 15) {
                    const int ch = 32 * wave + 16 * nt + kg * 4 + r;
                    dst[ch * 2] = t1;
                    dst[ch * 2 + 1] = t2;
                }
            }
    }
}

// returns -1 when the launch is not this kernel's shape: a = the ConvArgs ctg_conv_igemm built (one stride-2 3x3 window)
static int launch_strips2(const ConvArgs& a, float* stats, hipStream_t st, int* slabs_out) {
    static const bool off = getenv("CTG_NO_STRIPS2") != nullptr;     // A/B switch (scripts/ab.sh)
    if (off || a.ncls != 1 || a.Cin != 64 || a.Cout != 128 || a.os != 1 || a.is != 2 || a.oy0 || a.ox0 || a.frame ||
        a.bias != nullptr || a.act != ACT_NONE || a.pad_mode != PAD_ZERO || a.res != nullptr || a.fold != nullptr ||
        a.Hs != a.Ho || a.Ws != a.Wo || a.Hi != 2 * a.Ho || a.Wi != 2 * a.Wo || (a.Wo & 15) || a.Ho < 8 || (a.x_ld & 7) || (a.y_ld & 7) ||
        a.ntaps != 9)
        return -1;
    if ((long)a.B * a.Ho * a.Wo < (1L << 18) || (long)a.Hi * a.Wi * a.x_ld >= (1L << 30)) return -1;
    for (int t = 0; t < 9; ++t) {    // Conv2d(k=3, s=2, p=1): tap t = (ky, kx) reads input (2 oy + ky - 1, 2 ox + kx - 1), weight t
        const int tw = a.taps[t];
        if ((tw & 0xff) - 64 != t / 3 - 1 || ((tw >> 8) & 0xff) - 64 != t % 3 - 1 || (tw >> 16) != t) return -1;
    }
    StripS2Args s;
    s.x = (const bf16_t*)a.x; s.w = (const bf16_t*)a.w; s.y = (bf16_t*)a.y; s.stats = stats;
    s.B = a.B; s.Ho = a.Ho; s.Wo = a.Wo; s.x_ld = a.x_ld; s.y_ld = a.y_ld; s.w_tap_stride = a.w_tap_stride;
    s.nstrips = a.Wo / 16;
    const int n_cu = ctg_cu_count();
    static const int band_env = getenv("CTG_STRIPS2_BAND") ? atoi(getenv("CTG_STRIPS2_BAND")) : 0;     // A/B knob
    // two workgroups per CU are resident (registers): bands so that the grid fills the chip once
    long nb = (2L * n_cu) / ((long)a.B * s.nstrips);
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
        const int rc = ctg_lds_attr_once((const void*)conv_strips2_64_128_kernel, STRIPS_SMEM, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const dim3 grid((unsigned)((long)a.B * s.nbands * s.nstrips));
    hipLaunchKernelGGL(conv_strips2_64_128_kernel, grid, dim3(256), STRIPS_SMEM, st, s);
    return ctg_launch_status();
}
