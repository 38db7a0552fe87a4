// Sliding-window form of the stride-2 TRANSPOSED 3x3 convolution 128 -> 64 channels on large maps (gfx950, bf16):
// ConvTranspose2d(128, 64, 3, stride=2, padding=1, output_padding=1) of the generator's second up-sampling layer
// (Model/HdGan.py:93-95, [B,256,256,128] -> [B,512,512,64]) and, the same operator, the backward-data pass of its first
// down-sampling conv Conv2d(64, 128, 3, stride=2, padding=1) (:78-80).
//
// As four parity classes on conv_halo_kernel (one launch, ctg_conv_igemm_classes) these launches take 459 us for 155 GFLOP and
// 805 MB: 144 MFMAs per tile and wave inside ~2600 other instructions (profiles/r03_sq_table.md) -- bound by instruction issue at
// 340 TFLOP/s and 1.75 TB/s.  This kernel: 205-220 us (710-750 TFLOP/s, 3.7-3.9 TB/s; PMC traffic 1.00x the algorithmic bytes).
//   * a workgroup owns a 16-pixel-wide INPUT column strip (32 output columns) of one sample and slides down a band of input rows;
//     input row j and j+1 give output rows 2j and 2j+1, i.e. all four parity classes of the strip at once -- one input fetch
//     instead of four, no halo overlap between classes;
//   * its four WAVES split the 64 output channels (one 16-channel MFMA tile each), so the wave's share of the weights -- 9 taps x
//     4 k-steps = 36 A-fragments, 144 VGPRs -- stays in registers for the whole launch: no weight traffic after the prologue;
//   * the input rows (17 px x 256 B) go through one ring of 8 rows in LDS: every step each wave fetches a quarter of the row six
//     steps ahead by LDS-DMA (two instructions per wave and step, always: rows that do not exist are fetched from a zero chunk,
//     so the counted s_waitcnt is the same constant in every step); the 16-byte chunks of a pixel are XOR-swizzled so that the
//     fragment reads of the pixels p and p + 1 are both conflict-free under ds_read_b128's lane groups;
//   * per 64-channel half of K: 8 pixel fragments (2 rows x 2 columns x 2 k-steps), then the 9 (class, tap) pairs in the order
//     the class kernels use (tap, then k-step), four independent accumulation chains -- results are bit-identical to theirs;
//   * ONE barrier per step (36 MFMAs per wave): the waves put their 16 channels of the four classes into a staging tile
//     (double-buffered), and after the barrier wave q stores class q as whole 128-byte pixels.  (A first version without any
//     barrier -- a private ring per wave, each wave storing its own 32 bytes of a pixel -- ran at 310 us: the 32-byte partial
//     stores cost 1.24x write traffic and most of the time; knock-out timings in DESIGN.md section 8.)
//   * InstanceNorm moments accumulate from the fp32 accumulators over the band (packed fp32 adds / fmas).
#pragma once
#include "conv_halo.h"

#define STRIPT_R 8              // ring rows of the workgroup
#define STRIPT_D 6              // a row is fetched this many steps before the step that first reads it as its lower row
#define STRIPT_ROWB 4352        // 17 px x 256 B
#define STRIPT_STAGE 8192       // one step's output tile: 4 classes x 16 px x 64 channels, bf16
#define STRIPT_SMEM (STRIPT_R * STRIPT_ROWB + 2 * STRIPT_STAGE)        // 51200 B

struct StripTArgs {
    const bf16_t* x;            // [B][Hi][Wi][x_ld], 128 channels
    const bf16_t* w;            // packed [9][w_npad >= 64][128]
    bf16_t* y;                  // [B][2 Hi][2 Wi][y_ld], 64 channels
    float* stats;               // [B][slabs][64][2] or NULL
    int B, Hi, Wi, x_ld, y_ld, w_tap_stride;
    int band_rows, nbands, nstrips, xcd;
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// The arithmetic of one step of a wave: input rows at LDS r0 (row j) and r1 (row j + 1) -> its 16 channels of the four parity
// classes of output rows 2j, 2j + 1, as bf16 into the step's staging tile; InstanceNorm moments from the fp32 accumulators.
// MASKED: the strip reaches past the image's right edge (moments of columns >= Wi are dropped).
template <bool MASKED>
__device__ __forceinline__ void stript_mma(const u32x4 (&wf)[9][4], const char* r0, const char* r1, const int (&loff)[2][4], char* stage_w,
                                           bool col_ok, bool want_stats, f32x2_t (&s1)[2], f32x2_t (&s2)[2]) {
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // the 8 pixel fragments of this 64-channel half: f[dy][dx][ks]
        u32x4 f[2][2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) f[dy][dx][ks] = *reinterpret_cast<const u32x4*>((dy ? r1 : r0) + loff[dx][c * 2 + ks]);
#define STRIPT_TAP(Q, DY, DX, WIDX)                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                         \
        acc[Q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[WIDX][c * 2 + ks]),                  \
                                                         __builtin_bit_cast(bf16x8, f[DY][DX][ks]), acc[Q], 0, 0, 0);
        // class (0,0): tap (0,0) w4; (0,1): (0,1) w3, (0,0) w5; (1,0): (1,0) w1, (0,0) w7; (1,1): (1,1) w0, (1,0) w2, (0,1) w6,
        // (0,0) w8 -- engine._convT_classes(3, 1); interleaved over the classes, each class in its own list order
        STRIPT_TAP(0, 0, 0, 4)
        STRIPT_TAP(1, 0, 1, 3)
        STRIPT_TAP(2, 1, 0, 1)
        STRIPT_TAP(3, 1, 1, 0)
        STRIPT_TAP(1, 0, 0, 5)
        STRIPT_TAP(2, 0, 0, 7)
        STRIPT_TAP(3, 1, 0, 2)
        STRIPT_TAP(3, 0, 1, 6)
        STRIPT_TAP(3, 0, 0, 8)
#undef STRIPT_TAP
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[q][r];
        *reinterpret_cast<bf16x4*>(stage_w + q * 2048) = o;
    }
    if (want_stats) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2_t v = {acc[q][2 * h], acc[q][2 * h + 1]};
                if (MASKED && !col_ok) v = f32x2_t{0.f, 0.f};
                s1[h] += v;
                s2[h] = __builtin_elementwise_fma(v, v, s2[h]);
            }
    }
}

__global__ __launch_bounds__(256, 2) void conv_stript_128_64_kernel(const StripTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const int nt = wave;                                 // this wave's 16 output channels
    const int item = a.xcd ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int strip = item % a.nstrips;
    const int t2 = item / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    char* ring = smem;
    char* stage = smem + STRIPT_R * STRIPT_ROWB;
    const int i0 = strip * 16, jb = band * a.band_rows;
    const int nrows = min(a.band_rows, a.Hi - jb);       // input rows (steps) of this band
    const int nin = nrows + 1;                           // rows jb .. jb + nrows
    const int Hi = a.Hi, Wi = a.Wi, x_ld = a.x_ld;

    // ---- weights of (tap widx, k-step kk) for this wave's n-tile: rows nt*16 + (lane & 15), channels kk*32 + (lane >> 4)*8 ..
    u32x4 wf[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            wf[t][kk] = *reinterpret_cast<const u32x4*>(a.w + (size_t)t * a.w_tap_stride + (nt * 16 + p) * 128 + kk * 32 + kg * 8);

    // ---- the ring starts as zeros: the slots of columns past the image's right edge are never written again (zero padding)
    for (int i = tid; i < STRIPT_R * STRIPT_ROWB / 16; i += 256) *reinterpret_cast<u32x4*>(ring + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- a row's 17 pixels: wave w fetches pixels 4w .. 4w+3 (one DMA instruction: lane -> pixel 4w + lane/16, chunk lane%16) and
    //      every wave the 17th (16 lanes; four copies of the same 256 bytes: the instruction count per wave stays uniform).  Slot
    //      (px, cs) holds source chunk cs ^ 2 (px & 7) of column i0 + px.
    const int pxa = 4 * wave + (lane >> 4), csa = lane & 15;
    const unsigned voffa = (unsigned)(pxa * x_ld + (csa ^ ((pxa & 7) * 2)) * 8) * 2u;
    const bool voka = i0 + pxa < Wi;
    const bool has17 = i0 + 16 < Wi;                                                                    // uniform
    const unsigned voffb = (unsigned)(16 * x_ld + csa * 8) * 2u;
    const size_t rpitch = (size_t)Wi * x_ld * 2;                                                       // bytes per input row
    const char* __restrict__ X0 = reinterpret_cast<const char*>(a.x + (((size_t)n * Hi + jb) * Wi + i0) * x_ld);   // row jb, column i0
    // a row of a strip inside the image: two DMA instructions per wave, always (the wait counts rely on it) -- a row below the
    // image, or past the band's last row (never read), is fetched as zeros into its slot
    const int nvalid = min(nin, Hi - jb);
    auto issue_fast = [&](int slot, const char* rowp, int k) __attribute__((always_inline)) {
        const bool rv = k < nvalid;             // uniform
        const char* ra = rv ? rowp : reinterpret_cast<const char*>(g_zero_chunk);
        const char* rb = (rv && has17) ? rowp : reinterpret_cast<const char*>(g_zero_chunk);
        asm volatile("" : "+s"(ra), "+s"(rb));  // opaque: keeps the addresses "uniform row pointer + lane offset"
        char* dst = ring + slot * STRIPT_ROWB;
        __builtin_amdgcn_global_load_lds((gptr_t)(ra + (rv ? voffa : 0u)), (lptr_t)(dst + 1024 * wave), 16, 0, 0);
        if (lane < 16) __builtin_amdgcn_global_load_lds((gptr_t)(rb + ((rv && has17) ? voffb : 0u)), (lptr_t)(dst + 4096), 16, 0, 0);
    };
    // any row: columns past the edge are skipped, the row below the image is written as zeros
    auto issue_slow = [&](int k) __attribute__((always_inline)) {
        char* dst = ring + (k % STRIPT_R) * STRIPT_ROWB;
        const char* rowp = X0 + (size_t)k * rpitch;
        if (jb + k < Hi) {
            if (voka) __builtin_amdgcn_global_load_lds((gptr_t)(rowp + voffa), (lptr_t)(dst + 1024 * wave), 16, 0, 0);
            if (has17 && lane < 16) __builtin_amdgcn_global_load_lds((gptr_t)(rowp + voffb), (lptr_t)(dst + 4096), 16, 0, 0);
        } else {
            *reinterpret_cast<u32x4*>(dst + 1024 * wave + lane * 16) = u32x4{0u, 0u, 0u, 0u};
            if (lane < 16) *reinterpret_cast<u32x4*>(dst + 4096 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    };
    // ---- fragment byte offsets inside a ring row: pixel p + dx, logical chunk kk*4 + kg, physical chunk XORed with 2 (px & 7):
    //      ds_read_b128 serves the lanes in the groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, (+32) (MI355X_MICROARCH.md, LDS);
    //      with this XOR the 16 lanes of a group fall on 16 different chunks for the pixels p AND for the pixels p + 1 (XOR with
    //      px & 15 leaves the p + 1 reads 2-way conflicted)
    int loff[2][4];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int px = p + dx;
            loff[dx][kk] = (px * 16 + ((kk * 4 + kg) ^ ((px & 7) * 2))) * 16;
        }
    // ---- the staging tile [class][px][64 ch] (128-byte pixels, their 16-byte chunks XORed with px & 7): this lane writes its 4
    //      channels of pixel p of each class; after the barrier wave q stores class q as whole 128-byte pixels: lane -> pixel
    //      lane/8 (+8), chunk lane%8
    char* stage_w = stage + p * 128 + (((nt * 2 + (kg >> 1)) ^ (p & 7)) * 16) + (kg & 1) * 8;
    const int spx = lane >> 3, sch = lane & 7;
    const char* stage_r = stage + wave * 2048 + spx * 128 + ((sch ^ spx) * 16);       // pixel spx; pixel spx + 8 is 1024 bytes on
    const bool col_ok = i0 + p < Wi;
    const bool full_strip = i0 + 16 <= Wi;
    const bool st_ok0 = i0 + spx < Wi, st_ok1 = i0 + spx + 8 < Wi;
    const int Wo = 2 * Wi;
    // this lane's output pointer at step 0: class q = wave, row 2 jb + (q >> 1), column 2 (i0 + spx) + (q & 1), channels sch*8 ..
    bf16_t* __restrict__ yp = a.y + (((size_t)n * 2 * Hi + 2 * jb + (wave >> 1)) * Wo + 2 * (i0 + spx) + (wave & 1)) * a.y_ld + sch * 8;
    const size_t ystep = 2 * (size_t)Wo * a.y_ld;          // two output rows per step
    const size_t y8 = 16 * (size_t)a.y_ld;                 // pixel spx + 8: 16 output columns on
    f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    const bool want_stats = a.stats != nullptr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // the zeroed ring, before any DMA lands in it
    asm volatile("" ::: "memory");

    // A step j of the workgroup: fetch row j + D; multiply rows j, j+1 (visible since the barrier of step j - 1) into the staging
    // tile j & 1; wait for this wave's part of row j + 2; barrier; store the wave's class as whole pixels.  Between two barriers
    // nobody reads a row older than j, so row j + D may land in the slot of row j + D - R <= j - 2.
    int j = 0;
    if (full_strip && nrows >= STRIPT_D - 2) {
        // ---- the fast path of a strip inside the image.  Newer than the DMA of row j + 2 at the wait of step j are the DMA of rows
        //      j+3 .. j+D (2 each) and the stores of the last min(j, D-2) steps (2 each)
        const char* rowp = X0;
#pragma unroll
        for (int k = 0; k < STRIPT_D; ++k) { issue_fast(k, rowp, k); rowp += rpitch; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (STRIPT_D - 2)) : "memory");       // rows 0 and 1 (and the weights)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#define STRIPT_FAST_STEP(J, SLOT, PAR, NWAIT)                                                                                     \
    {                                                                                                                            \
        issue_fast(((SLOT) + STRIPT_D) % STRIPT_R, rowp, (J) + STRIPT_D);                                                        \
        rowp += rpitch;                                                                                                          \
        stript_mma<false>(wf, ring + (SLOT) * STRIPT_ROWB, ring + (((SLOT) + 1) % STRIPT_R) * STRIPT_ROWB, loff,                   \
                          stage_w + (PAR) * STRIPT_STAGE, true, want_stats, s1, s2);                                             \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");                                                  \
        __builtin_amdgcn_s_barrier();                                                                                            \
        asm volatile("" ::: "memory");                                                                                          \
        const u32x4 o0 = *reinterpret_cast<const u32x4*>(stage_r + (PAR) * STRIPT_STAGE);                                        \
        const u32x4 o1 = *reinterpret_cast<const u32x4*>(stage_r + (PAR) * STRIPT_STAGE + 1024);                                 \
        *reinterpret_cast<u32x4*>(yp) = o0;                                                                                      \
        *reinterpret_cast<u32x4*>(yp + y8) = o1;                                                                                 \
        yp += ystep;                                                                                                             \
    }
#pragma unroll
        for (int u = 0; u < STRIPT_D - 2; ++u) STRIPT_FAST_STEP(u, u, u & 1, 2 * (STRIPT_D - 2) + 2 * u)  // fewer stores in flight
        static_assert(((STRIPT_D - 2) & 1) == 0, "staging parity of the unrolled loop");
        for (j = STRIPT_D - 2; j < nrows; j += STRIPT_R) {
#pragma unroll
            for (int u = 0; u < STRIPT_R; ++u) {
                if (j + u >= nrows) break;
                STRIPT_FAST_STEP(j + u, (STRIPT_D - 2 + u) % STRIPT_R, u & 1, 4 * (STRIPT_D - 2))
            }
        }
        j = nrows;
#undef STRIPT_FAST_STEP
    } else {
        for (int k = 0; k < STRIPT_D && k < nin; ++k) issue_slow(k);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // ---- every step of a ragged strip or a very short band: nothing left in flight at a barrier
    for (; j < nrows; ++j) {
        if (j + STRIPT_D < nin) issue_slow(j + STRIPT_D);
        const int par = j & 1;
        stript_mma<true>(wf, ring + (j % STRIPT_R) * STRIPT_ROWB, ring + ((j + 1) % STRIPT_R) * STRIPT_ROWB, loff,
                         stage_w + par * STRIPT_STAGE, col_ok, want_stats, s1, s2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const u32x4 o0 = *reinterpret_cast<const u32x4*>(stage_r + par * STRIPT_STAGE);
        const u32x4 o1 = *reinterpret_cast<const u32x4*>(stage_r + par * STRIPT_STAGE + 1024);
        if (st_ok0) *reinterpret_cast<u32x4*>(yp) = o0;
        if (st_ok1) *reinterpret_cast<u32x4*>(yp + y8) = o1;
        yp += ystep;
    }
    if (want_stats) {
        const int slab = band * a.nstrips + strip, slabs = a.nbands * a.nstrips;
        float* dst = a.stats + (((size_t)n * slabs + slab) * 64) * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t1 = row16_sum_to_lane15(s1[r >> 1][r & 1]), t2 = row16_sum_to_lane15(s2[r >> 1][r & 1]);
            if (p == 15) {
                const int ch = nt * 16 + kg * 4 + r;
                dst[ch * 2] = t1;
                dst[ch * 2 + 1] = t2;
            }
        }
    }
}

// returns -1 when the launch is not this kernel's shape: a = the ConvArgs ctg_conv_igemm_classes built (4 classes)
static int launch_stript(const ConvArgs& a, hipStream_t st, int* tiles_out) {
    static const bool off = getenv("CTG_NO_STRIPT") != nullptr;      // A/B switch (scripts/ab.sh)
    if (off || a.ncls != 4 || a.Cin != 128 || a.Cout != 64 || a.os != 2 || a.is != 1 || a.bias != nullptr || a.act != ACT_NONE ||
        a.pad_mode != PAD_ZERO || a.Hs != a.Hi || a.Ws != a.Wi || a.Ho != 2 * a.Hi || a.Wo != 2 * a.Wi || (a.x_ld & 7) || (a.y_ld & 7))
        return -1;
    if ((long)a.B * a.Hi * a.Wi < (1L << 18) || a.Hi < 16 || a.Wi < 16) return -1;
    // the class / tap structure of ConvTranspose2d(k=3, s=2, p=1, output_padding=1): engine._convT_classes(3, 1)
    static const int want_n[4] = {1, 2, 2, 4}, want_oy[4] = {0, 0, 1, 1}, want_ox[4] = {0, 1, 0, 1};
    static const int want_t[9][3] = {{0, 0, 4}, {0, 1, 3}, {0, 0, 5}, {1, 0, 1}, {0, 0, 7}, {1, 1, 0}, {1, 0, 2}, {0, 1, 6}, {0, 0, 8}};
    int t = 0;
    for (int q = 0; q < 4; ++q) {
        if (a.c_ntaps[q] != want_n[q] || a.c_oy0[q] != want_oy[q] || a.c_ox0[q] != want_ox[q] || a.c_tap0[q] != t) return -1;
        for (int k = 0; k < want_n[q]; ++k, ++t) {
            const int tw = a.taps[t];
            if ((tw & 0xff) - 64 != want_t[t][0] || ((tw >> 8) & 0xff) - 64 != want_t[t][1] || (tw >> 16) != want_t[t][2]) return -1;
        }
    }
    StripTArgs s;
    s.x = (const bf16_t*)a.x; s.w = (const bf16_t*)a.w; s.y = (bf16_t*)a.y; s.stats = a.stats;
    s.B = a.B; s.Hi = a.Hi; s.Wi = a.Wi; s.x_ld = a.x_ld; s.y_ld = a.y_ld; s.w_tap_stride = a.w_tap_stride;
    s.nstrips = (a.Wi + 15) / 16;
    static const int xcd_env = getenv("CTG_STRIPT_XCD") ? atoi(getenv("CTG_STRIPT_XCD")) : 1;      // A/B knob
    s.xcd = xcd_env;
    const int n_cu = ctg_cu_count();
    static const int band_env = getenv("CTG_STRIPT_BAND") ? atoi(getenv("CTG_STRIPT_BAND")) : 0;      // A/B knob
    // two workgroups per CU are resident (registers): bands so that the grid fills the chip once (measured at B = 16, 256^2:
    // 1 / 2 / 4 / 8 bands per strip 240 / 205 / 222 / 232 us)
    long nb = (2L * n_cu) / ((long)a.B * s.nstrips);
    if (nb < 1) nb = 1;
    int band = (int)((a.Hi + nb - 1) / nb);
    if (band < 16) band = 16;
    if (band_env >= 8) band = band_env;      // (the caller sized the moments buffer for >= 8-row bands)
    s.band_rows = band;
    s.nbands = (a.Hi + band - 1) / band;
    if (tiles_out != nullptr) *tiles_out = s.nbands * s.nstrips;
    const int smem = STRIPT_SMEM;
    static unsigned long long attr_mask = 0;       // per device
    {
        const int rc = ctg_lds_attr_once((const void*)conv_stript_128_64_kernel, smem, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const dim3 grid((unsigned)((long)a.B * s.nbands * s.nstrips));
    hipLaunchKernelGGL(conv_stript_128_64_kernel, grid, dim3(256), smem, st, s);
    return ctg_launch_status();
}
