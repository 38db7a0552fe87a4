// Wave-autonomous sliding-window form of the stride-2 TRANSPOSED 3x3 convolution 128 -> 64 channels on large maps (gfx950, bf16):
// ConvTranspose2d(128, 64, 3, stride=2, padding=1, output_padding=1) of the generator's second up-sampling layer
// (Model/HdGan.py:93-95, [B,256,256,128] -> [B,512,512,64]) and, the same operator, the backward-data pass of its first
// down-sampling conv Conv2d(64, 128, 3, stride=2, padding=1) (:78-80).
//
// As four parity classes on conv_halo_kernel (one launch, ctg_conv_igemm_classes) these launches take 459 us for 155 GFLOP and
// 805 MB: 144 MFMAs per tile and wave inside ~2600 other instructions (profiles/r03_sq_table.md) -- bound by instruction issue at
// 340 TFLOP/s and 1.75 TB/s.  Same cure as conv_strip.h, adapted to the shape:
//   * a workgroup owns a 16-pixel-wide INPUT column strip (32 output columns) of one sample and slides down a band of input rows;
//     input row j and j+1 give output rows 2j and 2j+1, i.e. all four parity classes of the strip at once -- one input fetch
//     instead of four, no halo overlap between classes;
//   * its four WAVES split the 64 output channels (one 16-channel MFMA tile each), so the wave's share of the weights -- 9 taps x
//     4 k-steps = 36 A-fragments, 144 VGPRs -- stays in registers for the whole launch; the waves do not share LDS and never
//     meet at a barrier (each keeps its own ring of input rows: the input is fetched from L2 four times, from HBM once);
//   * one new input row (17 px x 256 B) per step by five LDS-DMA instructions, issued 2 rows ahead, retired by a counted
//     s_waitcnt; the 16-byte chunks of a pixel are XOR-swizzled by the pixel index, so the 256-byte pixel pitch does not put
//     the 16 pixels of a fragment read on one bank;
//   * per 64-channel half of K: 8 pixel fragments (2 rows x 2 columns x 2 k-steps), then the 9 (class, tap) pairs in the order
//     the class kernels use (tap, then k-step), four independent accumulation chains -- results are bit-identical to theirs;
//   * the lane that holds (pixel, 4 channels) of a class stores its 8 bytes; InstanceNorm moments accumulate over the band.
#pragma once
#include "conv_halo.h"

#define STRIPT_R 4              // ring rows per wave: two in use, two in flight
#define STRIPT_ROWB 4352        // 17 px x 256 B = 272 chunks: four full DMA instructions and one of 16 lanes; 4 waves x 4 rows = 68 KB,
                                // two workgroups per CU (the 236 + 20 registers allow two waves per SIMD as well)

struct StripTArgs {
    const bf16_t* x;            // [B][Hi][Wi][x_ld], 128 channels
    const bf16_t* w;            // packed [9][w_npad >= 64][128]
    bf16_t* y;                  // [B][2 Hi][2 Wi][y_ld], 64 channels
    float* stats;               // [B][slabs][64][2] or NULL
    int B, Hi, Wi, x_ld, y_ld, w_tap_stride;
    int band_rows, nbands, nstrips;
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// One step of a wave: input rows at LDS r0 (row j) and r1 (row j + 1) -> the four parity classes of output rows 2j, 2j + 1.
// MASKED: the strip reaches past the image's right edge (stores and moments of columns >= Wi are dropped).
template <bool MASKED>
__device__ __forceinline__ void stript_step(const u32x4 (&wf)[9][4], const char* r0, const char* r1, const int (&loff)[2][4], bf16_t* yp,
                                            size_t ypitch, int y_ld, bool col_ok, bool want_stats, f32x2_t (&s1)[2], f32x2_t (&s2)[2]) {
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
        if (!MASKED || col_ok) *reinterpret_cast<bf16x4*>(yp + (size_t)(q >> 1) * ypitch + (q & 1) * y_ld) = o;
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
    const int item = blockIdx.x;
    const int strip = item % a.nstrips;
    const int t2 = item / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    char* ring = smem + wave * (STRIPT_R * STRIPT_ROWB);
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
#pragma unroll
    for (int i = 0; i < STRIPT_R * STRIPT_ROWB / 1024; ++i) *reinterpret_cast<u32x4*>(ring + (i * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- per-lane byte offsets of an input row's 272 slots from the strip's first pixel of that row: slot (px, cs) holds source
    //      chunk cs ^ (px & 15) of column i0 + px
    unsigned voff[5];
    bool vok[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        const int s = lane + 64 * d;
        const int px = s >> 4, cs = s & 15;
        vok[d] = s < 272 && i0 + px < Wi;
        voff[d] = (unsigned)(px * x_ld + (cs ^ (px & 15)) * 8) * 2u;
    }
    const size_t rpitch = (size_t)Wi * x_ld * 2;                                                       // bytes per input row
    const char* __restrict__ X0 = reinterpret_cast<const char*>(a.x + (((size_t)n * Hi + jb) * Wi + i0) * x_ld);   // row jb, column i0
    // a row with every column inside the image: five DMA instructions off a uniform row pointer (the 17th pixel, 16 lanes, may be
    // the zero column right of the image -- the instruction count per row stays five, the wait counts below rely on it)
    const bool has17 = i0 + 16 < Wi;
    const unsigned voff4 = has17 ? voff[4] : 0u;
    auto issue_fast = [&](int slot, const char* rowp) __attribute__((always_inline)) {
        asm volatile("" : "+s"(rowp));          // opaque: keeps the address "uniform row pointer + 32-bit lane offset" (saddr form)
        char* dst = ring + slot * STRIPT_ROWB;
#pragma unroll
        for (int d = 0; d < 4; ++d) __builtin_amdgcn_global_load_lds((gptr_t)(rowp + voff[d]), (lptr_t)(dst + 1024 * d), 16, 0, 0);
        const char* r4 = has17 ? rowp : reinterpret_cast<const char*>(g_zero_chunk);
        if (lane < 16) __builtin_amdgcn_global_load_lds((gptr_t)(r4 + voff4), (lptr_t)(dst + 4096), 16, 0, 0);
    };
    // any row: columns past the edge are skipped, the row below the image is written as zeros
    auto issue_slow = [&](int k) __attribute__((always_inline)) {
        char* dst = ring + (k % STRIPT_R) * STRIPT_ROWB;
        const char* rowp = X0 + (size_t)k * rpitch;
        if (jb + k < Hi) {
#pragma unroll
            for (int d = 0; d < 5; ++d)
                if (vok[d]) __builtin_amdgcn_global_load_lds((gptr_t)(rowp + voff[d]), (lptr_t)(dst + 1024 * d), 16, 0, 0);
        } else {
#pragma unroll
            for (int d = 0; d < 5; ++d)
                if (d < 4 || lane < 16) *reinterpret_cast<u32x4*>(dst + (d * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    };
    // ---- fragment byte offsets inside a ring row: pixel p + dx, logical chunk kk*4 + kg, physical chunk XORed with the pixel
    int loff[2][4];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int px = p + dx;
            loff[dx][kk] = (px * 16 + ((kk * 4 + kg) ^ (px & 15))) * 16;
        }
    const bool col_ok = i0 + p < Wi;
    const bool full_strip = i0 + 16 <= Wi;
    const int Wo = 2 * Wi;
    // output pointer of class (py, px) at step 0: row 2 jb + py, column 2 (i0 + p) + px, channels nt*16 + kg*4 ..
    bf16_t* __restrict__ yp = a.y + (((size_t)n * 2 * Hi + 2 * jb) * Wo + 2 * (i0 + p)) * a.y_ld + nt * 16 + kg * 4;
    const size_t ypitch = (size_t)Wo * a.y_ld;
    f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    const bool want_stats = a.stats != nullptr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the zeroed ring, before any DMA lands in it

    int j = 0;
    if (full_strip && nrows >= 2 * STRIPT_R) {
        // ---- the fast path: every row it fetches is inside the image.  Step j fetches row j + R - 1 into the slot of row j - 1
        //      (last read one step ago), then retires row j + 1: newer than its DMA are the DMA of rows j+2 .. j+R-1 (5 each) and
        //      the stores of the last min(j, R-2) steps (4 each)
        const int nfast = nrows - (STRIPT_R - 1);            // steps j < nfast fetch a row < nrows
        const char* rowp = X0;
#pragma unroll
        for (int k = 0; k < STRIPT_R - 1; ++k) { issue_fast(k, rowp); rowp += rpitch; }
#pragma unroll
        for (int u = 0; u < STRIPT_R - 2; ++u) {             // the first R - 2 steps: fewer stores in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_fast((u + STRIPT_R - 1) % STRIPT_R, rowp);
            rowp += rpitch;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (STRIPT_R - 2) + 4 * u) : "memory");
            stript_step<false>(wf, ring + u * STRIPT_ROWB, ring + ((u + 1) % STRIPT_R) * STRIPT_ROWB, loff, yp, ypitch, a.y_ld, true,
                               want_stats, s1, s2);
            yp += 2 * ypitch;
        }
        for (j = STRIPT_R - 2; j + STRIPT_R <= nfast; j += STRIPT_R) {
#pragma unroll
            for (int u = 0; u < STRIPT_R; ++u) {
                const int slot = (STRIPT_R - 2 + u) % STRIPT_R;          // of row j + u
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_fast((slot + STRIPT_R - 1) % STRIPT_R, rowp);
                rowp += rpitch;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 * (STRIPT_R - 2)) : "memory");
                stript_step<false>(wf, ring + slot * STRIPT_ROWB, ring + ((slot + 1) % STRIPT_R) * STRIPT_ROWB, loff, yp, ypitch, a.y_ld,
                                   true, want_stats, s1, s2);
                yp += 2 * ypitch;
            }
        }
    } else {
        for (int k = 0; k < STRIPT_R - 1 && k < nin; ++k) issue_slow(k);
    }
    // ---- the remaining steps (the band's last rows; every step of a ragged strip or a short band): one row at a time
    for (; j < nrows; ++j) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j + STRIPT_R - 1 < nin) issue_slow(j + STRIPT_R - 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stript_step<true>(wf, ring + (j % STRIPT_R) * STRIPT_ROWB, ring + ((j + 1) % STRIPT_R) * STRIPT_ROWB, loff, yp, ypitch, a.y_ld,
                          col_ok, want_stats, s1, s2);
        yp += 2 * ypitch;
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
        a.pad_mode != PAD_ZERO || a.Hs != a.Hi || a.Ws != a.Wi || a.Ho != 2 * a.Hi || a.Wo != 2 * a.Wi || (a.x_ld & 7) || (a.y_ld & 3))
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
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    static const int band_env = getenv("CTG_STRIPT_BAND") ? atoi(getenv("CTG_STRIPT_BAND")) : 0;      // A/B knob
    // as few bands as fill the chip once (measured at B = 16, 256^2: 1 / 2 / 4 / 8 bands per strip 309 / 326 / 357 / 326 us)
    long nb = (long)n_cu / ((long)a.B * s.nstrips);
    if (nb < 1) nb = 1;
    int band = (int)((a.Hi + nb - 1) / nb);
    if (band < 16) band = 16;
    if (band_env >= 4) band = band_env;
    s.band_rows = band;
    s.nbands = (a.Hi + band - 1) / band;
    if (tiles_out != nullptr) *tiles_out = s.nbands * s.nstrips;
    const int smem = 4 * STRIPT_R * STRIPT_ROWB;
    static int attr_set = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_stript_128_64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return 1000 + (int)e;
        attr_set = 1;
    }
    const dim3 grid((unsigned)((long)a.B * s.nbands * s.nstrips));
    hipLaunchKernelGGL(conv_stript_128_64_kernel, grid, dim3(256), smem, st, s);
    return ctg_launch_status();
}
