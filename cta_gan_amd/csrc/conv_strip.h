// Wave-autonomous sliding-window 3x3 convolution for the 32 -> 32 channel layers on large maps (gfx950, bf16).
//
// Reg's full-resolution level (trainer/reg.py:42-48,65-70: ResnetBlock(32) at 512x512) moves 64 B in + 64 B out per pixel for
// 18 kFLOP: HBM-bound by 2x on paper, but conv_halo_kernel's tile machinery issues ~2000 instructions around the 86 MFMAs of a
// 16x16-pixel tile and wave (SQ counters, profiles/r03_sq_table.md): the launch is bound by instruction issue at 2.7-3.2 TB/s.
// A persistent variant of that machinery did not help (scripts/experiments/r03_persistent_narrow_conv.md).  This kernel drops
// the machinery instead:
//   * a WAVE owns a 16-pixel-wide column strip of one sample and slides down a band of rows; nothing is shared between
//     waves, so the kernel has NO workgroup barrier and no staging;
//   * all weights (9 taps x 32 x 32) live in the wave's registers (18 MFMA A-fragments = 72 VGPRs) for the whole launch;
//   * every new output row costs ONE new input row: 18 px x 64 B by two LDS-DMA instructions into the wave's private ring of 9
//     rows, issued 6 rows ahead and retired with a counted s_waitcnt (vmcnt is a FIFO over DMA and stores);
//   * reflection / zero padding of the columns is a per-lane source offset computed once per strip, of the rows a scalar;
//   * the 9 pixel fragments of an output row stay in registers across rows (consecutive rows share two of their three input
//     rows): three conflict-free ds_read_b128 at immediate offsets per row (the row loop is unrolled over the ring), issued one
//     row ahead; 18 MFMAs accumulate the row, and the lane that holds (pixel, 4 channels) stores its 8 bytes directly;
//   * InstanceNorm moments accumulate in registers over the band and leave once per wave.
// ~110 instructions per 16-pixel row and wave instead of ~470 (1870 per 4-row wave tile): measured 172 -> 122 us for the
// [16,512,512,32] forward launch (4.4 TB/s algorithmic; compute skeleton alone 68 us, loads and stores alone 78 / 85 us).
// Taps accumulate in the caller's list order (forward: dy ascending; backward-data: flipped), one k-step per tap, weights as the
// MFMA's A operand: results are bit-identical to conv_halo_kernel's.
#pragma once
#include "conv_halo.h"

#define STRIP_R 9           // ring rows per wave (a multiple of 3: the row loop is unrolled over the ring and the register rows)
#define STRIP_ROWB 1152     // 18 px x 64 B

struct StripArgs {
    const bf16_t* x;        // [B][H][W][x_ld]
    const bf16_t* w;        // packed [tap slice][w_npad][32]
    bf16_t* y;              // [B][H][W][y_ld]
    const float* bias;      // [32] or NULL
    float* stats;           // [B][slabs][32][2] or NULL
    int B, H, W, x_ld, y_ld, w_tap_stride;
    int pad_mode, act;
    int band_rows, nbands, nstrips;
    int widx[9];            // weight slice of tap t (list order)
    int x_lo, y_lo;         // split-pair kernel: element offsets of the lo planes of x / y (their pitch / 2)
};

// FLIP = false: tap t is (dy, dx) = (t / 3 - 1, t % 3 - 1) (forward); true: (1 - t / 3, 1 - t % 3) (backward-data of the same conv)
// EPI = false: no bias and no activation (an InstanceNorm follows: the common case) -- the row epilogue is convert + store
// XOR swizzle of the four 16-byte chunks of a 64-byte pixel in a ring row.  ds_read_b128 serves a wave in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, (+32) (MI355X_MICROARCH.md, LDS): with c ^ ((px >> 1) & 3) the fragment reads of the
// pixels p, p + 1 and p + 2 are all conflict-free; conv_halo's swz<4> (tiles read at one pixel offset only) leaves two of the
// three 2-way conflicted.
__device__ __forceinline__ int strip_swz(int px, int c) { return c ^ ((px >> 1) & 3); }

template <bool FLIP, bool EPI>
__global__ __launch_bounds__(256, 2) void conv_strip32_kernel(const StripArgs a) {
    __shared__ __attribute__((aligned(16))) char ring_all[4][STRIP_R][STRIP_ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const int item = blockIdx.x * 4 + wave;
    const int strip = item % a.nstrips;
    const int t2 = item / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    if (n >= a.B) return;                               // (wave-uniform; no barrier in this kernel)
    char* ring = ring_all[wave][0];
    const int x0 = strip * 16, yb = band * a.band_rows;
    const int nrows = min(a.band_rows, a.H - yb);       // output rows of this band
    const int nin = nrows + 2;                          // input rows yb-1 .. yb+nrows
    const int H = a.H, W = a.W, x_ld = a.x_ld;
    const bf16_t* __restrict__ Xn = a.x + (size_t)n * H * W * x_ld;

    // ---- weights: A fragment of (tap t, n-tile nt) = W[widx(t)][nt*16 + (lane & 15)][(lane >> 4) * 8 ..], resident in registers
    u32x4 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            wf[t][nt] = *reinterpret_cast<const u32x4*>(a.w + (size_t)a.widx[t] * a.w_tap_stride + (nt * 16 + p) * 32 + kg * 8);

    // ---- per-lane source offsets of a halo row's 72 slots: slot s = (px, c) holds source chunk swz(px, c) of image column
    // x0 - 1 + px; -1 = zero page (zero padding, or a column past a ragged strip's reflection range)
    int coloff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s = lane + 64 * j;
        const int px = s >> 2, c = s & 3;
        int col = x0 - 1 + px;
        bool ok = s < 72;
        if (a.pad_mode == PAD_REFLECT) {
            col = reflect_idx(col, W);
            ok = ok && (unsigned)col < (unsigned)W;
        } else {
            ok = ok && (unsigned)col < (unsigned)W;
        }
        coloff[j] = ok ? col * x_ld + strip_swz(px, c) * 8 : -1;
    }
    auto issue_row = [&](int j) __attribute__((always_inline)) {     // input row index j (image row yb - 1 + j) -> ring slot j % R
        int row = yb - 1 + j;
        bool rok = true;
        if (a.pad_mode == PAD_REFLECT) row = reflect_idx(row, H);
        rok = (unsigned)row < (unsigned)H;
        const bf16_t* xr = Xn + (size_t)(rok ? row : 0) * W * x_ld;
        char* dst = ring + (j % STRIP_R) * STRIP_ROWB;
        const bf16_t* s0 = (rok && coloff[0] >= 0) ? xr + coloff[0] : (const bf16_t*)g_zero_chunk;
        __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)dst, 16, 0, 0);
        if (lane < 8) {
            const bf16_t* s1 = (rok && coloff[1] >= 0) ? xr + coloff[1] : (const bf16_t*)g_zero_chunk;
            __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(dst + 1024), 16, 0, 0);
        }
    };

    // ---- fragment byte offsets inside a ring row for dx = -1, 0, +1 (px = p + dx + 1), output offsets, bias
    int loff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int px = p + k;
        loff[k] = (px * 4 + strip_swz(px, kg)) * 16;
    }
    const bool col_ok = x0 + p < W;
    float bv[EPI ? 2 : 1][4];
    if constexpr (EPI) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[nt][r] = a.bias != nullptr ? a.bias[nt * 16 + kg * 4 + r] : 0.f;
    }
    const bool full_strip = x0 + 16 <= W;               // (wave-uniform) no masked pixel columns
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    bf16_t* __restrict__ yrow = a.y + (((size_t)n * H + yb) * W + x0 + p) * a.y_ld + kg * 4;
    const size_t ypitch = (size_t)W * a.y_ld;
    const bool want_stats = a.stats != nullptr;
    const float neg_slope = a.act == ACT_NONE ? 1.f : (a.act == ACT_RELU ? 0.f : LRELU_SLOPE);      // (no tanh / sigmoid here: host-checked)

    // ---- the 9 pixel fragments of a row's three input rows live in registers: fr[j % 3][kx] = fragment of input row j shifted
    // by kx.  Consecutive output rows share two of their three input rows, so a new output row costs three new fragment reads
    // (issued one row ahead, behind the MFMAs that still use the slot they overwrite), not nine.
    u32x4 fr[3][3];
    auto read_row = [&](int slot, int rs) __attribute__((always_inline)) {      // ring slot (compile-time) -> register row rs
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            fr[rs][kx] = *reinterpret_cast<const u32x4*>(ring + slot * STRIP_ROWB + loff[kx]);
    };
    // ---- prologue: the whole ring in flight; rows 0..2 retired and read
#pragma unroll
    for (int j = 0; j < STRIP_R; ++j)
        if (j < nin) issue_row(j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (STRIP_R - 3)) : "memory");     // rows 0, 1, 2 (and the weight / bias loads)
    read_row(0, 0);
    read_row(1, 1);
    read_row(2, 2);

    for (int i0 = 0; i0 < nrows; i0 += STRIP_R) {
#pragma unroll
        for (int u = 0; u < STRIP_R; ++u) {
            const int i = i0 + u;                        // output row of the band; its input rows i, i+1, i+2 are in fr[]
            if (i < nrows) {
                // input row i + R goes where row i was (its fragments left LDS three rows ago); then retire row i + 3: newer
                // than its DMA are the DMA of rows i+4 .. i+R (2 each) and the stores of the last min(i, R-3) rows (2 each)
                if (i + STRIP_R < nin) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue_row(i + STRIP_R);
                    if (i < STRIP_R - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (STRIP_R - 3)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (STRIP_R - 3)) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // band tail: nothing left to overlap with
                }
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = FLIP ? 2 - t / 3 : t / 3, kx = FLIP ? 2 - t % 3 : t % 3;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t][nt]),
                                                                          __builtin_bit_cast(bf16x8, fr[(u + ky) % 3][kx]),
                                                                          acc[nt], 0, 0, 0);
                }
                if (i + 1 < nrows) read_row((u + 3) % STRIP_R, u % 3);      // input row i + 3, for the next output row
                bf16_t* yp = yrow + (size_t)i * ypitch;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[nt][r];
                        if constexpr (EPI) {
                            v += bv[nt][r];
                            v = v > 0.f ? v : v * neg_slope;
                        }
                        o[r] = (bf16_t)v;
                    }
                    if (col_ok) *reinterpret_cast<bf16x4*>(yp + nt * 16) = o;
                }
                if (want_stats) {       // (wave-uniform branches; EPI launches never ask for moments)
                    if (full_strip) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { s1[nt][r] += acc[nt][r]; s2[nt][r] += acc[nt][r] * acc[nt][r]; }
                    } else {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = col_ok ? acc[nt][r] : 0.f;
                                s1[nt][r] += v;
                                s2[nt][r] += v * v;
                            }
                    }
                }
            }
        }
    }
    if (want_stats) {
        // per-wave partial moments: 16-lane rows (the strip's pixels) by DPP, lane 15 of each row holds 4 channels x 2 n-tiles
        const int slab = band * a.nstrips + strip, slabs = a.nbands * a.nstrips;
        float* dst = a.stats + (((size_t)n * slabs + slab) * 32) * 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t1 = row16_sum_to_lane15(s1[nt][r]), t2 = row16_sum_to_lane15(s2[nt][r]);
                if (p == 15) {
                    const int c = nt * 16 + kg * 4 + r;
                    dst[c * 2] = t1;
                    dst[c * 2 + 1] = t2;
                }
            }
    }
}

// ===========================================================================
// The same layer in the split-bf16 ("bf16x3") mode: split-pair input and output, x_hi.w_hi + x_hi.w_lo + x_lo.w_hi.
//   * a ring row is 18 px x 128 B ([hi 4 chunks | lo 4 chunks] per pixel, XOR-swizzled by px & 7), six rows per wave
//     (55 KB per workgroup: two workgroups per CU), three LDS-DMA instructions per row, issued five rows ahead;
//   * both weight halves stay in registers (36 A fragments = 144 VGPRs); to make room the loop is INPUT-stationary: the six
//     fragments of one input row (3 shifts x hi / lo) feed the three output rows it belongs to (54 MFMAs into three live
//     accumulator rows), then the oldest output row is complete and leaves as hi / lo 8-byte pieces;
//   * the same counted vmcnt FIFO discipline (per step: 3 DMA + 4 stores), no workgroup barrier.
// ===========================================================================
#define STRIPP_R 6
#define STRIPP_ROWB 2304     // 18 px x 128 B

template <bool FLIP, bool EPI>
__global__ __launch_bounds__(256, 2) void conv_strip32p_kernel(const StripArgs a) {
    __shared__ __attribute__((aligned(16))) char ring_all[4][STRIPP_R][STRIPP_ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const int item = blockIdx.x * 4 + wave;
    const int strip = item % a.nstrips;
    const int t2 = item / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    if (n >= a.B) return;                               // (wave-uniform; no barrier in this kernel)
    char* ring = ring_all[wave][0];
    const int x0 = strip * 16, yb = band * a.band_rows;
    const int nrows = min(a.band_rows, a.H - yb);       // output rows of this band
    const int nin = nrows + 2;                          // input rows yb-1 .. yb+nrows
    const int H = a.H, W = a.W, x_ld = a.x_ld;
    const bf16_t* __restrict__ Xn = a.x + (size_t)n * H * W * x_ld;

    // ---- weights: rows of [w_hi 32 | w_lo 32]; A fragments of (tap t, n-tile nt), hi and lo halves, resident in registers
    u32x4 wh[9][2], wl[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bf16_t* wp = a.w + (size_t)a.widx[t] * a.w_tap_stride + (nt * 16 + p) * 64 + kg * 8;
            wh[t][nt] = *reinterpret_cast<const u32x4*>(wp);
            wl[t][nt] = *reinterpret_cast<const u32x4*>(wp + 32);
        }

    // ---- per-lane source offsets of a ring row's 144 slots: slot s = (px, c) holds source chunk c ^ (px & 7) of image column
    // x0 - 1 + px (chunks 0-3: hi plane, 4-7: lo plane); -1 = zero page
    int coloff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int s = lane + 64 * j;
        const int px = s >> 3, c = (s & 7) ^ (px & 7);
        int col = x0 - 1 + px;
        bool ok = s < 144;
        if (a.pad_mode == PAD_REFLECT) col = reflect_idx(col, W);
        ok = ok && (unsigned)col < (unsigned)W;
        coloff[j] = ok ? col * x_ld + (c & 3) * 8 + (c >> 2) * a.x_lo : -1;
    }
    auto issue_row = [&](int j) __attribute__((always_inline)) {     // input row index j (image row yb - 1 + j) -> ring slot j % R
        int row = yb - 1 + j;
        if (a.pad_mode == PAD_REFLECT) row = reflect_idx(row, H);
        const bool rok = (unsigned)row < (unsigned)H;
        const bf16_t* xr = Xn + (size_t)(rok ? row : 0) * W * x_ld;
        char* dst = ring + (j % STRIPP_R) * STRIPP_ROWB;
        const bf16_t* s0 = (rok && coloff[0] >= 0) ? xr + coloff[0] : (const bf16_t*)g_zero_chunk;
        __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)dst, 16, 0, 0);
        const bf16_t* s1 = (rok && coloff[1] >= 0) ? xr + coloff[1] : (const bf16_t*)g_zero_chunk;
        __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(dst + 1024), 16, 0, 0);
        if (lane < 16) {
            const bf16_t* s2 = (rok && coloff[2] >= 0) ? xr + coloff[2] : (const bf16_t*)g_zero_chunk;
            __builtin_amdgcn_global_load_lds((gptr_t)s2, (lptr_t)(dst + 2048), 16, 0, 0);
        }
    };

    // ---- fragment byte offsets inside a ring row for the three column shifts (px = p + kx), hi and lo halves
    int loh[3], lol[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int px = p + k;
        loh[k] = (px * 8 + (kg ^ (px & 7))) * 16;
        lol[k] = (px * 8 + ((kg + 4) ^ (px & 7))) * 16;
    }
    const bool col_ok = x0 + p < W;
    float bv[EPI ? 2 : 1][4];
    if constexpr (EPI) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[nt][r] = a.bias != nullptr ? a.bias[nt * 16 + kg * 4 + r] : 0.f;
    }
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nt][r] = 0.f; s2[nt][r] = 0.f; }
    bf16_t* __restrict__ yrow = a.y + (((size_t)n * H + yb) * W + x0 + p) * a.y_ld + kg * 4;
    const size_t ypitch = (size_t)W * a.y_ld;
    const bool want_stats = a.stats != nullptr;
    const float neg_slope = a.act == ACT_NONE ? 1.f : (a.act == ACT_RELU ? 0.f : LRELU_SLOPE);

    f32x4 acc[3][2];        // output rows j, j-1, j-2 (index = output row % 3) while input row j is swept
#pragma unroll
    for (int q = 0; q < 3; ++q) { acc[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- prologue: the whole ring in flight
#pragma unroll
    for (int j = 0; j < STRIPP_R; ++j)
        if (j < nin) issue_row(j);

    for (int j0 = 0; j0 < nin; j0 += STRIPP_R) {
#pragma unroll
        for (int u = 0; u < STRIPP_R; ++u) {
            const int j = j0 + u;                        // input row of the band
            if (j < nin) {
                // retire the DMA of row j: newer than it are (first ring pass) the prologue's later rows and the steps so far,
                // (later) the stores of the step that issued it and five whole steps of 3 DMA + 4 stores; band tail: everything
                if (j + STRIPP_R <= nin) {                   // every step between row j's DMA and now issued its row
                    if (j0 == 0) {
                        if (u < 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                        else if (u == 3) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
                        else if (u == 4) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // band tail: not every later step issues a row
                }
                u32x4 fh[3], fl[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    fh[kx] = *reinterpret_cast<const u32x4*>(ring + u * STRIPP_ROWB + loh[kx]);
                    fl[kx] = *reinterpret_cast<const u32x4*>(ring + u * STRIPP_ROWB + lol[kx]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (j + STRIPP_R < nin) issue_row(j + STRIPP_R);          // row j's fragments are in registers: its slot is free
                // input row j is row r of the window of output row j - r
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (j - r >= 0 && j - r < nrows) {                   // (wave-uniform)
                        const int q = (u + 3 - r) % 3;                   // (j - r) % 3: STRIPP_R is a multiple of 3
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int t = FLIP ? (2 - r) * 3 + (2 - kx) : r * 3 + kx;
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[t][nt]),
                                                                                 __builtin_bit_cast(bf16x8, fh[kx]), acc[q][nt], 0, 0, 0);
                                acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl[t][nt]),
                                                                                 __builtin_bit_cast(bf16x8, fh[kx]), acc[q][nt], 0, 0, 0);
                                acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[t][nt]),
                                                                                 __builtin_bit_cast(bf16x8, fl[kx]), acc[q][nt], 0, 0, 0);
                            }
                        }
                    }
                }
                if (j >= 2) {                                            // output row i = j - 2 is complete
                    const int i = j - 2;
                    const int q = (u + 1) % 3;                           // (j - 2) % 3
                    bf16_t* yp = yrow + (size_t)i * ypitch;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        bf16x4 hi, lo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[q][nt][r];
                            if constexpr (EPI) {
                                v += bv[nt][r];
                                v = v > 0.f ? v : v * neg_slope;
                            }
                            hi[r] = (bf16_t)v;
                            lo[r] = (bf16_t)(v - (float)hi[r]);
                        }
                        if (col_ok) {
                            *reinterpret_cast<bf16x4*>(yp + nt * 16) = hi;
                            *reinterpret_cast<bf16x4*>(yp + nt * 16 + a.y_lo) = lo;
                        }
                        if (want_stats) {       // (EPI launches never ask for moments)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = col_ok ? acc[q][nt][r] : 0.f;
                                s1[nt][r] += v;
                                s2[nt][r] += v * v;
                            }
                        }
                        acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
    }
    if (want_stats) {
        const int slab = band * a.nstrips + strip, slabs = a.nbands * a.nstrips;
        float* dst = a.stats + (((size_t)n * slabs + slab) * 32) * 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t1 = row16_sum_to_lane15(s1[nt][r]), t2 = row16_sum_to_lane15(s2[nt][r]);
                if (p == 15) {
                    const int c = nt * 16 + kg * 4 + r;
                    dst[c * 2] = t1;
                    dst[c * 2 + 1] = t2;
                }
            }
    }
}

// returns -1 when the shape is not served here
static int launch_strip32(const ConvArgs& a, hipStream_t st, int* tiles_out, bool pair) {
    static const bool off = getenv("CTG_NO_STRIP") != nullptr;       // A/B switch (scripts/ab.sh)
    if (off || a.Cin != (pair ? 64 : 32) || a.Cout != 32 || a.kh != 3 || a.kw != 3 || a.ntaps != 9 || a.is != 1 || a.os != 1 || a.ncls > 1 ||
        a.oy0 != 0 || a.ox0 != 0 || a.Ho != a.Hs || a.Wo != a.Ws || a.Hi != a.Hs || a.Wi != a.Ws || a.res != nullptr ||
        a.fold != nullptr || a.dy0 != -1 || a.dx0 != -1 || a.act > ACT_LRELU)
        return -1;
    if ((long)a.B * a.Hs * a.Ws < (1L << 20) || a.Hs < 32 || a.Ws < 32 || (a.y_ld & 3) || (a.x_ld & 7)) return -1;   // large maps only
    if (pair && (long)a.Hs * a.Ws * a.x_ld >= (1L << 31)) return -1;
    // tap order: forward (dy ascending, dx fastest) or flipped
    bool fwd = true, flip = true;
    for (int t = 0; t < 9; ++t) {
        const int dy = (a.taps[t] & 0xff) - 64, dx = ((a.taps[t] >> 8) & 0xff) - 64;
        fwd = fwd && dy == t / 3 - 1 && dx == t % 3 - 1;
        flip = flip && dy == 1 - t / 3 && dx == 1 - t % 3;
    }
    if (!fwd && !flip) return -1;
    StripArgs s;
    s.x = (const bf16_t*)a.x; s.w = (const bf16_t*)a.w; s.y = (bf16_t*)a.y; s.bias = a.bias; s.stats = a.stats;
    s.B = a.B; s.H = a.Hs; s.W = a.Ws; s.x_ld = a.x_ld; s.y_ld = a.y_ld; s.w_tap_stride = a.w_tap_stride;
    s.pad_mode = a.pad_mode; s.act = a.act;
    s.x_lo = a.pair_lo; s.y_lo = a.y_ld / 2;
    s.nstrips = (a.Ws + 15) / 16;
    // bands: one band per wave; as many waves as the chip holds at three per SIMD (one dispatch round, no tail), bands >= 32 rows
    static const int band_env = getenv("CTG_STRIP_BAND") ? atoi(getenv("CTG_STRIP_BAND")) : 0;      // A/B knob
    const int n_cu = ctg_cu_count();
    const long cap = (long)n_cu * 4 * (pair ? 2 : 3);       // (split-pair: two waves per SIMD)
    long nb = cap / ((long)a.B * s.nstrips);
    if (nb < 1) nb = 1;
    int band = (int)((a.Hs + nb - 1) / nb);
    if (band < 32) band = 32;
    if (band_env >= 8) band = band_env;
    s.band_rows = band;
    s.nbands = (a.Hs + band - 1) / band;
    for (int t = 0; t < 9; ++t) s.widx[t] = a.taps[t] >> 16;
    const long waves = (long)a.B * s.nbands * s.nstrips;
    if (tiles_out != nullptr) *tiles_out = s.nbands * s.nstrips;
    const dim3 grid((unsigned)((waves + 3) / 4));
    const bool epi = a.bias != nullptr || a.act != ACT_NONE;
    if (epi && a.stats != nullptr) return -1;
    if (pair) {
        if (fwd) {
            if (epi) hipLaunchKernelGGL((conv_strip32p_kernel<false, true>), grid, dim3(256), 0, st, s);
            else hipLaunchKernelGGL((conv_strip32p_kernel<false, false>), grid, dim3(256), 0, st, s);
        } else {
            if (epi) hipLaunchKernelGGL((conv_strip32p_kernel<true, true>), grid, dim3(256), 0, st, s);
            else hipLaunchKernelGGL((conv_strip32p_kernel<true, false>), grid, dim3(256), 0, st, s);
        }
        return ctg_launch_status();
    }
    if (fwd) {
        if (epi) hipLaunchKernelGGL((conv_strip32_kernel<false, true>), grid, dim3(256), 0, st, s);
        else hipLaunchKernelGGL((conv_strip32_kernel<false, false>), grid, dim3(256), 0, st, s);
    } else {
        if (epi) hipLaunchKernelGGL((conv_strip32_kernel<true, true>), grid, dim3(256), 0, st, s);
        else hipLaunchKernelGGL((conv_strip32_kernel<true, false>), grid, dim3(256), 0, st, s);
    }
    return ctg_launch_status();
}
