// Split-pair ("bf16x3") form of conv_strips2.h: the STRIDE-2 3x3 convolution 64 -> 128 channels on large maps with split-pair input,
// weights and output -- Conv2d(64, 128, 3, stride=2, padding=1) of the generator's first down-sampling layer (Model/HdGan.py:78-80,
// [B,512,512,64] -> [B,256,256,128]) and, the same operator, the backward-data pass of its second up-sampling layer
// ConvTranspose2d(128, 64, 3, stride=2, ...) (:93-95), with products x_hi.w_hi + x_hi.w_lo + x_lo.w_hi.
//
// As polyphase stride-1 slices on conv_halo_kernel<PK, S2D> these launches take 520 us at B = 16 (464 GFLOP of executed MFMA work,
// 1611 MB).  The design is conv_strips2.h's with the split-pair operands of conv_striptp.h:
//   * a workgroup owns a strip of 16 OUTPUT columns (33 input columns) of one sample and slides down a band of output rows; each
//     step fetches one PAIR of input rows (2j, 2j+1) and keeps row 2j-1 from the pair before;
//   * EIGHT waves, one 16-channel MFMA tile each: the wave's share of the split weights -- 9 taps x 2 slices x [w_hi | w_lo] = 36
//     A-fragments, 144 VGPRs -- stays in registers for the whole launch;
//   * the pairs go through one ring of 6 pair slots in LDS (2 x 33 px x 256 B each: per pixel two 32-channel slices [hi 32 | lo 32]),
//     fetched 4 steps ahead by LDS-DMA, three instructions per wave and step, always; a row is stored DE-INTERLEAVED (even input
//     columns, then odd ones) so the 16 pixels 2p + kx of a fragment read are 16 consecutive 256-byte slots, their chunks XORed
//     with 2 (slot & 7): conflict-free under ds_read_b128's lane groups at every slot offset;
//   * per step and wave 36 pixel fragments and 54 MFMAs in ONE accumulation chain, in the order conv_halo_kernel<PK, S2D> walks
//     its (phase, slice, tap) loop -- bit-identical results;
//   * ONE barrier per step: the waves split their fp32 accumulators into the hi / lo planes of a staging tile (double buffered);
//     after the barrier wave s stores four whole 256-byte pixel rows of plane s / 4;
//   * InstanceNorm moments accumulate from the fp32 accumulators over the band.
#pragma once
#include "conv_halo.h"

#define STRIPSP_R 6               // ring slots (input row pairs) of the workgroup
#define STRIPSP_D 4               // a pair is fetched this many steps before the step that reads it as "its" pair
#define STRIPSP_ROWB 8448         // 33 px x 256 B
#define STRIPSP_PAIRB (2 * STRIPSP_ROWB)
#define STRIPSP_STAGE 8192        // one step's output tile: 2 planes x 16 px x 128 channels, bf16
#define STRIPSP_SMEM (STRIPSP_R * STRIPSP_PAIRB + 2 * STRIPSP_STAGE)       // 117760 B

struct StripS2PArgs {
    const bf16_t* x;            // [B][2 Ho][2 Wo][x_ld] split pair, 64 channels (lo plane x_lo elements behind)
    const bf16_t* w;            // split pack [9][w_npad >= 128][128]: per 32 channels [w_hi 32 | w_lo 32]
    bf16_t* y;                  // [B][Ho][Wo][y_ld] split pair, 128 channels (lo plane y_lo elements behind)
    float* stats;               // [B][slabs][128][2] or NULL
    int B, Ho, Wo, x_ld, y_ld, x_lo, y_lo, w_tap_stride;
    int band_rows, nbands, nstrips;
};

typedef float f32x2sp_t __attribute__((ext_vector_type(2)));
// LDS-space pointers for the ring and the staging tile (through generic pointers the compiler keeps "base + offset" sums in VGPRs)
typedef const __attribute__((address_space(3))) char* ldsp_cptr_t;
typedef __attribute__((address_space(3))) char* ldsp_ptr_t;
typedef const __attribute__((address_space(3))) u32x4* ldsp_c4_t;
// (an opaque XOR: as plain C the loop-invariant fragment offsets are hoisted into registers the weights need)
__device__ __forceinline__ int stripsp_xor(int v, int k) {
    int r;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(v));
    return r;
}

// LDS slot of input column q (0 .. 32 <-> image column 2 ox0 - 1 + q) inside a ring row: even q first, then odd q
__device__ __forceinline__ int stripsp_pos(int q) { return (q & 1) ? 17 + (q >> 1) : (q >> 1); }

__global__ __launch_bounds__(512, 1) void conv_strips2p_64_128_kernel(const StripS2PArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const int item = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
    const int strip = item % a.nstrips;
    const int t2 = item / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    char* ring = smem;
    const int ox0 = strip * 16, oyb = band * a.band_rows;
    const int nrows = min(a.band_rows, a.Ho - oyb);      // output rows (steps) of this band
    const int Hi = 2 * a.Ho, Wi = 2 * a.Wo, x_ld = a.x_ld;

    // ---- weights of (tap t, slice c, half): rows 16 wave + (lane & 15), channels c*32 + (lane >> 4)*8 .. of the half
    u32x4 wf[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bf16_t* wr = a.w + (size_t)t * a.w_tap_stride + (16 * wave + p) * 128 + kg * 8;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            wf[t][c][0] = *reinterpret_cast<const u32x4*>(wr + c * 64);
            wf[t][c][1] = *reinterpret_cast<const u32x4*>(wr + c * 64 + 32);
        }
    }

    // ---- the ring starts as zeros: the slot of the column left of the image, and the pair above the image, are never written
    for (int i = tid; i < STRIPSP_R * STRIPSP_PAIRB / 16; i += 512) *reinterpret_cast<u32x4*>(ring + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- a pair's 1056 chunks as one linear array [row][slot][chunk]: DMA instruction i covers chunks 64 i .. 64 i + 63; wave w
    //      issues i = 2w and 2w + 1, and every wave the last 32 (eight copies of the same 512 bytes: the instruction count per wave
    //      stays uniform).  LDS chunk (row, slot, cs) holds the logical chunk L = cs ^ 2 (slot & 7) of the slot's column;
    //      L = slice * 8 + plane * 4 + k-group.
    unsigned voff[3];
    bool vok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int Li = d < 2 ? 128 * wave + 64 * d + lane : 1024 + (lane & 31);
        const int row = Li / 528, rem = Li % 528, slot = rem >> 4, cs = rem & 15;
        const int q = slot < 17 ? 2 * slot : 2 * (slot - 17) + 1;
        const int cx = 2 * ox0 - 1 + q;
        const int L = cs ^ ((slot & 7) * 2);
        vok[d] = cx >= 0 && cx < Wi;
        voff[d] = (unsigned)((row * Wi + q) * x_ld + (L >> 3) * 32 + (L & 3) * 8 + ((L >> 2) & 1) * a.x_lo) * 2u;
    }
    const size_t ppitch = 2 * (size_t)Wi * x_ld * 2;                                                    // bytes per input row pair
    // pair 0 of the band: input row 2 oyb, column 2 ox0 - 1 (a pointer only; the column left of the image is never dereferenced)
    const char* __restrict__ X0 = reinterpret_cast<const char*>(a.x + (((size_t)n * Hi + 2 * oyb) * Wi + 2 * ox0 - 1) * (long)x_ld);
    auto issue = [&](int slot, const char* pp, bool pv) __attribute__((always_inline)) {   // pv (uniform): the pair exists
        const char* r = pv ? pp : reinterpret_cast<const char*>(g_zero_chunk);
        asm volatile("" : "+s"(r));             // opaque: keeps the addresses "uniform pair pointer + lane offset"
        char* dst = ring + slot * STRIPSP_PAIRB;
        if (vok[0]) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[0] : 0u)), (lptr_t)(dst + 2048 * wave), 16, 0, 0);
        if (vok[1]) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[1] : 0u)), (lptr_t)(dst + 2048 * wave + 1024), 16, 0, 0);
        if (lane < 32) __builtin_amdgcn_global_load_lds((gptr_t)(r + (pv ? voff[2] : 0u)), (lptr_t)(dst + 16384), 16, 0, 0);
    };
    // ---- fragment byte offsets inside a ring row for kx = 0, 1, 2: input column q = 2p + kx, logical chunk kg (slice 0, hi plane);
    //      slice 1 is ^ 128, the lo plane ^ 64 (the swizzle 2 (slot & 7) touches bits 1-3 of the chunk index, the XORs bits 2 and 3:
    //      XOR is bitwise, so they commute)
    int loff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int slot = stripsp_pos(2 * p + kx);
        loff[kx] = slot * 256 + ((kg ^ ((slot & 7) * 2)) * 16);
    }
    // ---- the staging tile [plane][16 px][128 ch] (256-byte pixels, their 16-byte chunks XORed with the pixel): this lane writes its
    //      4 channels of pixel p to both planes; after the barrier wave s stores pixels 4 (s % 4) .. + 3 of plane s / 4 whole: lane ->
    //      pixel 4 (s % 4) + lane/16, chunk lane%16
    const ldsp_cptr_t ringl = (ldsp_cptr_t)smem;                                 // the same memory through LDS-space pointers
    const ldsp_ptr_t stagel = (ldsp_ptr_t)smem + STRIPSP_R * STRIPSP_PAIRB;
    const ldsp_ptr_t stage_w = stagel + p * 256 + (((wave * 2 + (kg >> 1)) ^ p) * 16) + (kg & 1) * 8;
    const int spl = wave >> 2, spx = 4 * (wave & 3) + (lane >> 4), sch = lane & 15;
    const ldsp_cptr_t stage_r = stagel + spl * 4096 + spx * 256 + ((sch ^ spx) * 16);
    bf16_t* __restrict__ yp = a.y + (((size_t)n * a.Ho + oyb) * a.Wo + ox0 + spx) * a.y_ld + sch * 8 + spl * a.y_lo;
    const size_t ystep = (size_t)a.Wo * a.y_ld;
    f32x2sp_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    u32x4 F[8];                                              // the ring of four fragment pairs
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
        if (oyb > 0) issue(STRIPSP_R - 1, X0 - ppitch, true);          // pair -1: its row B is input row 2 oyb - 1
#pragma unroll
        for (int k = 0; k < STRIPSP_D; ++k) { issue(k, pp, k < nrows); pp += ppitch; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (STRIPSP_D - 1)) : "memory");        // pairs -1 and 0 (and the weights)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // One (tap, slice) item: hi.w_hi, hi.w_lo, lo.w_hi -- conv_halo_kernel<PK>'s order -- on the pair of fragments F[I], F[I + 1].
        // The 18 items of a step run through a ring of four pairs: an item's fragments are requested three items (nine MFMAs)
        // before its MFMAs, into the registers of the item just finished; the first four right behind the barrier of the step
        // before (they lie in input row 2j: visible then), under the staging-tile store and the row-pair fetch.  Left to itself the
        // compiler sinks every read in front of its first use: 36 exposed LDS latencies per step.
#define STRIPSP_TAP(T, C, I)                                                                                                     \
    {                                                                                                                            \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][C][0]), __builtin_bit_cast(bf16x8, F[I]), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][C][1]), __builtin_bit_cast(bf16x8, F[I]), acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][C][0]), __builtin_bit_cast(bf16x8, F[(I) + 1]), acc, 0, 0, 0); \
    }
#define STRIPSP_LDP(I, ROWP, KX, C)                                                                                              \
    {                                                                                                                            \
        const int o_ = (C) ? stripsp_xor(loff[KX], 128) : loff[KX];                                                              \
        F[(I)] = *(ldsp_c4_t)((ROWP) + o_);                                                                                      \
        F[(I) + 1] = *(ldsp_c4_t)((ROWP) + (o_ ^ 64));                                                                           \
    }
#define STRIPSP_PREFETCH(SLOT)                                                                                                   \
    {                                                                                                                            \
        const ldsp_cptr_t r0 = ringl + (SLOT) * STRIPSP_PAIRB;                                                                   \
        STRIPSP_LDP(0, r0, 1, 0) STRIPSP_LDP(2, r0, 1, 1) STRIPSP_LDP(4, r0, 0, 0) STRIPSP_LDP(6, r0, 2, 0)                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
    }
#define STRIPSP_STEP(J, SLOT, PAR, NWAIT, PREV)                                                                                  \
    {                                                                                                                            \
        const ldsp_cptr_t rm = ringl + (((SLOT) + STRIPSP_R - 1) % STRIPSP_R) * STRIPSP_PAIRB + STRIPSP_ROWB;  /* row 2j - 1 */  \
        const ldsp_cptr_t r0 = ringl + (SLOT) * STRIPSP_PAIRB;                                                 /* row 2j     */  \
        const ldsp_cptr_t r1 = r0 + STRIPSP_ROWB;                                                              /* row 2j + 1 */  \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                                                        \
        u32x4 o0;                                                                                                                \
        /* the polyphase walk of conv_halo_kernel<S2D>: phase (row parity, column parity) major, then slice, then the phase's   \
           taps in ascending order -- phase (0,0): tap 4; (0,1): taps 3, 5; (1,0): taps 1, 7; (1,1): taps 0, 2, 6, 8.           \
           The step's chores sit BETWEEN its first MFMA groups (a wave issues them in the idle issue cycles behind an MFMA; in  \
           a phase of their own both waves of a SIMD would wait there together): read the staging tile of the step before,     \
           store it, fetch the pair D steps ahead -- the store before the fetch, so that the counted waits are                  \
           conv_strips2.h's */                                                                                                  \
        STRIPSP_TAP(4, 0, 0) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(0, r0, 0, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        if (PREV) { o0 = *(ldsp_c4_t)(stage_r + ((PAR) ^ 1) * STRIPSP_STAGE); } __builtin_amdgcn_sched_barrier(0);              \
        STRIPSP_TAP(4, 1, 2) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(2, r0, 2, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(3, 0, 4) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(4, rm, 1, 0) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(5, 0, 6) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(6, r1, 1, 0) __builtin_amdgcn_sched_barrier(0);                                                             \
        if (PREV) { *reinterpret_cast<u32x4*>(yp) = o0; yp += ystep; } __builtin_amdgcn_sched_barrier(0);                       \
        STRIPSP_TAP(3, 1, 0) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(0, rm, 1, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(5, 1, 2) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(2, r1, 1, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        issue(((SLOT) + STRIPSP_D) % STRIPSP_R, pp, (J) + STRIPSP_D < nrows); pp += ppitch; __builtin_amdgcn_sched_barrier(0);  \
        STRIPSP_TAP(1, 0, 4) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(4, rm, 0, 0) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(7, 0, 6) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(6, rm, 2, 0) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(1, 1, 0) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(0, r1, 0, 0) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(7, 1, 2) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(2, r1, 2, 0) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(0, 0, 4) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(4, rm, 0, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(2, 0, 6) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(6, rm, 2, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(6, 0, 0) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(0, r1, 0, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(8, 0, 2) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_LDP(2, r1, 2, 1) __builtin_amdgcn_sched_barrier(0);                                                             \
        STRIPSP_TAP(0, 1, 4) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_TAP(2, 1, 6) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_TAP(6, 1, 0) __builtin_amdgcn_sched_barrier(0);                                                                 \
        STRIPSP_TAP(8, 1, 2) __builtin_amdgcn_sched_barrier(0);                                                                 \
        {                                                                                                                        \
            bf16x4 h, l;                                                                                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                      \
                h[r] = (bf16_t)acc[r];                                                                                           \
                l[r] = (bf16_t)(acc[r] - (float)h[r]);                                                                           \
            }                                                                                                                    \
            *(__attribute__((address_space(3))) bf16x4*)(stage_w + (PAR) * STRIPSP_STAGE) = h;                                   \
            *(__attribute__((address_space(3))) bf16x4*)(stage_w + (PAR) * STRIPSP_STAGE + 4096) = l;                            \
        }                                                                                                                        \
        if (want_stats) {                                                                                                        \
            _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                                   \
                const f32x2sp_t v = {acc[2 * hh], acc[2 * hh + 1]};                                                              \
                s1[hh] += v;                                                                                                     \
                s2[hh] = __builtin_elementwise_fma(v, v, s2[hh]);                                                                \
            }                                                                                                                    \
        }                                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");                                                  \
        __builtin_amdgcn_s_barrier();                                                                                            \
        asm volatile("" ::: "memory");                                                                                          \
        STRIPSP_PREFETCH(((SLOT) + 1) % STRIPSP_R)                           /* the next step's first fragments */              \
    }
        STRIPSP_PREFETCH(0)
        int j = 0;
#pragma unroll
        for (int u = 0; u < STRIPSP_D - 1; ++u) {             // the first D - 1 steps: fewer stores in flight
            if (u < nrows) {
                if (u == 0) STRIPSP_STEP(u, u, u & 1, 3 * (STRIPSP_D - 1) + u, false)
                else STRIPSP_STEP(u, u, u & 1, 3 * (STRIPSP_D - 1) + u, true)
            }
        }
        for (j = STRIPSP_D - 1; j < nrows; j += STRIPSP_R) {
#pragma unroll
            for (int u = 0; u < STRIPSP_R; ++u) {
                if (j + u >= nrows) break;
                STRIPSP_STEP(j + u, (STRIPSP_D - 1 + u) % STRIPSP_R, (STRIPSP_D - 1 + u) & 1, 4 * (STRIPSP_D - 1), true)
            }
        }
        {   // the last step's tile (every wave is past that step's barrier)
            const u32x4 o0 = *(ldsp_c4_t)(stage_r + ((nrows - 1) & 1) * STRIPSP_STAGE);
            *reinterpret_cast<u32x4*>(yp) = o0;
        }
#undef STRIPSP_STEP
#undef STRIPSP_PREFETCH
#undef STRIPSP_LDP
#undef STRIPSP_TAP
    }
    if (want_stats) {
        const int slab = band * a.nstrips + strip, slabs = a.nbands * a.nstrips;
        float* dst = a.stats + (((size_t)n * slabs + slab) * 128) * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t1 = row16_sum_to_lane15(s1[r >> 1][r & 1]), t2 = row16_sum_to_lane15(s2[r >> 1][r & 1]);
            if (p == 15) {
                const int ch = 16 * wave + kg * 4 + r;
                dst[ch * 2] = t1;
                dst[ch * 2 + 1] = t2;
            }
        }
    }
}

// returns -1 when the launch is not this kernel's shape: a = the ConvArgs ctg_conv_igemm built for a split-pair launch (one
// stride-2 3x3 window; a.Cin = 2 x the channel count, a.pair_lo = the input's plane distance)
static int launch_strips2p(const ConvArgs& a, float* stats, hipStream_t st, int* slabs_out) {
    static const bool off = getenv("CTG_NO_STRIPS2P") != nullptr;     // A/B switch
    if (off || a.ncls != 1 || a.pair_lo == 0 || a.Cin != 128 || a.Cout != 128 || a.os != 1 || a.is != 2 || a.oy0 || a.ox0 || a.frame ||
        a.bias != nullptr || a.act != ACT_NONE || a.pad_mode != PAD_ZERO || a.res != nullptr || a.fold != nullptr ||
        a.Hs != a.Ho || a.Ws != a.Wo || a.Hi != 2 * a.Ho || a.Wi != 2 * a.Wo || (a.Wo & 15) || a.Ho < 8 || (a.x_ld & 15) || (a.y_ld & 15) ||
        a.x_ld < 128 || a.y_ld < 256 || a.ntaps != 9)
        return -1;
    if ((long)a.B * a.Ho * a.Wo < (1L << 18) || (long)a.Hi * a.Wi * a.x_ld >= (1L << 30)) return -1;
    for (int t = 0; t < 9; ++t) {    // Conv2d(k=3, s=2, p=1): tap t = (ky, kx) reads input (2 oy + ky - 1, 2 ox + kx - 1), weight t
        const int tw = a.taps[t];
        if ((tw & 0xff) - 64 != t / 3 - 1 || ((tw >> 8) & 0xff) - 64 != t % 3 - 1 || (tw >> 16) != t) return -1;
    }
    StripS2PArgs s;
    s.x = (const bf16_t*)a.x; s.w = (const bf16_t*)a.w; s.y = (bf16_t*)a.y; s.stats = stats;
    s.B = a.B; s.Ho = a.Ho; s.Wo = a.Wo; s.x_ld = a.x_ld; s.y_ld = a.y_ld; s.w_tap_stride = a.w_tap_stride;
    s.x_lo = a.pair_lo; s.y_lo = a.y_ld / 2;
    s.nstrips = a.Wo / 16;
    const int n_cu = ctg_cu_count();
    static const int band_env = getenv("CTG_STRIPS2P_BAND") ? atoi(getenv("CTG_STRIPS2P_BAND")) : 0;     // A/B knob
    // one 8-wave workgroup per CU is resident (registers): bands so that the grid fills the chip once
    long nb = (1L * n_cu) / ((long)a.B * s.nstrips);
    if (nb < 1) nb = 1;
    int band = (int)((a.Ho + nb - 1) / nb);
    if (band < 8) band = 8;
    if (band_env >= 8) band = band_env;      // (the caller sized the moments buffer for >= 8-row bands)
    s.band_rows = band;
    s.nbands = (a.Ho + band - 1) / band;
    if (stats != nullptr && slabs_out != nullptr) *slabs_out = s.nbands * s.nstrips;
    static unsigned long long attr_mask = 0;       // per device
    {
        const int rc = ctg_lds_attr_once((const void*)conv_strips2p_64_128_kernel, STRIPSP_SMEM, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const dim3 grid((unsigned)((long)a.B * s.nbands * s.nstrips));
    hipLaunchKernelGGL(conv_strips2p_64_128_kernel, grid, dim3(512), STRIPSP_SMEM, st, s);
    return ctg_launch_status();
}
