// Split-pair ("bf16x3") form of conv_stript.h: the stride-2 TRANSPOSED 3x3 convolution 128 -> 64 channels on large maps with
// split-pair input, weights and output -- ConvTranspose2d(128, 64, 3, stride=2, padding=1, output_padding=1) of the generator's
// second up-sampling layer (Model/HdGan.py:93-95, [B,256,256,128] -> [B,512,512,64]) and, the same operator, the backward-data
// pass of its first down-sampling conv Conv2d(64, 128, 3, stride=2, padding=1) (:78-80), in the mode whose products are
// x_hi.w_hi + x_hi.w_lo + x_lo.w_hi (three bf16 MFMAs per fp32-grade product).
//
// As four merged parity classes on conv_halo_kernel<PK, MC> (8-row tiles) these launches take 724 us at B = 16 for 464 GFLOP of
// executed MFMA work and 1611 MB: 65536 short-lived workgroups, each re-fetching its halo per class.  Here, as in conv_stript.h:
//   * a workgroup owns a 16-pixel-wide INPUT column strip of one sample and slides down a band of input rows; input rows j and
//     j + 1 give output rows 2j and 2j + 1, all four parity classes at once: one input fetch instead of four;
//   * the split weights are 295 KB ([w_hi 32 | w_lo 32] per 32 channels x 9 taps x 64 output channels): they fit the registers of
//     EIGHT waves when the waves split the output channels (four 16-channel MFMA tiles) AND the parity classes -- waves 0-3 own
//     the classes (0,0), (0,1), (1,0) (5 taps: 40 A-fragments, 160 VGPRs), waves 4-7 the class (1,1) (4 taps: 32 fragments).
//     The two waves of a SIMD are one of each kind: 60 + 48 MFMAs per step and SIMD.  No weight traffic after the prologue;
//   * the input rows (17 px x 512 B: per pixel four 32-channel slices [hi 32 | lo 32], the LDS image of conv_halo_kernel<PK>)
//     go through one ring of 8 rows: every step each wave fetches two pixels of the row six steps ahead by LDS-DMA (and the 17th
//     pixel: half a wave), two instructions per wave and step ALWAYS (absent rows come from a zero chunk), so every step's counted
//     s_waitcnt vmcnt is one constant; the 32 chunks of a pixel are XOR-swizzled with 2 (px & 7) -- conflict-free for the pixels p
//     and p + 1 under ds_read_b128's lane groups (a 512-byte pixel pitch aliases every pixel onto the same banks otherwise);
//   * per class: (32-channel slice, tap, [hi.w_hi, hi.w_lo, lo.w_hi]) in the order conv_halo_kernel<PK, MC> accumulates -- results
//     are bit-identical to the merged-class launch;
//   * ONE barrier per step: the waves split their fp32 accumulators into the hi / lo planes of a staging tile (double buffered),
//     and after the barrier wave s stores plane s / 4 of class s % 4 as whole 128-byte pixel rows;
//   * InstanceNorm moments from the fp32 accumulators; the two waves that share an output-channel tile write separate partials.
#pragma once
#include "conv_halo.h"

typedef float f32x2_t __attribute__((ext_vector_type(2)));      // (this header is compiled in its own translation unit: conv_pair_strips.hip)
// LDS-space pointers for the ring and the staging tile: through generic `char*` the compiler converts every address back with a null
// check and keeps "base + offset" sums in VGPRs instead of instruction immediates (27 spilled registers in this kernel)
typedef const __attribute__((address_space(3))) char* lds_cptr_t;
typedef __attribute__((address_space(3))) char* lds_ptr_t;
typedef const __attribute__((address_space(3))) u32x4* lds_c4_t;

#define STRIPTP_R 8               // ring rows of the workgroup
#define STRIPTP_D 6               // a row is fetched this many steps before the step that first reads it as its lower row
#define STRIPTP_ROWB 8704         // 17 px x 512 B
#define STRIPTP_STAGE 16384       // one step's output tile: 4 classes x 2 planes x 16 px x 64 channels, bf16
#define STRIPTP_SMEM (STRIPTP_R * STRIPTP_ROWB + 2 * STRIPTP_STAGE)        // 102400 B

struct StripTPArgs {
    const bf16_t* x;            // [B][Hi][Wi][x_ld] split pair, 128 channels: hi plane, lo plane x_lo elements behind
    const bf16_t* w;            // split pack [9][w_npad >= 64][256]: per 32 channels [w_hi 32 | w_lo 32]
    bf16_t* y;                  // [B][2 Hi][2 Wi][y_ld] split pair, 64 channels (lo plane y_lo elements behind)
    float* stats;               // [B][slabs][64][2] or NULL
    int B, Hi, Wi, x_ld, y_ld, x_lo, y_lo, w_tap_stride;
    int band_rows, nbands, nstrips, xcd;
};

// one (class, tap) of one 32-channel slice: hi.w_hi, hi.w_lo, lo.w_hi -- conv_halo_kernel<PK>'s order
#define STRIPTP_TAP(ACC, T, C, H, L)                                                                                           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][C][0]), __builtin_bit_cast(bf16x8, H), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][C][1]), __builtin_bit_cast(bf16x8, H), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[T][C][0]), __builtin_bit_cast(bf16x8, L), ACC, 0, 0, 0);

// split an accumulator into the hi / lo planes of the staging tile (class q: + q * 4096; lo plane + 2048) and add its moments
template <bool MASKED>
__device__ __forceinline__ void striptp_out(const f32x4 acc, lds_ptr_t stage_q, bool col_ok, bool want_stats, f32x2_t (&s1)[2], f32x2_t (&s2)[2]) {
    bf16x4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r] = (bf16_t)acc[r];
        l[r] = (bf16_t)(acc[r] - (float)h[r]);
    }
    *(__attribute__((address_space(3))) bf16x4*)(stage_q) = h;
    *(__attribute__((address_space(3))) bf16x4*)(stage_q + 2048) = l;
    if (want_stats) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x2_t v = {acc[2 * hh], acc[2 * hh + 1]};
            if (MASKED && !col_ok) v = f32x2_t{0.f, 0.f};
            s1[hh] += v;
            s2[hh] = __builtin_elementwise_fma(v, v, s2[hh]);
        }
    }
}

// One (hi, lo) pair of pixel fragments: 32-channel slice C of the pixels whose slice-0 hi chunk lies at byte LO of ring row R; slice C
// is ^ (C * 128), the lo plane ^ 64 (the swizzle only touches bits the XOR leaves alone).  F[I] = hi, F[I + 1] = lo.
// (the XOR is an opaque instruction: as plain C the sixteen loop-invariant offsets are hoisted into registers the weights need)
__device__ __forceinline__ int striptp_xor(int v, int k) {
    int r;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(v));
    return r;
}
#define STRIPTP_LD(R, O) (*(lds_c4_t)((R) + (O)))
#define STRIPTP_LDP(I, R, LO, C)                                                                                               \
    {                                                                                                                          \
        const int o_ = (C) ? striptp_xor(LO, (C) * 128) : (LO);                                                                \
        F[(I)] = STRIPTP_LD(R, o_);                                                                                            \
        F[(I) + 1] = STRIPTP_LD(R, o_ ^ 64);                                                                                   \
    }

// The first fragments of a step, requested as soon as the step's rows are visible (right behind the barrier of the step before):
// their LDS latency runs under the staging-tile stores and the row fetch that precede the step's first MFMA.
// Group 0 (classes (0,0), (0,1), (1,0)): slice 0 of f01 -> F[0..1], f10 -> F[2..3], f00 -> F[4..5] (f[dy][dx] = pixel p + dx of row j + dy);
// group 1 (class (1,1)): its first three (slice, tap) items f11, f10, f01 of slice 0 -> F[0..5].
template <int G>
__device__ __forceinline__ void striptp_prefetch(u32x4 (&F)[6], lds_cptr_t r0, lds_cptr_t r1, int lo0, int lo1) {
    if constexpr (G == 0) {
        STRIPTP_LDP(0, r0, lo1, 0)
        STRIPTP_LDP(2, r1, lo0, 0)
        STRIPTP_LDP(4, r0, lo0, 0)
    } else {
        STRIPTP_LDP(0, r1, lo1, 0)
        STRIPTP_LDP(2, r1, lo0, 0)
        STRIPTP_LDP(4, r0, lo1, 0)
    }
}

// The arithmetic of one step of a wave: F = its first fragments (striptp_prefetch) -> the fp32 accumulators of ITS parity classes of
// output rows 2j, 2j + 1.  Fragment reads run AHEAD of the MFMAs that use them, in registers that have just been released (left to
// itself the compiler sinks every read in front of its first use: 32 exposed LDS latencies per step -- the scheduling barriers pin
// the order):
//   group 0: per slice the taps run f01 (class (0,1)), f10 (class (1,0)), then the three taps on f00; each pair of the next slice is
//            requested into the registers of the pair just used (three pairs; 9-15 MFMAs between a request and its first use);
//   group 1: its 16 (slice, tap) items through a ring of three pairs, each re-requested right behind its tap (6 MFMAs ahead).
// (Four pairs spill: the 5-tap waves hold 160 weight registers, and a spill reload drains vmcnt -- the row prefetch -- every step.)
// Per class the (slice, tap, [hi.w_hi, hi.w_lo, lo.w_hi]) order is conv_halo_kernel<PK, MC>'s.
// CH_A / CH_B / CH_C: the step's chores -- reading the previous step's staging tile, storing it, fetching the row six steps ahead --
// as statements BETWEEN the first MFMA groups: a wave issues them in the 12 idle issue cycles behind each MFMA instead of in a
// phase of their own (with one barrier-synchronised workgroup per CU both waves of a SIMD would sit in that phase together).
#define STRIPTP_BODY0(CH_A, CH_B, CH_C)                                                                                          \
    STRIPTP_TAP(a1, 1, 0, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    CH_A __builtin_amdgcn_sched_barrier(0);                                                                                     \
    STRIPTP_LDP(0, r0, lo1, 1) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a2, 3, 0, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(2, r1, lo0, 1) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 0, 0, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    CH_B __builtin_amdgcn_sched_barrier(0);                                                                                     \
    STRIPTP_TAP(a1, 2, 0, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a2, 4, 0, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    CH_C __builtin_amdgcn_sched_barrier(0);                                                                                     \
    STRIPTP_LDP(4, r0, lo0, 1) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a1, 1, 1, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(0, r0, lo1, 2) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a2, 3, 1, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(2, r1, lo0, 2) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 0, 1, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a1, 2, 1, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a2, 4, 1, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(4, r0, lo0, 2) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a1, 1, 2, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(0, r0, lo1, 3) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a2, 3, 2, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(2, r1, lo0, 3) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 0, 2, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a1, 2, 2, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a2, 4, 2, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(4, r0, lo0, 3) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a1, 1, 3, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a2, 3, 3, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a0, 0, 3, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a1, 2, 3, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a2, 4, 3, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \

#define STRIPTP_BODY1(CH_A, CH_B, CH_C)                                                                                          \
    STRIPTP_TAP(a0, 0, 0, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(0, r0, lo0, 0) __builtin_amdgcn_sched_barrier(0);                                                               \
    CH_A __builtin_amdgcn_sched_barrier(0);                                                                                     \
    STRIPTP_TAP(a0, 1, 0, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(2, r1, lo1, 1) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 2, 0, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(4, r1, lo0, 1) __builtin_amdgcn_sched_barrier(0);                                                               \
    CH_B __builtin_amdgcn_sched_barrier(0);                                                                                     \
    STRIPTP_TAP(a0, 3, 0, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(0, r0, lo1, 1) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 0, 1, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(2, r0, lo0, 1) __builtin_amdgcn_sched_barrier(0);                                                               \
    CH_C __builtin_amdgcn_sched_barrier(0);                                                                                     \
    STRIPTP_TAP(a0, 1, 1, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(4, r1, lo1, 2) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 2, 1, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(0, r1, lo0, 2) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 3, 1, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(2, r0, lo1, 2) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 0, 2, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(4, r0, lo0, 2) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 1, 2, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(0, r1, lo1, 3) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 2, 2, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(2, r1, lo0, 3) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 3, 2, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(4, r0, lo1, 3) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 0, 3, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_LDP(0, r0, lo0, 3) __builtin_amdgcn_sched_barrier(0);                                                               \
    STRIPTP_TAP(a0, 1, 3, F[2], F[3]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a0, 2, 3, F[4], F[5]) __builtin_amdgcn_sched_barrier(0);                                                        \
    STRIPTP_TAP(a0, 3, 3, F[0], F[1]) __builtin_amdgcn_sched_barrier(0);                                                        \

template <int G, bool MASKED>
__device__ __forceinline__ void striptp_mma(const u32x4 (&wf)[5 - G][4][2], u32x4 (&F)[6], lds_cptr_t r0, lds_cptr_t r1, int lo0,
                                            int lo1, lds_ptr_t stage_w, bool col_ok, bool want_stats, f32x2_t (&s1)[2], f32x2_t (&s2)[2]) {
    if constexpr (G == 0) {
        // classes (0,0): tap (0,0) w4; (0,1): (0,1) w3, (0,0) w5; (1,0): (1,0) w1, (0,0) w7 -- wf slots 0..4 = w4, w3, w5, w1, w7
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
        STRIPTP_BODY0(, , )
        striptp_out<MASKED>(a0, stage_w, col_ok, want_stats, s1, s2);
        striptp_out<MASKED>(a1, stage_w + 4096, col_ok, want_stats, s1, s2);
        striptp_out<MASKED>(a2, stage_w + 8192, col_ok, want_stats, s1, s2);
    } else {
        // class (1,1): (1,1) w0, (1,0) w2, (0,1) w6, (0,0) w8 -- wf slots 0..3
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
        STRIPTP_BODY1(, , )
        striptp_out<MASKED>(a0, stage_w + 12288, col_ok, want_stats, s1, s2);
    }
}

// The whole pipeline of a wave of class group G (0: classes (0,0), (0,1), (1,0) -- five taps; 1: class (1,1) -- four), as one
// function per group: the two kinds of waves share no per-step branch (a runtime `grp` test inside the step lets the compiler merge
// the two groups' fragment reads into SGPR-selected addresses that it then hoists and spills), and the four-tap waves hold 128
// weight registers, not 160.  Both instantiations execute the same sequence of workgroup barriers.
template <int G>
__device__ __forceinline__ void striptp_run(const StripTPArgs& a, char* smem, const int wave) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = lane & 15, kg = lane >> 4;
    const int nt = wave & 3;                             // this wave's 16 output channels
    constexpr int grp = G;
    const int item = a.xcd ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int strip = item % a.nstrips;
    const int t2 = item / a.nstrips;
    const int band = t2 % a.nbands, n = t2 / a.nbands;
    char* ring = smem;
    const lds_cptr_t ringl = (lds_cptr_t)smem;                                   // the same memory through LDS-space pointers
    const lds_ptr_t stagel = (lds_ptr_t)smem + STRIPTP_R * STRIPTP_ROWB;
    const int i0 = strip * 16, jb = band * a.band_rows;
    const int nrows = min(a.band_rows, a.Hi - jb);       // input rows (steps) of this band
    const int nin = nrows + 1;                           // rows jb .. jb + nrows
    const int Hi = a.Hi, Wi = a.Wi, x_ld = a.x_ld;

    // ---- weights: slot t of this wave's group is tap widx; row nt*16 + (lane & 15); 32-channel slice c: hi chunk at c*64 + kg*8,
    //      lo chunk 32 elements on (ctg_split_weights: [w_hi 32 | w_lo 32] per 32 channels)
    u32x4 wf[5 - G][4][2];
    {
        const int t0[5] = {4, 3, 5, 1, 7}, t1[5] = {0, 2, 6, 8, 8};
#pragma unroll
        for (int t = 0; t < 5 - G; ++t) {
            const int widx = G ? t1[t] : t0[t];
            const bf16_t* wr = a.w + (size_t)widx * a.w_tap_stride + (nt * 16 + p) * 256 + kg * 8;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                wf[t][c][0] = *reinterpret_cast<const u32x4*>(wr + c * 64);
                wf[t][c][1] = *reinterpret_cast<const u32x4*>(wr + c * 64 + 32);
            }
        }
    }

    // ---- the ring starts as zeros: the slots of columns past the image's right edge are never written again (zero padding)
    for (int i = tid; i < STRIPTP_R * STRIPTP_ROWB / 16; i += 512) *reinterpret_cast<u32x4*>(ring + i * 16) = u32x4{0u, 0u, 0u, 0u};

    // ---- a row's 17 pixels: wave w fetches pixels 2w, 2w+1 (one DMA instruction: lane -> pixel 2w + lane/32, slot chunk lane%32) and
    //      every wave the 17th (32 lanes; eight copies of the same 512 bytes: the instruction count per wave stays uniform).  Slot
    //      (px, cs) holds the logical chunk L = cs ^ 2 (px & 7) of column i0 + px; L = slice * 8 + plane * 4 + k-group.
    const int pxa = 2 * wave + (lane >> 5), csa = lane & 31;
    auto src_off = [&](int px, int cs) __attribute__((always_inline)) {
        const int L = cs ^ ((px & 7) * 2);
        return (unsigned)(px * x_ld + (L >> 3) * 32 + (L & 3) * 8 + ((L >> 2) & 1) * a.x_lo) * 2u;
    };
    const unsigned voffa = src_off(pxa, csa);
    const bool voka = i0 + pxa < Wi;
    const bool has17 = i0 + 16 < Wi;                                                                    // uniform
    const unsigned voffb = src_off(16, csa);
    const size_t rpitch = (size_t)Wi * x_ld * 2;                                                       // bytes per input row
    const char* __restrict__ X0 = reinterpret_cast<const char*>(a.x + (((size_t)n * Hi + jb) * Wi + i0) * x_ld);   // row jb, column i0
    const int nvalid = min(nin, Hi - jb);
    auto issue_fast = [&](int slot, const char* rowp, int k) __attribute__((always_inline)) {
        const bool rv = k < nvalid;             // uniform
        const char* ra = rv ? rowp : reinterpret_cast<const char*>(g_zero_chunk);
        const char* rb = (rv && has17) ? rowp : reinterpret_cast<const char*>(g_zero_chunk);
        asm volatile("" : "+s"(ra), "+s"(rb));  // opaque: keeps the addresses "uniform row pointer + lane offset"
        char* dst = ring + slot * STRIPTP_ROWB;
        __builtin_amdgcn_global_load_lds((gptr_t)(ra + (rv ? voffa : 0u)), (lptr_t)(dst + 1024 * wave), 16, 0, 0);
        if (lane < 32) __builtin_amdgcn_global_load_lds((gptr_t)(rb + ((rv && has17) ? voffb : 0u)), (lptr_t)(dst + 8192), 16, 0, 0);
    };
    auto issue_slow = [&](int k) __attribute__((always_inline)) {
        char* dst = ring + (k % STRIPTP_R) * STRIPTP_ROWB;
        const char* rowp = X0 + (size_t)k * rpitch;
        if (jb + k < Hi) {
            if (voka) __builtin_amdgcn_global_load_lds((gptr_t)(rowp + voffa), (lptr_t)(dst + 1024 * wave), 16, 0, 0);
            if (has17 && lane < 32) __builtin_amdgcn_global_load_lds((gptr_t)(rowp + voffb), (lptr_t)(dst + 8192), 16, 0, 0);
        } else {
            *reinterpret_cast<u32x4*>(dst + 1024 * wave + lane * 16) = u32x4{0u, 0u, 0u, 0u};
            if (lane < 32) *reinterpret_cast<u32x4*>(dst + 8192 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    };
    // ---- fragment byte offsets inside a ring row: pixel p + dx, logical chunk kg (slice 0, hi plane), physical chunk XORed with
    //      2 (px & 7): under ds_read_b128's lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, (+32) the 16 lanes of a group fall on
    //      16 different chunk columns for the pixels p AND p + 1 (conv_stript.h; a 512-byte pixel is two bank rows, so only the low
    //      four chunk bits matter)
    const int lo0 = (p * 32 + (kg ^ ((p & 7) * 2))) * 16;
    const int lo1 = ((p + 1) * 32 + (kg ^ (((p + 1) & 7) * 2))) * 16;
    // ---- the staging tile [class][plane][px][64 ch] (128-byte pixels, their 16-byte chunks XORed with px & 7): this lane writes its 4
    //      channels of pixel p; after the barrier wave s stores plane s/4 of class s%4 as whole 128-byte pixels: lane -> pixel
    //      lane/8 (+8), chunk lane%8
    const lds_ptr_t stage_w = stagel + p * 128 + (((nt * 2 + (kg >> 1)) ^ (p & 7)) * 16) + (kg & 1) * 8;
    const int spx = lane >> 3, sch = lane & 7;
    const int sq = wave & 3, spl = wave >> 2;
    const lds_cptr_t stage_r = stagel + sq * 4096 + spl * 2048 + spx * 128 + ((sch ^ spx) * 16);       // pixel spx; pixel spx + 8 is 1024 bytes on
    const bool col_ok = i0 + p < Wi;
    const bool full_strip = i0 + 16 <= Wi;
    const bool st_ok0 = i0 + spx < Wi, st_ok1 = i0 + spx + 8 < Wi;
    const int Wo = 2 * Wi;
    // this lane's output pointer at step 0: class sq, plane spl, row 2 jb + (sq >> 1), column 2 (i0 + spx) + (sq & 1), channels sch*8 ..
    bf16_t* __restrict__ yp = a.y + (((size_t)n * 2 * Hi + 2 * jb + (sq >> 1)) * Wo + 2 * (i0 + spx) + (sq & 1)) * a.y_ld + sch * 8 + spl * a.y_lo;
    const size_t ystep = 2 * (size_t)Wo * a.y_ld;          // two output rows per step
    const size_t y8 = 16 * (size_t)a.y_ld;                 // pixel spx + 8: 16 output columns on
    f32x2_t s1[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    u32x4 F[6];                                              // the coming step's first pixel fragments (striptp_prefetch)
    const bool want_stats = a.stats != nullptr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // the zeroed ring, before any DMA lands in it
    asm volatile("" ::: "memory");

    // A step j of the workgroup: fetch row j + D; multiply rows j, j+1 (visible since the barrier of step j - 1) into the staging
    // tile j & 1; wait for this wave's part of row j + 2; barrier; store the wave's (class, plane) as whole pixels.  Between two
    // barriers nobody reads a row older than j, so row j + D may land in the slot of row j + D - R <= j - 2.
    int j = 0;
    if (full_strip && nrows >= STRIPTP_D - 2) {
        // ---- the fast path of a strip inside the image.  Newer than the DMA of row j + 2 at the wait of step j are the DMA of rows
        //      j+3 .. j+D (2 each) and the stores of the last min(j, D-2) steps (2 each)
        const char* rowp = X0;
#pragma unroll
        for (int k = 0; k < STRIPTP_D; ++k) { issue_fast(k, rowp, k); rowp += rpitch; }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (STRIPTP_D - 2)) : "memory");       // rows 0 and 1 (and the weights)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        striptp_prefetch<G>(F, ringl, ringl + STRIPTP_ROWB, lo0, lo1);
        // chores of a step, issued between its first MFMA groups (STRIPTP_BODY*): read the staging tile of the step before, store it
        // (two whole 128-byte pixels of this wave's class and plane), fetch row j + D.  Stores before the fetch: newer than the
        // fetch of row j + 2 at the wait of step j are then exactly the fetches of rows j+3 .. j+D and the stores issued in the
        // steps j-(D-3) .. j (2 + 2 per step), as in conv_stript.h
#define STRIPTP_CH_A(PAR) o0 = *(lds_c4_t)(stage_r + ((PAR) ^ 1) * STRIPTP_STAGE);
#define STRIPTP_CH_B(PAR)                                                                                                        \
    *reinterpret_cast<u32x4*>(yp) = o0;                                                                                          \
    o0 = *(lds_c4_t)(stage_r + ((PAR) ^ 1) * STRIPTP_STAGE + 1024);
#define STRIPTP_CH_C(J, SLOT)                                                                                                    \
    *reinterpret_cast<u32x4*>(yp + y8) = o0;                                                                                     \
    yp += ystep;                                                                                                                 \
    STRIPTP_CH_DMA(J, SLOT)
#define STRIPTP_CH_DMA(J, SLOT)                                                                                                  \
    issue_fast(((SLOT) + STRIPTP_D) % STRIPTP_R, rowp, (J) + STRIPTP_D);                                                         \
    rowp += rpitch;
#define STRIPTP_FAST_STEP(J, SLOT, PAR, NWAIT, PREV)                                                                             \
    {                                                                                                                            \
        const lds_cptr_t r0 = ringl + (SLOT) * STRIPTP_ROWB, r1 = ringl + (((SLOT) + 1) % STRIPTP_R) * STRIPTP_ROWB;             \
        const lds_ptr_t sw = stage_w + (PAR) * STRIPTP_STAGE;                                                                    \
        u32x4 o0;                                                                                                                \
        if constexpr (G == 0) {                                                                                                  \
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;                                                                   \
            if (PREV) { STRIPTP_BODY0(STRIPTP_CH_A(PAR), STRIPTP_CH_B(PAR), STRIPTP_CH_C(J, SLOT)) }                              \
            else { STRIPTP_BODY0(, , STRIPTP_CH_DMA(J, SLOT)) }                                                                  \
            striptp_out<false>(a0, sw, true, want_stats, s1, s2);                                                                \
            striptp_out<false>(a1, sw + 4096, true, want_stats, s1, s2);                                                         \
            striptp_out<false>(a2, sw + 8192, true, want_stats, s1, s2);                                                         \
        } else {                                                                                                                 \
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f};                                                                                     \
            if (PREV) { STRIPTP_BODY1(STRIPTP_CH_A(PAR), STRIPTP_CH_B(PAR), STRIPTP_CH_C(J, SLOT)) }                              \
            else { STRIPTP_BODY1(, , STRIPTP_CH_DMA(J, SLOT)) }                                                                  \
            striptp_out<false>(a0, sw + 12288, true, want_stats, s1, s2);                                                        \
        }                                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");                                                  \
        __builtin_amdgcn_s_barrier();                                                                                            \
        asm volatile("" ::: "memory");                                                                                          \
        striptp_prefetch<G>(F, ringl + (((SLOT) + 1) % STRIPTP_R) * STRIPTP_ROWB, ringl + (((SLOT) + 2) % STRIPTP_R) * STRIPTP_ROWB,  \
                         lo0, lo1);                                          /* the next step's first fragments */              \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
    }
#pragma unroll
        for (int u = 0; u < STRIPTP_D - 2; ++u) {            // fewer stores in flight
            if (u == 0) STRIPTP_FAST_STEP(u, u, u & 1, 2 * (STRIPTP_D - 2) + 2 * u, false)
            else STRIPTP_FAST_STEP(u, u, u & 1, 2 * (STRIPTP_D - 2) + 2 * u, true)
        }
        static_assert(((STRIPTP_D - 2) & 1) == 0, "staging parity of the unrolled loop");
        for (j = STRIPTP_D - 2; j < nrows; j += STRIPTP_R) {
#pragma unroll
            for (int u = 0; u < STRIPTP_R; ++u) {
                if (j + u >= nrows) break;
                STRIPTP_FAST_STEP(j + u, (STRIPTP_D - 2 + u) % STRIPTP_R, u & 1, 4 * (STRIPTP_D - 2), true)
            }
        }
        {   // the last step's tile (every wave is past that step's barrier)
            const int par = (nrows - 1) & 1;
            const u32x4 o0 = *(lds_c4_t)(stage_r + par * STRIPTP_STAGE);
            const u32x4 o1 = *(lds_c4_t)(stage_r + par * STRIPTP_STAGE + 1024);
            *reinterpret_cast<u32x4*>(yp) = o0;
            *reinterpret_cast<u32x4*>(yp + y8) = o1;
            yp += ystep;
        }
        j = nrows;
#undef STRIPTP_CH_A
#undef STRIPTP_CH_B
#undef STRIPTP_CH_C
#undef STRIPTP_CH_DMA
#undef STRIPTP_FAST_STEP
    } else {
        for (int k = 0; k < STRIPTP_D && k < nin; ++k) issue_slow(k);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // ---- every step of a ragged strip or a very short band: nothing left in flight at a barrier
    for (; j < nrows; ++j) {
        if (j + STRIPTP_D < nin) issue_slow(j + STRIPTP_D);
        const int par = j & 1;
        striptp_prefetch<G>(F, ringl + (j % STRIPTP_R) * STRIPTP_ROWB, ringl + ((j + 1) % STRIPTP_R) * STRIPTP_ROWB, lo0, lo1);
        striptp_mma<G, true>(wf, F, ringl + (j % STRIPTP_R) * STRIPTP_ROWB, ringl + ((j + 1) % STRIPTP_R) * STRIPTP_ROWB, lo0, lo1,
                          stage_w + par * STRIPTP_STAGE, col_ok, want_stats, s1, s2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const u32x4 o0 = *(lds_c4_t)(stage_r + par * STRIPTP_STAGE);
        const u32x4 o1 = *(lds_c4_t)(stage_r + par * STRIPTP_STAGE + 1024);
        if (st_ok0) *reinterpret_cast<u32x4*>(yp) = o0;
        if (st_ok1) *reinterpret_cast<u32x4*>(yp + y8) = o1;
        yp += ystep;
    }
    if (want_stats) {
        // the two class groups of an output-channel tile: separate partial slabs (2 per band and strip)
        const int slab = (band * a.nstrips + strip) * 2 + grp, slabs = a.nbands * a.nstrips * 2;
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

__global__ __launch_bounds__(512, 1) void conv_striptp_128_64_kernel(const StripTPArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) striptp_run<0>(a, smem, wave);
    else striptp_run<1>(a, smem, wave);
}

// returns -1 when the launch is not this kernel's shape: a = the ConvArgs ctg_conv_igemm_classes built for a split-pair launch
// (4 classes; a.Cin = 2 x the channel count, a.pair_lo = the input's plane distance)
static int launch_striptp(const ConvArgs& a, hipStream_t st, int* tiles_out) {
    static const bool off = getenv("CTG_NO_STRIPTP") != nullptr;      // A/B switch
    if (off || a.ncls != 4 || a.pair_lo == 0 || a.Cin != 256 || a.Cout != 64 || a.os != 2 || a.is != 1 || a.bias != nullptr ||
        a.act != ACT_NONE || a.pad_mode != PAD_ZERO || a.Hs != a.Hi || a.Ws != a.Wi || a.Ho != 2 * a.Hi || a.Wo != 2 * a.Wi ||
        (a.x_ld & 15) || (a.y_ld & 15) || a.x_ld < 256 || a.y_ld < 128)
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
    StripTPArgs s;
    s.x = (const bf16_t*)a.x; s.w = (const bf16_t*)a.w; s.y = (bf16_t*)a.y; s.stats = a.stats;
    s.B = a.B; s.Hi = a.Hi; s.Wi = a.Wi; s.x_ld = a.x_ld; s.y_ld = a.y_ld; s.w_tap_stride = a.w_tap_stride;
    s.x_lo = a.pair_lo; s.y_lo = a.y_ld / 2;
    s.nstrips = (a.Wi + 15) / 16;
    static const int xcd_env = getenv("CTG_STRIPT_XCD") ? atoi(getenv("CTG_STRIPT_XCD")) : 1;      // A/B knob
    s.xcd = xcd_env;
    const int n_cu = ctg_cu_count();
    static const int band_env = getenv("CTG_STRIPTP_BAND") ? atoi(getenv("CTG_STRIPTP_BAND")) : 0;      // A/B knob
    static const int wgs_per_cu = getenv("CTG_STRIPTP_WGS") ? atoi(getenv("CTG_STRIPTP_WGS")) : 1;      // A/B knob
    // one 8-wave workgroup per CU is resident (registers): bands so that the grid fills the chip once
    long nb = ((long)wgs_per_cu * n_cu) / ((long)a.B * s.nstrips);
    if (nb < 1) nb = 1;
    int band = (int)((a.Hi + nb - 1) / nb);
    if (band < 16) band = 16;
    if (band_env >= 8) band = band_env;      // (the caller sized the moments buffer for >= 8-row bands, 4 slots per 8 x 16 tile)
    s.band_rows = band;
    s.nbands = (a.Hi + band - 1) / band;
    if (tiles_out != nullptr) *tiles_out = 2 * s.nbands * s.nstrips;
    const int smem = STRIPTP_SMEM;
    static unsigned long long attr_mask = 0;       // per device
    {
        const int rc = ctg_lds_attr_once((const void*)conv_striptp_128_64_kernel, smem, &attr_mask);
        if (rc != CTG_OK) return rc;
    }
    const dim3 grid((unsigned)((long)a.B * s.nbands * s.nstrips));
    hipLaunchKernelGGL(conv_striptp_128_64_kernel, grid, dim3(512), smem, st, s);
    return ctg_launch_status();
}
