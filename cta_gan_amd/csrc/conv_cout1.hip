// The PatchGAN's last layer: Conv2d(512, 1, 4, padding=1) -- Model/HdGan.py:136-137 (every scale of Discriminator_m),
// Model/HdGan.py:217-219 (NLayerDiscriminator), Model/CycleGan.py:104 (Discriminator) -- forward, input gradient and weight gradient.
//
// On the matrix-core kernels this layer is a GEMM with ONE useful column: the forward rode conv_halo_kernel<BN = 16> (15 of 16
// output columns zero), the backward zero-padded the 1-channel gradient to 32 channels for a 32 -> 512 conv and a 32 x 512 weight
// gradient: 95 + 68 + 113 us per call in bf16 and 200 + 120 + 340 us on split-pair operands (rounds 1-4; 0.08-0.23 of their
// bounds, profiles/r05_conv_roofline_*.md) for a layer whose whole input is 67 / 134 MB.  It is a matrix-VECTOR product per pixel;
// these kernels do it on the vector ALUs in fp32, reading the wide tensor once:
//   * a lane owns 8 of the 512 channels (one 16-byte chunk per pixel and plane), a wave a run of 16 pixels of one row;
//   * forward: 16 output accumulators per lane (its 8-channel share of each output pixel of the run), every input pixel of the
//     4 x 19 window loaded once and used for up to 4 outputs; ONE cross-lane reduction per 16 outputs;
//   * input gradient: the 4 x 19 patch of the scalar gradient map broadcast to every lane, dx[pixel][8 channels] = sum over the 16
//     taps of g x w: 128 FMAs per pixel and lane, stored as one chunk per plane;
//   * weight gradient: the same patch, dw[tap][8 channels] += g x x over the wave's pixels (128 accumulators per lane), the four
//     waves of a workgroup summed through LDS in a fixed order, one fp32 partial per workgroup for ctg_wgrad_reduce.
// Weights stay fp32 (the master copy: no rounding at all); split-pair operands enter as hi + lo (exact in fp32), so in the
// "bf16x3" mode these three launches are closer to the fp32 reference than the three-product MFMA form they replace.
#include "common.h"

typedef float c1_f2 __attribute__((ext_vector_type(2)));      // two fp32 lanes of one v_pk_fma_f32 / v_pk_mul_f32

// acc += a * b as ONE v_pk_fma_f32 on whole register pairs.  Written as inline assembly on purpose: from `c1_f2{g, g} * w + acc` the
// compiler folds the splat into the instruction's op_sel / op_sel_hi modifiers (both halves of a source read from ONE register of a
// pair) -- and that form returned wrong products in lanes 48-63 whenever the wave shared a CU with workgroups of
// conv_halo_kernel<BN <= 32> on another stream (scripts/lds_neighbour_stress.py: 3e5 wrong elements per run; the same instruction on
// a materialised {g, g} pair, or two v_fma_f32: none; found through tests/test_step_parity_gpu.py's bit-repeatability of the
// CycleGan step).  The cause is not understood, so since round 6 the whole library is compiled with the
// packed-fp32 target feature OFF (cta_gan_amd/build.py: the compiler cannot form these instructions at all, modifiers or not) and the
// three kernels of this file -- bound by their vector instructions -- write every packed operation they want as inline assembly on
// whole register pairs; tests/test_isa_gate.py disassembles the library and fails on any v_pk_*_f32 that carries a modifier.
// The assembler refuses the mnemonic in a function without the feature, so these three kernels (and only they) re-enable it with a
// target attribute (C1_PK_F32); what the compiler then forms by itself inside them is kept modifier-free by hand: accumulators start
// from registers, not inline constants (op_sel_hi:[1,1,0]), and the hi/lo split of the stores subtracts with v_sub_f32 (c1_sub: the
// SLP vectoriser pairs plain subtractions into v_pk_add_f32 neg_lo neg_hi).
#if defined(__HIP_DEVICE_COMPILE__)
#define C1_PK_F32 __attribute__((target("packed-fp32-ops")))
#else
#define C1_PK_F32
#endif
__device__ __forceinline__ void c1_pk_fma(c1_f2& acc, const c1_f2 a, const c1_f2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ c1_f2 c1_pk_add(const c1_f2 a, const c1_f2 b) {
    c1_f2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float c1_sub(const float a, const float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Chunk<T>::store for this file's kernels (see C1_PK_F32)
__device__ __forceinline__ void c1_store(bf16_t* p, int, const c1_f2 (&d)[4]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)d[i >> 1][i & 1];
    *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void c1_store(bfpair_t* p, int ld, const c1_f2 (&d)[4]) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = d[i >> 1][i & 1];
        h[i] = (bf16_t)v;                          // RNE
        l[i] = (bf16_t)c1_sub(v, (float)h[i]);     // the remainder is exact in fp32
    }
    *reinterpret_cast<bf16x8*>(p) = h;
    *reinterpret_cast<bf16x8*>(p + (ld >> 1)) = l;
}

#define C1_CIN 512
#define C1_RUN 16
#define C1_WB 8       // pixels per load batch of the weight-gradient kernel

// a pixel's 8-channel chunk as loaded (one 16-byte word per plane): kept raw so that a whole row of them can be REQUESTED before the
// first is converted -- with the loads inside per-pixel branches the compiler waits for each one in turn and a wave spends its
// life in 76 memory latencies (first version of the forward kernel: 107 us for 65 MB)
template <typename T> struct RawChunk;
template <> struct RawChunk<bf16_t> {
    u32x4 h;
    __device__ __forceinline__ void load(const bf16_t* p, int) { h = *reinterpret_cast<const u32x4*>(p); }
    __device__ __forceinline__ void keep_if(bool k) { if (!k) h = u32x4{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ void to(c1_f2 (&v)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = c1_f2{__uint_as_float(h[i] << 16), __uint_as_float(h[i] & 0xffff0000u)};
    }
};
template <> struct RawChunk<bfpair_t> {
    u32x4 h, l;
    __device__ __forceinline__ void load(const bfpair_t* p, int ld) {
        h = *reinterpret_cast<const u32x4*>(p);
        l = *reinterpret_cast<const u32x4*>(p + (ld >> 1));
    }
    __device__ __forceinline__ void keep_if(bool k) { if (!k) { h = u32x4{0u, 0u, 0u, 0u}; l = h; } }
    __device__ __forceinline__ void to(c1_f2 (&v)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = c1_pk_add(c1_f2{__uint_as_float(h[i] << 16), __uint_as_float(h[i] & 0xffff0000u)},
                             c1_f2{__uint_as_float(l[i] << 16), __uint_as_float(l[i] & 0xffff0000u)});
    }
};

// (bf16: three waves per SIMD -- the kernel is a chain of load batches per wave, and what hides their latency is the number of
//  waves; the split-pair instantiation needs the registers of two)
template <typename T, int KS>
__global__ C1_PK_F32 __launch_bounds__(256, (sizeof(RawChunk<T>) > 16 ? 2 : 3)) void cout1_fwd_kernel(const T* __restrict__ x, int x_ld, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int act,
                                                            int Hi, int Wi, int pad, int Ho, int Wo, int ntask) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: SGPR
    // every input row feeds KS output rows: consecutive workgroups (= consecutive output rows) must share an L2, or each XCD fetches
    // its own copy of the row from HBM (measured: 132 us for 130 MB with the dispatcher's round-robin order)
    const int task = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;
    if (task >= ntask) return;      // (no barrier in this kernel)
    const int segs = (Wo + C1_RUN - 1) / C1_RUN;
    const int seg = task % segs, oy = (task / segs) % Ho, b = task / (segs * Ho);
    const int ox0 = seg * C1_RUN;
    // packed fp32 arithmetic throughout (v_pk_fma_f32: the kernel is bound by its vector instructions once the loads are batched):
    // an output's accumulator is a PAIR of partial sums (even / odd channels of the lane's chunk), added at the end
    c1_f2 acc[C1_RUN];
#pragma unroll
    for (int o = 0; o < C1_RUN; ++o) acc[o] = c1_f2{0.f, 0.f};
    const T* __restrict__ X = x + (size_t)b * Hi * Wi * x_ld + 8 * lane;
    static_for<KS>([&](auto kyc) {
        constexpr int ky = decltype(kyc)::value;
        const int iy = oy + ky - pad;
        if ((unsigned)iy < (unsigned)Hi) {      // wave-uniform
            // this kernel row's weights and the pixels of the input row are requested in batches before anything of a batch is used
            c1_f2 wr[KS][4];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(w + (ky * KS + kx) * C1_CIN + 8 * lane);
                const f32x4 c = *reinterpret_cast<const f32x4*>(w + (ky * KS + kx) * C1_CIN + 8 * lane + 4);
                wr[kx][0] = c1_f2{a[0], a[1]}; wr[kx][1] = c1_f2{a[2], a[3]};
                wr[kx][2] = c1_f2{c[0], c[1]}; wr[kx][3] = c1_f2{c[2], c[3]};
            }
            // batches of 10 (bf16) / 5 (split pair) pixels: 40 raw registers at most
            constexpr int PW = C1_RUN + KS - 1, NBATCH = sizeof(RawChunk<T>) > 16 ? 4 : 2, HB = (PW + NBATCH - 1) / NBATCH;
            static_for<NBATCH>([&](auto hc) {
                constexpr int j0 = decltype(hc)::value * HB;
                constexpr int nb = j0 + HB <= PW ? HB : PW - j0;
                RawChunk<T> raw[nb];
#pragma unroll
                for (int j = 0; j < nb; ++j) {
                    int ix = ox0 + j0 + j - pad;
                    ix = ix < 0 ? 0 : (ix > Wi - 1 ? Wi - 1 : ix);      // out-of-row pixels read a valid address and are zeroed below
                    raw[j].load(X + (size_t)(iy * Wi + ix) * x_ld, x_ld);
                }
                static_for<nb>([&](auto jc) {
                    constexpr int j = j0 + decltype(jc)::value;
                    raw[j - j0].keep_if((unsigned)(ox0 + j - pad) < (unsigned)Wi);
                    c1_f2 v[4];
                    raw[j - j0].to(v);
                    static_for<KS>([&](auto kxc) {
                        constexpr int kx = decltype(kxc)::value;
                        constexpr int o = j - kx;       // ox - ox0 of the output this input pixel feeds through tap (ky, kx)
                        if constexpr (o >= 0 && o < C1_RUN) {
                            c1_f2 s = acc[o];
#pragma unroll
                            for (int q = 0; q < 4; ++q) c1_pk_fma(s, v[q], wr[kx][q]);
                            acc[o] = s;
                        }
                    });
                });
                asm volatile("" ::: "memory");      // (as in the weight-gradient kernel: one batch of raw words live at a time)
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    });
    float mine = 0.f;
#pragma unroll
    for (int o = 0; o < C1_RUN; ++o) {
        const float s = wave_sum(acc[o][0] + acc[o][1]);
        if (lane == o) mine = s;
    }
    if (lane < C1_RUN && ox0 + lane < Wo) {
        const float v = mine + (bias != nullptr ? bias[0] : 0.f);
        y[((size_t)b * Ho + oy) * Wo + ox0 + lane] = act_apply(v, act);
    }
}

// the KS x (16 + KS - 1) patch of the scalar map g around a run of 16 input pixels (row iy, columns ix0 ..): entry [ky][c] =
// g[iy - ky + pad][ix0 + c - (KS - 1) + pad] (0 outside the map) -- input pixel ix0 + i meets it through tap (ky, kx) at c = i - kx + KS - 1
template <int KS>
__device__ __forceinline__ void cout1_patch(const float* __restrict__ G, int Ho, int Wo, int iy, int ix0, int pad, int lane,
                                            float* sp) {
    constexpr int PW = C1_RUN + KS - 1;
    for (int i = lane; i < KS * PW; i += 64) {
        const int ky = i / PW, c = i - ky * PW;
        const int oy = iy - ky + pad, ox = ix0 + c - (KS - 1) + pad;
        sp[i] = ((unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo) ? G[(size_t)oy * Wo + ox] : 0.f;
    }
    __syncthreads();        // (every wave of the workgroup, live task or not)
}

template <typename T, int KS>
__global__ C1_PK_F32 __launch_bounds__(256, 2) void cout1_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                            T* __restrict__ dx, int dx_ld, int Hi, int Wi, int pad, int Ho,
                                                            int Wo, int ntask) {
    constexpr int PW = C1_RUN + KS - 1;
    __shared__ float spatch[4][KS * PW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: SGPR
    const int task = blockIdx.x * 4 + wave;
    const bool live = task < ntask;         // (every wave reaches the barrier in cout1_patch)
    const int segs = (Wi + C1_RUN - 1) / C1_RUN;
    const int t2 = live ? task : 0;
    const int seg = t2 % segs, iy = (t2 / segs) % Hi, b = t2 / (segs * Hi);
    const int ix0 = seg * C1_RUN;
    cout1_patch<KS>(g + (size_t)b * Ho * Wo, Ho, Wo, iy, ix0, pad, lane, spatch[wave]);
    if (!live) return;
    const float* sp = spatch[wave];      // (NOT restrict: the patch is written through another pointer)
    c1_f2 wr[KS * KS][4];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + t * C1_CIN + 8 * lane);
        const f32x4 c = *reinterpret_cast<const f32x4*>(w + t * C1_CIN + 8 * lane + 4);
        wr[t][0] = c1_f2{a[0], a[1]}; wr[t][1] = c1_f2{a[2], a[3]};
        wr[t][2] = c1_f2{c[0], c[1]}; wr[t][3] = c1_f2{c[2], c[3]};
    }
    T* __restrict__ D = dx + ((size_t)(b * Hi + iy) * Wi) * dx_ld + 8 * lane;
    static_for<C1_RUN>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (ix0 + i < Wi) {     // wave-uniform
            c1_f2 d[4] = {c1_f2{0.f, 0.f}, c1_f2{0.f, 0.f}, c1_f2{0.f, 0.f}, c1_f2{0.f, 0.f}};
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float g1 = sp[ky * PW + i - kx + KS - 1];      // a broadcast LDS read at a constant offset
                    const c1_f2 gv = c1_f2{g1, g1};
#pragma unroll
                    for (int q = 0; q < 4; ++q) c1_pk_fma(d[q], gv, wr[ky * KS + kx][q]);
                }
            c1_store(D + (size_t)(ix0 + i) * dx_ld, dx_ld, d);
        }
    });
}

// (one wave per SIMD: 128 accumulators + a batch of raw words + the compiler's address registers do not fit 256 registers in the
//  split-pair instantiation, and four waves per CU with 8-16 loads in flight each are enough to stream)
template <typename T, int KS>
__global__ C1_PK_F32 __launch_bounds__(256, 1) void cout1_wgrad_kernel(const float* __restrict__ g, const T* __restrict__ x, int x_ld,
                                                              float* __restrict__ part, int Hi, int Wi, int pad, int Ho, int Wo,
                                                              int ntask) {
    constexpr int PW = C1_RUN + KS - 1;
    __shared__ float spatch[4][KS * PW];
    __shared__ float red[KS * KS * C1_CIN];     // 32 KB for 4x4 taps
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: SGPR
    const int segs = (Wi + C1_RUN - 1) / C1_RUN;
    c1_f2 acc[KS * KS][4];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = c1_f2{0.f, 0.f};
    const int rounds = (ntask + gridDim.x * 4 - 1) / (gridDim.x * 4);      // the same for every wave: uniform barrier counts
    for (int r = 0; r < rounds; ++r) {
        const int task = (r * gridDim.x + blockIdx.x) * 4 + wave;
        const bool live = task < ntask;
        const int t2 = live ? task : 0;
        const int seg = t2 % segs, iy = (t2 / segs) % Hi, b = t2 / (segs * Hi);
        const int ix0 = seg * C1_RUN;
        __syncthreads();        // the previous round's broadcast reads are done before the patch is overwritten
        cout1_patch<KS>(g + (size_t)b * Ho * Wo, Ho, Wo, iy, ix0, pad, lane, spatch[wave]);
        const float* sp = spatch[wave];      // (NOT restrict: the patch is written through another pointer)
        if (live) {
            const T* __restrict__ X = x + ((size_t)(b * Hi + iy) * Wi) * x_ld + 8 * lane;
            // batches of C1_WB pixels: every load of a batch is requested before the first is used (see RawChunk)
            constexpr int WB = C1_WB;
            static_for<C1_RUN / WB>([&](auto hc) {
                constexpr int i0 = decltype(hc)::value * WB;
                RawChunk<T> raw[WB];
#pragma unroll
                for (int i = 0; i < WB; ++i) {
                    const int ix = ix0 + i0 + i;
                    raw[i].load(X + (size_t)(ix < Wi ? ix : Wi - 1) * x_ld, x_ld);
                }
                static_for<WB>([&](auto ic) {
                    constexpr int i = i0 + decltype(ic)::value;
                    raw[i - i0].keep_if(ix0 + i < Wi);
                    c1_f2 v[4];
                    raw[i - i0].to(v);
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            const float g1 = sp[ky * PW + i - kx + KS - 1];      // a broadcast LDS read at a constant offset
                            const c1_f2 gv = c1_f2{g1, g1};
#pragma unroll
                            for (int q = 0; q < 4; ++q) c1_pk_fma(acc[ky * KS + kx][q], gv, v[q]);
                        }
                });
                // the next batch's loads stay behind this batch's use: hoisted, all 16 pixels' raw words are live beside the 128
                // accumulators and the kernel spills
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    // the four waves' sums in a fixed order (wave 0 stores, waves 1..3 add one after the other)
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < KS * KS; ++t)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float* p = red + t * C1_CIN + 8 * lane + q;
                    *p = wv == 0 ? acc[t][q >> 1][q & 1] : *p + acc[t][q >> 1][q & 1];
                }
        }
    }
    __syncthreads();
    float* __restrict__ dst = part + (size_t)blockIdx.x * KS * KS * C1_CIN;
    for (int i = threadIdx.x; i < KS * KS * C1_CIN; i += 256) dst[i] = red[i];
}

static bool cout1_shape_ok(int dtype, int B, int Hi, int Wi, int Cin, int k, int pad, int Ho, int Wo) {
    return (dtype == DT_BF16 || dtype == DT_PAIR) && Cin == C1_CIN && k == 4 && pad >= 0 && pad < k && B >= 1 && Hi >= 1 && Wi >= 1 &&
           Ho == Hi + 2 * pad - k + 1 && Wo == Wi + 2 * pad - k + 1 && Ho >= 1 && Wo >= 1 &&
           (long)B * Hi * Wi * 2 * Cin < (1L << 31);
}

extern "C" int ctg_conv_cout1_fwd(int dtype, const void* x, int x_ld, const float* w, const float* bias, float* y, int act, int B,
                                  int Hi, int Wi, int Cin, int k, int pad, int Ho, int Wo, void* stream) {
    CTG_ENTER();
    if (x == nullptr || w == nullptr || y == nullptr || !cout1_shape_ok(dtype, B, Hi, Wi, Cin, k, pad, Ho, Wo)) return CTG_EINVAL;
    if (x_ld % 8 || x_ld < (dtype == DT_PAIR ? 2 : 1) * Cin || (dtype == DT_PAIR && x_ld % 16) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15))
        return CTG_EINVAL;
    const long ntask = (long)B * Ho * ((Wo + C1_RUN - 1) / C1_RUN);
    if (ntask >= (1L << 30)) return CTG_EINVAL;
    const dim3 grid((unsigned)((ntask + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_PAIR)
        hipLaunchKernelGGL((cout1_fwd_kernel<bfpair_t, 4>), grid, dim3(256), 0, st, (const bfpair_t*)x, x_ld, w, bias, y, act, Hi, Wi, pad,
                           Ho, Wo, (int)ntask);
    else
        hipLaunchKernelGGL((cout1_fwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, x_ld, w, bias, y, act, Hi, Wi, pad, Ho,
                           Wo, (int)ntask);
    return ctg_launch_status();
}

extern "C" int ctg_conv_cout1_bwd(int dtype, const float* g, const float* w, void* dx, int dx_ld, int B, int Hi, int Wi, int Cin,
                                  int k, int pad, int Ho, int Wo, void* stream) {
    CTG_ENTER();
    if (g == nullptr || w == nullptr || dx == nullptr || !cout1_shape_ok(dtype, B, Hi, Wi, Cin, k, pad, Ho, Wo)) return CTG_EINVAL;
    if (dx_ld % 8 || dx_ld < (dtype == DT_PAIR ? 2 : 1) * Cin || (dtype == DT_PAIR && dx_ld % 16) || ((uintptr_t)dx & 15) || ((uintptr_t)w & 15))
        return CTG_EINVAL;
    const long ntask = (long)B * Hi * ((Wi + C1_RUN - 1) / C1_RUN);
    if (ntask >= (1L << 30)) return CTG_EINVAL;
    const dim3 grid((unsigned)((ntask + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_PAIR)
        hipLaunchKernelGGL((cout1_bwd_kernel<bfpair_t, 4>), grid, dim3(256), 0, st, g, w, (bfpair_t*)dx, dx_ld, Hi, Wi, pad, Ho, Wo,
                           (int)ntask);
    else
        hipLaunchKernelGGL((cout1_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, g, w, (bf16_t*)dx, dx_ld, Hi, Wi, pad, Ho, Wo, (int)ntask);
    return ctg_launch_status();
}

extern "C" int ctg_conv_cout1_wgrad(int dtype, const float* g, const void* x, int x_ld, float* part, int Z, int B, int Hi, int Wi,
                                    int Cin, int k, int pad, int Ho, int Wo, void* stream) {
    CTG_ENTER();
    if (g == nullptr || x == nullptr || part == nullptr || Z < 1 || Z > 65535 || !cout1_shape_ok(dtype, B, Hi, Wi, Cin, k, pad, Ho, Wo))
        return CTG_EINVAL;
    if (x_ld % 8 || x_ld < (dtype == DT_PAIR ? 2 : 1) * Cin || (dtype == DT_PAIR && x_ld % 16) || ((uintptr_t)x & 15)) return CTG_EINVAL;
    const long ntask = (long)B * Hi * ((Wi + C1_RUN - 1) / C1_RUN);
    if (ntask >= (1L << 30)) return CTG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DT_PAIR)
        hipLaunchKernelGGL((cout1_wgrad_kernel<bfpair_t, 4>), dim3(Z), dim3(256), 0, st, g, (const bfpair_t*)x, x_ld, part, Hi, Wi, pad, Ho,
                           Wo, (int)ntask);
    else
        hipLaunchKernelGGL((cout1_wgrad_kernel<bf16_t, 4>), dim3(Z), dim3(256), 0, st, g, (const bf16_t*)x, x_ld, part, Hi, Wi, pad, Ho, Wo,
                           (int)ntask);
    return ctg_launch_status();
}
