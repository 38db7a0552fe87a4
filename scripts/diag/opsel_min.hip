// Minimal reproducer attempt for the packed-fp32 `op_sel` hazard (LAB_NOTES.md section 9): nothing but the
// instruction form that failed in conv_cout1's first input-gradient kernel -- v_pk_fma_f32 whose LOW half selects the HIGH register
// of a VGPR source pair (op_sel:[1,0,0]) -- fed the way that kernel fed it (a patch of floats written to LDS once, read back with
// broadcast ds_read2_b32 into register pairs, 4 accumulator pairs per tap), beside a scalar v_fma_f32 chain on the same operands in
// the same lane.  Each lane counts the accumulator halves whose packed and scalar results differ (they are bit-identical when the
// instruction does what the ISA says: both are single fused multiply-adds).  Built with the packed-fp32 feature ON, outside the
// product library (scripts/diag/opsel_min.py compiles it into cta_gan_amd/_build/diag/).
//   form 0: op_sel:[1,0,0]          low half <- pair's HIGH register   (the 16 instructions that failed)
//   form 1: op_sel_hi:[0,1,1]       high half <- pair's LOW register   (the 1008 that did not)
//   form 2: a materialised {g, g} pair, no modifier                    (the shipped form)
#include <hip/hip_runtime.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int FORM>
__device__ __forceinline__ void pk(f2& acc, const f2 g, const f2 w) {
    if constexpr (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(g), "v"(w));
    else if constexpr (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(g), "v"(w));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(g), "v"(w));
}

#define NACC 16      // accumulator pairs per lane (the failing kernel held 4 per output pixel, 16 pixels in flight over its life)
#define NPATCH 76    // the failing kernel's 4 x 19 patch

template <int FORM>
__global__ __launch_bounds__(256, 2) void opsel_kernel(const float* __restrict__ patch_src, const float* __restrict__ wsrc, int iters,
                                                        unsigned* __restrict__ bad_by_lane, float* __restrict__ sink) {
    __shared__ float sp[4][NPATCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < NPATCH; i += 64) sp[wave][i] = patch_src[(blockIdx.x * 4 + wave) % 64 * NPATCH + i];
    __syncthreads();
    f2 w[NACC], acc[NACC];
    float sa[NACC][2];
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        w[q] = f2{wsrc[(lane * 32 + q) * 2], wsrc[(lane * 32 + q) * 2 + 1]};
        acc[q] = f2{0.f, 0.f};
        sa[q][0] = 0.f; sa[q][1] = 0.f;
    }
    unsigned bad = 0;
    // the whole patch is read into register pairs ONCE, before the loop (broadcast LDS reads, every lane the same address), and the
    // pairs are consumed long after -- as in the failing kernel, where the last patch column waited ~1600 instructions for its use
    f2 gpr[NPATCH / 2];
#pragma unroll
    for (int t = 0; t < NPATCH / 2; ++t) gpr[t] = *reinterpret_cast<const f2*>(&sp[wave][2 * t]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NPATCH / 2; ++t) {
            const f2 gp = gpr[t];
            // the value the splat takes: form 0 -> the pair's HIGH register, form 1 -> its LOW register, form 2 -> a copy of the high one
            const float g1 = FORM == 1 ? gp[0] : gp[1];
            const f2 gm = FORM == 2 ? f2{g1, g1} : gp;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = (t * 4 + q) % NACC;
                pk<FORM>(acc[a], gm, w[a]);
                sa[a][0] = __builtin_fmaf(g1, w[a][0], sa[a][0]);
                sa[a][1] = __builtin_fmaf(g1, w[a][1], sa[a][1]);
            }
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            bad += (__float_as_uint(acc[q][0]) != __float_as_uint(sa[q][0])) + (__float_as_uint(acc[q][1]) != __float_as_uint(sa[q][1]));
            acc[q] = f2{sa[q][0] * 0.5f, sa[q][1] * 0.5f};      // keep the chains bounded and in step
            sa[q][0] = acc[q][0]; sa[q][1] = acc[q][1];
        }
        // a 16-byte store per lane and iteration: global memory traffic in flight beside the packed instructions, as the failing
        // kernel's per-pixel stores were
        *reinterpret_cast<float4*>(sink + 4 + ((size_t)(blockIdx.x * 256 + threadIdx.x) * 4)) = float4{acc[0][0], acc[1][0], acc[2][1], acc[3][1]};
    }
    if (bad) atomicAdd(&bad_by_lane[lane], bad);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1];
    if (s == 123.456f) sink[0] = s;      // keeps the packed chain alive
}

extern "C" int opsel_launch(int form, int blocks, int iters, const float* patch, const float* w, unsigned* bad_by_lane, float* sink,
                            void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (form == 0) hipLaunchKernelGGL(opsel_kernel<0>, dim3(blocks), dim3(256), 0, st, patch, w, iters, bad_by_lane, sink);
    else if (form == 1) hipLaunchKernelGGL(opsel_kernel<1>, dim3(blocks), dim3(256), 0, st, patch, w, iters, bad_by_lane, sink);
    else hipLaunchKernelGGL(opsel_kernel<2>, dim3(blocks), dim3(256), 0, st, patch, w, iters, bad_by_lane, sink);
    return (int)hipGetLastError();
}
