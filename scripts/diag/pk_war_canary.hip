// Stand-alone diagnostic (not part of the library): does an LDS read that overwrites the source pair of the v_pk_fma_f32 issued just
// before it disturb that instruction?  FORM 0: the FMA reads the pair through op_sel / op_sel_hi (both halves from the pair's high
// register); FORM 1: the FMA reads a copy of the value made by v_mov (plain form).  Every lane also accumulates the same products
// with v_fma_f32 from a copy taken BEFORE the packed instruction.  Built and run by scripts/pk_war_probe.py beside a conv launch.
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int FORM>
__global__ __launch_bounds__(256) void pk_war_kernel(int iters, const float* __restrict__ gsrc, unsigned* __restrict__ report) {
    __shared__ float tab[4][128];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    tab[wave][lane] = gsrc[lane];
    tab[wave][64 + lane] = gsrc[64 + lane];
    __syncthreads();
    const f2 w = f2{1.0f + 0.001f * (float)lane, 0.5f - 0.002f * (float)lane};
    const unsigned base = (unsigned)(size_t)(&tab[wave][0]);
    for (int it = 0; it < iters; ++it) {
        f2 acc = f2{0.f, 0.f};
        float r0 = 0.f, r1 = 0.f;
        asm volatile("ds_read_b64 v[200:201], %0\n\ts_waitcnt lgkmcnt(0)" :: "v"(base) : "v200", "v201", "memory");
#pragma unroll
        for (int i = 0; i < 60; ++i) {
            const unsigned next = base + 8u * (unsigned)(i + 1);
            if constexpr (FORM == 0) {
                asm volatile("v_mov_b32 v202, v201\n\t"
                             "v_pk_fma_f32 %0, v[200:201], %3, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                             "ds_read_b64 v[200:201], %4\n\t"
                             "v_fma_f32 %1, v202, %5, %1\n\t"
                             "v_fma_f32 %2, v202, %6, %2\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "+v"(acc), "+v"(r0), "+v"(r1) : "v"(w), "v"(next), "v"(w[0]), "v"(w[1]) : "v200", "v201", "v202", "memory");
            } else {
                asm volatile("v_mov_b32 v202, v201\n\t"
                             "v_mov_b32 v203, v201\n\t"
                             "v_pk_fma_f32 %0, v[202:203], %3, %0\n\t"
                             "ds_read_b64 v[200:201], %4\n\t"
                             "v_fma_f32 %1, v202, %5, %1\n\t"
                             "v_fma_f32 %2, v202, %6, %2\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "+v"(acc), "+v"(r0), "+v"(r1) : "v"(w), "v"(next), "v"(w[0]), "v"(w[1]) : "v200", "v201", "v202", "v203", "memory");
            }
        }
        if (acc[0] != r0 || acc[1] != r1) {
            atomicAdd(&report[0], 1u);
            atomicAdd(&report[4 + (lane >> 4)], 1u);
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

extern "C" int pk_war_launch(int form, int blocks, int iters, const float* gsrc, unsigned* report, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (form == 0) hipLaunchKernelGGL(pk_war_kernel<0>, dim3(blocks), dim3(256), 0, st, iters, gsrc, report);
    else hipLaunchKernelGGL(pk_war_kernel<1>, dim3(blocks), dim3(256), 0, st, iters, gsrc, report);
    return (int)hipGetLastError();
}
