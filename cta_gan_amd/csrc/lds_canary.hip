// Diagnostic (round 5): a workgroup that fills its LDS allocation with a pattern and keeps re-reading it.  Run on a second stream beside
// any other launch, it reports every word of ITS allocation that something else changed -- i.e. a kernel on the same CU writing LDS
// outside its own allocation (LDS-DMA destinations are computed by hand in this library).  tests/test_lds_canary_gpu.py runs it beside
// the training step of every mode.
#include "common.h"
#include "../../include/ctagan_hip_diag.h"      // diagnostics: not part of the product ABI

#define CANARY_WORDS 2048      // 8 KB per workgroup: small enough to be resident beside the 50-150 KB workgroups of the conv kernels

__global__ __launch_bounds__(256) void lds_canary_kernel(int spins, unsigned* __restrict__ report, int report_cap) {
    __shared__ unsigned words[CANARY_WORDS];
    const unsigned tag = 0xC0DE0000u;
    for (int i = threadIdx.x; i < CANARY_WORDS; i += 256) words[i] = tag | (unsigned)i;
    __syncthreads();
    for (int s = 0; s < spins; ++s) {
        // broadcast reads (every lane the same word): the access pattern of a kernel that shares a small table through LDS
        for (int i = 0; i < 304; ++i) {
            const unsigned v = *reinterpret_cast<volatile unsigned*>(&words[i]);
            if (v != (tag | (unsigned)i)) {
                const unsigned slot = atomicAdd(&report[1], 1u);
                if ((int)slot < report_cap && (threadIdx.x & 63) == 63) {
                    unsigned* r = report + 4 + 4 * slot;
                    r[0] = (unsigned)i | 0x80000000u;
                    r[1] = v;
                    r[2] = (unsigned)s;
                    r[3] = blockIdx.x * 256 + threadIdx.x;
                }
            }
        }
        for (int i = threadIdx.x; i < CANARY_WORDS; i += 256) {
            const unsigned v = *reinterpret_cast<volatile unsigned*>(&words[i]);
            if (v != (tag | (unsigned)i)) {
                const unsigned slot = atomicAdd(&report[0], 1u);
                if ((int)slot < report_cap) {
                    unsigned* r = report + 4 + 4 * slot;
                    r[0] = (unsigned)i;
                    r[1] = v;
                    r[2] = (unsigned)s;
                    r[3] = blockIdx.x;
                }
                words[i] = tag | (unsigned)i;       // re-arm
            }
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

// report: 4 + 4 * report_cap unsigned words, zeroed by the caller: report[0] = number of foreign writes seen, then (word index, value
// found, spin, workgroup) per event.  blocks workgroups of 256 threads and 8 KB of LDS each; spins x ~1 us of residence.
extern "C" int ctg_lds_canary(int blocks, int spins, unsigned* report, int report_cap, void* stream) {
    CTG_ENTER();
    if (blocks < 1 || blocks > 65535 || spins < 1 || report == nullptr || report_cap < 0) return CTG_EINVAL;
    hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, spins, report, report_cap);
    return ctg_launch_status();
}
