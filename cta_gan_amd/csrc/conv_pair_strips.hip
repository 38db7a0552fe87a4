// Split-pair ("bf16x3") sliding-window kernels of the generator's stride-2 layers at 512^2 / 256^2 (Model/HdGan.py:78-80, 93-95), in
// their own translation unit: the kernels are self-contained (a few seconds to compile) while conv_igemm.hip carries every
// instantiation of the halo / gather kernels.
#define g_zero_chunk g_ps_zero_chunk      // this unit's own zero page (device globals are per translation unit without -fgpu-rdc)
#include "conv_pair_strips.h"
#include "conv_striptp.h"
#include "conv_strips2p.h"

__device__ __attribute__((aligned(16))) unsigned g_zero_chunk[4];

int pairstrip_launch_t(const ConvArgs& a, hipStream_t st, int* tiles_out) { return launch_striptp(a, st, tiles_out); }
int pairstrip_launch_s2(const ConvArgs& a, float* stats, hipStream_t st, int* slabs_out) { return launch_strips2p(a, stats, st, slabs_out); }
