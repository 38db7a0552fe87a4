// Host entry points of the split-pair ("bf16x3") sliding-window kernels (conv_pair_strips.hip: conv_striptp.h, ...), called by the
// dispatchers in conv_igemm.hip.  Each returns -1 when the launch is not its kernel's shape (the caller falls through to the
// halo / gather kernels), else CTG_OK / an error status like every launcher.
#pragma once
#include "conv_halo.h"

// ConvTranspose2d(128, 64, 3, s2) / backward-data of Conv2d(64, 128, 3, s2): a = the ConvArgs ctg_conv_igemm_classes built for a
// split-pair launch (4 classes); *tiles_out = InstanceNorm partial slabs per sample
int pairstrip_launch_t(const ConvArgs& a, hipStream_t st, int* tiles_out);

// Conv2d(64, 128, 3, s2) / backward-data of ConvTranspose2d(128, 64, 3, s2): a = the ConvArgs ctg_conv_igemm built for a split-pair
// launch (one stride-2 3x3 window); stats / *slabs_out as for launch_strips2
int pairstrip_launch_s2(const ConvArgs& a, float* stats, hipStream_t st, int* slabs_out);
