// The Generator's last layer: ReflectionPad2d(3) + Conv2d(64, 1, 7) + Tanh (Model/HdGan.py:108-111 ==
// Model/CycleGan.py:66-69), forward, bf16 or exact-fp32 operands (gfx950).
//
// One output channel leaves 15 of the 16 MFMA rows idle when the conv is run as "Cout padded to 16" (that launch
// took 0.82 ms for B=16 at 512^2, 2.6 % of the whole training step, twice per step).  Here the 16 rows of an MFMA
// tile carry TWO output columns x 7 kernel rows, the 16 MFMA columns are 16 consecutive image ROWS at one input
// column, and the sum over kernel rows is a lane shift afterwards:
//   D[o*8 + ky][y'] = sum_{j<8} sum_ci  Xpad[y'][x + j][ci] * A_j[o*8 + ky][ci],   A_j[o*8+ky] = w[ky][j - o] (or 0)
//   out[y][x + o]   = tanh(bias + sum_ky D[o*8 + ky][y + ky])
// A lane holds D[4g + r][y'] (g = lane >> 4): rows g = 0/1 are column x (ky 0-3 / 4-6), g = 2/3 column x+1; the
// shifts are DPP row_shl by ky inside the 16-lane rows, the g-pair sum one cross-row shuffle.  16 input rows give 10
// complete output rows, so 14/16 rows x 7/8 columns x 10/16 lanes of the MFMA work are useful (48 % instead of 6 %).
//
// Workgroup = 4 waves = 2 row groups x 2 column halves: 20 x 32 output pixels; the (26 x 38)-pixel reflection-padded
// halo of a 32-channel slice sits in LDS (LDS-DMA; row pitch 41 pixels and an XOR of the chunk index with
// (row >> 2) & 3 make the 16-rows-apart fragment reads conflict free); accumulators of all 8 column pairs of a wave
// persist over the two channel slices; each halo column is read once per slice and feeds up to four pairs.
#include "common.h"

#define TL_ROWS 20
#define TL_COLS 32
#define TL_HR (TL_ROWS + 6)     // 26 halo rows
#define TL_HC 41                // halo row pitch in pixels (38 used)
#define TL_HCU (TL_COLS + 6)    // 38 halo columns used

struct TailArgs {
    const void* x;       // T [B][H][W][x_ld], 64 channels
    const void* wp;      // T [NS slices][8 j][16 rows][SC ci]  (bf16: 2 x 32 channels; fp32: 4 x 16)
    const float* bias;   // 1 float or null
    float* y;            // fp32 [B][H][W]
    int B, H, W, x_ld, act;
    int pair_lo;         // PK: x is a split pair, its lo plane pair_lo elements behind; wp = [2 (hi, lo)][NS][8][16][SC]
};

typedef const __attribute__((address_space(1))) void* tl_gptr_t;
typedef __attribute__((address_space(3))) void* tl_lptr_t;
__device__ __attribute__((aligned(16))) unsigned g_tl_zero_chunk[4];

template <int N> __device__ __forceinline__ float dpp_row_shl(float v) {
    if constexpr (N == 0) return v;
    else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}

// PK (T = bf16, the "bf16x3" mode): split-pair input and split weights, x_hi.w_hi + x_hi.w_lo + x_lo.w_hi -- four halo passes
// (hi plane slices with both weight halves, lo plane slices with the hi half) instead of two.
template <typename T, bool PK = false>
__global__ __launch_bounds__(256, 2) void conv_tail_kernel(const TailArgs a) {
    constexpr int EPC = VecOf<T>::N;          // elements per 16-byte chunk
    constexpr int SC = 4 * EPC;               // channels per slice: 4 chunks = 64 bytes per halo pixel
    constexpr int NS = 64 / SC;               // slices
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HSLOTS = TL_HR * TL_HC * 4;                 // 16-byte chunks of one halo slice (SC channels)
    char* sH = smem;
    float* sO = reinterpret_cast<float*>(smem + HSLOTS * 16);  // [TL_ROWS][TL_COLS] raw sums

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int n = blockIdx.y;
    const int tx_n = (a.W + TL_COLS - 1) / TL_COLS;
    const int sp = xcd_contiguous(blockIdx.x, gridDim.x);
    const int Y0 = (sp / tx_n) * TL_ROWS, X0 = (sp % tx_n) * TL_COLS;
    const T* __restrict__ X = (const T*)a.x + (size_t)n * a.H * a.W * a.x_ld;
    const int H = a.H, W = a.W, x_ld = a.x_ld;

    auto issue_halo = [&](int slice) __attribute__((always_inline)) {
        for (int base = 0; base < HSLOTS; base += 256) {
            if (base + 64 * wave < HSLOTS) {          // wave-uniform
                const int sl = base + tid;
                const int pix = sl >> 2;
                const int hr = pix / TL_HC, hc = pix - hr * TL_HC;
                const int kc = (sl & 3) ^ ((hr >> 2) & 3);
                const int iy = reflect_idx(Y0 + hr - 3, H), ix = reflect_idx(X0 + hc - 3, W);
                const bool ok = sl < HSLOTS && hc < TL_HCU && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                // (slice < 0: slice -1 - slice of the split pair's lo plane)
                const int coff = slice >= 0 ? slice * SC : a.pair_lo + (-1 - slice) * SC;
                const T* src = ok ? X + ((size_t)(iy * W + ix) * x_ld + coff + kc * EPC) : (const T*)g_tl_zero_chunk;
                __builtin_amdgcn_global_load_lds((tl_gptr_t)src, (tl_lptr_t)(sH + (base + 64 * wave) * 16), 16, 0, 0);
            }
        }
    };

    f32x4 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int hrow = wr * 10 + (lane & 15);                     // halo row of this lane's MFMA column
    const int kcl = (lane >> 4) ^ ((hrow >> 2) & 3);            // swizzled chunk of its k-group
    const char* bbase = sH + ((hrow * TL_HC + wc * 16) * 4 + kcl) * 16;
    const T* __restrict__ WP = (const T*)a.wp;

#pragma unroll 1
    for (int pass = 0; pass < (PK ? 2 * NS : NS); ++pass) {
        const int slice = PK ? pass % NS : pass;
        const bool lo_plane = PK && pass >= NS;
        if (pass) __syncthreads();                 // every wave is done reading the previous slice
        issue_halo(lo_plane ? -1 - slice : slice);
        u32x4 fa[8];                               // A_j fragments of this slice: lane = (row lane&15, k-group lane>>4)
        u32x4 fl[PK ? 8 : 1];                      // PK: the weights' lo halves (used with the input's hi plane)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            fa[j] = *reinterpret_cast<const u32x4*>(WP + ((slice * 8 + j) * 16 + (lane & 15)) * SC + (lane >> 4) * EPC);
            if constexpr (PK)
                fl[j] = *reinterpret_cast<const u32x4*>(WP + (((NS + slice) * 8 + j) * 16 + (lane & 15)) * SC + (lane >> 4) * EPC);
        }
        __syncthreads();                           // halo landed (vmcnt 0)
#pragma unroll
        for (int c = 0; c < 22; ++c) {             // wave-local input columns: outputs 0..15 need columns 0..21
            const u32x4 fb = *reinterpret_cast<const u32x4*>(bbase + c * 64);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int j = c - 2 * p;
                if (j >= 0 && j < 8) {
                    if constexpr (sizeof(T) == 2) {
                        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[j]),
                                                                         __builtin_bit_cast(bf16x8, fb), acc[p], 0, 0, 0);
                        if constexpr (PK) {
                            if (!lo_plane)
                                acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fl[j]),
                                                                                 __builtin_bit_cast(bf16x8, fb), acc[p], 0, 0, 0);
                        }
                    } else {   // exact fp32: the 4 floats of a chunk feed 4 MFMAs (same K order for A and B)
                        const f32x4 va = __builtin_bit_cast(f32x4, fa[j]), vb = __builtin_bit_cast(f32x4, fb);
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
                            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[qq], vb[qq], acc[p], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- sum over kernel rows: lane i of a 16-lane row needs D[..ky..][i + ky]
    const bool odd_g = (lane >> 4) & 1;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const f32x4 v = acc[p];
        float s = odd_g ? dpp_row_shl<4>(v[0]) : v[0];
        s += odd_g ? dpp_row_shl<5>(v[1]) : dpp_row_shl<1>(v[1]);
        s += odd_g ? dpp_row_shl<6>(v[2]) : dpp_row_shl<2>(v[2]);
        s += odd_g ? 0.f : dpp_row_shl<3>(v[3]);                  // ky = 7 does not exist
        s += __shfl_down(s, 16, 64);                               // g = 0 += g = 1, g = 2 += g = 3
        const int i = lane & 15;
        if (i < 10 && (lane & 16) == 0) {
            const int o = lane >> 5;
            sO[(wr * 10 + i) * TL_COLS + wc * 16 + 2 * p + o] = s;
        }
    }
    __syncthreads();
    const float bias = a.bias != nullptr ? a.bias[0] : 0.f;
    float* __restrict__ Y = a.y + (size_t)n * H * W;
    for (int e = tid; e < TL_ROWS * TL_COLS; e += 256) {
        const int r = e / TL_COLS, c = e - r * TL_COLS;
        const int oy = Y0 + r, ox = X0 + c;
        if (oy < H && ox < W) Y[(size_t)oy * W + ox] = act_apply(sO[e] + bias, a.act);
    }
}

// C ABI.  y[B][H][W] (fp32) = act(bias + conv7x7(reflection_pad3(x))) for x [B][H][W][x_ld] (dtype) with 64 channels and ONE
// output channel; wp (dtype) = [NS][8][16][SC] with SC = 32 (bf16) / 16 (fp32) channels per slice, NS = 64 / SC:
// wp[s][j][o*8 + ky][ci] = weight[0][SC s + ci][ky][j - o] for 0 <= j - o <= 6, ky <= 6, else 0.  Replaces nn.ReflectionPad2d(3) + nn.Conv2d(64, output_nc = 1, 7) + nn.Tanh() (Model/HdGan.py:108-111).
extern "C" int ctg_conv_tail7(int dtype, const void* x, int x_ld, const void* wp, const float* bias, float* y, int act,
                              int B, int H, int W, void* stream) {
    CTG_ENTER();
    if (dtype != DT_BF16 && dtype != DT_F32 && dtype != DT_PAIR) return CTG_EINVAL;
    if (x == nullptr || wp == nullptr || y == nullptr || B < 1 || H < 4 || W < 4) return CTG_EINVAL;
    // DT_PAIR: x a split pair (x_ld its pitch), wp = [2][2][8][16][32] bf16: the packed operand's hi halves, then its lo halves
    if (dtype == DT_PAIR && (x_ld % 16 || x_ld < 128)) return CTG_EINVAL;
    const int epc = dtype == DT_F32 ? 4 : 8;
    if (x_ld % epc || x_ld < 64 || ((uintptr_t)x & 15) || ((uintptr_t)wp & 15)) return CTG_EINVAL;
    if ((long)H * W * x_ld >= (1L << 31)) return CTG_EINVAL;
    TailArgs a;
    a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.B = B; a.H = H; a.W = W; a.x_ld = x_ld; a.act = act;
    a.pair_lo = x_ld / 2;
    const int smem = TL_HR * TL_HC * 4 * 16 + TL_ROWS * TL_COLS * 4;
    static unsigned long long m0 = 0, m1 = 0, m2 = 0;       // per device
    {
        int rc = ctg_lds_attr_once((const void*)conv_tail_kernel<bf16_t>, smem, &m0);
        if (rc == CTG_OK) rc = ctg_lds_attr_once((const void*)conv_tail_kernel<float>, smem, &m1);
        if (rc == CTG_OK) rc = ctg_lds_attr_once((const void*)(conv_tail_kernel<bf16_t, true>), smem, &m2);
        if (rc != CTG_OK) return rc;
    }
    const int tiles = ((H + TL_ROWS - 1) / TL_ROWS) * ((W + TL_COLS - 1) / TL_COLS);
    if (dtype == DT_PAIR) hipLaunchKernelGGL((conv_tail_kernel<bf16_t, true>), dim3(tiles, B), dim3(256), smem, (hipStream_t)stream, a);
    else if (dtype == DT_BF16) hipLaunchKernelGGL(conv_tail_kernel<bf16_t>, dim3(tiles, B), dim3(256), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_tail_kernel<float>, dim3(tiles, B), dim3(256), smem, (hipStream_t)stream, a);
    return ctg_launch_status();
}
