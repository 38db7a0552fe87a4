// Shared device/host helpers for the CTA-GAN gfx950 kernel library.
//
// Layout convention for every activation tensor: NHWC ("channels last"),
// [B][H][W][ld] with `ld` >= C the per-pixel pitch in elements, so a channel
// slice of a wider buffer (the U-Net concat buffers of trainer/reg.py:91-95) is
// addressed without a copy.  Element type T is float or __bf16; reductions,
// statistics and accumulators are always fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "../../include/ctagan_hip.h"  // definitions are checked against the published C ABI

#define CTG_OK 0
#define CTG_EINVAL 1

// DT_MIX ("bf16x3f": split-pair forward, plain bf16 backward): the operand that is a SAVED FORWARD activation (`x` of the InstanceNorm
// backward and of the max-pool backward) is a split pair, every other operand plain bf16 -- activation masks and xhat are taken from
// the full-precision value hi + lo, where the hi plane alone flips ~0.3 % of the masks (an O(1) gradient change per flipped element)
enum { DT_F32 = 0, DT_BF16 = 1, DT_PAIR = 2, DT_MIX = 3 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };
enum { PAD_ZERO = 0, PAD_REFLECT = 1 };

typedef __bf16 bf16_t;
// Split-pair storage of the "bf16x3" compute mode (DT_PAIR): a value x lives as two bf16 planes of its pixel row,
// hi = bf16(x) at p[c] and lo = bf16(x - hi) at p[c + ld / 2] (ld = the row pitch in bf16 elements, 2 x the channels of the
// buffer) -- x = hi + lo to 2^-17 relative, 4 bytes per element like fp32, and BOTH planes are ordinary bf16 NHWC tensors the
// MFMA kernels load without a conversion pass: a convolution contracts hi.w_hi + hi.w_lo + lo.w_hi straight from them.
struct bfpair_t { bf16_t v; };
typedef float f32x4 __attribute__((ext_vector_type(4)));
// native vector for 16-byte register chunks: arrays of HIP's struct-based uint4 are not always
// promoted out of scratch by hipcc (ROCm 7.2), arrays of ext vectors are.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define LRELU_SLOPE 0.2f

// hipGetLastError() is sticky per thread: an unrelated, already-handled error of an earlier HIP call (e.g. a
// device-count probe made before the runtime was initialised) would otherwise be reported by our next launch.
#define CTG_ENTER() (void)hipGetLastError()

static inline int ctg_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CTG_OK : 1000 + (int)e;
}

// per-DEVICE launch prerequisites (a process may drive more than one GPU: `static int attr_set` would only serve the first)
static inline int ctg_cu_count() {
    static int n_cu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    dev &= 63;
    if (n_cu[dev] == 0) {
        int n = 0;
        n_cu[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return n_cu[dev];
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device and kernel; *mask is the caller's static bit set
static inline int ctg_lds_attr_once(const void* fn, int bytes, unsigned long long* mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (*mask & bit) return CTG_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return 1000 + (int)e;
    *mask |= bit;
    return CTG_OK;
}

template <typename T> struct VecOf;  // elements per 16-byte chunk
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };
template <> struct VecOf<bfpair_t> { static constexpr int N = 8; };

// ---- 16-byte chunk <-> fp32 lanes -----------------------------------------
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
    // (p, ld): the row pitch only matters to the split-pair specialisation below
    __device__ __forceinline__ void load(const float* p, int) { load(p); }
    __device__ __forceinline__ void store(float* p, int) const { store(p); }
    __device__ __forceinline__ void zero() { v[0] = v[1] = v[2] = v[3] = 0.f; }
};
template <> struct Chunk<bf16_t> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(t[i] << 16);
            v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
        *reinterpret_cast<bf16x8*>(p) = o;
    }
    __device__ __forceinline__ void load(const bf16_t* p, int) { load(p); }
    __device__ __forceinline__ void store(bf16_t* p, int) const { store(p); }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
};
// split-pair: 8 channels = one 16-byte chunk of the hi plane + one of the lo plane, ld / 2 elements further
template <> struct Chunk<bfpair_t> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const bfpair_t* p, int ld) {
        const u32x4 h = *reinterpret_cast<const u32x4*>(p);
        const u32x4 l = *reinterpret_cast<const u32x4*>(p + (ld >> 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(h[i] << 16) + __uint_as_float(l[i] << 16);
            v[2 * i + 1] = __uint_as_float(h[i] & 0xffff0000u) + __uint_as_float(l[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void store(bfpair_t* p, int ld) const {
        bf16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h[i] = (bf16_t)v[i];                       // RNE
            l[i] = (bf16_t)(v[i] - (float)h[i]);       // the remainder is exact in fp32
        }
        *reinterpret_cast<bf16x8*>(p) = h;
        *reinterpret_cast<bf16x8*>(p + (ld >> 1)) = l;
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
};
// planes of a dense row: a dense split-pair tensor of C channels has row pitch 2 C
template <typename T> struct PlanesOf { static constexpr int N = 1; };
template <> struct PlanesOf<bfpair_t> { static constexpr int N = 2; };

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)v; }

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_LRELU: return x > 0.f ? x : LRELU_SLOPE * x;
        case ACT_TANH: return tanhf(x);
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        default: return x;
    }
}
// derivative expressed through the activation's OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_LRELU: return y > 0.f ? 1.f : LRELU_SLOPE;
        case ACT_TANH: return 1.f - y * y;
        case ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) -- the index is a
// constant expression inside f, so it can feed instruction immediates (inline-asm "n" operands), not just array indices
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2): map the id so
// that every XCD owns ONE contiguous run of logical tiles (any grid size, not only multiples of 8).
#define IN_EPS 1e-5f      // nn.InstanceNorm2d / nn.BatchNorm2d default eps

__device__ __forceinline__ int xcd_contiguous(int id, int G) {
    const int x = id & 7, q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + (id >> 3);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every global
// load and STORE in flight -- in a persistent workgroup that exposes the store latency of the previous tile.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// XOR swizzle of the 16-byte chunk column inside an LDS tile row of KCH chunks (KCH = 8: 128-byte rows, KCH = 4:
// 64-byte rows): the 16 rows a fragment read touches spread over all banks without padding the rows.
// ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, (+32) (MI355X_MICROARCH.md, LDS):
// with these XORs a fragment read (lane -> row r0 + (lane & 15), chunk lane >> 4) is conflict-free for EVERY row offset r0 -- the
// halo kernels read pixel rows at the tap's column offset 0, 1, 2.  (Rounds 1-2 used (row >> 1) & 7 and -(row >> 2) & 3:
// conflict-free at offset 0 only, 2-way at offsets 1 and 2.)
template <int KCH> __device__ __forceinline__ int swz(int row, int c) {
    if constexpr (KCH == 4) return c ^ ((row >> 1) & 3);
    else return c ^ (row & 7);
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// sum over each 16-lane row of the wave with 4 DPP adds (row_shr 1, 2, 4, 8; zeros shift in): the total of a row
// ends up in its lane 15.  Pure VALU -- __shfl_xor lowers to ds_bpermute (LDS pipe) for these patterns.
__device__ __forceinline__ float row16_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

// wave-wide sum (64 lanes) via DPP-lowered shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
