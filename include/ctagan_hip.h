/*
 * ctagan_hip.h -- C ABI of libctagan_hip.so, the gfx950 (MI355X) kernel library
 * behind the CTA-GAN G+D training-step hot path.
 *
 * The reference (yml-bit/CTA-GAN) has no native interface of its own: its hot
 * path is a chain of stock torch.nn layers (Model/HdGan.py, Model/CycleGan.py,
 * trainer/layers.py, trainer/reg.py, trainer/transformer.py) dispatched to
 * ATen/cuDNN.  Each entry point below names the reference call site(s) whose
 * ATen dispatch it replaces.  The Python binding a maintainer adds is the
 * ctypes table in cta_gan_amd/_lib.py.
 *
 * Conventions
 *   - plain device pointers, ints and a hipStream_t passed as void*; the library
 *     never allocates, frees or synchronises: buffers and workspaces are borrowed
 *     for the duration of the call, kernels are enqueued on `stream`.
 *   - return value: 0 ok, 1 invalid argument (CTG_EINVAL), 1000 + hipError_t.
 *   - activations are NHWC: [B][H][W][ld], `ld` >= C is the per-pixel pitch in
 *     ELEMENTS (a channel slice of a wider buffer needs no copy).
 *   - dtype: 0 = fp32, 1 = bf16 (storage type of activations / packed weights;
 *     accumulation, statistics and parameters are always fp32), 2 = split pair
 *     (the "bf16x3" mode's storage, accepted by the entry points that say so): a
 *     value x is two bf16 planes of its pixel row, hi = bf16(x) at p[c] and
 *     lo = bf16(x - hi) at p[c + ld / 2]; pointers and `ld` are in bf16 ELEMENTS,
 *     ld % 16 == 0, a dense tensor of C channels has ld = 2 C; 4 bytes per value
 *     like fp32, x = hi + lo to 2^-17, and both planes are ordinary bf16 NHWC
 *     tensors the MFMA kernels load without a conversion pass.  3 (ABI 11; ctg_in_bwd,
 *     ctg_in_bwd_partial, ctg_in_bwd_apply, ctg_in_bwd_stats, ctg_maxpool2_bwd only) = mixed:
 *     the SAVED FORWARD activation `x` is a split pair, every other operand plain bf16 --
 *     the plain bf16 backward of a split-pair forward ("bf16x3f") takes activation masks,
 *     the max-pool argmax and the InstanceNorm backward's xhat from hi + lo.
 *   - act: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 tanh, 4 sigmoid.  pad_mode: 0 zero, 1 reflect.
 *   - taps: ntaps ints, each (dy + 64) | (dx + 64) << 8 | weight_slice << 16.
 */
#ifndef CTAGAN_HIP_H
#define CTAGAN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever the signature or the meaning of an existing entry point changes: a binding compares it with the value it
 * was written against before it makes any other call (cta_gan_amd/_lib.py does), so a stale library is an error, not a
 * mis-typed call. */
#define CTG_ABI_VERSION 11
int ctg_abi_version(void);

/* ---- convolution: forward / backward-data / transposed, as one gather-GEMM ----
 * Y[n, j*os+oy0, i*os+ox0, co] = act(bias[co] + sum_t sum_ci X[n, pad(j*is+dy_t), pad(i*is+dx_t), ci] * W[t][co][ci])
 * for (j, i) in Hs x Ws (frame != 0: only the 1-pixel frame of that grid -- the ring of a padded-grid backward-data
 * pass whose interior is a second, tile-aligned call; stats_part must be NULL with it).  W is packed [slices][w_npad][Cin] (ctg_weight_pack).  out_f32 != 0 stores fp32
 * output (only for Cout <= 16).  Cin % 16 (fp32) / % 32 (bf16) == 0.
 * stats_part / stats_slabs_out (both may be NULL): when the call is served by the halo-resident kernel and has no
 * bias / activation, per-(sample, spatial tile, channel) partial (sum, sum of squares) of the results are written
 * to stats_part[B][slabs][Cout][2] and *stats_slabs_out = slabs (<= ceil(Hs/8)*ceil(Ws/16), the caller sizes the
 * buffer for that bound and reads it with the returned slab count); otherwise *stats_slabs_out = 0 and the caller runs ctg_in_stats.
 * epi->res / epi->fold (both may be NULL; unit-stride launches covering the whole output: os == is == 1, oy0 == ox0 == 0, Ho == Hs,
 * Wo == Ws, dtype-typed, served by the halo-resident kernel only -- CTG_EINVAL otherwise): res[B][Hs][Ws][res_ld] is added
 * to the rounded result (the skip gradient of a residual block, Model/HdGan.py:62); fold[B][Hs+2][Ws+2][fold_ld] is a
 * gradient on the 1-pixel reflection-padded grid of which only the frame is read and added to the interior pixels it
 * mirrors, so the backward-data pass of ReflectionPad2d(1) + Conv2d emits the unpadded gradient directly.
 * epi->bstats (bf16 launches with res / fold only): the emitted gradient g belongs to out = act(InstanceNorm(z)) [+ skip]
 * (Model/HdGan.py:54-59); the two sums of that InstanceNorm's backward are accumulated per (sample, tile, channel) while
 * g is stored -- ctg_in_bwd_stats then replaces ctg_in_bwd's own statistics pass; *stats_slabs_out returns the tile count.
 * dtype 2 (split pair, "bf16x3"): x is a split-pair tensor of Cin channels (Cin % 32 == 0), W the packed weights split along K
 * by ctg_split_weights -- 2 Cin bf16 per row, [w_hi 32 | w_lo 32] per 32 channels --, and each K step contracts
 * x_hi.w_hi + x_hi.w_lo + x_lo.w_hi on the bf16 matrix cores from ONE halo tile [hi 32 | lo 32] and one weight tile:
 * nn.Conv2d's fp32 product to ~1e-5 relative at a third of the bf16 MFMA rate.  out_f32 == 0: split-pair result (y_ld its
 * pitch; Cout % 8 == 0; epi->res / fold / bz split pairs too, and epi->bstats is served), 1: fp32 result.
 * Replaces: nn.Conv2d / nn.ConvTranspose2d (+ nn.ReflectionPad2d, bias, LeakyReLU / Tanh) forward and the
 * input-gradient half of their backward -- Model/HdGan.py:53-59,69-72,78-80,93-95,100-102,120-136,156-175;
 * Model/CycleGan.py:10-16,27-60,78-94; trainer/layers.py:85,97-104,282,295.                                  */
/* optional fused epilogue of ctg_conv_igemm (NULL = none); plain data, every member may be NULL / 0 */
typedef struct ctg_conv_epilogue {
    const void* res;      /* [B][Hs][Ws][res_ld] added to the result                                            */
    const void* fold;     /* [B][Hs+2][Ws+2][fold_ld] padded-grid gradient whose frame is folded into the result */
    const void* bz;       /* [B][Hs][Ws][bz_ld] input z of the InstanceNorm whose OUTPUT's gradient this launch emits */
    const float* bmean;   /* [B][Cout] mean / rstd of that InstanceNorm                                          */
    const float* brstd;
    float* bstats;        /* out: [B][tiles][Cout][2] partial (sum g m, sum g m xhat), tiles = the stats_slabs_out count */
    int res_ld, fold_ld, bz_ld;
    int bact;             /* activation fused behind that InstanceNorm: 0 none, 1 ReLU, 2 LeakyReLU(0.2)           */
    /* InstanceNorm of THIS launch's result in its epilogue: y = nie_act(IN(conv(x))) [+ res]; the conv result itself is never
     * stored (forward passes that keep nothing for a backward: Model/HdGan.py:53-63 under torch.no_grad(), HdTrainer.py:742-743).
     * nie_sync: 1 + nie_groups 64-bit words (nie_groups >= B * ceil(Cout / 128), else CTG_EINVAL), zeroed ONCE by the caller and then
     * owned by the library (monotonic arrival counters); one buffer per stream AND per nie_tiles = the group size the counters count
     * in: ceil(Hs / 16) * ceil(Ws / 16), or ceil(Hs / 8) * ceil(Ws / 16) for dtype 1 launches of fewer than 384 such tiles x
     * ceil(Cout / 128) x B (the 8-row tile variant; a mismatch is CTG_EINVAL).
     * Needs stats_part / stats_slabs_out; served for 3x3 unit-stride windows, Cout % 128 == 0, dtype 1 / 2 -- otherwise the call
     * returns 2 with nothing launched.
     *
     * What the in-launch exchange ASSUMES, and what happens when an assumption breaks:
     *  (1) Dispatch order.  The workgroups of a grid are placed in blockIdx order (x fastest, then y = the sample), and a placed
     *      workgroup keeps its slot until it ends.  A workgroup that waits for its sample's statistics therefore only waits for
     *      workgroups that were placed BEFORE any workgroup of a later sample -- provided every workgroup of ONE sample can be
     *      resident at once.  Inside a sample the ceil(Cout / 128) statistics groups are interleaved in dispatch order (the
     *      channel tile is the fastest tile index and the XCD-contiguous tile map deals the ids over eight runs), so the set that
     *      must fit is nie_tiles * ceil(Cout / 128) workgroups, not nie_tiles.
     *  (2) Residency.  That set must not exceed the chip's workgroup slots for this kernel (occupancy x compute units, queried
     *      per device) divided by the number of such launches that may wait at the same time -- streams of one process, or
     *      processes sharing the card: CTG_NIE_SHARE, default 2 (the reference-side callers issue these launches from one stream).  A launch that fails the test returns 2 (nothing launched; the
     *      caller runs the unfused conv + ctg_in_finalize + ctg_in_apply).  Kernels that never wait (everything else in this
     *      library) always drain, so they can delay a group but not starve it.
     *  (3) Memory ordering.  Tile moments and arrival counters are exchanged with relaxed AGENT-scope atomic accesses only
     *      (write-through stores, L2-bypassing loads); a workgroup waits for the acknowledgement of its moment stores
     *      (s_waitcnt vmcnt(0) + workgroup barrier) BEFORE its arrival increment, which is what orders "moments visible" before
     *      "arrival counted".  No release / acquire fence is used: on this multi-L2 part an agent-scope fence writes back and
     *      invalidates the whole L2 of the XCD (measured 4x on the launch).  The moments buffer (stats_part) must not be read by
     *      the caller as if it were coherent with its own cached loads -- ctg_in_finalize is not needed after such a launch.
     *  (4) Failure.  The wait is bounded: nie_budget polls of ~0.3 us (0 = the default 2^22, about a second).  A workgroup whose
     *      wait runs out stores NaN results for its tile and sets nie_sync[0] = 1.  The flag is sticky until the caller zeroes it;
     *      a caller MUST read it wherever it synchronises with the device anyway and treat a non-zero value as an error (the
     *      Python trainers raise and stop fusing for the rest of the process: cta_gan_amd/ops.py nie_check). */
    void* nie_sync;
    int nie_act;
    int nie_tiles;
    int nie_groups;       /* counter words behind nie_sync[0]                                                       */
    int nie_budget;       /* polls before a waiting workgroup gives up; 0 = default                                 */
} ctg_conv_epilogue;

int ctg_conv_igemm(int dtype, int out_f32, const void* x, const void* w, void* y, const float* bias,
                   int B, int Hi, int Wi, int Cin, int x_ld, int Ho, int Wo, int Cout, int y_ld,
                   int Hs, int Ws, int oy0, int ox0, int os, int is, int frame, int pad_mode, int act,
                   int w_npad, int ntaps, const int* taps_host, float* stats_part, int* stats_slabs_out,
                   const ctg_conv_epilogue* epi, void* stream);
/* The four parity classes of a stride-2 transposed conv (nn.ConvTranspose2d(k=3, s=2, p=1, output_padding=1), Model/HdGan.py:93-95)
 * or of the backward-data pass of a stride-2 conv (Model/HdGan.py:78-80, 124-131) in ONE launch: class q writes
 * out[2 j + cls_oy0[q], 2 i + cls_ox0[q]] = sum over its cls_ntaps[q] taps (taps = the classes' tap words back to back, encoded as
 * for ctg_conv_igemm), j < Hs, i < Ws.  The four workgroups of a spatial tile run back to back on one XCD and share the input
 * halo through L2.  bf16 in and out.  Returns 0, 1, 1000+hipError_t, or 2 = shape not served here (launch the classes one by one
 * with ctg_conv_igemm).  stats_part (optional, no bias / activation): B * 4 * ceil(Hs/8) * ceil(Ws/16) * Cout * 2 floats of
 * InstanceNorm partial moments, *stats_slabs_out = partials per sample. */
int ctg_conv_igemm_classes(int dtype, const void* x, const void* w, void* y, const float* bias, int B, int Hi, int Wi,
                           int Cin, int x_ld, int Ho, int Wo, int Cout, int y_ld, int Hs, int Ws, int pad_mode, int act,
                           int w_npad, const int* cls_ntaps, const int* cls_oy0, const int* cls_ox0, const int* taps,
                           float* stats_part, int* stats_slabs_out, void* stream);

/* ---- convolution weight gradient (split over pixel slabs, deterministic reduce) ----
 * part[z][t][m][c] = sum over slab z of G[n, j, i, m] * X[n, pad(j*is+dy_t), pad(i*is+dx_t), c];
 * part holds B*ceil(Hs*Ws/slab)*ntaps*Mc*Nc floats.  Mc, Nc % 32 == 0.
 * dtype 2 (split pairs, "bf16x3"): G_hi X_hi + G_hi X_lo + G_lo X_hi as three sweeps of every pixel tile into the same
 * accumulators (one partial, one launch: return 0) or, on small grids, in three workgroups writing three partials (return 3);
 * returns 2 when the shape is not served that way (the caller then makes three bf16 calls on the hi / lo plane views).  `part`
 * is sized three times as large in this mode.
 * Replaces: the weight-gradient half of convolution_backward for the same call sites.                      */
int ctg_conv_wgrad(int dtype, const void* g, const void* x, float* part, int B, int Hs, int Ws, int Mc, int g_ld,
                   int Hi, int Wi, int Nc, int x_ld, int is, int pad_mode, int slab, int ntaps,
                   const int* taps_host, void* stream);
/* dst[m*sm + c*sn + t*stp] (+)= sum_z part[z][t][m][c] for m < Mreal, c < Nreal */
int ctg_wgrad_reduce(const float* part, int Z, int ntaps, int Mc, int Nc, float* dst, int Mreal, int Nreal,
                     long sm, long sn, long stp, int accumulate, void* stream);
/* the same for `count` reductions given as parallel host arrays: every weight gradient of a network's backward */
int ctg_wgrad_reduce_multi(int count, const void* const* part, void* const* dst, const int* Z, const int* ntaps,
                           const int* Mc, const int* Nc, const int* Mreal, const int* Nreal, const long* sm,
                           const long* sn, const long* stp, const int* accumulate, void* stream);

/* ---- InstanceNorm2d(affine=False, eps=1e-5) fused with its neighbours ----
 * Replaces: nn.InstanceNorm2d + nn.ReLU / nn.LeakyReLU(0.2) + the residual add, forward and backward --
 * Model/HdGan.py:55-56,59,63,71-72,79-80,94-95,124-125,128-129,132-133,164,171-172; trainer/layers.py:14,282,295,299.
 * `part` = B*nslabs*C*2 floats scratch (nslabs <= 128); mean/rstd/s1/s2 = B*C floats.
 * pad > 0: `dout` lives on the reflection-padded grid (H+2pad, W+2pad) and is folded on load.                */
/* mean == NULL: only the partial moments part[B][nslabs][C][2] are produced (ctg_in_apply_part finalizes them) */
int ctg_in_stats(int dtype, const void* x, int x_ld, int B, int H, int W, int C, int nslabs, float* part,
                 float* mean, float* rstd, void* stream);
/* mode 0: mean / rstd from partial moments [B][nslabs][C][2] (any nslabs), e.g. those of ctg_conv_igemm; mode 1: the two plain
 * means (sum / HW) of the InstanceNorm backward from its partial sums */
int ctg_in_finalize(const float* part, int B, int C, int nslabs, int HW, int mode, float* mean, float* rstd, void* stream);
int ctg_in_apply(int dtype, const void* x, int x_ld, const float* mean, const float* rstd, int act,
                 const void* res, int r_ld, void* out, int o_ld, int B, int H, int W, int C, void* stream);
/* ctg_in_finalize + ctg_in_apply in one launch: (mean, rstd) come from the partial moments part[B][nslabs][C][2]
 * (nslabs <= 128; ctg_conv_igemm's stats_part or ctg_in_stats') in the kernel's prologue -- each workgroup owns one sample, one
 * group of 64 (bf16) / 32 (fp32) channels and a strip of pixels -- and are also written to mean / rstd [B][C] for the backward. */
int ctg_in_apply_part(int dtype, const void* x, int x_ld, const float* part, int nslabs, float* mean, float* rstd, int act,
                      const void* res, int r_ld, void* out, int o_ld, int B, int H, int W, int C, void* stream);
/* InstanceNorm backward, piecewise: the statistics pass (part[B][nslabs][C][2] = per-slab (sum g m, sum g m xhat), nslabs <= 128),
 * ctg_in_finalize(mode 1) into s1 / s2 [B][C], and the elementwise pass dx = rstd (g m - s1 - xhat s2); ctg_in_bwd = all three. */
int ctg_in_bwd_partial(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad, const float* mean,
                       const float* rstd, int act, int B, int H, int W, int C, int nslabs, float* part, void* stream);
int ctg_in_bwd_apply(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad, const float* mean,
                     const float* rstd, const float* s1, const float* s2, int act, void* dx, int dx_ld, int B, int H, int W,
                     int C, void* stream);
int ctg_in_bwd(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad, const float* mean,
               const float* rstd, int act, void* dx, int dx_ld, int B, int H, int W, int C, int nslabs,
               float* part, float* s1, float* s2, void* stream);
/* ctg_in_finalize(mode 1) + ctg_in_bwd_apply in one launch from partial sums part[B][nslabs <= 128][C][2] that ctg_in_bwd_partial
 * or a fused conv epilogue (ctg_conv_epilogue.bstats) produced. */
int ctg_in_bwd_stats(int dtype, const void* x, int x_ld, const void* dout, int d_ld, int pad, const float* mean,
                     const float* rstd, int act, void* dx, int dx_ld, int B, int H, int W, int C, int nslabs,
                     const float* part, void* stream);
/* out = a + fold(b), then * act'(yact) (yact = saved activation OUTPUT); any of a / b / yact may be NULL.
 * Replaces: autograd's gradient accumulation at fan-out points, ReflectionPad2d backward and the
 * LeakyReLU / Tanh backward (Model/HdGan.py:63,102,121; trainer/layers.py:60-62,299).                       */
int ctg_grad_combine(int dtype, const void* a, int a_ld, const void* b, int b_ld, int pad, const void* yact,
                     int y_ld, int act, void* out, int o_ld, int B, int H, int W, int C, void* stream);
/* out[B][H][W][C] = fold(dp[B][H+2p][W+2p][C]) for tiny-channel fp32 maps (ReflectionPad2d backward of the
 * generator's 7x7 head when its input carries a gradient: CycTrainer.py:153,156) */
int ctg_fold_f32(const float* dp, float* out, int B, int H, int W, int C, int pad, void* stream);
/* out[i] = g[i] * act'(y[i]) over n fp32 elements of any count (y = the saved activation OUTPUT): Tanh / LeakyReLU
 * backward of the 1-/2-channel fp32 maps (Model/HdGan.py:102 on images whose pixel count is not a multiple of 4) */
int ctg_act_bwd_f32(const float* g, const float* y, int act, float* out, long n, void* stream);
/* the same with sum_out[0] (+)= sum(out): Tanh backward of the generator's 1-channel output and the bias gradient of its
 * last conv (Model/HdGan.py:100-102) in one pass; part >= 4096 floats scratch */
int ctg_act_bwd_sum_f32(const float* g, const float* y, int act, float* out, long n, float* part, float* sum_out,
                        int accumulate, void* stream);
/* db[c] (+)= sum_{n,y,x} fold(g)[n,y,x,c]: bias gradient of convs not followed by an InstanceNorm */
int ctg_bias_grad(int dtype, const void* g, int g_ld, int pad, int B, int H, int W, int C, int Creal, int nslabs,
                  float* part, float* db, int accumulate, void* stream);

/* the two in one pass for a conv + bias + (Leaky)ReLU layer (trainer/layers.py:97-104): gout = fold(g) * act'(yact) (yact = the
 * layer's saved output) is written for the conv's backward passes and db[c] (+)= its sum; act in {1, 2} */
int ctg_bias_grad_act(int dtype, const void* g, int g_ld, int pad, const void* yact, int y_ld, int act, void* gout,
                      int go_ld, int B, int H, int W, int C, int Creal, int nslabs, float* part, float* db,
                      int accumulate, void* stream);

/* ---- registration U-Net plumbing: nn.MaxPool2d(2) (trainer/layers.py:172), F.interpolate(bilinear,
 * align_corners=False) x2 (trainer/reg.py:93), torch.cat (reg.py:77,94) ---- */
int ctg_maxpool2_fwd(int dtype, const void* x, int x_ld, void* out, int o_ld, int B, int H, int W, int C, void* stream);
int ctg_maxpool2_bwd(int dtype, const void* x, int x_ld, const void* dout, int d_ld, void* dx, int dx_ld,
                     int accumulate, int B, int H, int W, int C, void* stream);
int ctg_bilinear_fwd(int dtype, const void* x, int x_ld, void* out, int o_ld, int B, int Hi, int Wi, int Ho, int Wo,
                     int C, void* stream);
int ctg_bilinear_bwd(int dtype, const void* dout, int d_ld, void* dx, int dx_ld, int B, int Hi, int Wi, int Ho,
                     int Wo, int C, void* stream);
int ctg_copy_channels(int dtype, const void* src, int s_ld, void* dst, int d_ld, int C, long P, void* stream);
/* fp32 [P][x_ld] (C channels, C % 32 == 0) -> bf16 [P][2C]: packed weights as the operand of the split-bf16 ("bf16x3")
 * convolutions: hi = bf16(w), lo = bf16(w - hi), per 32 channels [hi 32 | lo 32] = one K step of ctg_conv_igemm(dtype 2),
 * ctg_conv_igemm_classes(dtype 2) and ctg_conv_smallcin(dtype 2). */
int ctg_split_weights(const float* x, long x_ld, void* out, int C, long P, void* stream);
/* ctg_split_weights for `count` dense packs (x_ld == C) given as parallel host arrays, ONE launch per 32 packs: every stale split of a
 * network after an optimiser step (Model/HdGan.py: 24 convs in the Generator alone) instead of one small launch per pack. */
int ctg_split_weights_multi(int count, const void* const* x, void* const* out, const int* C, const long* P, void* stream);
/* fp32 rows <-> split-pair rows (C channels of P pixels): dir 0: src fp32 (pitch s_ld floats) -> dst split pair (pitch d_ld bf16
 * elements); dir 1: src split pair -> dst fp32.  Where "bf16x3" tensors meet fp32 ones: wide network inputs / outputs at the
 * Python boundary (a stand-alone ResidualBlock, Model/HdGan.py:49-63; the feature maps Discriminator_m returns, :229-256). */
int ctg_pair_convert(int dir, const void* src, long s_ld, void* dst, long d_ld, int C, long P, void* stream);
/* packers that put 1-/2-channel tensors on the MFMA path (first / last layers) */
int ctg_chan_pad(int dtype, const float* src, int Cs, void* dst, int Cpad, long P, void* stream);
int ctg_im2col_pack(int dtype, const float* s0, const float* s1, int Cin, int B, int Hi, int Wi, int kh, int kw,
                    int stride, int pad, int pad_mode, void* dst, int Ho, int Wo, int Kpad, void* stream);
/* First-layer conv (Cin in {1,2} fp32 image planes [B][Hi][Wi]) with the im2col tile built in LDS: replaces
 * nn.Conv2d(1|2, C, k) at Model/HdGan.py:70,120,158, Model/CycleGan.py:28,78, trainer/reg.py:77 and the
 * input-gradient of the 1-channel tail conv (HdGan.py:110).  w = [w_npad][Kpad] from ctg_weight_pack of
 * weight.view(Cout, Cin*kh*kw); Cout <= 64; Kpad in {32, 64}; stats_part/stats_slabs_out as in ctg_conv_igemm. */
/* w_layout 0: k = (c*kh + ky)*kw + kx as above; 1 ("kx window": one plane, stride 1, kh, kw <= 8, Kpad 64, Cout > 32, dtype 1 / 2):
 * w[n][8 ky + kx] -- a chunk of the im2col row is then 8 consecutive pixels of one patch row and the tile is assembled from
 * register windows (the 7x7 head of the generator and the input gradient of its 7x7 tail).  dtype 2: y a split pair, w split by
 * ctg_split_weights. */
int ctg_conv_smallcin(int dtype, const float* s0, const float* s1, int Cin, int B, int Hi, int Wi, int kh, int kw,
                      int stride, int pad, int pad_mode, const void* w, int w_npad, int Kpad, int w_layout, const float* bias,
                      int act, void* y, int y_ld, int Ho, int Wo, int Cout, float* stats_part, int* stats_slabs_out,
                      void* stream);
/* The Generator's last layer forward: y[B][H][W] fp32 = act(bias + conv7x7(reflection_pad3(x))), x (dtype) [B][H][W][x_ld]
 * with 64 channels, ONE output channel; wp (dtype) = [NS][8][16][SC], SC = 32 (bf16) / 16 (fp32) channels per slice,
 * wp[s][j][o*8+ky][ci] = weight[0][SC*s+ci][ky][j-o] (0 <= j-o <= 6, ky <= 6, else 0).
 * Replaces ReflectionPad2d(3) + Conv2d(64, 1, 7) + Tanh (Model/HdGan.py:108-111). */
int ctg_conv_tail7(int dtype, const void* x, int x_ld, const void* wp, const float* bias, float* y, int act, int B, int H,
                   int W, void* stream);
/* The PatchGAN's last layer, Conv2d(512, 1, 4, padding=1) (Model/HdGan.py:136-137, 217-219; Model/CycleGan.py:104), on the vector
 * ALUs in fp32 (ABI 10; csrc/conv_cout1.hip): dtype 1 (bf16) / 2 (split pair), Cin = 512, k = 4, zero padding 0 <= pad < k, stride 1,
 * Ho = Hi + 2 pad - 3.  w = fp32 [16 taps][512] (w[ky*4+kx][ci] = weight[0][ci][ky][kx]: the master weights, unrounded).
 *   fwd:   y[B][Ho][Wo] fp32 = act(bias[0] + conv(x)), x (dtype) [B][Hi][Wi][x_ld]; replaces F.conv2d of that layer.
 *   bwd:   dx (dtype) [B][Hi][Wi][dx_ld] = the input gradient for g = dL/dy fp32 [B][Ho][Wo] (the dX half of convolution_backward).
 *   wgrad: part[Z][16][512] fp32, Z workgroups' shares of dW[tap][ci] = sum g x; finish with
 *          ctg_wgrad_reduce(part, Z, 1, 16, 512, dW, 16, 512, sm = 1, sn = 16, ...) into weight[0][ci][ky][kx]. */
int ctg_conv_cout1_fwd(int dtype, const void* x, int x_ld, const float* w, const float* bias, float* y, int act, int B, int Hi,
                       int Wi, int Cin, int k, int pad, int Ho, int Wo, void* stream);
int ctg_conv_cout1_bwd(int dtype, const float* g, const float* w, void* dx, int dx_ld, int B, int Hi, int Wi, int Cin, int k,
                       int pad, int Ho, int Wo, void* stream);
int ctg_conv_cout1_wgrad(int dtype, const float* g, const void* x, int x_ld, float* part, int Z, int B, int Hi, int Wi, int Cin,
                         int k, int pad, int Ho, int Wo, void* stream);
/* Weight gradient of the same first layers and of the 1-channel tail conv (HdGan.py:110), bf16:
 * part[(n*wgs + w)][m][k] = workgroup w's share of C[m][k] = sum_q Gpad[q][m] * Ipad[q + tap_k] over the grid
 * [0,Hs) x [0,Ws); Gpad[q] = g[pad_g(q - gpad)] (g bf16 [B][Gh][Gw][g_ld], Mc in {32,64} channels), Ipad[j] =
 * image[pad_i(j - ipad)] (1|2 fp32 planes), k = (c*kh + ky)*kw + kx < 64.  Finish with
 * ctg_wgrad_reduce(part, B*wgs, 1, Mc, 64, ...).  Replaces the weight-gradient half of convolution_backward.
 * g_ld < 0 (ABI 8): g is a split pair of pitch -g_ld (a multiple of 16, >= 2 Mc; lo plane -g_ld / 2 elements behind hi) and the
 * launch accumulates g_hi.I_hi + g_hi.I_lo + g_lo.I_hi with I_hi = bf16(I), I_lo = bf16(I - I_hi): each plane of g is read once. */
int ctg_corr_smallcin(const void* g, int Gh, int Gw, int g_ld, int Mc, int gpad, int g_pad_mode, const float* i0,
                      const float* i1, int Cin, int Ih, int Iw, int kh, int kw, int ipad, int i_pad_mode, int B,
                      int Hs, int Ws, float* part, int wgs, void* stream);
/* dst[t][n][k] = src[n*sn + k*sk + t*stp], zero padded to [ntaps][Npad][Kpad]; fp32 master -> dtype */
int ctg_weight_pack(int dtype, const float* src, long sn, long sk, long stp, int Nreal, int Kreal, void* dst,
                    int ntaps, int Npad, int Kpad, void* stream);
/* the same for `count` tensors given as parallel host arrays: all packs of a network after an optimiser step */
int ctg_weight_pack_multi(int dtype, int count, const void* const* src, void* const* dst, const long* sn,
                          const long* sk, const long* stp, const int* nreal, const int* kreal, const int* ntaps,
                          const int* npad, const int* kpad, void* stream);

/* ---- spatial transformer: Transformer_2D.forward (trainer/transformer.py:11-31) = pixel grid + flow ->
 * F.grid_sample(bilinear, align_corners=True, padding_mode="border"); 1-channel src, 2-channel flow given by
 * element strides (n, c, y, x).  dsrc is zeroed then scatter-added; dflow uses flow's strides. ---- */
int ctg_warp_fwd(const float* src, const float* flow, long fs_n, long fs_c, long fs_y, long fs_x, float* out,
                 int B, int H, int W, void* stream);
/* det_ws (may be NULL): B*H*W + 1 eight-byte words of workspace.  With it d_src is scattered in 64-bit fixed point (integer
 * atomics are associative: two runs are bit-identical whatever the arrival order) instead of with float atomics, whose result
 * depends on the order in the last bits -- the deterministic test / debug mode (CTG_DETERMINISTIC=1). */
int ctg_warp_bwd(const float* src, const float* flow, long fs_n, long fs_c, long fs_y, long fs_x,
                 const float* gout, float* dsrc, float* dflow, int B, int H, int W, void* det_ws, void* stream);

/* ---- losses.  `part` >= 4096 floats scratch; `out` / `gscale` are 1-element device buffers ----
 * smoothness: smooothing_loss (trainer/utils.py:165-173).  l1: nn.L1Loss (HdTrainer.py:721; CycTrainer.py:154,157);
 * with mask != NULL the stage-2 masked variant of HdTrainer.py:726-735.  avgpool: F.avg_pool2d over the whole
 * PatchGAN map (Model/HdGan.py:145,279,288).                                                                */
/* `weight`: the loss weight (Smooth_lamda / Corr_lamda1 / Corr_lamda2 of Yaml/HdGan.yaml:10-15) folded into the reduction's scale
 * (forward) and into the gradient (backward, on top of gscale) -- no scalar multiply launches around the loss */
int ctg_smooth_fwd(const float* f, long sn, long sc, long sy, long sx, int B, int C, int H, int W, float weight, float* part,
                   float* out, void* stream);
int ctg_smooth_bwd(const float* f, long sn, long sc, long sy, long sx, int B, int C, int H, int W, float weight,
                   const float* gscale, float* df, int accumulate, void* stream);
int ctg_l1_fwd(const float* a, const float* b, const float* mask, long n, float weight, float* part, float* out, void* stream);
int ctg_l1_bwd(const float* a, const float* b, const float* mask, long n, float weight, const float* gscale, float* da,
               int accumulate, void* stream);
/* LSGAN loss of GANLoss (Model/HdGan.py:276-285) on a 1-channel PatchGAN map x[B][HW], fused: out = sum_b s_b (mean(x[b]) - t_b)^2,
 * (t_b, s_b) = (t0, s0) for b < nb else (t1, s1) -- s = loss weight * w_i / group size; nb < B serves the fake and the real half
 * of the D step's batched pass (HdTrainer.py:745-747) at once.  pooled[B]: the per-sample means, kept for ctg_lsgan_bwd
 * (dx[b][i] = gscale * 2 s_b (pooled_b - t_b) / HW).  mode 0: the squared error above (nn.MSELoss, use_lsgan=True); mode 1:
 * nn.BCELoss's -(t log p + (1 - t) log(1 - p)), logs clamped at -100 (use_lsgan=False, Model/HdGan.py:266-267). */
int ctg_lsgan_fwd(const float* x, int B, int HW, int nb, float t0, float s0, float t1, float s1, int mode, float* pooled,
                  float* out, void* stream);
int ctg_lsgan_bwd(const float* pooled, int B, int HW, int nb, float t0, float s0, float t1, float s1, int mode,
                  const float* gscale, float* dx, void* stream);
/* out = scalars[0] + ... + scalars[count-1] (count <= 8 one-element device buffers): the sum of the step's loss terms
 * (HdTrainer.py:736) in one launch */
int ctg_sum_scalars(int count, const void* const* scalars, float* out, void* stream);
int ctg_avgpool_fwd(const float* x, int B, int HW, float* out, void* stream);
int ctg_avgpool_bwd(const float* gout, int B, int HW, float* dx, void* stream);

/* ---- evaluation path of test()/validation: to_windowdata (trainer/HdTrainer.py:41-64,
 * CycTrainer.py:34-57) over B slices of HW pixels with per-slice window centre / width; and the windowed + raw
 * MAE / PSNR / UQI of HdTrainer.py:1008-1050,1089-1125: out[B][2][3] doubles = {windowed, raw} x {MAE, PSNR, UQI};
 * part = B*nblk*20 doubles of workspace.  aliased = 1 reproduces trainer/CycTrainer.py:288-298, where `bb = b` / `cc = c`
 * are aliases and the windowed pair degenerates to the two +-1 masks. ---- */
int ctg_to_windowdata(const float* img, const float* wc, const float* ww, float* out, int B, long HW, void* stream);
/* Mean structural similarity of B slice pairs [B][H][W] (ABI 9): what skimage.measure.compare_ssim(x, y) returns with its
 * defaults (7x7 uniform window, sample covariance, K1 0.01, K2 0.03, float64) -- the validation pass of every trainer's train()
 * (trainer/HdTrainer.py:242-258, 765-781; CycTrainer.py:203-218; p2pTrainer.py:153-166; RegTrainer.py:206-221) and the SSIM /
 * SSIMw lines of test() (HdTrainer.py:1028, 1053).  data_range: 2 for the reference's float images.  mode 0: the pair as given,
 * out[B] doubles (wc / ww unused); mode 1: the two masked pairs ctg_window_metrics builds, out[B][2] = {windowed, raw}.
 * part: 2 * B * ceil((H-6)/16) * ceil((W-6)/16) doubles of workspace.  H, W >= 7. */
int ctg_ssim(const float* fake, const float* real, const float* wc, const float* ww, int B, int H, int W, int mode, int aliased,
             double data_range, double* part, double* out, void* stream);
int ctg_window_metrics(const float* fake, const float* real, const float* wc, const float* ww, int B, long HW,
                       int nblk, int aliased, double* part, double* out, void* stream);
/* input pipeline arithmetic (section 8f rank 2): read_ori_w after the DICOM read (trainer/datasets.py:36-71): raw HU
 * (int16, SimpleITK convention) -> windowed image (centre wc, width ww; the reference hard-codes 50 / 400) and the
 * full-range image, both in [-1, 1]; Resize = F.interpolate(mode="nearest") (trainer/utils.py:13-32) on B planes. */
int ctg_hu_to_inputs(const short* hu, float wc, float ww, float* win, float* full, long n, void* stream);
int ctg_resize_nearest(const float* src, int B, int Hi, int Wi, float* dst, int Ho, int Wo, void* stream);

/* ---- torch.optim.Adam(lr, betas=(0.5, 0.999)) step over `count` fp32 tensors (HdTrainer.py:612-616,738-739,751;
 * CycTrainer.py:67-73,162,178,197).  Host arrays of device pointers; `step` is 1-based. ---- */
int ctg_adam_step(int count, void* const* params, const void* const* grads, void* const* exp_avg,
                  void* const* exp_avg_sq, const long* numel, float lr, float beta1, float beta2, float eps,
                  int step, const float* dev_state3, void* stream);
/* graph-capturable form: dev_state3 = {step, 1-b1^step, sqrt(1-b2^step)} lives on the device, ctg_adam_tick advances
 * it by one step inside the stream, ctg_adam_step(..., step ignored, dev_state3) reads the corrections from it */
int ctg_adam_tick(float* dev_state3, float beta1, float beta2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTAGAN_HIP_H */
