"""Layer-level executor of the hot path: a small define-by-run tape over the HIP ops.

Why not one torch.autograd.Function per op: the kernels exchange things torch's
autograd has no slot for -- gradients that live on a reflection-PADDED grid and
are folded by their consumer, raw conv outputs that are normalised on the fly,
channel slices of shared concat buffers -- and a whole network runs as ONE
autograd node (nets.py), so torch only sees the network boundary.

`Act` is an activation handle: `.t` is the physical NHWC tensor view
`[B, H, W, C]`, `.grad` a `(tensor, pad)` pair accumulated during backward.
Every function here records its backward closure on the tape when gradients are
needed; `Tape.backward()` replays them in reverse.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from . import dp, ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, PAD_REFLECT, PAD_ZERO, pack_tap


class Act:
    __slots__ = ("t", "grad", "req", "moments", "in_src", "in_plain", "grad_stats")

    def __init__(self, t: torch.Tensor, req: bool = False):
        self.t = t          # [B, H, W, C] NHWC view
        self.grad = None    # (tensor, pad) or None
        self.req = req      # does anything upstream want d/d(this)?
        self.moments = None # (partials [B, slabs, C, 2], slabs) emitted by the producing conv's epilogue
        self.in_src = None      # (z, mean, rstd, act) when this activation is act(InstanceNorm(z)) [+ skip]
        self.in_plain = False   # ... and no skip was added: self.t IS act(xhat)
        self.grad_stats = None  # (gradient tensor, partial IN-backward sums) from the conv epilogue that wrote that gradient

    @property
    def shape(self):
        return tuple(self.t.shape)


class Tape:
    def __init__(self, enabled: bool):
        self.enabled = enabled
        self.nodes: List[Callable[[], None]] = []

    def record(self, fn: Callable[[], None]):
        if self.enabled:
            self.nodes.append(fn)

    def backward(self):
        for fn in reversed(self.nodes):
            fn()
        self.nodes = []


# ----------------------------------------------------------------------------- gradient plumbing
# Developer experiment ("split forward, bf16 backward"): numerically emulate a backward whose gradient tensors
# are stored in bf16 and whose backward-data convs use the bf16 half of the weights, inside the split-pair mode and with its kernels
# (so the cost is unchanged; only the arithmetic is what the cheaper backward would compute).  1 = gradients only, 2 = also w_lo = 0.
_EMU_BF16_BWD = int(os.environ.get("CTG_EMU_BF16_BWD", "0") or 0)


def add_grad(a: Act, g: torch.Tensor, pad: int = 0):
    """Accumulate a gradient (optionally on the reflection-padded grid) into `a`."""
    if not a.req:
        return
    if a.grad is None:
        a.grad = (g, pad)
        return
    g0, p0 = a.grad
    if p0 and pad:  # both on a padded grid (does not occur on this path): fold one first
        g0, p0 = _fold(g0, p0, a.t.shape), 0
    out = ops.empty_act(a.t.shape, g.dtype, g.device)
    if p0:
        ops.grad_combine(g, g0, p0, None, ACT_NONE, out)
    else:
        ops.grad_combine(g0, g, pad, None, ACT_NONE, out)
    a.grad = (out, 0)


def _fold(g, pad, shape):
    out = ops.empty_act(shape, g.dtype, g.device)
    ops.grad_combine(None, g, pad, None, ACT_NONE, out)
    return out


def take_grad(a: Act, allow_pad: bool = False):
    """Gradient of `a` as (tensor, pad); folded to the unpadded grid unless the consumer folds on load."""
    if a.grad is None:
        return None, 0
    g, pad = a.grad
    a.grad = None
    if _EMU_BF16_BWD and ops.is_pair(g):
        ops.pair_lo(g).zero_()      # experiment: what a bf16-STORED gradient would hand its consumer (hi = RNE bf16 of the value)
    if pad and not allow_pad:
        return _fold(g, pad, a.t.shape), 0
    return g, pad


# ----------------------------------------------------------------------------- weight packing cache
class PackCache:
    """Packed (dtype-converted, tap-major) copies of the fp32 master weights, keyed by the parameter's
    version counter: re-packed only after an optimiser step really changed the parameter.

    Packs made through `get_pack` remember their recipe, so `refresh()` (called at the start of a network's forward)
    re-packs every stale one of them in ONE multi-tensor launch instead of ~50 tiny ones spread over the step."""

    def __init__(self):
        self.store = {}

    @staticmethod
    def _ver(param):
        return (param._version, getattr(param, "_ctg_version", 0))  # the HIP Adam bumps _ctg_version

    def get(self, param: torch.Tensor, kind: str, dtype, maker):
        key = (id(param), kind, dtype)
        ver = self._ver(param)
        hit = self.store.get(key)
        if hit is not None and hit[0] == ver and hit[2] == param.data_ptr():
            return hit[1]
        val = maker()
        self.store[key] = (ver, val, param.data_ptr(), None, None)
        return val

    def get_pack(self, param: torch.Tensor, kind: str, dtype, recipe):
        """recipe = (ntaps, nreal, kreal, npad, kpad, sn, sk, stp) of ops.weight_pack."""
        key = (id(param), kind, dtype)
        ver = self._ver(param)
        hit = self.store.get(key)
        if hit is not None and hit[0] == ver and hit[2] == param.data_ptr():
            return hit[1]
        ntaps, nreal, kreal, npad, kpad, sn, sk, stp = recipe
        val = ops.weight_pack(param, dtype, ntaps, nreal, kreal, npad, kpad, sn, sk, stp)
        self.store[key] = (ver, val, param.data_ptr(), recipe, param)
        return val

    def refresh(self):
        """Re-pack, in one launch per dtype, every remembered pack whose parameter changed since it was made."""
        stale = {}
        for key, (ver, val, ptr, recipe, param) in self.store.items():
            if recipe is None or param is None:
                continue
            nv = self._ver(param)
            if nv != ver and ptr == param.data_ptr() and param.is_contiguous():
                stale.setdefault(key[2], []).append((key, nv, val, recipe, param))
        for dtype, items in stale.items():
            ops.weight_pack_multi([(param.detach(), val, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7])
                                   for (_, _, val, r, param) in items])
            for key, nv, val, recipe, param in items:
                self.store[key] = (nv, val, param.data_ptr(), recipe, param)
            if ops.PAIR and dtype == torch.float32:
                # the split-bf16 copies of these packs (ops.split_w_pair) are stale now: all of them re-split in one launch
                # (one small launch per pack and step otherwise: ~80 launches, 0.7 ms of the split-pair step)
                ops.split_w_pair_refresh([val for (_, _, val, _, _) in items])


_NO_IN_FUSE = bool(os.environ.get("CTG_NO_IN_FUSE"))   # A/B switch (scripts/ab.sh)
_NO_BIAS_ACT = bool(os.environ.get("CTG_NO_BIAS_ACT"))   # A/B switch
_NO_FRAME = bool(os.environ.get("CTG_NO_FRAME"))   # A/B switch (scripts/ab.sh)


def _round_up(v, m):
    return (v + m - 1) // m * m


def _bn_for(cout):
    return 128 if cout > 64 else 64 if cout > 32 else 32 if cout > 16 else 16


# ----------------------------------------------------------------------------- convolution
@dataclass
class ConvSpec:
    """One nn.Conv2d / nn.ConvTranspose2d of the reference with its fused neighbours."""
    cin: int
    cout: int
    k: int
    stride: int = 1
    pad: int = 0
    reflect: bool = False        # ReflectionPad2d(pad) in front instead of zero padding
    transposed: bool = False     # ConvTranspose2d(k=3, s=2, p=1, output_padding=1)
    use_bias: bool = True        # False: an affine-free InstanceNorm follows, the bias cancels exactly
    act: int = ACT_NONE          # activation fused into the epilogue (only when no norm follows)
    out_f32: bool = False        # final 1-/2-channel maps are kept in fp32

    @property
    def kk(self):
        return self.k * self.k


def _taps_fwd(spec: ConvSpec):
    return [pack_tap(ky - spec.pad, kx - spec.pad, ky * spec.k + kx) for ky in range(spec.k) for kx in range(spec.k)]


def _taps_1d_transposed(k, pad, parity):
    """(ky, d) pairs of a stride-2 transposed gather: out index 2*j + parity reads in index j + d."""
    return [(ky, (parity + pad - ky) // 2) for ky in range(k) if (parity + pad - ky) % 2 == 0]


def _convT_classes(k, pad):
    """Parity classes of `out[2j+py, 2i+px] = sum in[j+dy, i+dx] * W[ky, kx]` (stride-2 transposed conv)."""
    classes = []
    for py in (0, 1):
        for px in (0, 1):
            taps = [pack_tap(dy, dx, ky * k + kx)
                    for ky, dy in _taps_1d_transposed(k, pad, py) for kx, dx in _taps_1d_transposed(k, pad, px)]
            classes.append((py, px, taps))
    return classes


def _pack_dtype(dtype):
    """dtype of the packed weight copies: fp32 in the split-pair mode (the conv wrapper splits the pack into its bf16 halves,
    cached on the pack; the exact-f32 first-layer / tail kernels use it as it is)."""
    return torch.float32 if (ops.PAIR and dtype == torch.bfloat16) else dtype


def _pack_fwd(cache: PackCache, spec: ConvSpec, w: torch.Tensor, dtype, kpad=None):
    """[tap][Cout_pad][Cin] for the forward gather-GEMM."""
    dtype = _pack_dtype(dtype)
    npad = _round_up(spec.cout, _bn_for(spec.cout))
    kk = spec.kk
    if kpad is not None:  # im2col-packed first layer: one slice, K = Cin*k*k padded
        return cache.get_pack(w, "fwd_packed", dtype, (1, spec.cout, spec.cin * kk, npad, kpad, spec.cin * kk, 1, 0)), npad
    if spec.transposed:   # master (Cin, Cout, kh, kw)
        return cache.get_pack(w, "fwd", dtype, (kk, spec.cout, spec.cin, npad, spec.cin, kk, spec.cout * kk, 1)), npad
    return cache.get_pack(w, "fwd", dtype, (kk, spec.cout, spec.cin, npad, spec.cin, spec.cin * kk, kk, 1)), npad


def _pack_bwd(cache: PackCache, spec: ConvSpec, w: torch.Tensor, dtype, kpad=None):
    """[tap][Cin_pad][Cout(_pad)] for the backward-data gather-GEMM (N = Cin, K = Cout)."""
    dtype = _pack_dtype(dtype)
    npad = _round_up(spec.cin, _bn_for(spec.cin))
    kk = spec.kk
    kdim = spec.cout if kpad is None else kpad
    if spec.transposed:   # master (Cin, Cout, kh, kw): element (n=ci, k=co, t)
        return cache.get_pack(w, "bwd", dtype, (kk, spec.cin, spec.cout, npad, kdim, spec.cout * kk, kk, 1)), npad
    return cache.get_pack(w, "bwd", dtype, (kk, spec.cin, spec.cout, npad, kdim, kk, spec.cin * kk, 1)), npad


def conv_out_hw(spec: ConvSpec, h, w):
    if spec.transposed:
        return 2 * h, 2 * w
    return (h + 2 * spec.pad - spec.k) // spec.stride + 1, (w + 2 * spec.pad - spec.k) // spec.stride + 1


def _cout1(spec, dtype):
    return ops.cout1_ok(spec.cin, spec.cout, spec.k, spec.stride, spec.reflect, spec.pad, spec.transposed, dtype)



def conv_forward(tape: Tape, cache: PackCache, spec: ConvSpec, x: Act, weight, bias, dtype,
                 img_sources=None) -> Act:
    """y = act(conv(x) + bias).  `img_sources` = (s0, s1|None) dense fp32 [B,H,W] images for the Cin<=2
    first layers (then `x` is ignored for compute and only carries the gradient request)."""
    dev = weight.device
    pad_mode = PAD_REFLECT if spec.reflect else PAD_ZERO
    b_eff = bias if spec.use_bias else None
    odt = torch.float32 if spec.out_f32 else dtype
    packed_x = None
    moments = None
    # which parameter gradients the backward owes is decided NOW: a `_frozen(net)` context around the forward has been
    # left again (requires_grad restored) by the time loss.backward() replays the tape
    preq = (bool(weight.requires_grad), bool(bias is not None and bias.requires_grad))
    if img_sources is not None:
        s0, s1 = img_sources
        bsz, hi, wi = s0.shape
        ho, wo = conv_out_hw(spec, hi, wi)
        kpad = _round_up(spec.cin * spec.kk, 32)  # multiple of 32: legal K for igemm (both dtypes) and N for wgrad
        wp, npad = _pack_fwd(cache, spec, weight, dtype, kpad=kpad)
        y = ops.empty_act((bsz, ho, wo, spec.cout), odt, dev)

        def packed_x():   # the im2col matrix only exists for the weight gradient (built when the backward asks)
            return ops.im2col_pack(s0, s1, spec.k, spec.stride, spec.pad, pad_mode, dtype, kpad)
        packed_x.sources = (s0, s1)
        if ops.smallcin_ok(spec.cin, spec.cout, spec.k, dtype, odt, spec.stride):
            # im2col tile assembled in LDS: no packed detour through HBM
            kxw = ops.kxw_ok(spec.cin, spec.cout, spec.k, spec.stride, dtype) and odt == dtype
            if kxw:      # the 7x7 head: weights in kx-window order (8 ky + kx), the tile assembled from register windows
                pdt = _pack_dtype(dtype)
                wp = cache.get(weight, "fwd_kxw", pdt, lambda: ops.kxw_pack(weight.reshape(spec.cout, spec.kk), spec.k, npad, pdt))
            moments = ops.conv_smallcin(s0, s1, spec.k, spec.stride, spec.pad, pad_mode, wp, npad, b_eff, spec.act, y,
                                        spec.cout, want_stats=not spec.use_bias, kxw=kxw)
        else:
            px = packed_x()
            packed_x = lambda: px
            packed_x.sources = (s0, s1)
            ops.conv_igemm(px, wp, npad, y, b_eff, spec.cout, ho, wo, 0, 0, 1, 1, PAD_ZERO, spec.act,
                           [pack_tap(0, 0, 0)])
    else:
        bsz, hi, wi, cin = x.t.shape
        assert cin == spec.cin, (cin, spec.cin)
        ho, wo = conv_out_hw(spec, hi, wi)
        y = ops.empty_act((bsz, ho, wo, spec.cout), odt, dev)
        if spec.out_f32 and ops.conv_tail7_ok(spec.cin, spec.cout, spec.k, spec.stride, spec.reflect, spec.pad, dtype, hi, wi):
            # the 64 -> 1 channel 7x7 tail: column pairs x kernel rows on the MFMA rows (csrc/conv_tail.hip)
            tdt = "pair" if (ops.PAIR and dtype == torch.bfloat16) else dtype
            wp7 = cache.get(weight, "tail7", tdt, lambda: ops.tail7_pack(weight, tdt))
            ops.conv_tail7(x.t, wp7, b_eff, y, spec.act)
            out = Act(y, req=tape.enabled)
            if tape.enabled:
                tape.record(lambda: _conv_backward(cache, spec, x, out, weight, bias, dtype, None, preq))
            return out
        if spec.out_f32 and _cout1(spec, dtype):
            # the PatchGAN's 512 -> 1 channel 4x4 last layer: a matrix-vector product per pixel, on the vector ALUs in fp32
            # (csrc/conv_cout1.hip) instead of a GEMM with one live column
            w16 = cache.get(weight, "cout1", torch.float32, lambda: ops.cout1_pack(weight))
            ops.conv_cout1_fwd(x.t, w16, b_eff, y, spec.act, spec.pad)
            out = Act(y, req=tape.enabled)
            if tape.enabled:
                tape.record(lambda: _conv_backward(cache, spec, x, out, weight, bias, dtype, None, preq))
            return out
        wp, npad = _pack_fwd(cache, spec, weight, dtype)
        if spec.transposed:
            # four parity classes of the output; each emits the InstanceNorm moments of ITS pixels (concatenated below)
            classes = _convT_classes(spec.k, spec.pad)
            merged = ops.conv_igemm_classes(x.t, wp, npad, y, b_eff, spec.cout, hi, wi, classes, PAD_ZERO, spec.act,
                                            want_stats=not spec.use_bias)
            if merged is not None:
                # all four classes in one launch: their workgroups share the input halo through L2
                if merged[0] is not None and merged[1] > 0:
                    moments = merged
            else:
                parts = []
                for py, px, taps in classes:
                    parts.append(ops.conv_igemm(x.t, wp, npad, y, b_eff, spec.cout, hi, wi, py, px, 2, 1, PAD_ZERO,
                                                spec.act, taps, want_stats=not spec.use_bias))
                if all(p is not None and p[1] > 0 for p in parts):
                    moments = (torch.cat([p[0] for p in parts], dim=1), sum(p[1] for p in parts))
        else:
            # a conv without live bias / activation feeds an InstanceNorm: ask for its moments from the epilogue
            moments = ops.conv_igemm(x.t, wp, npad, y, b_eff, spec.cout, ho, wo, 0, 0, 1, spec.stride, pad_mode,
                                     spec.act, _taps_fwd(spec), want_stats=not spec.use_bias)
    out = Act(y, req=tape.enabled)
    if moments is not None and moments[1] > 0:
        out.moments = moments
    if tape.enabled:
        tape.record(lambda: _conv_backward(cache, spec, x, out, weight, bias, dtype, packed_x, preq))
    return out


def _conv_backward(cache, spec: ConvSpec, x: Act, out: Act, weight, bias, dtype, packed_x, preq=(True, True)):
    wreq, breq = preq
    g, _ = take_grad(out)
    if g is None:
        return
    dev = weight.device
    bsz, ho, wo, cout = out.t.shape
    pad_mode = PAD_REFLECT if spec.reflect else PAD_ZERO
    kk = spec.kk
    # 1. through the fused epilogue activation
    tail_db = fused_db = None
    if (spec.act != ACT_NONE and spec.out_f32 and cout == 1 and spec.use_bias and bias is not None and breq):
        # 1-channel fp32 output (the generator's Tanh tail): the activation backward and the bias gradient (the plain sum of
        # the masked gradient) in one pass
        tail_db = _grad_like(bias)
        g = ops.act_bwd_sum_f32(g.contiguous(), out.t.contiguous(), spec.act, tail_db)
    elif spec.act != ACT_NONE:
        if spec.out_f32 and g.numel() % 4:
            # odd-sized 1-/2-channel fp32 maps (the networks' own sizes are multiples of 4): the 16-byte-chunk kernel
            # does not apply, the scalar one does
            gg = ops.act_bwd_f32(g.contiguous(), out.t.contiguous(), spec.act)
        elif spec.out_f32:
            gg = torch.empty_like(g)
            n4 = g.numel() // 4
            ops.grad_combine(g.view(1, 1, n4, 4), None, 0, out.t.view(1, 1, n4, 4), spec.act, gg.view(1, 1, n4, 4))
        elif (spec.act in (ACT_RELU, ACT_LRELU) and spec.use_bias and bias is not None and breq and not _NO_BIAS_ACT
              and g.shape[-1] % 8 == 0):
            # conv + bias + LeakyReLU (Reg's plain convs, D's first layer): the masked gradient and the bias gradient (its
            # per-channel sum) in ONE pass over g and the saved output
            gg = ops.empty_like_act(g)
            fused_db = _grad_like(bias)
            ops.bias_grad_act(g, 0, out.t, spec.act, gg, cout, fused_db)
        else:
            gg = ops.empty_like_act(g)
            ops.grad_combine(g, None, 0, out.t, spec.act, gg)
        g = gg
    # 2. tiny-channel outputs ride the MFMA kernels zero-padded to 32 channels -- except the 1-channel 7x7 tail, whose
    #    gradients are "image x wide tensor" correlations served straight from the fp32 gradient plane
    # (the input gradient alone also goes that way in the split-pair mode, whose weight gradient has no image-correlation kernel)
    tail_dx_small = (spec.out_f32 and cout == 1 and spec.reflect and packed_x is None and spec.k * spec.k <= 64
                     and spec.stride == 1 and ops.smallcin_ok(1, spec.cin, spec.k, dtype, dtype))
    tail_small = tail_dx_small and ops.corr_smallcin_ok(1, spec.cin, spec.k, spec.stride, dtype)
    tail_dx_small = tail_small or (tail_dx_small and ops.PAIR)
    cout1 = spec.out_f32 and packed_x is None and _cout1(spec, dtype)
    if cout1:
        # the PatchGAN's last layer: bias gradient, weight gradient and input gradient straight from the fp32 gradient plane
        g1 = g.contiguous()
        if spec.use_bias and bias is not None and breq:
            db = tail_db
            if db is None:
                db = _grad_like(bias)
                torch.sum(g1.reshape(1, -1), dim=1, out=db)
            _store_param_grad(bias, db)
        if wreq:
            dw = _grad_like(weight)
            ops.conv_cout1_wgrad(g1, x.t, dw, spec.pad, defer=_REDUCE_JOBS)
            _store_param_grad(weight, dw)
        if x.req:
            bsz, hi, wi, cin = x.t.shape
            dx = ops.empty_act((bsz, hi, wi, cin), dtype, dev)
            ops.conv_cout1_bwd(g1, cache.get(weight, "cout1", torch.float32, lambda: ops.cout1_pack(weight)), dx, spec.pad)
            add_grad(x, dx, 0)
        return
    if tail_small:
        gm, m_c = None, cout
    elif spec.out_f32:
        gm = ops.chan_pad(g.contiguous(), cout, dtype, 32)
        m_c = 32
    else:
        gm = g
        m_c = cout
    # 3. bias gradient (only where the bias is live)
    if spec.use_bias and bias is not None and breq:
        if tail_db is not None or fused_db is not None:
            db = tail_db if tail_db is not None else fused_db      # already summed above
        elif tail_small:
            db = _grad_like(bias)          # (1,): the data-parallel bucket slot when an exchange is active
            torch.sum(g.reshape(1, -1), dim=1, out=db)
        else:
            db = _grad_like(bias)
            ops.bias_grad(gm, 0, cout, db)
        _store_param_grad(bias, db)
    # 4. weight gradient
    if wreq:
        dw = _grad_like(weight)
        if tail_small:
            # dW[0][ci][ky][kx] = sum_q rpad(X)[q][ci] * zpad(dY)[q + (2p-ky, 2p-kx)]: taps come out flipped (48 - k)
            p = spec.pad
            ops.corr_smallcin(x.t, p, PAD_REFLECT, g.reshape(bsz, ho, wo), None, spec.k, 2 * p, PAD_ZERO,
                              ho + 2 * p, wo + 2 * p, dw, kk - 1, spec.cin, kk, kk, -1, defer=_REDUCE_JOBS)
        elif packed_x is not None and ops.corr_smallcin_ok(spec.cin, cout, spec.k, spec.stride, dtype) and not spec.out_f32:
            s0, s1 = packed_x.sources
            ops.corr_smallcin(gm, 0, PAD_ZERO, s0, s1, spec.k, spec.pad, pad_mode, ho, wo, dw, 0, cout,
                              spec.cin * kk, spec.cin * kk, 1, defer=_REDUCE_JOBS)
        elif packed_x is not None:
            ops.conv_wgrad(gm, packed_x(), [pack_tap(0, 0, 0)], 1, PAD_ZERO, dw, cout, spec.cin * kk,
                           spec.cin * kk, 1, 0, defer=_REDUCE_JOBS)
        elif spec.transposed:
            # roles swap: G = layer input (Cin, on its own grid), X = dL/dy read at (2*iy - pad + ky)
            taps = [pack_tap(ky - spec.pad, kx - spec.pad, ky * spec.k + kx)
                    for ky in range(spec.k) for kx in range(spec.k)]
            ops.conv_wgrad(x.t, gm, taps, 2, PAD_ZERO, dw, spec.cin, cout, cout * kk, kk, 1, defer=_REDUCE_JOBS)
        else:
            # 1-/2-channel gradients: only the first 16 of the 32 zero-padded channels carry data
            g_w = gm[..., :16] if (spec.out_f32 and dtype == torch.bfloat16 and spec.stride == 1 and spec.k == 7
                                   and spec.cin % 64 == 0 and ho >= 8 and wo >= 16) else gm
            ops.conv_wgrad(g_w, x.t, _taps_fwd(spec), spec.stride, pad_mode, dw, cout, spec.cin, spec.cin * kk, kk, 1,
                           defer=_REDUCE_JOBS)
        _store_param_grad(weight, dw)
    # 5. input gradient
    if not x.req:
        return
    if packed_x is not None:
        # d/d(image): a (transposed) conv with Cout_eff = Cin <= 2, fp32 result [B, Hi, Wi, Cin]
        bsz, hi, wi, cin = x.t.shape
        wb, npad = _pack_bwd(cache, spec, weight, dtype, kpad=m_c if spec.out_f32 else None)
        if spec.reflect:
            # padded-grid gradient, then the scalar fold (CycleGAN back-propagates through G's input)
            p = spec.pad
            assert spec.stride == 1
            dxp = torch.empty((bsz, hi + 2 * p, wi + 2 * p, cin), dtype=torch.float32, device=dev)
            taps = [pack_tap(-ky, -kx, ky * spec.k + kx) for ky in range(spec.k) for kx in range(spec.k)]
            ops.conv_igemm(gm, wb, npad, dxp, None, cin, hi + 2 * p, wi + 2 * p, 0, 0, 1, 1, PAD_ZERO, ACT_NONE, taps)
            add_grad(x, ops.fold_f32(dxp, p), 0)
            return
        dx = torch.empty((bsz, hi, wi, cin), dtype=torch.float32, device=dev)
        _bwd_data_launch(spec, gm, wb, npad, dx, hi, wi, cin)
        add_grad(x, dx, 0)
        return
    bsz, hi, wi, cin = x.t.shape
    if tail_dx_small:
        # dX_pad[q][ci] = sum_k' zpad(dY)[q + k' - 2p] * W[0][ci][flipped k']: a first-layer-style conv of the gradient plane
        p = spec.pad
        kxw = ops.kxw_ok(1, cin, spec.k, 1, dtype)
        if kxw:
            wflip = cache.get(weight, "tail_bwd_kxw", _pack_dtype(dtype), lambda: ops.kxw_pack(
                weight.detach()[0].flip(-1, -2).reshape(cin, kk), spec.k, _round_up(cin, 32), _pack_dtype(dtype)))
        else:
            wflip = cache.get(weight, "tail_bwd_small", _pack_dtype(dtype), lambda: ops.weight_pack(
                weight.detach()[0].flip(-1, -2).reshape(cin, kk).contiguous(), _pack_dtype(dtype), 1, cin, kk, _round_up(cin, 32), 64,
                kk, 1, 0))
        dxp = ops.empty_act((bsz, hi + 2 * p, wi + 2 * p, cin), dtype, dev)
        ops.conv_smallcin(g.reshape(bsz, ho, wo), None, spec.k, 1, 2 * p, PAD_ZERO, wflip, _round_up(cin, 32), None,
                          ACT_NONE, dxp, cin, kxw=kxw)
        add_grad(x, dxp, p)
        return
    wb, npad = _pack_bwd(cache, spec, weight, dtype, kpad=m_c if spec.out_f32 else None)
    if spec.transposed:
        # dX[iy] = sum_ky dY[2*iy - pad + ky] * W[ci, co, ky]: a stride-2 forward-style gather over dY
        dx = ops.empty_act((bsz, hi, wi, cin), dtype, dev)
        taps = [pack_tap(ky - spec.pad, kx - spec.pad, ky * spec.k + kx) for ky in range(spec.k) for kx in range(spec.k)]
        ops.conv_igemm(gm, wb, npad, dx, None, cin, hi, wi, 0, 0, 1, 2, PAD_ZERO, ACT_NONE, taps)
        add_grad(x, dx, 0)
    elif spec.reflect:
        # gradient w.r.t. the reflection-PADDED input; the consumer folds it (norm_act.hip: fold_load)
        p = spec.pad
        dxp = ops.empty_act((bsz, hi + 2 * p, wi + 2 * p, cin), dtype, dev)
        assert spec.stride == 1
        taps = [pack_tap(-ky, -kx, ky * spec.k + kx) for ky in range(spec.k) for kx in range(spec.k)]
        if p == 1 and hi % 16 == 0 and wi % 16 == 0 and hi >= 32 and wi >= 32 and not _NO_FRAME:
            # the padded grid (H+2) is two pixels off the 16x16 tiling of the halo kernel (17 ragged tiles of 81 at
            # 128^2): tile-aligned interior on the halo kernel, the 1-pixel frame as one small gather launch
            taps_in = [pack_tap(p - ky, p - kx, ky * spec.k + kx) for ky in range(spec.k) for kx in range(spec.k)]
            if ops.conv_fusable(cin, hi, wi):
                # frame first; the interior launch then folds it in from its epilogue and adds the gradient already
                # waiting at x (the skip path of a residual block): the unpadded, complete gradient leaves the conv
                # and no combine pass follows
                ops.conv_igemm(gm, wb, npad, dxp, None, cin, hi + 2 * p, wi + 2 * p, 0, 0, 1, 1, PAD_ZERO, ACT_NONE, taps,
                               frame=True)
                res = None
                if x.grad is not None and x.grad[1] == 0 and x.grad[0].dtype == dtype \
                        and tuple(x.grad[0].shape) == (bsz, hi, wi, cin):
                    res, x.grad = x.grad[0], None
                dx = ops.empty_act((bsz, hi, wi, cin), dtype, dev)
                # x = act(IN(z)) [+ skip] and this launch writes its complete gradient: the sums of that InstanceNorm's
                # backward are taken while the gradient is stored (bf16; `grad_stats` is dropped if another gradient is
                # accumulated onto x later, see inorm_forward)
                want_in = (x.in_src is not None and x.grad is None and dtype == torch.bfloat16 and not _NO_IN_FUSE)
                in_arg = x.in_src if want_in else None
                if want_in and ops.PAIR_BWD_ACTIVE:
                    # bf16x3f: this (plain bf16) launch would take the mask and xhat of the sums from z's hi plane, and a mask
                    # taken from bf16(z) flips where |z - mean| is below z's bf16 rounding (~0.3 % of the elements).  For
                    # out = ReLU(xhat) the saved OUTPUT is the operand instead: its hi plane is > 0 exactly where xhat is, and where
                    # it is it equals xhat to bf16 rounding (statistics 0 / 1) -- the rounding of xhat only enters the two SUMS, where
                    # it averages out over the sample's pixels (the elementwise pass, DT_MIX, uses z = hi + lo).  Without an
                    # activation there is no mask and z's hi plane serves; anything else runs the separate DT_MIX statistics pass.
                    zact = x.in_src[3]
                    if zact == ACT_RELU and x.in_plain:
                        in_arg = (x.t, _unit_stats(x.t, 0.0), _unit_stats(x.t, 1.0), ACT_RELU)
                    elif zact != ACT_NONE:
                        want_in, in_arg = False, None
                part, slabs = ops.conv_igemm(gm, wb, npad, dx, None, cin, hi, wi, 0, 0, 1, 1, PAD_ZERO, ACT_NONE, taps_in,
                                             res=res, fold=dxp, in_bwd=in_arg)
                add_grad(x, dx, 0)
                if want_in and slabs > 0:
                    x.grad_stats = (dx, part)
                return
            ops.conv_igemm(gm, wb, npad, dxp, None, cin, hi, wi, p, p, 1, 1, PAD_ZERO, ACT_NONE, taps_in)
            ops.conv_igemm(gm, wb, npad, dxp, None, cin, hi + 2 * p, wi + 2 * p, 0, 0, 1, 1, PAD_ZERO, ACT_NONE, taps,
                           frame=True)
        else:
            ops.conv_igemm(gm, wb, npad, dxp, None, cin, hi + 2 * p, wi + 2 * p, 0, 0, 1, 1, PAD_ZERO, ACT_NONE, taps)
        add_grad(x, dxp, p)
    else:
        dx = ops.empty_act((bsz, hi, wi, cin), dtype, dev)
        _bwd_data_launch(spec, gm, wb, npad, dx, hi, wi, cin)
        add_grad(x, dx, 0)


_UNIT_STATS = {}


def _unit_stats(t, value):
    """[B, C] fp32 statistics filled with `value` (mean 0 / rstd 1: "the operand already is xhat"), cached per shape and device."""
    key = (t.device, t.shape[0], t.shape[3], value)
    hit = _UNIT_STATS.get(key)
    if hit is None:
        hit = _UNIT_STATS[key] = torch.full((t.shape[0], t.shape[3]), value, dtype=torch.float32, device=t.device)
    return hit


def _bwd_data_launch(spec: ConvSpec, gm, wb, npad, dx, hi, wi, cin):
    """Backward-data of a zero-padded conv (stride 1 or 2) as gather-GEMM launches over dY."""
    if spec.stride == 1:
        taps = [pack_tap(spec.pad - ky, spec.pad - kx, ky * spec.k + kx) for ky in range(spec.k) for kx in range(spec.k)]
        ops.conv_igemm(gm, wb, npad, dx, None, cin, hi, wi, 0, 0, 1, 1, PAD_ZERO, ACT_NONE, taps)
    else:
        assert spec.stride == 2
        classes = _convT_classes(spec.k, spec.pad)
        if hi % 2 == 0 and wi % 2 == 0 and dx.dtype == gm.dtype and \
                ops.conv_igemm_classes(gm, wb, npad, dx, None, cin, hi // 2, wi // 2, classes, PAD_ZERO, ACT_NONE) is not None:
            return     # the four parity classes of the input gradient in one launch
        for py, px, taps in classes:
            hs, ws = (hi - py + 1) // 2, (wi - px + 1) // 2
            if hs > 0 and ws > 0 and taps:
                ops.conv_igemm(gm, wb, npad, dx, None, cin, hs, ws, py, px, 2, 1, PAD_ZERO, ACT_NONE, taps)
            elif hs > 0 and ws > 0:
                ops.zero_act(dx[:, py::2, px::2, :])


def _grad_like(param):
    """Where a parameter's gradient is written: its slot in a data-parallel gradient bucket when an exchange is active
    (dp.GradSync: the all-reduce then runs in place, nothing is packed), else a fresh tensor."""
    if dp._ACTIVE:
        buf = dp.grad_buffer(param)
        if buf is not None:
            return buf
    return torch.empty_like(param)


def fire_mark(net, tag):
    """A point of `net`'s backward after which every parameter gradient produced so far is final: with a data-parallel
    exchange active the queued split-K reductions are flushed and the buckets this completes start their all-reduce
    (overlapping the rest of the backward); otherwise nothing happens."""
    if dp._ACTIVE and (tag == "done" or dp.wants_mark(net, tag)):
        flush_reduces()
        dp.fire_mark(net, tag)


_PARAM_GRADS = None  # set by nets._NetFn while a network's backward runs
_REDUCE_JOBS = None  # pending split-K reductions of that backward (summed in one launch at its end)


def flush_reduces():
    """Sum every queued weight-gradient partial (one multi-tensor launch); called at the end of a network's backward."""
    if _REDUCE_JOBS:
        ops.wgrad_reduce_multi(_REDUCE_JOBS)
        del _REDUCE_JOBS[:]


def _store_param_grad(param, grad):
    if _PARAM_GRADS is None:
        raise RuntimeError("parameter gradient produced outside a network backward")
    prev = _PARAM_GRADS.get(id(param))
    if prev is not None:
        flush_reduces()      # a shared parameter: both gradients must be complete before they are added
    _PARAM_GRADS[id(param)] = grad if prev is None else prev + grad


# ----------------------------------------------------------------------------- instance norm (+act, +residual)
_INPLACE_NOGRAD = not __import__("os").environ.get("CTG_NO_INPLACE_IN")      # A/B switch


def conv_inorm_forward(tape: Tape, cache: PackCache, spec: ConvSpec, x: Act, weight, bias, dtype, act: int,
                       res: Optional[Act] = None, out_t: Optional[torch.Tensor] = None) -> Act:
    """act(IN(conv(x))) [+ res] -- conv_forward + inorm_forward, as ONE launch where nothing has to be kept for a backward pass
    (a forward under torch.no_grad(): trainer/HdTrainer.py:742-743, the test loops) and the shape is `ops.conv_in_fusable`: the
    workgroups of a sample exchange their tile moments and normalise the accumulators in registers, so the conv result is never
    stored or re-read (csrc/conv_halo.h, NIE)."""
    if (not tape.enabled) and (not spec.use_bias) and spec.act == ACT_NONE and (not spec.transposed) and (not spec.out_f32) \
            and x.t.shape[-1] == spec.cin:
        bsz, hi, wi, _ = x.t.shape
        ho, wo = conv_out_hw(spec, hi, wi)
        if ops.conv_in_fusable(x.t, spec.cin, spec.cout, spec.k, spec.stride, ho, wo) and (ho, wo) == (hi, wi):
            wp, npad = _pack_fwd(cache, spec, weight, dtype)
            o = out_t if out_t is not None else ops.empty_act((bsz, ho, wo, spec.cout), dtype, weight.device)
            r = ops.conv_igemm(x.t, wp, npad, o, None, spec.cout, ho, wo, 0, 0, 1, 1, PAD_REFLECT if spec.reflect else PAD_ZERO,
                               ACT_NONE, _taps_fwd(spec), want_stats=True, res=res.t if res is not None else None, in_fuse=act)
            if r is not None:
                return Act(o, req=False)
    y = conv_forward(tape, cache, spec, x, weight, bias, dtype)
    return inorm_forward(tape, y, act, res=res, out_t=out_t)


def inorm_forward(tape: Tape, y: Act, act: int, res: Optional[Act] = None, out_t: Optional[torch.Tensor] = None) -> Act:
    """out = act(IN(y)) [+ res].  `out_t` lets the result land in a slice of a concat buffer."""
    if y.moments is not None:
        part, nsl = y.moments
        y.moments = None
    else:
        part, nsl = ops.in_partial(y.t)
    if out_t is not None:
        o = out_t
    elif not tape.enabled and _INPLACE_NOGRAD:
        o = y.t      # nothing reads the conv result again: normalise it where it lies (the lines are still in the memory-side cache)
    else:
        o = ops.empty_like_act(y.t)
    if ops.fin_fusable(nsl):
        # the elementwise kernel finalizes the partial moments of its channel group in its prologue: no finalize launch
        mean, rstd = ops.in_apply_part(y.t, part, act, res.t if res is not None else None, o)
    else:
        mean, rstd = ops.in_finalize(part, nsl, y.t.shape[1] * y.t.shape[2])
        ops.in_apply(y.t, mean, rstd, act, res.t if res is not None else None, o)
    out = Act(o, req=tape.enabled)
    if tape.enabled:
        out.in_src = (y.t, mean, rstd, act)
        out.in_plain = res is None

        def bwd():
            g, pad = take_grad(out, allow_pad=True)
            st, out.grad_stats = out.grad_stats, None
            if g is None:
                return
            if res is not None and res.req:
                # the skip branch sees the same gradient (folded if it came from a reflect-padded conv)
                if pad:
                    g = _fold(g, pad, out.t.shape)
                    pad = 0
                add_grad(res, g, 0)
            if y.req:
                dy = ops.empty_like_act(y.t)
                if st is not None and st[0] is g and pad == 0:
                    # the conv that wrote g already summed (g m, g m xhat) in its epilogue: no statistics pass
                    ops.in_bwd_stats(y.t, g, mean, rstd, act, dy, st[1])
                else:
                    ops.in_bwd(y.t, g, pad, mean, rstd, act, dy)
                add_grad(y, dy, 0)
        tape.record(bwd)
    return out


def bnorm_forward(tape: Tape, y: Act, act: int, bn) -> Act:
    """out = act(BatchNorm2d(y)) for nn.BatchNorm2d's defaults (affine, running statistics, eps 1e-5, momentum 0.1) --
    NLayerDiscriminator's own norm_layer default (Model/HdGan.py:149; every trainer overrides it with InstanceNorm, :208).
    Built on the InstanceNorm kernels: the per-(sample, slab) moments are finalized over the WHOLE batch, and the affine map is
    folded into the statistics the elementwise kernels take -- gamma xhat + beta = (y - mean') rstd' with rstd' = gamma rstd,
    mean' = mean - beta / rstd' -- so `in_apply` writes act(gamma xhat + beta) and the backward kernels see the true
    pre-activation (their ReLU mask) while the batch-wide terms of the BatchNorm backward enter through s1 / s2."""
    b, h, w, c = y.t.shape
    n = b * h * w
    training = bool(bn.training)
    gamma, beta = bn.weight.detach(), bn.bias.detach()
    with torch.no_grad():
        if training:
            part, nsl = y.moments if y.moments is not None else ops.in_partial(y.t)
            y.moments = None
            mean, rstd = ops.in_finalize(part.reshape(1, b * nsl, c, 2), b * nsl, n)      # [1, C]: over samples and pixels
            mean, rstd = mean[0], rstd[0]
            var = 1.0 / (rstd * rstd) - bn.eps
            bn.running_mean.mul_(1.0 - bn.momentum).add_(bn.momentum * mean)
            bn.running_var.mul_(1.0 - bn.momentum).add_(bn.momentum * var * (n / max(n - 1, 1)))
            bn.num_batches_tracked += 1
        else:
            mean, rstd = bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)
        gs = torch.where(gamma.abs() < 1e-30, torch.full_like(gamma, 1e-30), gamma)      # (a zero scale: out = beta all the same)
        rstd_f = (rstd * gs).contiguous()
        mean_f = (mean - beta / rstd_f).contiguous()
        mean_bc, rstd_bc = mean_f.expand(b, c).contiguous(), rstd_f.expand(b, c).contiguous()
    o = ops.empty_like_act(y.t)
    ops.in_apply(y.t, mean_bc, rstd_bc, act, None, o)
    out = Act(o, req=tape.enabled)
    if tape.enabled:
        greq = (bool(bn.weight.requires_grad), bool(bn.bias.requires_grad))

        def bwd():
            g, pad = take_grad(out, allow_pad=True)
            if g is None:
                return
            part = ops.in_bwd_partial(y.t, g, pad, mean_bc, rstd_bc, act)       # per (sample, slab): (sum g m, sum g m (gamma xhat + beta))
            sums = part.sum(dim=(0, 1))                                           # [C, 2]: batch-wide
            s1 = sums[:, 0]
            s2x = (sums[:, 1] - beta * s1) / gs                                   # sum g m xhat
            if greq[0]:
                dg = _grad_like(bn.weight)
                dg.copy_(s2x)
                _store_param_grad(bn.weight, dg)
            if greq[1]:
                db = _grad_like(bn.bias)
                db.copy_(s1)
                _store_param_grad(bn.bias, db)
            if y.req:
                if training:      # dx = rstd gamma (g m - mean(g m) - xhat mean(g m xhat)), in the folded variables
                    a2 = s2x / (gs * n)
                    a1 = s1 / n - beta * a2
                else:             # running statistics are constants
                    a1 = a2 = torch.zeros_like(s1)
                dy = ops.empty_like_act(y.t)
                ops.in_bwd_apply(y.t, g, pad, mean_bc, rstd_bc, a1.expand(b, c).contiguous(), a2.expand(b, c).contiguous(), act, dy)
                add_grad(y, dy, 0)
        tape.record(bwd)
    return out


# ----------------------------------------------------------------------------- U-Net pieces
def maxpool_forward(tape: Tape, x: Act) -> Act:
    b, h, w, c = x.t.shape
    o = ops.empty_act((b, h // 2, w // 2, c), x.t.dtype, x.t.device)
    ops.maxpool2_fwd(x.t, o)
    out = Act(o, req=tape.enabled)
    if tape.enabled:
        def bwd():
            g, _ = take_grad(out)
            if g is None or not x.req:
                return
            if x.grad is not None and x.grad[1] == 0 and x.grad[0].shape == x.t.shape:
                ops.maxpool2_bwd(x.t, g, x.grad[0], True)   # accumulate onto the decoder's gradient in place
            else:
                dx = ops.empty_act(x.t.shape, g.dtype, g.device)
                ops.maxpool2_bwd(x.t, g, dx, False)
                add_grad(x, dx, 0)
        tape.record(bwd)
    return out


def upsample_concat_forward(tape: Tape, x: Act, skip: Act, buf: Optional[torch.Tensor] = None) -> Act:
    """cat([bilinear_x2(x), skip], channel) -- trainer/reg.py:91-94.  The concat buffer is written directly; with
    `buf` the skip already LIVES in its channel slice [c1, c1+c2) (its producer wrote it there) and its gradient is
    handed back as a view of the buffer's gradient: no copy in either direction."""
    b, h, w, c1 = x.t.shape
    _, hs, ws, c2 = skip.t.shape
    in_place = buf is not None
    if in_place:
        assert buf.shape == (b, hs, ws, c1 + c2) and skip.t.data_ptr() == buf[..., c1:].data_ptr()
    else:
        buf = ops.empty_act((b, hs, ws, c1 + c2), x.t.dtype, x.t.device)
        ops.copy_channels(skip.t, buf[..., c1:])
    ops.bilinear_fwd(x.t, buf[..., :c1])
    out = Act(buf, req=tape.enabled)
    if tape.enabled:
        def bwd():
            g, _ = take_grad(out)
            if g is None:
                return
            if x.req:
                dx = ops.empty_act(x.t.shape, g.dtype, g.device)
                ops.bilinear_bwd(g[..., :c1], dx)
                add_grad(x, dx, 0)
            if skip.req:
                if in_place:
                    add_grad(skip, g[..., c1:], 0)
                else:
                    ds = ops.empty_act(skip.t.shape, g.dtype, g.device)
                    ops.copy_channels(g[..., c1:], ds)
                    add_grad(skip, ds, 0)
        tape.record(bwd)
    return out
