"""Thin typed wrappers over the C ABI (include/ctagan_hip.h) for torch tensors.

PyTorch is used here only for device memory (the caching allocator) and the
current HIP stream; every function below enqueues hand-written gfx950 kernels
and returns immediately.  Activations are physical NHWC tensors `[B, H, W, C]`
(possibly a channel-narrowed view of a wider buffer: `ld = t.stride(2)`).
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Sequence

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REFLECT = 0, 1
_DT = {torch.float32: 0, torch.bfloat16: 1}
_EPC = {torch.float32: 4, torch.bfloat16: 8}


def dt(dtype) -> int:
    return _DT[dtype]


def epc(dtype) -> int:
    return _EPC[dtype]


DT_PAIR = 2
DT_MIX = 3


def dtc(t) -> int:
    """dtype code of an activation tensor for the C ABI: 0 fp32, 1 bf16, 2 split pair (see PAIR)."""
    if _pair_now() and t.dtype == torch.bfloat16:
        return DT_PAIR
    return _DT[t.dtype]


def dtc_saved(t) -> int:
    """dtype code for a launch whose `x` operand is a SAVED FORWARD activation (InstanceNorm / max-pool backward): in the plain
    bf16 backward of a split-pair forward ("bf16x3f") that operand is still a split pair and the launch is DT_MIX -- masks, argmax
    and xhat from hi + lo, gradients in and out plain bf16."""
    if _PAIR_MODE and getattr(_TLS, "bwd_plain", False) and t.dtype == torch.bfloat16:
        return DT_MIX
    return dtc(t)


# bench.py sets this to a dict to collect (start, end) HIP events around every launch of the dominant conv shape
# (256 -> 256 channels, 9 taps) on the launch stream, keyed "fwd" / "bwd_data" (fused-epilogue instantiation) /
# "wgrad"; None = no instrumentation.
KERNEL_EVENTS = None


KERNEL_EVENT_STRIDE = 3      # every 3rd launch of a kernel is bracketed: an event pair costs the stream a little (A/B of all
_event_count = {}            # launches vs none: 0.3 ms of the 52 ms step), and 24 launches per step and kernel are timed anyway
# HBM-bound kernels timed the same way (bench.py's `roofline.hbm`): key -> algorithmic bytes of ONE launch, filled in by the
# wrappers below for the shapes they bracket (the InstanceNorm elementwise kernels on the residual blocks' 256-channel maps, the
# 32 -> 32 channel 3x3 convs of Reg's full-resolution level)
KERNEL_BYTES = {}


def _timed_begin(key, nbytes=None):
    if KERNEL_EVENTS is None or key is None:
        return None
    n = _event_count.get(key, 0)
    _event_count[key] = n + 1
    if n % KERNEL_EVENT_STRIDE:
        return None
    if nbytes is not None:
        KERNEL_BYTES[key] = nbytes
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _timed_end(key, e0):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        KERNEL_EVENTS.setdefault(key, []).append((e0, e1))


# scripts/conv_roofline.py sets this to a list: every conv-class launch (forward / backward-data / weight gradient / first- and
# last-layer kernels) is then bracketed by HIP events and logged as (label, flop, algorithmic bytes, start event, end event) --
# the per-launch roofline table of profiles/.  None = no instrumentation (the product path).
OP_LOG = None


def _log_begin(label, flop, nbytes):
    if OP_LOG is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return (label, float(flop), float(nbytes), e0)


def _log_end(tok):
    if tok is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        OP_LOG.append(tok + (e1,))


def _esz(t):
    return 4 if is_pair(t) else t.element_size()


def _hbm_key(name, x, c_want=256):
    """Event key "name|BxHxWxC" of an InstanceNorm-chain launch on a 256-channel map (the generator's residual blocks; bench.py
    picks the shape it reports), else None."""
    if KERNEL_EVENTS is None or x.shape[-1] != c_want:
        return None
    return "%s|%dx%dx%dx%d" % ((name,) + tuple(x.shape))


# Split-bf16 ("bf16x3") compute mode (nets.set_default_compute_dtype("bf16x3")): statistics, accumulators, parameters and losses
# are fp32 exactly as in the fp32 mode, but every MFMA contraction runs on the bf16 matrix cores as hi.hi + hi.lo + lo.hi of
# operands split into two bf16 halves -- fp32-grade products (~1e-5 relative; only lo.lo is dropped) at a third of the bf16
# MFMA rate instead of the 1/16 of v_mfma_f32_16x16x4_f32.  Rounds 2-3 kept fp32 activations and a [hi | hi | lo] copy per
# conv operand (106 slices/s); since round 4 the storage itself is split:
# SPLIT-PAIR STORAGE.  Every wide activation / gradient is stored as two bf16 planes
# per pixel row, [hi C | lo C] (x = hi + lo to 2^-17; 4 bytes per value, like fp32), written as such by its producer (conv
# epilogues, the InstanceNorm / pooling / upsampling / gradient kernels) and read as such by its consumers: a convolution
# contracts hi.w_hi + hi.w_lo + lo.w_hi straight from the planes (ctg_conv_igemm dtype 2), the weight gradient runs on the
# hi / lo plane views.  No fp32 activation tensor, no split pass.  While PAIR is set, every bf16 NHWC activation handle IS a
# split pair: a view [B, H, W, C] with pixel pitch ld = 2 * (channels of its buffer) whose lo plane lies ld / 2 elements behind
# (`empty_act` allocates them; channel slices of a pair buffer are pair views).  1-/2-channel maps stay fp32.
# `ops.PAIR` is READ as a module attribute everywhere (module __getattr__ below); it is the process-wide mode AND NOT "this thread is
# inside the plain backward of a split-pair forward".
_PAIR_MODE = False
# "bf16x3f" (nets.set_default_compute_dtype): split-pair forward, plain bf16 backward.  nets._NetFn.backward enters
# `plain_backward()` for the duration of a network's backward: on THAT thread (autograd's engine thread) every bf16 handle is then a
# plain bf16 tensor with its own pixel pitch -- a saved split-pair activation is read as its hi plane -- while a forward running on
# another thread at the same time still sees the split-pair mode (thread-local: tests/test_neighbour_stress_gpu.py runs exactly that).
PAIR_BWD_PLAIN = False
_TLS = threading.local()


def set_pair_mode(pair: bool, bwd_plain: bool = False):
    global _PAIR_MODE, PAIR_BWD_PLAIN
    _PAIR_MODE, PAIR_BWD_PLAIN = bool(pair), bool(pair and bwd_plain)


def _pair_now() -> bool:
    return _PAIR_MODE and not getattr(_TLS, "bwd_plain", False)


class plain_backward:
    """Context of a network's backward: a no-op unless the mode is "bf16x3f"."""

    def __enter__(self):
        self.prev = getattr(_TLS, "bwd_plain", False)
        if PAIR_BWD_PLAIN:
            _TLS.bwd_plain = True
        return self

    def __exit__(self, *exc):
        _TLS.bwd_plain = self.prev
        return False


def __getattr__(name):
    if name == "PAIR":
        return _pair_now()
    if name == "PAIR_BWD_ACTIVE":       # inside such a backward (PAIR reads False, saved forward activations are still split pairs)
        return _PAIR_MODE and getattr(_TLS, "bwd_plain", False)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def is_pair(t) -> bool:
    return _pair_now() and t.dtype == torch.bfloat16


def empty_act(shape, dtype, device):
    """A new NHWC activation [B, H, W, C] of the compute dtype; in the split-pair mode a bf16 request is the hi-plane view of a
    [B, H, W, 2C] buffer (pixel pitch 2C, lo plane C elements behind)."""
    if _pair_now() and dtype == torch.bfloat16:
        b, h, w, c = shape
        if c % 8:
            raise RuntimeError("split-pair activations need a multiple of 8 channels, got %d" % c)
        return torch.empty((b, h, w, 2 * c), dtype=dtype, device=device)[..., :c]
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def empty_like_act(t):
    return empty_act(t.shape, t.dtype, t.device)


def pair_lo(t):
    """The lo-plane view of a split-pair handle (same shape and pitch)."""
    ld = _nhwc(t)[4]
    return t.as_strided(t.shape, t.stride(), t.storage_offset() + ld // 2)


def to_pair(x32, out=None):
    """fp32 NHWC [B, H, W, C] (any pixel pitch) -> split-pair tensor (a new one, or `out`)."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x32)
    assert x32.dtype == torch.float32 and _pair_now()
    if out is None:
        out = empty_act((b, h, w, c), torch.bfloat16, x32.device)
    assert tuple(out.shape) == (b, h, w, c) and is_pair(out)
    _lib.check(lib.ctg_pair_convert(0, _p(x32), ld, _p(out), _nhwc(out)[4], c, b * h * w, _stream()), "ctg_pair_convert")
    return out


def from_pair(xp):
    """split-pair handle -> new dense fp32 NHWC tensor."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(xp)
    assert is_pair(xp)
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=xp.device)
    _lib.check(lib.ctg_pair_convert(1, _p(xp), ld, _p(out), c, c, b * h * w, _stream()), "ctg_pair_convert")
    return out


def zero_act(t):
    """Zero a (possibly strided) activation view; both planes of a split pair."""
    t.zero_()
    if is_pair(t):
        pair_lo(t).zero_()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def split_w_pair(w_packed, cin):
    """Packed fp32 weights [T, N, K] -> bf16 [T, N, 2K], per 32 channels [hi 32 | lo 32]: one K step of a conv on a split-pair
    input (ctg_split_weights); cached on the pack (engine.PackCache clears it when it re-packs in place)."""
    hit = getattr(w_packed, "_ctg_split3w", None)
    if hit is not None:
        return hit
    lib = _lib.load()
    assert w_packed.dtype == torch.float32 and w_packed.is_contiguous() and w_packed.shape[-1] == cin and cin % 32 == 0
    out = torch.empty(tuple(w_packed.shape[:-1]) + (2 * cin,), dtype=torch.bfloat16, device=w_packed.device)
    _lib.check(lib.ctg_split_weights(_p(w_packed), cin, _p(out), cin, w_packed.numel() // cin, _stream()), "ctg_split_weights")
    w_packed._ctg_split3w = out
    return out


def split_w_pair_refresh(packs):
    """Re-split, in ONE launch, every pack of `packs` (fp32, freshly re-packed in place) that already carries a split copy from an
    earlier step: the copies are rewritten where they lie (engine.PackCache.refresh; a pack without one is split on first use)."""
    jobs = [(w, w._ctg_split3w_old) for w in packs if getattr(w, "_ctg_split3w_old", None) is not None]
    if not jobs:
        return
    lib = _lib.load()
    n = len(jobs)
    vp, it, lg = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_long * n
    cs = [w.shape[-1] for w, _ in jobs]
    _lib.check(lib.ctg_split_weights_multi(n, vp(*[w.data_ptr() for w, _ in jobs]), vp(*[o.data_ptr() for _, o in jobs]), it(*cs),
                                           lg(*[w.numel() // c for (w, _), c in zip(jobs, cs)]), _stream()), "ctg_split_weights_multi")
    for w, o in jobs:
        w._ctg_split3w = o
        w._ctg_split3w_old = None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _nhwc(t: torch.Tensor):
    """(B, H, W, C, ld) of a physical NHWC view; checks the strides really are NHWC with pitch ld."""
    b, h, w, c = t.shape
    if w > 1:
        ld = t.stride(2)
    elif h > 1:
        ld = t.stride(1)
    elif b > 1:
        ld = t.stride(0)
    else:
        ld = c
    if c > 1 and t.stride(3) != 1:
        raise RuntimeError("activation is not channels-innermost")
    exp = (h * w * ld, w * ld, ld)
    for dim, e, g in zip((b, h, w), exp, t.stride()[:3]):
        if dim > 1 and e != g:
            raise RuntimeError("activation is not a dense NHWC view: strides %s, expected %s" % (t.stride(), exp))
    return b, h, w, c, ld


def pack_tap(dy: int, dx: int, widx: int) -> int:
    assert -64 <= dy < 64 and -64 <= dx < 64 and 0 <= widx < 256
    return (dy + 64) | ((dx + 64) << 8) | (widx << 16)


def _tap_array(taps: Sequence[int]):
    return (ctypes.c_int * len(taps))(*taps)


# ---------------------------------------------------------------------------- conv
class ConvEpilogue(ctypes.Structure):
    """`ctg_conv_epilogue` of include/ctagan_hip.h."""
    _fields_ = [("res", ctypes.c_void_p), ("fold", ctypes.c_void_p), ("bz", ctypes.c_void_p), ("bmean", ctypes.c_void_p),
                ("brstd", ctypes.c_void_p), ("bstats", ctypes.c_void_p), ("res_ld", ctypes.c_int), ("fold_ld", ctypes.c_int),
                ("bz_ld", ctypes.c_int), ("bact", ctypes.c_int), ("nie_sync", ctypes.c_void_p), ("nie_act", ctypes.c_int),
                ("nie_tiles", ctypes.c_int), ("nie_groups", ctypes.c_int), ("nie_budget", ctypes.c_int)]


# InstanceNorm in the conv epilogue (ctg_conv_epilogue.nie_*): monotonic arrival counters, one buffer per (device, stream, group
# size), zeroed once and owned by the kernels afterwards
_NIE_SYNC = {}
_NO_NIE = bool(os.environ.get("CTG_NO_NIE"))      # A/B switch
NIE_GROUPS = 4096


NIE_POISON = bool(os.environ.get("CTG_NIE_POISON"))      # fill the moment buffer of a fused launch with NaNs first (tests)
NIE_MAX_WGS = int(os.environ.get("CTG_NIE_MAX_WGS", "1024"))
NIE_MAX_WGS_PAIR = int(os.environ.get("CTG_NIE_MAX_WGS_PAIR", "2048"))
NIE_BUDGET = int(os.environ.get("CTG_NIE_BUDGET", "0"))      # polls before a waiting workgroup gives up (0: the library's 2^22, ~1 s); tests shrink it
# launches that may wait for their groups at the same time (streams of this process, processes sharing the card); mirrors the
# library's residency test (csrc/conv_halo.h launch_halo_cfg reads the same variable and the kernel's real occupancy).  Default 2:
# the trainers issue these launches from ONE stream (the D step's no-grad generator forward), so one is in flight per process; two
# leaves room for a second process on the card and keeps the 8-row-tile launches of batch sizes 1-2 (256 workgroups per sample)
# fusable.  More tenants per card: CTG_NIE_SHARE=<n> (a launch that cannot be guaranteed runs unfused; one that is starved anyway
# ends in nie_check's RuntimeError, not in silent NaNs).
NIE_SHARE = max(1, int(os.environ.get("CTG_NIE_SHARE", "2") or 2))
NIE_SLOTS = 512                                               # 2 workgroups of the 128-channel-tile kernel per CU x 256 CUs


def _nie_sync(device, tiles):
    """The arrival counters of one group size on the current stream.  Inside a stream capture only buffers that
    `nie_prepare_capture` made beforehand are handed out (None otherwise: the caller launches the unfused kernels)."""
    cap = torch.cuda.is_current_stream_capturing()
    key = (device.index, "graph" if cap else torch.cuda.current_stream(device).cuda_stream, tiles)
    buf = _NIE_SYNC.get(key)
    if buf is None and not cap:
        buf = _NIE_SYNC[key] = torch.zeros(1 + NIE_GROUPS, dtype=torch.int64, device=device)
    return buf


def nie_prepare_capture(device):
    """Before capturing a step into a hipGraph (trainer/HdTrainer.py): counters for every group size the eager warm-up steps
    used, in memory that does not belong to the graph's pool.  Replays of captured steps are assumed not to overlap."""
    dev = torch.device(device)
    for (idx, _, tiles) in list(_NIE_SYNC.keys()):
        if idx == dev.index and (idx, "graph", tiles) not in _NIE_SYNC:
            _NIE_SYNC[(idx, "graph", tiles)] = torch.zeros(1 + NIE_GROUPS, dtype=torch.int64, device=dev)


_NIE_REFUSED = set()      # (device, pair, B, Cin, Cout, H, W) of launches the library answered "2 = not served" (see conv_igemm)
NIE_ON_FAILURE = []      # weak references to callables run by nie_check before it raises (trainers: drop the captured step graph)


def nie_failures():
    """Number of sync buffers whose bounded wait ran out (ctg_conv_epilogue.nie_sync[0]); synchronises.  Tests / smoke / bench."""
    return sum(int(b[0].item() != 0) for b in _NIE_SYNC.values())


def nie_check(where=""):
    """The trainers' check of the fused conv + InstanceNorm launches, made wherever they synchronise with the device anyway
    (`sync_losses`, end-of-epoch checkpoints, the end of train() / test()): ONE device read of every buffer's failure flag.  A
    bounded wait that ran out left NaN tiles in a generator output -- that is an error, never a silent NaN: the flags are cleared,
    fusion is switched off for the rest of the process (the unfused conv + finalize + in_apply launches take over) and RuntimeError is
    raised."""
    global _NO_NIE
    bufs = list(_NIE_SYNC.values())
    if not bufs:
        return
    by_dev = {}
    for b in bufs:
        by_dev.setdefault(b.device, []).append(b)
    bad = 0
    for dev, bs in by_dev.items():
        flags = torch.stack([b[0] for b in bs])
        n = int((flags != 0).sum().item())
        if n:
            bad += n
            for b in bs:
                b[0].zero_()
    # data parallel: every rank learns of a failure on ANY rank and raises with it (a lone raising rank would leave the others
    # waiting in the next gradient exchange)
    from . import dp
    total = bad
    if dp.enabled():
        t = torch.tensor([bad], dtype=torch.int64, device=bufs[0].device)
        torch.distributed.all_reduce(t)
        total = int(t.item())
    if total:
        _NO_NIE = True
        # a captured hipGraph has the fused launches and their 'graph' counters baked in: every holder drops its graph, so that the
        # next step is captured again with fusion off (trainers register `_drop_graph`)
        for ref in list(NIE_ON_FAILURE):
            fn = ref()
            if fn is None:
                NIE_ON_FAILURE.remove(ref)
            else:
                fn()
        raise RuntimeError(
            "cta_gan_amd: %d fused conv + InstanceNorm launch group(s) gave up waiting for their statistics%s%s -- the affected "
            "generator output holds NaNs, and optimiser steps taken since the last check saw them: continue from the last checkpoint, "
            "not from the current weights.  The dispatch-order / residency assumption of the in-launch exchange was broken "
            "(include/ctagan_hip.h, ctg_conv_epilogue); fusion is now OFF for the rest of this process (CTG_NO_NIE=1 starts that way) "
            "and captured step graphs were dropped (the next step is captured again, unfused)."
            % (total, (" (" + where + ")") if where else "", "" if total == bad else " (%d on this rank)" % bad))


def conv_in_fusable(x, cin, cout, k, stride, hs, ws):
    """Shapes whose conv + InstanceNorm (+ activation, + skip) run as ONE launch when nothing is kept for a backward pass:
    3x3 unit-stride convs of bf16 / split-pair activations on 128-channel tiles, <= 128 tiles per sample (see below) -- and at
    most NIE_MAX_WGS workgroups in the launch: a workgroup that waits for its sample's statistics holds its slot of the chip, which
    costs nothing while the launch fits the chip about twice (B <= 8 at 128^2: -3 launches per conv, 19.2 -> 17.7 ms per step at
    B = 4) and ~50 us per launch once four rounds of workgroups queue for the slots (bf16, B = 16: 339 vs 245 + 69 us)."""
    if _NO_NIE or x.dtype != torch.bfloat16 or k != 3 or stride != 1 or cout % 128 or cin % 64 or hs < 16 or ws < 16:
        return False
    if _NIE_REFUSED and (x.device.index, is_pair(x), x.shape[0], cin, cout, hs, ws) in _NIE_REFUSED:
        return False
    tiles = _nie_tiles(x, cout, hs, ws)
    groups = x.shape[0] * (cout // 128)
    # (split pair: the two launches saved move twice the bytes, and since the pair epilogue stages half of every wave's channel
    # tiles per round -- no scratch spills -- the fused launch wins at the bench shape too: B = 16, 2048 workgroups, 139.3 -> 140.6
    # slices/s; with the spilling epilogue it lost 2.4 % at 1024)
    limit = NIE_MAX_WGS_PAIR if is_pair(x) else NIE_MAX_WGS
    # residency: a waiting workgroup holds one of the chip's slots, and every workgroup of ONE sample (its channel-tile groups are
    # interleaved in dispatch order: tiles x channel tiles) must become resident in full -- NIE_SHARE such launches (streams,
    # processes sharing the card) can then be in flight at once without starving each other (the library re-checks with the
    # kernel's real occupancy and answers 2 = not served)
    return tiles * (cout // 128) * NIE_SHARE <= NIE_SLOTS and groups <= NIE_GROUPS and tiles * groups <= limit \
        and _nie_sync(x.device, tiles) is not None


_TH8_WGS = 0 if os.environ.get("CTG_NO_TH8") is not None else int(os.environ.get("CTG_TH8_WGS", "384"))


def _nie_tiles(x, cout, hs, ws):
    """Workgroups per statistics group = spatial tiles per sample of the halo kernel: 16x16 pixels, 8x16 for the small bf16 grids
    csrc/conv_halo.h (launch_halo_t) serves with its 8-row variant."""
    tiles = ((hs + 15) // 16) * ((ws + 15) // 16)
    if not is_pair(x) and tiles * ((cout + 127) // 128) * x.shape[0] < _TH8_WGS:
        tiles = ((hs + 7) // 8) * ((ws + 15) // 16)
    return tiles


def conv_igemm(x, w_packed, w_npad, y, bias, cout, hs, ws, oy0, ox0, os_, is_, pad_mode, act, taps, want_stats=False,
               frame=False, res=None, fold=None, in_bwd=None, in_fuse=None):
    """One conv launch (csrc/conv_igemm.hip, conv_halo.h).  x, y: NHWC views; y may be fp32 when cout <= 16.

    want_stats: ask the kernel to emit InstanceNorm partial moments from its epilogue; returns (part, nslabs)
    with nslabs == 0 when the shape was not served by the halo-resident kernel (caller then runs in_stats).
    res (B, hs, ws, cout): added to the result in the epilogue; fold (B, hs+2, ws+2, cout): padded-grid gradient whose
    frame is folded into the result (see include/ctagan_hip.h); both only for `conv_fusable` launches.
    in_bwd = (z, mean, rstd, act) (bf16, with res / fold): y is the gradient of act(IN(z)) [+ skip]; the partial sums of
    that InstanceNorm's backward are returned instead of the forward moments (same (part, nslabs) convention).
    in_fuse = act (`conv_in_fusable` shapes, want_stats): y = act(InstanceNorm(conv(x))) [+ res] in this one launch; returns None
    when the library does not serve the shape (nothing was launched), else True."""
    lib = _lib.load()
    b, hi, wi, cin, x_ld = _nhwc(x)
    cin0 = cin
    pair_in = is_pair(x)
    if pair_in:
        # split-pair input: hi.w_hi + hi.w_lo + lo.w_hi straight from the planes; the pack is fp32 and split here (cached)
        assert w_packed.dtype == torch.float32 and cin % 32 == 0
        w_packed = split_w_pair(w_packed, cin)
    b2, ho, wo, cy, y_ld = _nhwc(y)
    res_ld = fold_ld = 0
    if res is not None:
        rb, rh, rw, rc, res_ld = _nhwc(res)
        assert (rb, rh, rw, rc) == (b, hs, ws, cout) and res.dtype == y.dtype
    if fold is not None:
        fb, fh, fw, fc, fold_ld = _nhwc(fold)
        assert (fb, fh, fw, fc) == (b, hs + 2, ws + 2, cout) and fold.dtype == y.dtype
    assert b == b2 and cy == cout and w_packed.dtype == x.dtype
    out_f32 = int(y.dtype == torch.float32 and x.dtype != torch.float32)
    if y.dtype != x.dtype and not out_f32:
        raise RuntimeError("output dtype must equal the compute dtype (or fp32 for cout <= 16)")
    if out_f32 and cout > 16 and not pair_in:
        raise RuntimeError("fp32 output from a bf16 conv only for cout <= 16 (or in the split-bf16 mode)")
    arr = _tap_array(taps)
    tkey = None
    if KERNEL_EVENTS is not None and cin0 == 256 and cout == 256 and len(taps) == 9 and not frame and os_ == 1 and is_ == 1:
        # the fused-epilogue launches are other kernels
        tkey = "fwd_in" if in_fuse is not None else "fwd" if (res is None and fold is None) else "bwd_data"
    part, slabs = None, ctypes.c_int(0)
    if want_stats and bias is None and act == ACT_NONE and cout > 16:
        part = torch.empty(b * ((hs + 7) // 8) * ((ws + 15) // 16) * cout * 2, dtype=torch.float32, device=x.device)
    epi = None
    nie_out = None
    if in_fuse is not None:
        assert part is not None and fold is None and in_bwd is None
        if NIE_POISON:      # tests: a tile moment that is read before its workgroup published it would be a NaN, not last step's value
            part.fill_(float("nan"))
        tiles = _nie_tiles(x, cout, hs, ws)
        nie_out = True
        epi = ConvEpilogue(_p(res), None, None, None, None, None, res_ld, 0, 0, 0, _p(_nie_sync(x.device, tiles)), in_fuse, tiles,
                           NIE_GROUPS, NIE_BUDGET)
    elif res is not None or fold is not None:
        epi = ConvEpilogue(_p(res), _p(fold), None, None, None, None, res_ld, fold_ld, 0, 0, None, 0, 0, 0, 0)
        if in_bwd is not None:
            z, mean, rstd, zact = in_bwd
            zb, zh, zw, zc, z_ld = _nhwc(z)
            assert (zb, zh, zw, zc) == (b, hs, ws, cout) and z.dtype == y.dtype == torch.bfloat16 and not want_stats
            assert is_pair(z) == pair_in
            part = torch.empty(b * ((hs + 7) // 8) * ((ws + 15) // 16) * cout * 2, dtype=torch.float32, device=x.device)   # sized for 8-row tiles
            epi.bz, epi.bmean, epi.brstd, epi.bstats, epi.bz_ld, epi.bact = _p(z), _p(mean), _p(rstd), _p(part), z_ld, zact
    tbytes = None
    if KERNEL_EVENTS is not None and tkey is None and cin0 == 32 and cout == 32 and len(taps) == 9 and not frame \
            and os_ == 1 and is_ == 1 and res is None and fold is None and y.dtype == x.dtype and hs * ws >= 512 * 512:
        # the 32 -> 32 channel reflect convs of Reg's full-resolution residual blocks: HBM-bound (in + out once)
        tkey = "conv32"
        tbytes = b * hs * ws * (cin0 + cout) * _esz(x) + w_packed.numel() * w_packed.element_size()
    if KERNEL_EVENTS is not None and tkey is None and cin0 == 64 and cout == 128 and len(taps) == 9 and not frame and os_ == 1 \
            and is_ == 2 and res is None and fold is None and y.dtype == x.dtype == torch.bfloat16 and bias is None \
            and hs * ws >= 256 * 256:
        # the 64 -> 128 channel stride-2 conv (d1 forward, u2 backward-data) on csrc/conv_strips2.h: in + out once
        tkey = "convs2"
        tbytes = (x.shape.numel() * _esz(x) + y.shape.numel() * _esz(y)) + w_packed.numel() * w_packed.element_size()
    tok = None
    if OP_LOG is not None:
        npx = (2 * ws + 2 * (hs - 2)) if frame else hs * ws
        nb_ = b * hi * wi * cin0 * _esz(x) + b * npx * cout * _esz(y) * (1 + (res is not None) + (in_bwd is not None)) \
            + len(taps) * cout * cin0 * (4 if pair_in else x.element_size())
        kind = "frame" if frame else "conv+IN" if in_fuse is not None else "fused bwd-data" if (res is not None or fold is not None) else "conv"
        tok = _log_begin("%s %d->%d %dtaps is%d os%d @%dx%d%s" % (kind, cin0, cout, len(taps), is_, os_, hs, ws,
                                                                   " +INsums" if in_bwd is not None else ""),
                         2.0 * b * npx * cout * cin0 * len(taps), nb_)
    e0 = _timed_begin(tkey, tbytes)
    st = lib.ctg_conv_igemm(DT_PAIR if pair_in else dt(x.dtype), out_f32, _p(x), _p(w_packed), _p(y), _p(bias), b, hi, wi, cin, x_ld,
                            ho, wo, cout, y_ld, hs, ws, oy0, ox0, os_, is_, int(frame), pad_mode, act, w_npad, len(taps), arr,
                            _p(part) if in_bwd is None else None, ctypes.addressof(slabs) if part is not None else None,
                            ctypes.addressof(epi) if epi is not None else None, _stream())
    _timed_end(tkey, e0)
    if in_fuse is not None and st == 2:
        # the library's own residency test (the kernel's REAL occupancy x the device's CU count) refused what the Python-side gate
        # (NIE_SLOTS: two workgroups per CU x 256 CUs) admitted -- a smaller or partitioned part, an occupancy of 1: remembered per
        # shape, so that the moments buffer and the launch attempt are not repeated on every call
        _NIE_REFUSED.add((x.device.index, pair_in, b, cin0, cout, hs, ws))
        return None
    _log_end(tok)
    _lib.check(st, "ctg_conv_igemm")
    if in_fuse is not None:
        return nie_out
    if part is not None and slabs.value > 0:
        part = part[:b * slabs.value * cout * 2].view(b, slabs.value, cout, 2)
    return part, slabs.value


def conv_igemm_classes(x, w_packed, w_npad, y, bias, cout, hs, ws, classes, pad_mode, act, want_stats=False):
    """The four parity classes [(oy0, ox0, taps)] of a stride-2 transposed conv / stride-2 backward-data pass in ONE launch
    (ctg_conv_igemm_classes; bf16).  Returns None when the shape is not served (the caller launches the classes one by
    one), else (part, nslabs) like conv_igemm."""
    if x.dtype != torch.bfloat16 or y.dtype != torch.bfloat16 or len(classes) != 4 or any(not c[2] for c in classes):
        return None
    if hs < 16 or ws < 16 or cout <= 16:
        return None
    lib = _lib.load()
    b, hi, wi, cin, x_ld = _nhwc(x)
    b2, ho, wo, cy, y_ld = _nhwc(y)
    if is_pair(x):
        w_packed = split_w_pair(w_packed, cin)
    assert b == b2 and cy == cout and w_packed.dtype == x.dtype
    taps = [t for c in classes for t in c[2]]
    i4 = ctypes.c_int * 4
    part, slabs = None, ctypes.c_int(0)
    if want_stats and bias is None and act == ACT_NONE:
        part = torch.empty(b * 4 * ((hs + 7) // 8) * ((ws + 15) // 16) * cout * 2, dtype=torch.float32, device=x.device)      # (8-row tiles possible)
    tkey = tbytes = None
    if KERNEL_EVENTS is not None and cin == 128 and cout == 64 and len(taps) == 9 and bias is None and hs * ws >= 256 * 256:
        # the 128 -> 64 channel stride-2 transposed conv (u2 forward, d1 backward-data) on csrc/conv_stript.h: in + out once
        tkey = "convt64"
        tbytes = (x.shape.numel() * _esz(x) + y.shape.numel() * _esz(y)) + w_packed.numel() * w_packed.element_size()
    tok = None
    if OP_LOG is not None:
        tok = _log_begin("conv 4 parity classes %d->%d %dtaps @%dx%d" % (cin, cout, len(taps), hs, ws),
                         2.0 * b * hs * ws * cout * cin * len(taps),
                         (x.shape.numel() * _esz(x) + y.shape.numel() * _esz(y)) + len(taps) * cout * cin * _esz(x))
    e0 = _timed_begin(tkey, tbytes)
    st = lib.ctg_conv_igemm_classes(dtc(x), _p(x), _p(w_packed), _p(y), _p(bias), b, hi, wi, cin, x_ld, ho, wo, cout,
                                    y_ld, hs, ws, pad_mode, act, w_npad, i4(*[len(c[2]) for c in classes]),
                                    i4(*[c[0] for c in classes]), i4(*[c[1] for c in classes]), _tap_array(taps), _p(part),
                                    ctypes.addressof(slabs) if part is not None else None, _stream())
    _timed_end(tkey, e0)
    if st == 2:
        return None
    _log_end(tok)
    _lib.check(st, "ctg_conv_igemm_classes")
    if part is not None and slabs.value > 0:
        part = part[:b * slabs.value * cout * 2].view(b, slabs.value, cout, 2)
    return part, slabs.value


def conv_fusable(cout, hs, ws):
    """Launches whose epilogue can take `res` / `fold`: the shapes ctg_conv_igemm hands to the halo-resident kernel
    (full-window stride-1 taps are the caller's business)."""
    return cout > 16 and hs >= 16 and ws >= 16 and not os.environ.get("CTG_NO_HALO") and not os.environ.get("CTG_NO_EPI_FUSE")


def weight_pack(master, dtype, ntaps, nreal, kreal, npad, kpad, sn, sk, stp):
    lib = _lib.load()
    assert master.dtype == torch.float32 and master.is_contiguous()
    out = torch.empty((ntaps, npad, kpad), dtype=dtype, device=master.device)
    _lib.check(lib.ctg_weight_pack(dt(dtype), _p(master), sn, sk, stp, nreal, kreal, _p(out), ntaps, npad, kpad,
                                   _stream()), "ctg_weight_pack")
    return out


def weight_pack_multi(jobs):
    """jobs: list of (master fp32 tensor, out tensor [ntaps, npad, kpad], ntaps, nreal, kreal, npad, kpad, sn, sk, stp),
    all outputs of one dtype -- ONE launch per 24 tensors instead of one each."""
    if not jobs:
        return
    lib = _lib.load()
    n = len(jobs)
    vp, lg, it = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    dtype = jobs[0][1].dtype
    assert all(j[1].dtype == dtype and j[0].dtype == torch.float32 and j[0].is_contiguous() for j in jobs)
    for j in jobs:
        # the pack is rewritten in place: its cached split (split_w_pair) is stale -- kept aside so that the caller can refresh
        # every stale split in one launch (split_w_pair_refresh)
        old = getattr(j[1], "_ctg_split3w", None)
        if old is not None:
            j[1]._ctg_split3w_old = old
        j[1]._ctg_split3w = None
    _lib.check(lib.ctg_weight_pack_multi(
        dt(dtype), n, vp(*[j[0].data_ptr() for j in jobs]), vp(*[j[1].data_ptr() for j in jobs]),
        lg(*[j[7] for j in jobs]), lg(*[j[8] for j in jobs]), lg(*[j[9] for j in jobs]),
        it(*[j[3] for j in jobs]), it(*[j[4] for j in jobs]), it(*[j[2] for j in jobs]),
        it(*[j[5] for j in jobs]), it(*[j[6] for j in jobs]), _stream()), "ctg_weight_pack_multi")


_NO_WG_GROUPS_FIX = bool(os.environ.get("CTG_NO_WG_GROUPS_FIX"))      # A/B switch
_WG_S2_FIX = not os.environ.get("CTG_NO_WG_S2_FIX")      # A/B switch
_WG_FIX_BELOW = int(os.environ.get("CTG_WG_FIX_BELOW", "512"))      # A/B knob (256: the first form of the fix)


def conv_wgrad(g, x, taps, is_, pad_mode, dst, mreal, nreal, sm, sn, stp, accumulate=False, target_blocks=768,
               defer=None):
    """dst[m*sm + c*sn + t*stp] (+)= sum_pixels g[.., m] * x[tap t .., c]  (csrc/conv_wgrad.hip).
    Split-pair operands: g_hi x_hi + g_hi x_lo + g_lo x_hi, as three sweeps of every pixel tile inside one launch."""
    lib = _lib.load()
    b, hs, ws, mc, g_ld = _nhwc(g)
    b2, hi, wi, nc, x_ld = _nhwc(x)
    assert b == b2 and g.dtype == x.dtype and dst.dtype == torch.float32
    pairs = [(g, x)]
    fused_pair = is_pair(g)
    if fused_pair:
        # split-pair operands: ONE launch that sweeps every pixel tile three times (g_hi x_hi, g_hi x_lo, g_lo x_hi) into one
        # partial (ctg_conv_wgrad dtype 2); shapes it does not serve: three bf16 launches on the plane views, one reduce
        assert is_pair(x)
    cdt = pairs[0][0].dtype
    bm = 128 if mc % 128 == 0 else 64 if mc % 64 == 0 else 32 if mc % 32 == 0 else 16
    bn = 128 if nc % 128 == 0 else 64 if nc % 64 == 0 else 32
    tiles = (mc // bm) * (nc // bn)
    hw = hs * ws
    # taps swept inside one workgroup (mirrors launch_wg_t in csrc/conv_wgrad.hip)
    nt_blk = 9 if (len(taps) == 9 and (bm, bn) in ((32, 32), (64, 32), (32, 64))) else \
        49 if (len(taps) == 49 and bm == 16) else 7 if (len(taps) == 49 and (bm, bn) == (32, 64)) else 1
    groups = tiles * (len(taps) // nt_blk) * b
    if is_ == 1 and cdt == torch.bfloat16 and len(taps) in (1, 4, 9) and hs >= 8 and ws >= 16 and not _NO_WG_GROUPS_FIX:
        # the halo-resident kernel (what serves these launches): 64-wide channel tiles, the whole window in one workgroup.
        # (Rounds 1-3 sized the slabs of the 64 x 64-channel layers for nine tap groups that kernel does not have: 96
        # workgroups for 256 CUs on every 64-channel 3x3 layer of the registration U-Net.)
        # Applied where that left the chip under-filled (< 512 workgroups: two per CU; first only < 256, the wider form is
        # +0.4 % in both modes); the residual blocks' 256 x 256 layers keep the grid they were tuned on (512 workgroups: 768
        # measured 0.7 % slower in the step)
        halo_groups = (mc // min(bm, 64)) * (nc // min(bn, 64)) * b
        old_sps = max(1, min((target_blocks + groups - 1) // groups, (hw + 63) // 64))
        if halo_groups * old_sps < _WG_FIX_BELOW and not (mc == 256 and nc == 256):
            groups = halo_groups
            if _WG_FIX_BELOW > 256:
                target_blocks = 512
    if is_ == 2 and cdt == torch.bfloat16 and pad_mode == PAD_ZERO and hs >= 8 and ws >= 16 and mc % 32 == 0:
        # stride-2 weight gradients run as one halo launch per polyphase component with 64-wide tiles (conv_wgrad.hip):
        # size the pixel slabs for THAT grid (the slab count only steers performance, any value is correct)
        # (until late in round 4 only where the old sizing left the chip under 256 workgroups; the 256 x 128 and 128 x 64 layers
        # of the generator then still ran their four polyphase launches on 256 workgroups each: 512 is +0.55 % on the step)
        halo_groups = (mc // min(bm, 64)) * (nc // min(bn, 64)) * b
        old_sps = max(1, min((target_blocks + groups - 1) // groups, (hw + 63) // 64))
        if halo_groups * old_sps < 256 or _WG_S2_FIX:
            groups = halo_groups
        target_blocks = 512
    sps = max(1, min((target_blocks + groups - 1) // groups, (hw + 63) // 64))
    slab = (((hw + sps - 1) // sps) + 63) // 64 * 64
    sps = (hw + slab - 1) // slab
    z1 = b * sps
    arr = _tap_array(taps)
    tok = _log_begin("wgrad %dx%d %dtaps is%d @%dx%d" % (mc, nc, len(taps), is_, hs, ws), 2.0 * b * hs * ws * mc * nc * len(taps),
                     (g.shape.numel() * _esz(g) + x.shape.numel() * _esz(x)) + z1 * len(taps) * mc * nc * 4.0) \
        if OP_LOG is not None else None
    e0 = _timed_begin("wgrad" if (mc == 256 and nc == 256 and len(taps) == 9 and is_ == 1) else None)
    part = None
    z = z1
    if fused_pair:
        part = torch.empty((3 * z1, len(taps), mc, nc), dtype=torch.float32, device=g.device)     # (room for the split form)
        st = lib.ctg_conv_wgrad(DT_PAIR, _p(g), _p(x), _p(part), b, hs, ws, mc, g_ld, hi, wi, nc, x_ld, is_, pad_mode, slab,
                                len(taps), arr, _stream())
        if st == 2:       # not served in one launch: three launches on the plane views
            pairs = [(g, x), (g, pair_lo(x)), (pair_lo(g), x)]
            z = 3 * z1
        elif st == 3:     # small grid: the three sweeps ran in three workgroups each, 3 x the partials
            pairs = []
            z = 3 * z1
        else:
            _lib.check(st, "ctg_conv_wgrad")
            pairs = []
    if part is None:
        part = torch.empty((z, len(taps), mc, nc), dtype=torch.float32, device=g.device)
    for i, (gg, xx) in enumerate(pairs):
        st = lib.ctg_conv_wgrad(dt(cdt), _p(gg), _p(xx), _p(part[i * z1]), b, hs, ws, mc, _nhwc(gg)[4], hi, wi, nc,
                                _nhwc(xx)[4], is_, pad_mode, slab, len(taps), arr, _stream())
        _lib.check(st, "ctg_conv_wgrad")
    _timed_end("wgrad", e0)
    _log_end(tok)
    if defer is not None:   # summed later, together with the network's other weight gradients (wgrad_reduce_multi)
        defer.append((part, dst.data_ptr(), z, len(taps), mc, nc, mreal, nreal, sm, sn, stp, int(accumulate), dst))
        return
    _lib.check(lib.ctg_wgrad_reduce(_p(part), z, len(taps), mc, nc, _p(dst), mreal, nreal, sm, sn, stp,
                                    int(accumulate), _stream()), "ctg_wgrad_reduce")


def wgrad_reduce_multi(jobs):
    """jobs: tuples (part, dst_ptr, Z, ntaps, Mc, Nc, Mreal, Nreal, sm, sn, stp, accumulate, keepalive) queued by
    conv_wgrad / corr_smallcin(defer=...): ONE launch per 24 reductions."""
    if not jobs:
        return
    lib = _lib.load()
    n = len(jobs)
    vp, lg, it = ctypes.c_void_p * n, ctypes.c_long * n, ctypes.c_int * n
    _lib.check(lib.ctg_wgrad_reduce_multi(
        n, vp(*[j[0].data_ptr() for j in jobs]), vp(*[j[1] for j in jobs]), it(*[j[2] for j in jobs]),
        it(*[j[3] for j in jobs]), it(*[j[4] for j in jobs]), it(*[j[5] for j in jobs]), it(*[j[6] for j in jobs]),
        it(*[j[7] for j in jobs]), lg(*[j[8] for j in jobs]), lg(*[j[9] for j in jobs]), lg(*[j[10] for j in jobs]),
        it(*[j[11] for j in jobs]), _stream()), "ctg_wgrad_reduce_multi")


# ---------------------------------------------------------------------------- norm / elementwise
_NO_SMALLB = bool(os.environ.get("CTG_NO_SMALLB"))     # A/B switch (also read by csrc/norm_act.hip: pix_grid)


def _nslabs(b, hw):
    if b < 16 and not _NO_SMALLB:
        # small batches (the reference's yaml ships batchSize 1): 64-pixel slabs, up to 128 per sample.  At B = 4 the 64 slabs x 4
        # samples of a statistics pass left the chip under one workgroup per CU, each walking 8 dependent trips: 71 us for a
        # quarter of the bytes that take 128 us at B = 16.  With pix_grid's shorter lanes (norm_act.hip): B=1 12.1 -> 10.5 ms
        # (graph replay), B=2 14.0 -> 12.6, B=4 18.05 -> 17.4, B=8 28.3 -> 27.8 ms/step; B = 16 launches are unchanged.
        return int(max(1, min(128, (1024 + b - 1) // b, (hw + 63) // 64)))
    return int(max(1, min(64, (1024 + b - 1) // b, (hw + 255) // 256)))


def in_stats(x):
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    ns = _nslabs(b, h * w)
    part = torch.empty((b, ns, c, 2), dtype=torch.float32, device=x.device)
    mean = torch.empty((b, c), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    _lib.check(lib.ctg_in_stats(dtc(x), _p(x), ld, b, h, w, c, ns, _p(part), _p(mean), _p(rstd), _stream()),
               "ctg_in_stats")
    return mean, rstd


def in_partial(x):
    """Partial moments [B, nslabs, C, 2] of x (nslabs <= 128: `_nslabs`), to be finalized by in_apply_part / in_finalize."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    ns = _nslabs(b, h * w)
    part = torch.empty((b, ns, c, 2), dtype=torch.float32, device=x.device)
    _lib.check(lib.ctg_in_stats(dtc(x), _p(x), ld, b, h, w, c, ns, _p(part), None, None, _stream()), "ctg_in_stats")
    return part, ns


def in_finalize(part, nslabs, hw, mode=0):
    """mode 0: (mean, rstd) from partial moments [B, nslabs, C, 2] (a conv epilogue's, in_partial's); mode 1: the two plain
    means of the InstanceNorm backward from its partial sums."""
    lib = _lib.load()
    b, ns, c, _ = part.shape
    assert ns == nslabs and part.is_contiguous()
    mean = torch.empty((b, c), dtype=torch.float32, device=part.device)
    rstd = torch.empty_like(mean)
    _lib.check(lib.ctg_in_finalize(_p(part), b, c, nslabs, hw, mode, _p(mean), _p(rstd), _stream()), "ctg_in_finalize")
    return mean, rstd


FUSED_MAX_SLABS = 128      # csrc/norm_act.hip: partial counts the elementwise kernels can finalize in their prologue
# Round-3 experiment, OFF by default: finalize fused into the elementwise kernels (one launch less per InstanceNorm).  Measured
# (interleaved A/B, B=16): 53.00 / 53.14 / 53.01 ms with it against 52.81 / 53.04 / 52.86 without -- the prologue has to be
# amortised over a (sample, 64-channel group) strip, and 128-byte-per-pixel accesses stream at 4.2 TB/s where the full 512-byte
# pixels of the plain kernels reach 5.2; what the ~100 saved launches per step gain, the slower passes lose.  CTG_FIN_FUSE=1.
_FIN_FUSE = bool(os.environ.get("CTG_FIN_FUSE"))


def fin_fusable(nslabs):
    return nslabs <= FUSED_MAX_SLABS and _FIN_FUSE


def in_apply(x, mean, rstd, act, res, out):
    """out = act((x - mean) * rstd) [+ res]."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    _, _, _, _, o_ld = _nhwc(out)
    r_ld = _nhwc(res)[4] if res is not None else 0
    key = _hbm_key("in_apply_res" if res is not None else "in_apply", x)
    e0 = _timed_begin(key, x.numel() * _esz(x) * (3 if res is not None else 2))
    _lib.check(lib.ctg_in_apply(dtc(x), _p(x), ld, _p(mean), _p(rstd), act, _p(res), r_ld, _p(out), o_ld, b, h, w, c, _stream()),
               "ctg_in_apply")
    _timed_end(key, e0)


def in_apply_part(x, part, act, res, out):
    """out = act(InstanceNorm(x)) [+ res] straight from the partial moments [B, nslabs <= 128, C, 2]; returns (mean, rstd)."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    _, _, _, _, o_ld = _nhwc(out)
    r_ld = _nhwc(res)[4] if res is not None else 0
    ns = part.shape[1]
    assert part.shape == (b, ns, c, 2) and part.is_contiguous() and ns <= FUSED_MAX_SLABS
    mean = torch.empty((b, c), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    key = _hbm_key("in_apply_res" if res is not None else "in_apply", x)
    e0 = _timed_begin(key, x.numel() * _esz(x) * (3 if res is not None else 2))
    _lib.check(lib.ctg_in_apply_part(dtc(x), _p(x), ld, _p(part), ns, _p(mean), _p(rstd), act, _p(res), r_ld, _p(out),
                                     o_ld, b, h, w, c, _stream()), "ctg_in_apply_part")
    _timed_end(key, e0)
    return mean, rstd


def in_bwd(x, dout, pad, mean, rstd, act, dx):
    """InstanceNorm backward: the statistics pass over (x, dout), then the elementwise pass (finalize fused in)."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    d_ld = _nhwc(dout)[4]
    assert dout.shape[1] == h + 2 * pad and dout.shape[2] == w + 2 * pad and dout.dtype == x.dtype
    ns = _nslabs(b, h * w)
    part = torch.empty((b, ns, c, 2), dtype=torch.float32, device=x.device)
    key = _hbm_key("in_bwd_partial", x) if pad == 0 else None
    e0 = _timed_begin(key, x.numel() * x.element_size() * 2)
    _lib.check(lib.ctg_in_bwd_partial(dtc_saved(x), _p(x), ld, _p(dout), d_ld, pad, _p(mean), _p(rstd), act, b, h, w, c, ns,
                                      _p(part), _stream()), "ctg_in_bwd_partial")
    _timed_end(key, e0)
    in_bwd_stats(x, dout, mean, rstd, act, dx, part, pad=pad)


def in_bwd_partial(x, dout, pad, mean, rstd, act):
    """The statistics pass of a normalisation's backward alone: partial sums [B, nslabs, C, 2] of (g m, g m xhat) with
    xhat = (x - mean) rstd, m = act'(xhat), g = fold(dout)."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    d_ld = _nhwc(dout)[4]
    ns = _nslabs(b, h * w)
    part = torch.empty((b, ns, c, 2), dtype=torch.float32, device=x.device)
    _lib.check(lib.ctg_in_bwd_partial(dtc_saved(x), _p(x), ld, _p(dout), d_ld, pad, _p(mean), _p(rstd), act, b, h, w, c, ns,
                                      _p(part), _stream()), "ctg_in_bwd_partial")
    return part


def in_bwd_apply(x, dout, pad, mean, rstd, s1, s2, act, dx):
    """dx = rstd (g m - s1 - xhat s2) with caller-supplied per-(sample, channel) terms s1 / s2 [B, C] (the elementwise pass of
    the InstanceNorm backward; BatchNorm feeds it batch-wide terms)."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    _lib.check(lib.ctg_in_bwd_apply(dtc_saved(x), _p(x), ld, _p(dout), _nhwc(dout)[4], pad, _p(mean), _p(rstd), _p(s1), _p(s2), act,
                                    _p(dx), _nhwc(dx)[4], b, h, w, c, _stream()), "ctg_in_bwd_apply")


def in_bwd_stats(x, dout, mean, rstd, act, dx, part, pad=0):
    """IN backward from partial sums [B, nslabs, C, 2] (in_bwd's own pass, or a fused conv epilogue's:
    `conv_igemm(..., in_bwd=...)`): finalize launch + elementwise pass (or one launch with CTG_FIN_FUSE)."""
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    d_ld = _nhwc(dout)[4]
    dx_ld = _nhwc(dx)[4]
    assert tuple(dout.shape) == (b, h + 2 * pad, w + 2 * pad, c) and dout.dtype == x.dtype
    assert part.shape[0] == b and part.shape[2] == c and part.is_contiguous()
    key = _hbm_key("in_bwd_apply", x) if pad == 0 else None
    nbytes = x.numel() * _esz(x) * 3
    if fin_fusable(part.shape[1]):
        e0 = _timed_begin(key, nbytes)
        _lib.check(lib.ctg_in_bwd_stats(dtc_saved(x), _p(x), ld, _p(dout), d_ld, pad, _p(mean), _p(rstd), act, _p(dx), dx_ld, b,
                                        h, w, c, part.shape[1], _p(part), _stream()), "ctg_in_bwd_stats")
        _timed_end(key, e0)
        return
    s1, s2 = in_finalize(part, part.shape[1], h * w, mode=1)
    e0 = _timed_begin(key, nbytes)
    _lib.check(lib.ctg_in_bwd_apply(dtc_saved(x), _p(x), ld, _p(dout), d_ld, pad, _p(mean), _p(rstd), _p(s1), _p(s2), act,
                                    _p(dx), dx_ld, b, h, w, c, _stream()), "ctg_in_bwd_apply")
    _timed_end(key, e0)


def grad_combine(a, b, pad, yact, act, out):
    """out = a + fold(b) [* act'(yact)]; shapes from `out` ([B,H,W,C] NHWC)."""
    lib = _lib.load()
    bsz, h, w, c, o_ld = _nhwc(out)
    a_ld = _nhwc(a)[4] if a is not None else 0
    b_ld = _nhwc(b)[4] if b is not None else 0
    y_ld = _nhwc(yact)[4] if yact is not None else 0
    if b is not None:
        assert b.shape[1] == h + 2 * pad and b.shape[2] == w + 2 * pad
    _lib.check(lib.ctg_grad_combine(dtc(out), _p(a), a_ld, _p(b), b_ld, pad, _p(yact), y_ld, act, _p(out), o_ld,
                                    bsz, h, w, c, _stream()), "ctg_grad_combine")


def fold_f32(dp, pad):
    """fp32 [B, H+2p, W+2p, C] padded-grid gradient -> [B, H, W, C] (transpose of ReflectionPad2d)."""
    lib = _lib.load()
    b, hp, wp, c = dp.shape
    assert dp.dtype == torch.float32 and dp.is_contiguous()
    out = torch.empty((b, hp - 2 * pad, wp - 2 * pad, c), dtype=torch.float32, device=dp.device)
    _lib.check(lib.ctg_fold_f32(_p(dp), _p(out), b, hp - 2 * pad, wp - 2 * pad, c, pad, _stream()), "ctg_fold_f32")
    return out


def act_bwd_f32(g, y, act):
    """g * act'(y) for dense fp32 tensors of any element count (y = the activation's saved output)."""
    lib = _lib.load()
    assert g.dtype == torch.float32 and y.dtype == torch.float32 and g.is_contiguous() and y.is_contiguous()
    assert g.numel() == y.numel()
    out = torch.empty_like(g)
    _lib.check(lib.ctg_act_bwd_f32(_p(g), _p(y), act, _p(out), g.numel(), _stream()), "ctg_act_bwd_f32")
    return out


def bias_grad(g, pad, creal, db, accumulate=False):
    lib = _lib.load()
    b, hp, wp, c, ld = _nhwc(g)
    h, w = hp - 2 * pad, wp - 2 * pad
    ns = _nslabs(b, h * w)
    part = torch.empty((b, ns, c, 2), dtype=torch.float32, device=g.device)
    _lib.check(lib.ctg_bias_grad(dtc(g), _p(g), ld, pad, b, h, w, c, creal, ns, _p(part), _p(db), int(accumulate),
                                 _stream()), "ctg_bias_grad")


def bias_grad_act(g, pad, yact, act, gout, creal, db, accumulate=False):
    """gout = fold(g) * act'(yact), db (+)= its per-channel sum: activation backward + bias gradient of a conv + bias +
    (Leaky)ReLU layer in one pass."""
    lib = _lib.load()
    b, hp, wp, c, ld = _nhwc(g)
    h, w = hp - 2 * pad, wp - 2 * pad
    assert tuple(yact.shape) == (b, h, w, c) == tuple(gout.shape) and yact.dtype == g.dtype == gout.dtype
    ns = _nslabs(b, h * w)
    part = torch.empty((b, ns, c, 2), dtype=torch.float32, device=g.device)
    _lib.check(lib.ctg_bias_grad_act(dtc(g), _p(g), ld, pad, _p(yact), _nhwc(yact)[4], act, _p(gout), _nhwc(gout)[4], b, h, w, c,
                                     creal, ns, _p(part), _p(db), int(accumulate), _stream()), "ctg_bias_grad_act")


# ---------------------------------------------------------------------------- spatial
def maxpool2_fwd(x, out):
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    _lib.check(lib.ctg_maxpool2_fwd(dtc(x), _p(x), ld, _p(out), _nhwc(out)[4], b, h, w, c, _stream()),
               "ctg_maxpool2_fwd")


def maxpool2_bwd(x, dout, dx, accumulate):
    lib = _lib.load()
    b, h, w, c, ld = _nhwc(x)
    _lib.check(lib.ctg_maxpool2_bwd(dtc_saved(x), _p(x), ld, _p(dout), _nhwc(dout)[4], _p(dx), _nhwc(dx)[4],
                                    int(accumulate), b, h, w, c, _stream()), "ctg_maxpool2_bwd")


def bilinear_fwd(x, out):
    lib = _lib.load()
    b, hi, wi, c, ld = _nhwc(x)
    _, ho, wo, _, o_ld = _nhwc(out)
    _lib.check(lib.ctg_bilinear_fwd(dtc(x), _p(x), ld, _p(out), o_ld, b, hi, wi, ho, wo, c, _stream()),
               "ctg_bilinear_fwd")


def bilinear_bwd(dout, dx):
    lib = _lib.load()
    b, ho, wo, c, d_ld = _nhwc(dout)
    _, hi, wi, _, dx_ld = _nhwc(dx)
    _lib.check(lib.ctg_bilinear_bwd(dtc(dx), _p(dout), d_ld, _p(dx), dx_ld, b, hi, wi, ho, wo, c, _stream()),
               "ctg_bilinear_bwd")


def copy_channels(src, dst):
    lib = _lib.load()
    b, h, w, c, s_ld = _nhwc(src)
    _lib.check(lib.ctg_copy_channels(dtc(src), _p(src), s_ld, _p(dst), _nhwc(dst)[4], c, b * h * w, _stream()),
               "ctg_copy_channels")


def chan_pad(src_f32, cs, dtype, cpad):
    """fp32 [B,H,W,cs] (dense) -> dtype [B,H,W,cpad] zero padded."""
    lib = _lib.load()
    b, h, w, c = src_f32.shape
    assert c == cs and src_f32.is_contiguous() and src_f32.dtype == torch.float32
    out = empty_act((b, h, w, cpad), dtype, src_f32.device)
    _lib.check(lib.ctg_chan_pad(dtc(out), _p(src_f32), cs, _p(out), cpad, b * h * w, _stream()), "ctg_chan_pad")
    return out


def im2col_pack(s0, s1, k, stride, pad, pad_mode, dtype, kpad):
    """s0, s1: dense fp32 [B,H,W] single-channel images (s1 optional) -> [B,Ho,Wo,kpad]."""
    lib = _lib.load()
    b, hi, wi = s0.shape
    cin = 1 if s1 is None else 2
    ho = (hi + 2 * pad - k) // stride + 1
    wo = (wi + 2 * pad - k) // stride + 1
    assert s0.is_contiguous() and (s1 is None or s1.is_contiguous())
    out = empty_act((b, ho, wo, kpad), dtype, s0.device)
    _lib.check(lib.ctg_im2col_pack(dtc(out), _p(s0), _p(s1), cin, b, hi, wi, k, k, stride, pad, pad_mode, _p(out),
                                   ho, wo, kpad, _stream()), "ctg_im2col_pack")
    return out


_NO_KXW = bool(os.environ.get("CTG_NO_KXW"))      # A/B switch


def kxw_ok(cin, cout, k, stride, dtype):
    """First-layer shapes served with the kx-window weight layout (csrc/conv_small.hip KXW): one image plane, stride 1,
    a 5..8 wide window (Kpad 64), 33..64 output channels, bf16 or split-pair compute."""
    return (not _NO_KXW) and cin == 1 and stride == 1 and 5 <= k <= 8 and 32 < cout <= 64 and dtype == torch.bfloat16


def kxw_pack(w2d, k, npad, dtype):
    """fp32 [N, k*k] (row-major ky, kx) -> [1, npad, 64] with column 8 ky + kx (zero elsewhere): the kx-window operand."""
    n = w2d.shape[0]
    out = torch.zeros((npad, 8, 8), dtype=torch.float32, device=w2d.device)
    out[:n, :k, :k] = w2d.detach().reshape(n, k, k)
    return out.reshape(1, npad, 64).to(dtype).contiguous()


def conv_smallcin(s0, s1, k, stride, pad, pad_mode, w_packed, w_npad, bias, act, y, cout, want_stats=False, kxw=False):
    """First-layer conv straight from fp32 image planes (csrc/conv_small.hip): y[B,Ho,Wo,:cout] = act(conv + bias).
    s0, s1: dense fp32 [B,H,W] (s1 optional); w_packed: [1][w_npad][Kpad] of the compute dtype.
    Returns (part, nslabs) InstanceNorm partial moments when want_stats (and no bias / activation)."""
    lib = _lib.load()
    b, hi, wi = s0.shape
    cin = 1 if s1 is None else 2
    b2, ho, wo, cy, y_ld = _nhwc(y)
    kpad = w_packed.shape[-1]
    assert b == b2 and cy == cout and (y.dtype == w_packed.dtype or (is_pair(y) and w_packed.dtype == torch.float32))
    assert w_packed.shape[-2] == w_npad
    assert s0.is_contiguous() and s0.dtype == torch.float32 and (s1 is None or (s1.is_contiguous() and s1.shape == s0.shape))
    part, slabs = None, ctypes.c_int(0)
    if want_stats and bias is None and act == ACT_NONE:
        part = torch.empty(b * ((ho + 15) // 16) * ((wo + 15) // 16) * cout * 2, dtype=torch.float32, device=y.device)
    if is_pair(y):
        # split-pair mode: the im2col tile is built as bf16 hi / lo halves, split-bf16 MFMA against the split pack, and the
        # fp32 accumulators leave as a split pair (ctg_conv_smallcin dtype 2)
        w_packed = split_w_pair(w_packed, kpad)
    tok = _log_begin("first-layer conv %d->%d %dx%d s%d @%dx%d" % (cin, cout, k, k, stride, ho, wo),
                     2.0 * b * ho * wo * cout * cin * k * k, b * hi * wi * cin * 4.0 + b * ho * wo * cout * _esz(y)) \
        if OP_LOG is not None else None
    _lib.check(lib.ctg_conv_smallcin(dtc(y), _p(s0), _p(s1), cin, b, hi, wi, k, k, stride, pad, pad_mode,
                                     _p(w_packed), w_npad, kpad, int(kxw), _p(bias), act, _p(y), y_ld, ho, wo, cout, _p(part),
                                     ctypes.addressof(slabs) if part is not None else None, _stream()),
               "ctg_conv_smallcin")
    _log_end(tok)
    if part is not None and slabs.value > 0:
        part = part.view(b, slabs.value, cout, 2)
    return part, slabs.value


_TAIL_IDX = {}


def tail7_pack(weight, dtype=torch.bfloat16):
    """weight (1, 64, 7, 7) fp32 master -> [NS][8][16][SC] operand of ctg_conv_tail7 (one gather); SC = 32 channels per
    slice in bf16, 16 in fp32."""
    dev = weight.device
    sc = 32 if dtype in (torch.bfloat16, "pair") else 16
    ns = 64 // sc
    idx = _TAIL_IDX.get((dev, sc))
    if idx is None:
        import numpy as np
        ix = np.full((ns, 8, 16, sc), 64 * 49, dtype=np.int64)      # default: the appended zero
        for s in range(ns):
            for j in range(8):
                for o in range(2):
                    kx = j - o
                    if 0 <= kx <= 6:
                        for ky in range(7):
                            ix[s, j, o * 8 + ky, :] = (np.arange(sc) + sc * s) * 49 + ky * 7 + kx
        idx = torch.from_numpy(ix).to(dev)
        _TAIL_IDX[(dev, sc)] = idx
    flat = torch.cat([weight.detach().reshape(-1), weight.new_zeros(1)])
    if dtype == "pair":       # split-pair mode: [2][NS][8][16][SC], the operand's bf16 hi halves, then its lo halves
        g = flat[idx]
        hi = g.to(torch.bfloat16)
        return torch.stack([hi, (g - hi.float()).to(torch.bfloat16)]).contiguous()
    return flat[idx].to(dtype).contiguous()


def conv_tail7(x, wp, bias, y, act):
    """y[B,H,W] fp32 = act(bias + conv7x7(reflection_pad3(x))), x NHWC (bf16 or fp32) with 64 channels (csrc/conv_tail.hip)."""
    lib = _lib.load()
    b, h, w, c, x_ld = _nhwc(x)
    assert c == 64 and wp.dtype == x.dtype and y.dtype == torch.float32 and y.is_contiguous() and y.numel() == b * h * w
    assert not is_pair(x) or wp.shape[0] == 2         # split-pair input: the (hi, lo) operand of tail7_pack(weight, "pair")
    tok = _log_begin("tail conv 64->1 7x7 @%dx%d" % (h, w), 2.0 * b * h * w * 64 * 49, b * h * w * (64 * _esz(x) + 4.0)) \
        if OP_LOG is not None else None
    _lib.check(lib.ctg_conv_tail7(dtc(x), _p(x), x_ld, _p(wp), _p(bias), _p(y), act, b, h, w, _stream()),
               "ctg_conv_tail7")
    _log_end(tok)


def conv_tail7_ok(spec_cin, spec_cout, k, stride, reflect, pad, dtype, h, w):
    if os.environ.get("CTG_NO_TAIL7"):
        return False
    return (spec_cin == 64 and spec_cout == 1 and k == 7 and stride == 1 and reflect and pad == 3
            and dtype in (torch.bfloat16, torch.float32) and h >= 4 and w >= 4)


def lds_canary(blocks=1024, spins=2000, cap=64, stream=None):
    """Launch the LDS canary (csrc/lds_canary.hip) on `stream` (default: the current one); returns the report tensor
    (int32 [4 + 4 cap]): [0] = foreign writes seen so far (read it after a synchronize)."""
    lib = _lib.load()
    rep = torch.zeros(4 + 4 * cap, dtype=torch.int32, device="cuda")
    st = _stream() if stream is None else stream.cuda_stream
    _lib.check(lib.ctg_lds_canary(blocks, spins, _p(rep), cap, st), "ctg_lds_canary")
    return rep


# ---- the PatchGAN's last layer, Conv2d(512, 1, 4, padding=1), on the vector ALUs (csrc/conv_cout1.hip) -------------------------
def cout1_ok(cin, cout, k, stride, reflect, pad, transposed, dtype):
    """Layers ctg_conv_cout1_* serve: 512 -> 1 channels, 4x4, stride 1, zero padding, bf16 / split-pair activations."""
    if os.environ.get("CTG_NO_COUT1"):      # A/B switch (the MFMA kernels with 15 / 31 zero columns)
        return False
    return (cin == 512 and cout == 1 and k == 4 and stride == 1 and not reflect and not transposed and 0 <= pad < 4
            and dtype == torch.bfloat16)


def cout1_pack(weight):
    """fp32 [16][512]: w[ky*4 + kx][ci] = weight[0][ci][ky][kx] (the unrounded master weights)."""
    assert tuple(weight.shape[:1]) == (1,) and weight.shape[1] == 512 and weight.shape[2] == weight.shape[3] == 4
    return weight.detach()[0].permute(1, 2, 0).reshape(16, 512).float().contiguous()


def conv_cout1_fwd(x, w16, bias, y, act, pad):
    """y[B,Ho,Wo] fp32 = act(bias + conv4x4(zero_pad(x))) for x NHWC with 512 channels (bf16 or split pair)."""
    lib = _lib.load()
    b, hi, wi, c, x_ld = _nhwc(x)
    ho, wo = hi + 2 * pad - 3, wi + 2 * pad - 3
    assert c == 512 and x.dtype == torch.bfloat16 and w16.dtype == torch.float32 and tuple(w16.shape) == (16, 512)
    assert y.dtype == torch.float32 and y.is_contiguous() and y.numel() == b * ho * wo
    tok = _log_begin("conv 512->1 16taps is1 os1 @%dx%d" % (ho, wo), 2.0 * b * ho * wo * 512 * 16, b * hi * wi * 512 * _esz(x) + 4.0 * b * ho * wo) \
        if OP_LOG is not None else None
    _lib.check(lib.ctg_conv_cout1_fwd(dtc(x), _p(x), x_ld, _p(w16), _p(bias), _p(y), act, b, hi, wi, 512, 4, pad, ho, wo, _stream()),
               "ctg_conv_cout1_fwd")
    _log_end(tok)


def conv_cout1_bwd(g, w16, dx, pad):
    """dx (NHWC, 512 channels, bf16 or split pair) = the input gradient of that layer for g = dL/dy fp32 [B,Ho,Wo]."""
    lib = _lib.load()
    b, hi, wi, c, dx_ld = _nhwc(dx)
    ho, wo = hi + 2 * pad - 3, wi + 2 * pad - 3
    assert c == 512 and dx.dtype == torch.bfloat16 and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == b * ho * wo
    tok = _log_begin("conv 1->512 16taps is1 os1 @%dx%d" % (hi, wi), 2.0 * b * hi * wi * 512 * 16, b * hi * wi * 512 * _esz(dx) + 4.0 * b * ho * wo) \
        if OP_LOG is not None else None
    _lib.check(lib.ctg_conv_cout1_bwd(dtc(dx), _p(g), _p(w16), _p(dx), dx_ld, b, hi, wi, 512, 4, pad, ho, wo, _stream()),
               "ctg_conv_cout1_bwd")
    _log_end(tok)


def conv_cout1_wgrad(g, x, dw, pad, defer=None):
    """dw (fp32 [1,512,4,4], contiguous) = the weight gradient of that layer; `defer`: queue the split reduction (see conv_wgrad)."""
    lib = _lib.load()
    b, hi, wi, c, x_ld = _nhwc(x)
    ho, wo = hi + 2 * pad - 3, wi + 2 * pad - 3
    assert c == 512 and x.dtype == torch.bfloat16 and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == b * ho * wo
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == 512 * 16
    ntask = b * hi * ((wi + 15) // 16)
    z = max(1, min(256, (ntask + 3) // 4))
    part = torch.empty((z, 1, 16, 512), dtype=torch.float32, device=x.device)
    tok = _log_begin("wgrad 1x512 16taps is1 @%dx%d" % (ho, wo), 2.0 * b * ho * wo * 512 * 16, b * hi * wi * 512 * _esz(x) + 4.0 * b * ho * wo) \
        if OP_LOG is not None else None
    _lib.check(lib.ctg_conv_cout1_wgrad(dtc(x), _p(g), _p(x), x_ld, _p(part), z, b, hi, wi, 512, 4, pad, ho, wo, _stream()),
               "ctg_conv_cout1_wgrad")
    _log_end(tok)
    # part[z][tap][ci] -> dw[0][ci][ky][kx]: tap stride 1, channel stride 16
    if defer is not None:
        defer.append((part, dw.data_ptr(), z, 1, 16, 512, 16, 512, 1, 16, 0, 0, dw))
        return
    _lib.check(lib.ctg_wgrad_reduce(_p(part), z, 1, 16, 512, dw.data_ptr(), 16, 512, 1, 16, 0, 0, _stream()), "ctg_wgrad_reduce")


def corr_smallcin(g, gpad, g_pad_mode, i0, i1, k, ipad, i_pad_mode, hs, ws, dst, dst_off, mreal, nreal, sm, sn,
                  defer=None):
    """dst.view(-1)[dst_off + m*sm + kk*sn] = sum over the hs x ws grid of Gpad[q][m] * Ipad[q + tap_kk]
    (csrc/corr_small.hip).  g: bf16 NHWC [B,Gh,Gw,Mc] (Mc in {32,64}); i0/i1: dense fp32 [B,Ih,Iw] planes."""
    lib = _lib.load()
    b, gh, gw, mc, g_ld = _nhwc(g)
    assert g.dtype == torch.bfloat16 and dst.dtype == torch.float32 and dst.is_contiguous()
    assert i0.is_contiguous() and i0.dtype == torch.float32 and (i1 is None or (i1.is_contiguous() and i1.shape == i0.shape))
    cin = 1 if i1 is None else 2
    ntiles = ((hs + 15) // 16) * ((ws + 15) // 16)
    wgs = max(1, min(ntiles, (512 + b - 1) // b))
    # split-pair mode: g_hi.I_hi + g_hi.I_lo + g_lo.I_hi.  Round 5: ONE launch (a negative pitch tells the library that g is a pair;
    # it reads each plane of g once and splits the image planes itself); CTG_CORR_3RUN=1 keeps rounds 3-4's three launches (fed I
    # the kernel uses I_hi, fed the exactly representable remainder I - bf16(I) it uses I_lo) for A/B runs and the parity test
    runs = [(g, i0, i1, g_ld)]
    if is_pair(g):
        if os.environ.get("CTG_CORR_3RUN"):
            lo = [None if t is None else t - t.to(torch.bfloat16).float() for t in (i0, i1)]
            runs = [(g, i0, i1, g_ld), (g, lo[0], lo[1], g_ld), (pair_lo(g), i0, i1, g_ld)]
        else:
            runs = [(g, i0, i1, -g_ld)]
    z = b * wgs
    tok = _log_begin("image-correlation wgrad %dch x %d taps @%dx%d" % (mc, cin * k * k, hs, ws), 2.0 * b * hs * ws * mc * cin * k * k,
                     b * hs * ws * (mc * _esz(g) + cin * 4.0)) if OP_LOG is not None else None
    part = torch.empty((len(runs) * z, 1, mc, 64), dtype=torch.float32, device=g.device)
    for r, (gg, a0, a1, ld) in enumerate(runs):
        _lib.check(lib.ctg_corr_smallcin(_p(gg), gh, gw, ld, mc, gpad, g_pad_mode, _p(a0), _p(a1), cin, i0.shape[1],
                                         i0.shape[2], k, k, ipad, i_pad_mode, b, hs, ws, _p(part[r * z]), wgs, _stream()),
                   "ctg_corr_smallcin")
    z *= len(runs)
    _log_end(tok)
    if defer is not None:
        defer.append((part, dst.data_ptr() + 4 * dst_off, z, 1, mc, 64, mreal, nreal, sm, sn, 0, 0, dst))
        return
    _lib.check(lib.ctg_wgrad_reduce(_p(part), z, 1, mc, 64, dst.data_ptr() + 4 * dst_off, mreal, nreal, sm, sn, 0,
                                    0, _stream()), "ctg_wgrad_reduce")


def corr_smallcin_ok(cin_img, m_ch, k, stride, dtype):
    """Shapes ctg_corr_smallcin serves (bf16, stride 1, <= 64 taps, 32 or 64 wide-tensor channels)."""
    if os.environ.get("CTG_NO_SMALLCIN"):
        return False
    return dtype == torch.bfloat16 and stride == 1 and cin_img * k * k <= 64 and k <= 8 and m_ch in (32, 64)


def smallcin_ok(cin, cout, k, dtype, out_dtype, stride=1):
    """Shapes ctg_conv_smallcin serves (otherwise: im2col_pack + 1x1 gather-GEMM)."""
    if cin * (15 * stride + k) ** 2 > 1280:     # input patch of a 16x16 output tile: 5 elements per thread (conv_small.hip)
        return False
    epc = 8 if dtype == torch.bfloat16 else 4
    if os.environ.get("CTG_NO_SMALLCIN"):   # A/B switch (scripts/ab.sh)
        return False
    return cin * k * k <= 64 and cout <= 64 and cout % epc == 0 and out_dtype == dtype


# ---------------------------------------------------------------------------- STN / losses / Adam
def _flow_strides(flow):
    """flow: logical (B, 2, H, W) fp32 tensor of any strides."""
    assert flow.dim() == 4 and flow.shape[1] == 2 and flow.dtype == torch.float32
    return flow.stride(0), flow.stride(1), flow.stride(2), flow.stride(3)


def warp_fwd(src, flow):
    lib = _lib.load()
    b, _, h, w = flow.shape
    src = src.contiguous()
    out = torch.empty((b, 1, h, w), dtype=torch.float32, device=src.device)
    sn, sc, sy, sx = _flow_strides(flow)
    _lib.check(lib.ctg_warp_fwd(_p(src), _p(flow), sn, sc, sy, sx, _p(out), b, h, w, _stream()), "ctg_warp_fwd")
    return out


# Deterministic mode (tests / debugging; CTG_DETERMINISTIC=1 or ops.DETERMINISTIC = True): the one order-dependent kernel of the
# step -- the float-atomic scatter of the warp backward (trainer/transformer.py:29, grid_sample's backward) -- runs in 64-bit
# fixed point, so two runs of a step are bit-identical.  ~3 extra passes over 1-channel maps; off by default.
DETERMINISTIC = bool(os.environ.get("CTG_DETERMINISTIC"))


def warp_bwd(src, flow, gout, need_src, need_flow):
    lib = _lib.load()
    b, _, h, w = flow.shape
    src = src.contiguous()
    gout = gout.contiguous()
    dsrc = torch.empty_like(src) if need_src else None
    dflow = torch.empty_strided(flow.shape, flow.stride(), dtype=torch.float32, device=flow.device) if need_flow else None
    sn, sc, sy, sx = _flow_strides(flow)
    det = torch.empty(b * h * w + 1, dtype=torch.int64, device=src.device) if (DETERMINISTIC and need_src) else None
    _lib.check(lib.ctg_warp_bwd(_p(src), _p(flow), sn, sc, sy, sx, _p(gout), _p(dsrc), _p(dflow), b, h, w, _p(det), _stream()),
               "ctg_warp_bwd")
    return dsrc, dflow


def _scratch(dev):
    return torch.empty(4096, dtype=torch.float32, device=dev)


def smooth_fwd(f, weight=1.0):
    lib = _lib.load()
    b, c, h, w = f.shape
    out = torch.empty((), dtype=torch.float32, device=f.device)
    _lib.check(lib.ctg_smooth_fwd(_p(f), f.stride(0), f.stride(1), f.stride(2), f.stride(3), b, c, h, w, float(weight),
                                  _p(_scratch(f.device)), _p(out), _stream()), "ctg_smooth_fwd")
    return out


def smooth_bwd(f, gscale, weight=1.0):
    lib = _lib.load()
    b, c, h, w = f.shape
    df = torch.empty_strided(f.shape, f.stride(), dtype=torch.float32, device=f.device)
    _lib.check(lib.ctg_smooth_bwd(_p(f), f.stride(0), f.stride(1), f.stride(2), f.stride(3), b, c, h, w, float(weight),
                                  _p(gscale), _p(df), 0, _stream()), "ctg_smooth_bwd")
    return df


def l1_fwd(a, b, mask, weight=1.0):
    lib = _lib.load()
    out = torch.empty((), dtype=torch.float32, device=a.device)
    _lib.check(lib.ctg_l1_fwd(_p(a), _p(b), _p(mask), a.numel(), float(weight), _p(_scratch(a.device)), _p(out), _stream()),
               "ctg_l1_fwd")
    return out


def l1_bwd(a, b, mask, gscale, weight=1.0):
    lib = _lib.load()
    da = torch.empty_like(a)
    _lib.check(lib.ctg_l1_bwd(_p(a), _p(b), _p(mask), a.numel(), float(weight), _p(gscale), _p(da), 0, _stream()),
               "ctg_l1_bwd")
    return da


def lsgan_fwd(x, nb, t0, s0, t1, s1, mode=0):
    """x: dense fp32 (B, 1, H, W) PatchGAN map -> (loss scalar, pooled [B]): sum_b s_b loss(mean(x[b]), t_b) with (t0, s0) for
    the first nb samples and (t1, s1) for the rest; mode 0: squared error (LSGAN), 1: binary cross entropy (ctg_lsgan_fwd)."""
    lib = _lib.load()
    b = x.shape[0]
    hw = x.numel() // b
    pooled = torch.empty((b,), dtype=torch.float32, device=x.device)
    out = torch.empty((), dtype=torch.float32, device=x.device)
    _lib.check(lib.ctg_lsgan_fwd(_p(x), b, hw, nb, t0, s0, t1, s1, mode, _p(pooled), _p(out), _stream()), "ctg_lsgan_fwd")
    return out, pooled


def lsgan_bwd(pooled, shape, nb, t0, s0, t1, s1, gscale, mode=0):
    lib = _lib.load()
    b = shape[0]
    dx = torch.empty(shape, dtype=torch.float32, device=pooled.device)
    _lib.check(lib.ctg_lsgan_bwd(_p(pooled), b, dx.numel() // b, nb, t0, s0, t1, s1, mode, _p(gscale), _p(dx), _stream()),
               "ctg_lsgan_bwd")
    return dx


def sum_scalars(scalars):
    """Sum of up to 8 one-element fp32 device tensors, one launch."""
    lib = _lib.load()
    n = len(scalars)
    assert 1 <= n <= 8 and all(t.numel() == 1 and t.dtype == torch.float32 and t.is_cuda for t in scalars)
    out = torch.empty((), dtype=torch.float32, device=scalars[0].device)
    vp = ctypes.c_void_p * n
    _lib.check(lib.ctg_sum_scalars(n, vp(*[t.data_ptr() for t in scalars]), _p(out), _stream()), "ctg_sum_scalars")
    return out


def act_bwd_sum_f32(g, y, act, sum_out, accumulate=False):
    """g * act'(y) for dense fp32 tensors, and sum_out[0] (+)= its sum (the bias gradient of a 1-channel conv) in the same pass."""
    lib = _lib.load()
    assert g.dtype == torch.float32 and y.dtype == torch.float32 and g.is_contiguous() and y.is_contiguous()
    assert g.numel() == y.numel() and sum_out.numel() == 1 and sum_out.dtype == torch.float32
    out = torch.empty_like(g)
    _lib.check(lib.ctg_act_bwd_sum_f32(_p(g), _p(y), act, _p(out), g.numel(), _p(_scratch(g.device)), _p(sum_out),
                                       int(accumulate), _stream()), "ctg_act_bwd_sum_f32")
    return out


def avgpool_fwd(x):
    """x: dense fp32 (B, 1, H, W) -> (B, 1)."""
    lib = _lib.load()
    b = x.shape[0]
    hw = x.numel() // b
    out = torch.empty((b, 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.ctg_avgpool_fwd(_p(x), b, hw, _p(out), _stream()), "ctg_avgpool_fwd")
    return out


def avgpool_bwd(gout, shape):
    lib = _lib.load()
    b = shape[0]
    dx = torch.empty(shape, dtype=torch.float32, device=gout.device)
    hw = dx.numel() // b
    _lib.check(lib.ctg_avgpool_bwd(_p(gout.contiguous()), b, hw, _p(dx), _stream()), "ctg_avgpool_bwd")
    return dx


def adam_tick(state3, beta1, beta2):
    lib = _lib.load()
    _lib.check(lib.ctg_adam_tick(_p(state3), beta1, beta2, _stream()), "ctg_adam_tick")


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, dev_state=None):
    lib = _lib.load()
    n = len(params)
    if n == 0:
        return
    vp = ctypes.c_void_p * n
    numel = (ctypes.c_long * n)(*[p.numel() for p in params])
    _lib.check(lib.ctg_adam_step(n, vp(*[p.data_ptr() for p in params]), vp(*[g.data_ptr() for g in grads]),
                                 vp(*[m.data_ptr() for m in exp_avg]), vp(*[v.data_ptr() for v in exp_avg_sq]), numel,
                                 lr, beta1, beta2, eps, step, _p(dev_state), _stream()), "ctg_adam_step")


# ---------------------------------------------------------------------------- evaluation (test() loop)
def _win_vec(v, b, dev):
    t = torch.as_tensor(v, dtype=torch.float32, device=dev).reshape(-1)
    return t.expand(b).contiguous() if t.numel() == 1 else t.contiguous()


def to_windowdata(img, wc, ww):
    """to_windowdata (trainer/HdTrainer.py:41-64) of B slices: img (B, ..., H, W) fp32 on the GPU; wc / ww scalars or
    per-slice vectors.  Returns a new tensor of the same shape."""
    lib = _lib.load()
    if not img.is_cuda:
        raise RuntimeError("to_windowdata: CPU tensors are not supported (no CPU fallback)")
    x = img.float().contiguous()
    b = x.shape[0]
    hw = x.numel() // b
    out = torch.empty_like(x)
    wcv, wwv = _win_vec(wc, b, x.device), _win_vec(ww, b, x.device)
    assert wcv.numel() == b and wwv.numel() == b
    _lib.check(lib.ctg_to_windowdata(_p(x), _p(wcv), _p(wwv), _p(out), b, hw, _stream()), "ctg_to_windowdata")
    return out


def window_metrics(fake, real, wc, ww, aliased=False):
    """Windowed and raw MAE / PSNR / UQI of B slices (HdTrainer.py:1008-1050, 1089-1125): fake, real (B, ..., H, W)
    fp32 on the GPU -> float64 tensor [B, 2, 3] = {windowed, raw} x {MAE, PSNR, UQI} (on the GPU, no sync).
    aliased=True: the CycTrainer.py:288-298 variant (its `bb = b` / `cc = c` aliases make the windowed pair binary)."""
    lib = _lib.load()
    if not (fake.is_cuda and real.is_cuda):
        raise RuntimeError("window_metrics: CPU tensors are not supported (no CPU fallback)")
    f, r = fake.float().contiguous(), real.float().contiguous()
    assert f.shape == r.shape
    b = f.shape[0]
    hw = f.numel() // b
    nblk = int(max(1, min(64, hw // 4096)))
    part = torch.empty((b, nblk, 20), dtype=torch.float64, device=f.device)
    out = torch.empty((b, 2, 3), dtype=torch.float64, device=f.device)
    wcv, wwv = _win_vec(wc, b, f.device), _win_vec(ww, b, f.device)
    assert wcv.numel() == b and wwv.numel() == b
    _lib.check(lib.ctg_window_metrics(_p(f), _p(r), _p(wcv), _p(wwv), b, hw, nblk, int(aliased), _p(part), _p(out), _stream()),
               "ctg_window_metrics")
    return out


def _ssim_launch(fake, real, wc, ww, mode, aliased, data_range):
    lib = _lib.load()
    if not (fake.is_cuda and real.is_cuda):
        raise RuntimeError("ssim: CPU tensors are not supported (no CPU fallback)")
    f, r = fake.float().contiguous(), real.float().contiguous()
    assert f.shape == r.shape and f.dim() >= 3
    b, h, w = f.shape[0], f.shape[-2], f.shape[-1]
    if f.numel() != b * h * w:
        raise RuntimeError("ssim: one plane per slice (B, [1,] H, W)")
    if h < 7 or w < 7:
        raise ValueError("win_size exceeds image extent (7x7 windows need H, W >= 7)")     # skimage's own error for this case
    nblk = ((h - 6 + 15) // 16) * ((w - 6 + 15) // 16)
    part = torch.empty((b, nblk, 2), dtype=torch.float64, device=f.device)
    out = torch.empty((b, 2 if mode else 1), dtype=torch.float64, device=f.device)
    wcv = wwv = None
    if mode:
        wcv, wwv = _win_vec(wc, b, f.device), _win_vec(ww, b, f.device)
        assert wcv.numel() == b and wwv.numel() == b
    _lib.check(lib.ctg_ssim(_p(f), _p(r), _p(wcv), _p(wwv), b, h, w, mode, int(aliased), float(data_range), _p(part), _p(out),
                            _stream()), "ctg_ssim")
    return out


def ssim(fake, real, data_range=2.0):
    """Mean structural similarity of B slice pairs (B, [1,] H, W) fp32 on the GPU -> float64 [B] on the GPU (no sync): what
    `skimage.measure.compare_ssim(fake, real)` returns for float images with its defaults -- the validation pass of the trainers
    (trainer/HdTrainer.py:779, CycTrainer.py:216, p2pTrainer.py:164, RegTrainer.py:219)."""
    return _ssim_launch(fake, real, None, None, 0, False, data_range)[:, 0]


def window_ssim(fake, real, wc, ww, aliased=False, data_range=2.0):
    """SSIMw / SSIM of the test() loop (HdTrainer.py:1028, 1053): the masked pairs of `window_metrics` -> float64 [B, 2] =
    {windowed (c, b), raw (fake cc, real bb)}."""
    return _ssim_launch(fake, real, wc, ww, 1, aliased, data_range)


def val_psnr(fake, real):
    """`PSNR(fake, real)` of the trainers (HdTrainer.py:566-580) per slice -> float64 [B] on the GPU: mean squared difference of
    (x + 1) / 2 over the pixels where real != -1 (all pixels + 1e-10 if there are none), 100 below 1e-10."""
    b = fake.shape[0]
    f, r = fake.float().reshape(b, -1), real.float().reshape(b, -1)
    m = r != -1
    d2 = (((f + 1) / 2. - (r + 1) / 2.) ** 2).double()
    cnt = m.sum(1)
    mse = torch.where(cnt > 0, (d2 * m).sum(1) / cnt.clamp_min(1), d2.mean(1) + 1e-10)
    return torch.where(mse < 1.0e-10, torch.full_like(mse, 100.0), 20 * torch.log10(1 / (torch.sqrt(mse) + 1e-10)))


# ---------------------------------------------------------------------------- input pipeline (datasets.py / utils.py)
def hu_to_inputs(hu, wc=50.0, ww=400.0):
    """read_ori_w (trainer/datasets.py:36-71) after the DICOM read: raw HU int16 tensor (SimpleITK convention) on the GPU
    -> (windowed image, full-range image), both fp32 in [-1, 1] and of hu's shape."""
    lib = _lib.load()
    if not hu.is_cuda:
        raise RuntimeError("hu_to_inputs: CPU tensors are not supported (no CPU fallback)")
    h = hu.to(torch.int16).contiguous()
    win = torch.empty(h.shape, dtype=torch.float32, device=h.device)
    full = torch.empty_like(win)
    _lib.check(lib.ctg_hu_to_inputs(_p(h), float(wc), float(ww), _p(win), _p(full), h.numel(), _stream()),
               "ctg_hu_to_inputs")
    return win, full


def resize_nearest(x, size):
    """F.interpolate(x, size=size) (mode 'nearest', trainer/utils.py:13-32) of (..., H, W) fp32 planes on the GPU."""
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError("resize_nearest: CPU tensors are not supported (no CPU fallback)")
    xs = x.float().contiguous()
    hi, wi = xs.shape[-2:]
    ho, wo = int(size[0]), int(size[1])
    b = xs.numel() // (hi * wi)
    out = torch.empty(xs.shape[:-2] + (ho, wo), dtype=torch.float32, device=xs.device)
    _lib.check(lib.ctg_resize_nearest(_p(xs), b, hi, wi, _p(out), ho, wo, _stream()), "ctg_resize_nearest")
    return out
