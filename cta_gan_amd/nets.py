"""The networks of the hot path as single autograd nodes over the HIP tape.

Each network (`GeneratorNet`, `PatchDiscriminatorNet`, `RegNet`) is an
`nn.Module` that owns fp32 master parameters under exactly the reference's
`state_dict` keys and whose forward runs `engine` ops on the
current HIP stream.  torch.autograd sees ONE node per call (`_NetFn`); inside,
the tape in engine.py owns saved tensors and the backward schedule.

The public, reference-named classes live in `cta_gan_amd/Model/*.py` and
`cta_gan_amd/trainer/*.py`; they subclass or wrap what is here.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Sequence

import torch
import torch.nn as nn

from . import engine as E
from . import ops
from .engine import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, Act, ConvSpec, Tape

_DEFAULT_DTYPE = torch.float32


def set_default_compute_dtype(dtype):
    """torch.float32 (exact-f32 MFMA; the parity mode), torch.bfloat16 (bf16 storage + bf16 MFMA, fp32 accumulate /
    statistics / parameters; the throughput mode of BASELINE.json configs[2]) or "bf16x3" (split-pair storage -- every wide
    tensor as [hi | lo] bf16 planes -- and every conv contraction as hi.hi + hi.lo + lo.hi on the bf16 matrix cores: the
    north_star's 1e-3 rel-L2 at a third of the bf16 MFMA rate; see ops.PAIR)."""
    global _DEFAULT_DTYPE
    pair = isinstance(dtype, str) and dtype in ("bf16x3", "bf16x3f")
    bwd_plain = pair and (dtype == "bf16x3f" or bool(os.environ.get("CTG_X3F")))
    if pair:
        dtype = torch.bfloat16
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError('compute dtype must be torch.float32, torch.bfloat16, "bf16x3" or "bf16x3f"')
    _DEFAULT_DTYPE = dtype
    ops.set_pair_mode(pair, bwd_plain)


def default_compute_dtype():
    return _DEFAULT_DTYPE


def compute_mode():
    """"fp32", "bf16", "bf16x3" or "bf16x3f"."""
    if ops.PAIR:
        return "bf16x3f" if ops.PAIR_BWD_PLAIN else "bf16x3"
    return "bf16" if _DEFAULT_DTYPE == torch.bfloat16 else "fp32"


# ----------------------------------------------------------------------------- parameter tree helpers
class _Slot(nn.Module):
    """A weight/bias pair living at one index of a reference nn.Sequential (keeps the state_dict key)."""

    def __init__(self, wshape, bshape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(wshape))
        self.bias = nn.Parameter(torch.empty(bshape))


class _BNSlot(nn.Module):
    """nn.BatchNorm2d(C)'s state at its index of a reference nn.Sequential: affine weight / bias, running statistics."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = eps, momentum


class _Tree(nn.Module):
    """Anonymous container: children are attached by (possibly numeric) name."""


def _attach(root: nn.Module, dotted: str, mod: nn.Module):
    parts = dotted.split(".")
    cur = root
    for p in parts[:-1]:
        if p not in cur._modules:
            cur.add_module(p, _Tree())
        cur = cur._modules[p]
    cur.add_module(parts[-1], mod)


def _default_conv_init(slot: _Slot):
    """nn.Conv2d / nn.ConvTranspose2d.reset_parameters (the reference never applies weights_init_normal)."""
    nn.init.kaiming_uniform_(slot.weight, a=math.sqrt(5))
    fan_in, _ = nn.init._calculate_fan_in_and_fan_out(slot.weight)
    bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
    nn.init.uniform_(slot.bias, -bound, bound)


def _to_nhwc(x: torch.Tensor, dtype) -> torch.Tensor:
    """logical (B, C, H, W) -> dense physical [B, H, W, C] of `dtype` (no copy when already so); in the split-pair mode a wide
    tensor becomes a split pair."""
    t = x.permute(0, 2, 3, 1)
    if ops.PAIR and dtype == torch.bfloat16:
        return ops.to_pair(t.float().contiguous())
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _to_nchw_view(t: torch.Tensor) -> torch.Tensor:
    if ops.is_pair(t):      # a wide tensor leaving the network (a feature map, a stand-alone block's result): fp32 again
        t = ops.from_pair(t)
    return t.permute(0, 3, 1, 2)


_NO_MULTI_REDUCE = bool(os.environ.get("CTG_NO_MULTI_REDUCE"))   # A/B switch (scripts/ab.sh)


def _require_cuda(x):
    if not x.is_cuda:
        raise RuntimeError("cta_gan_amd runs on an MI355X only: got a %s tensor and there is no CPU fallback "
                           "(use oracle/ for CPU reference results in tests)" % x.device)


# ----------------------------------------------------------------------------- the autograd node
class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, n_in, *tensors):
        n_in, grad_on = n_in      # (number of data inputs, was grad mode on at the call: inside forward() it never is)
        inputs = tensors[:n_in]
        # outputs nobody differentiates (D_m's intermediate feature maps) get grad None, not a materialised zero map
        ctx.set_materialize_grads(False)
        # `needs_input_grad` only mirrors `requires_grad`: under torch.no_grad() nothing will ever ask for a backward, so no
        # tape is recorded and no activation is kept for it (round 1 built and dropped one in every no-grad forward)
        need_in = [bool(f) and grad_on for f in ctx.needs_input_grad[2:2 + n_in]]
        need_any = grad_on and any(ctx.needs_input_grad[2:])
        tape = Tape(need_any)
        net._cache.refresh()      # all weight packs an optimiser step invalidated, in one launch
        out_acts, in_acts, finish = net._run(tape, inputs, need_in)
        ctx.tape, ctx.out_acts, ctx.in_acts, ctx.finish = tape, out_acts, in_acts, finish
        ctx.params = tensors[n_in:]
        ctx.net = net
        ctx.n_in = n_in
        ctx.in_shapes = [tuple(t.shape) for t in inputs]
        outs = tuple(_to_nchw_view(a.t) for a in out_acts)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        tape = ctx.tape
        if tape is None:
            raise RuntimeError("this network call was already back-propagated (the tape keeps no second copy)")
        grads: Dict[int, torch.Tensor] = {}
        E._PARAM_GRADS = grads
        E._REDUCE_JOBS = [] if not _NO_MULTI_REDUCE else None
        # "bf16x3f": the backward of a split-pair forward runs in the plain bf16 mode.  Inside ops.plain_backward() ops.PAIR reads
        # False on this thread and every bf16 handle is taken as a plain bf16 tensor with its own pixel pitch -- so a saved split-pair
        # activation (pitch 2C) is read as its hi plane, new gradient tensors are plain bf16, weight packs are bf16 and every launch
        # is the bf16 mode's (except the DT_MIX ones: ops.dtc_saved).
        try:
            with ops.plain_backward():
                for a, g in zip(ctx.out_acts, gouts):
                    if g is None:
                        continue
                    E.add_grad(a, _to_nhwc(g, a.t.dtype), 0)
                tape.backward()
                E.flush_reduces()     # every split-K weight-gradient partial of this backward, one launch
                E.fire_mark(ctx.net, "done")
                in_grads = ctx.finish(ctx.in_acts) if ctx.finish is not None else [None] * ctx.n_in
        finally:
            E._PARAM_GRADS = None
            E._REDUCE_JOBS = None
        res = [None, None]
        for i in range(ctx.n_in):
            res.append(in_grads[i] if ctx.needs_input_grad[2 + i] else None)
        for p in ctx.params:
            res.append(grads.get(id(p)))
        ctx.tape = None
        ctx.net = None
        ctx.out_acts = ctx.in_acts = None
        return tuple(res)


class HipNet(nn.Module):
    """Base: parameters under reference keys + a `_run(tape, inputs, need_in)` that executes HIP ops."""

    def __init__(self):
        super().__init__()
        self.compute_dtype = None  # None -> nets.default_compute_dtype() at call time
        self._cache = E.PackCache()

    @property
    def dtype_(self):
        return self.compute_dtype or _DEFAULT_DTYPE

    def _call(self, *inputs):
        for x in inputs:
            _require_cuda(x)
        params = [p for p in self.parameters()]
        return _NetFn.apply(self, (len(inputs), torch.is_grad_enabled()), *inputs, *params)

    def _p(self, key: str):
        mod = self
        for part in key.split("."):
            mod = mod._modules[part] if part in mod._modules else getattr(mod, part)
        return mod


def _img_plane(x: torch.Tensor, c: int) -> torch.Tensor:
    """channel c of a logical (B, C, H, W) fp32 tensor as a dense [B, H, W] image."""
    t = x[:, c]
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _image_grad_finish(n_ch):
    def finish(in_acts):
        a = in_acts[0]
        g, _ = E.take_grad(a)
        if g is None:
            return [None]
        return [_to_nchw_view(g)]  # [B,H,W,C] fp32 -> logical (B,C,H,W)
    return finish


# ----------------------------------------------------------------------------- Generator
class GeneratorNet(HipNet):
    """9-block ResNet generator -- Model/HdGan.py:65-113 == Model/CycleGan.py:23-71."""

    def grad_buckets(self):
        """Parameters in the two halves `fire_mark(self, "mid")` separates, in backward order: [(params, tag)]."""
        late, early = [], []
        split = self.n_blocks // 2
        for key, p in self.named_parameters():
            parts = key.split(".")
            is_late = parts[0] == "model_tail" or (parts[0] == "model_body" and int(parts[1]) > split)
            (late if is_late else early).append(p)
        return [(late, "mid"), (early, "done")]

    def __init__(self, input_nc: int, output_nc: int, n_residual_blocks: int = 9):
        super().__init__()
        if input_nc not in (1, 2) or not 1 <= output_nc <= 4:
            raise NotImplementedError("HIP generator supports input_nc in {1,2}, output_nc <= 4 (reference uses 1/1)")
        self.input_nc, self.output_nc, self.n_blocks = input_nc, output_nc, n_residual_blocks
        # state_dict keys: model_head.{1,4,7}, model_body.{i}.conv_block.{1,5}, model_tail.{0,3,7}
        shapes = [("model_head.1", (64, input_nc, 7, 7), 64), ("model_head.4", (128, 64, 3, 3), 128),
                  ("model_head.7", (256, 128, 3, 3), 256)]
        for i in range(n_residual_blocks):
            shapes += [("model_body.%d.conv_block.1" % i, (256, 256, 3, 3), 256),
                       ("model_body.%d.conv_block.5" % i, (256, 256, 3, 3), 256)]
        shapes += [("model_tail.0", (256, 128, 3, 3), 128), ("model_tail.3", (128, 64, 3, 3), 64),
                   ("model_tail.7", (output_nc, 64, 7, 7), output_nc)]
        for key, ws, bs in shapes:
            slot = _Slot(ws, (bs,))
            _attach(self, key, slot)
            _default_conv_init(slot)
        self.s_head = ConvSpec(input_nc, 64, 7, 1, 3, reflect=True, use_bias=False)
        self.s_d1 = ConvSpec(64, 128, 3, 2, 1, use_bias=False)
        self.s_d2 = ConvSpec(128, 256, 3, 2, 1, use_bias=False)
        self.s_res = ConvSpec(256, 256, 3, 1, 1, reflect=True, use_bias=False)
        self.s_u1 = ConvSpec(256, 128, 3, 2, 1, transposed=True, use_bias=False)
        self.s_u2 = ConvSpec(128, 64, 3, 2, 1, transposed=True, use_bias=False)
        self.s_tail = ConvSpec(64, output_nc, 7, 1, 3, reflect=True, use_bias=True, act=ACT_TANH, out_f32=True)

    def forward(self, x):
        return self._call(x)[0]

    def _run(self, tape: Tape, inputs, need_in):
        (x,) = inputs
        dt, cache = self.dtype_, self._cache
        b, c, h, w = x.shape
        assert c == self.input_nc
        if h % 4 or w % 4:
            raise RuntimeError("generator input height/width must be multiples of 4")
        x_act = Act(torch.zeros(1, device=x.device).expand(b, h, w, c), req=need_in[0])
        srcs = (_img_plane(x, 0), _img_plane(x, 1) if c == 2 else None)

        def wb(key):
            s = self._p(key)
            return s.weight, s.bias

        a = E.conv_forward(tape, cache, self.s_head, x_act, *wb("model_head.1"), dt, img_sources=srcs)
        a = E.inorm_forward(tape, a, ACT_RELU)
        a = E.conv_forward(tape, cache, self.s_d1, a, *wb("model_head.4"), dt)
        a = E.inorm_forward(tape, a, ACT_RELU)
        a = E.conv_forward(tape, cache, self.s_d2, a, *wb("model_head.7"), dt)
        a = E.inorm_forward(tape, a, ACT_RELU)
        for i in range(self.n_blocks):
            a = _res_block(tape, cache, self.s_res, a, wb("model_body.%d.conv_block.1" % i),
                           wb("model_body.%d.conv_block.5" % i), dt)
            if i == self.n_blocks // 2:
                # backward passes this point with the gradients of the later blocks and the tail complete: a
                # data-parallel run starts their all-reduce here, behind the remaining half of the backward (dp.py)
                tape.record(lambda: E.fire_mark(self, "mid"))
        a = E.conv_forward(tape, cache, self.s_u1, a, *wb("model_tail.0"), dt)
        a = E.inorm_forward(tape, a, ACT_RELU)
        a = E.conv_forward(tape, cache, self.s_u2, a, *wb("model_tail.3"), dt)
        a = E.inorm_forward(tape, a, ACT_RELU)
        a = E.conv_forward(tape, cache, self.s_tail, a, *wb("model_tail.7"), dt)
        return [a], [x_act], _image_grad_finish(c)


def _res_block(tape, cache, spec, x: Act, wb1, wb2, dt, out_t=None) -> Act:
    """x + IN(conv(rpad(relu(IN(conv(rpad(x)))))))  -- Model/HdGan.py:49-63; trainer/layers.py:243-300.
    `out_t`: where the block's output lands (a channel slice of a U-Net concat buffer)."""
    h = E.conv_inorm_forward(tape, cache, spec, x, wb1[0], wb1[1], dt, ACT_RELU)
    return E.conv_inorm_forward(tape, cache, spec, h, wb2[0], wb2[1], dt, ACT_NONE, res=x, out_t=out_t)


class ResidualBlockNet(HipNet):
    """Stand-alone residual block (Model/HdGan.py:49-63): NCHW in/out at the compute dtype's precision."""

    def __init__(self, in_features: int):
        super().__init__()
        c = in_features
        for idx in (1, 5):
            slot = _Slot((c, c, 3, 3), (c,))
            _attach(self, "conv_block.%d" % idx, slot)
            _default_conv_init(slot)
        self.spec = ConvSpec(c, c, 3, 1, 1, reflect=True, use_bias=False)

    def forward(self, x):
        return self._call(x)[0].to(x.dtype)

    def _run(self, tape, inputs, need_in):
        (x,) = inputs
        dt = self.dtype_
        xa = Act(_to_nhwc(x, dt), req=need_in[0])
        s1, s5 = self._p("conv_block.1"), self._p("conv_block.5")
        out = _res_block(tape, self._cache, self.spec, xa, (s1.weight, s1.bias), (s5.weight, s5.bias), dt)

        def finish(in_acts):
            g, _ = E.take_grad(in_acts[0])
            return [None if g is None else _to_nchw_view(g).to(x.dtype)]
        return [out], [xa], finish


# ----------------------------------------------------------------------------- PatchGAN discriminator
class PatchStack:
    """The 4x4 conv stack of Discriminator / NLayerDiscriminator (Model/HdGan.py:115-205) with InstanceNorm.
    Not a Module: its parameter slots are attached to `owner` under `keys` (the reference's state_dict
    prefixes of the convs), and `owner` (a HipNet) is the autograd node."""

    def __init__(self, owner: nn.Module, input_nc: int, keys: Sequence[str], ndf: int = 64, n_layers: int = 3,
                 sigmoid: bool = False, norm: str = "instance", norm_keys: Sequence[str] = ()):
        """norm: "instance" (affine-free nn.InstanceNorm2d: what every trainer of the reference passes) or "batch"
        (nn.BatchNorm2d, NLayerDiscriminator's own default: `norm_keys` = the state_dict prefixes of the norm layers behind
        convs 1 .. n_layers; the conv biases are live then)."""
        if input_nc not in (1, 2):
            raise NotImplementedError("HIP discriminator supports input_nc in {1, 2}")
        self.norm = norm
        self.input_nc = input_nc
        chans = [input_nc, ndf]
        nf = ndf
        for _ in range(1, n_layers):
            nf = min(nf * 2, 512)
            chans.append(nf)
        chans.append(min(nf * 2, 512))
        chans.append(1)
        self.chans = chans
        nconv = len(chans) - 1
        assert len(keys) == nconv
        self.keys = list(keys)
        root = owner
        self._root = [root]
        for i, key in enumerate(keys):
            slot = _Slot((chans[i + 1], chans[i], 4, 4), (chans[i + 1],))
            _attach(root, key, slot)
            _default_conv_init(slot)
        bn = norm == "batch"      # (a BatchNorm in eval mode does not cancel the conv bias)
        self.norm_keys = list(norm_keys)
        if bn:
            assert len(self.norm_keys) == nconv - 2
            for i, key in enumerate(self.norm_keys):
                _attach(root, key, _BNSlot(chans[i + 2]))
        self.specs = [ConvSpec(chans[0], chans[1], 4, 2, 1, use_bias=True, act=ACT_LRELU)]
        for i in range(1, nconv - 2):
            self.specs.append(ConvSpec(chans[i], chans[i + 1], 4, 2, 1, use_bias=bn))
        self.specs.append(ConvSpec(chans[nconv - 2], chans[nconv - 1], 4, 1, 1, use_bias=bn))
        # use_sigmoid (Model/HdGan.py:177-178): nn.Sigmoid() behind the last conv, fused into its epilogue
        self.specs.append(ConvSpec(chans[nconv - 1], 1, 4, 1, 1, use_bias=True, out_f32=True,
                                   act=ACT_SIGMOID if sigmoid else ACT_NONE))

    def _slots(self):
        root = self._root[0]
        out = []
        for key in self.keys:
            mod = root
            for part in key.split("."):
                mod = mod._modules[part]
            out.append(mod)
        return out

    def _norm_slot(self, j):
        mod = self._root[0]
        for part in self.norm_keys[j].split("."):
            mod = mod._modules[part]
        return mod

    def run_stack(self, tape: Tape, cache, x: torch.Tensor, need_in: bool, dt):
        """x: logical (B, C, H, W) fp32.  Returns (feature Acts [5], input Act)."""
        b, c, h, w = x.shape
        assert c == self.input_nc
        slots = self._slots()
        x_act = Act(torch.zeros(1, device=x.device).expand(b, h, w, c), req=need_in)
        srcs = (_img_plane(x, 0), _img_plane(x, 1) if c == 2 else None)
        feats = []
        a = E.conv_forward(tape, cache, self.specs[0], x_act, slots[0].weight, slots[0].bias, dt, img_sources=srcs)
        feats.append(a)
        for i in range(1, len(self.specs) - 1):
            a = E.conv_forward(tape, cache, self.specs[i], a, slots[i].weight, slots[i].bias, dt)
            if self.norm == "batch":
                a = E.bnorm_forward(tape, a, ACT_LRELU, self._norm_slot(i - 1))
            else:
                a = E.inorm_forward(tape, a, ACT_LRELU)
            feats.append(a)
        a = E.conv_forward(tape, cache, self.specs[-1], a, slots[-1].weight, slots[-1].bias, dt)
        feats.append(a)
        return feats, x_act


# ----------------------------------------------------------------------------- registration U-Net
class RegNet(HipNet):
    """ResUnet cfg 'A' -- trainer/reg.py:31-99 on top of trainer/layers.py:71-300."""

    NDF = [32, 64, 64, 64, 64, 64, 64]
    NUF = [64, 64, 64, 64, 64, 64, 32]

    def __init__(self, nc_a: int, nc_b: int):
        super().__init__()
        if nc_a != 1 or nc_b != 1:
            raise NotImplementedError("HIP Reg supports single-channel image pairs (the reference's configuration)")
        self.add_module("offset_map", _Tree())

        def conv(key, cin, cout, k, act_lrelu=True, zeros=False):
            slot = _Slot((cout, cin, k, k), (cout,))
            _attach(self._modules["offset_map"], key + ".conv2d", slot)
            if zeros:
                nn.init.normal_(slot.weight, mean=0.0, std=1e-5)        # layers.py:44-45 ('zeros')
            elif act_lrelu:
                nn.init.kaiming_normal_(slot.weight, a=0.2, nonlinearity="leaky_relu", mode="fan_in")
            else:
                nn.init.kaiming_normal_(slot.weight, a=0.0, nonlinearity="relu", mode="fan_in")
            slot.bias.data.zero_()

        def resblocks(key, dim, n):
            for i in range(n):
                for idx in (1, 5):
                    slot = _Slot((dim, dim, 3, 3), (dim,))
                    _attach(self._modules["offset_map"], "%s.model.%d.conv_block.%d" % (key, i, idx), slot)
                    nn.init.kaiming_normal_(slot.weight, a=0.0, nonlinearity="relu", mode="fan_in")  # layers.py:226
                    slot.bias.data.zero_()

        cin = nc_a + nc_b
        # construction order == the reference's module order, so a seeded default init draws the same stream
        for i, c in enumerate(self.NDF, start=1):
            conv("down_%d.conv_0" % i, cin, c, 3)
            resblocks("down_%d.conv_0.resnet_block" % i, c, 1)
            cin = c
        conv("c1", cin, 2 * cin, 1)
        resblocks("t", 2 * cin, 3)
        conv("c2", 2 * cin, cin, 1)
        n = len(self.NDF)
        for i, c in zip(range(n, 0, -1), self.NUF):
            conv("up_%d" % i, cin + self.NDF[i - 1], c, 3)
            cin = c
        resblocks("refine.0", cin, 1)
        conv("refine.1", cin, cin, 1)
        conv("output", cin, 2, 3, act_lrelu=False, zeros=True)

    def _wb(self, key):
        mod = self._modules["offset_map"]
        for part in key.split("."):
            mod = mod._modules[part]
        return mod.weight, mod.bias

    def forward(self, img_a, img_b):
        return self._call(img_a, img_b)[0]

    def _resblocks(self, tape, x, key, dim, n, dt, out_t=None):
        spec = ConvSpec(dim, dim, 3, 1, 1, reflect=True, use_bias=False)
        for i in range(n):
            x = _res_block(tape, self._cache, spec, x, self._wb("%s.model.%d.conv_block.1" % (key, i)),
                           self._wb("%s.model.%d.conv_block.5" % (key, i)), dt, out_t=out_t if i == n - 1 else None)
        return x

    def _run(self, tape: Tape, inputs, need_in):
        img_a, img_b = inputs
        dt, cache = self.dtype_, self._cache
        b, _, h, w = img_a.shape
        if h % 128 or w % 128 or h < 256 or w < 256:
            raise RuntimeError("Reg input must be a multiple of 128 and >= 256 pixels: 7 pooling levels and a reflection-"
                               "padded bottleneck need >= 2x2 there (the reference fails the same way below 256)")
        need = need_in[0] or need_in[1]
        x_act = Act(torch.zeros(1, device=img_a.device).expand(b, h, w, 2), req=need)
        srcs = (_img_plane(img_a, 0), _img_plane(img_b, 0))
        skips = []
        x = None
        cin = 2
        # channels arriving from below at decoder stage i (trainer/reg.py:91-95): the encoder output of level i is
        # written straight into the channel slice [c_up, c_up + NDF[i-1]) of that stage's concat buffer
        n_lv = len(self.NDF)
        c_up, cprev = {}, self.NDF[-1]
        for i, c in zip(range(n_lv, 0, -1), self.NUF):
            c_up[i], cprev = cprev, c
        hh, ww = h, w
        for i, c in enumerate(self.NDF, start=1):
            spec = ConvSpec(cin, c, 3, 1, 1, use_bias=True, act=ACT_LRELU)
            wgt, bias = self._wb("down_%d.conv_0.conv2d" % i)
            if i == 1:
                x = E.conv_forward(tape, cache, spec, x_act, wgt, bias, dt, img_sources=srcs)
            else:
                x = E.conv_forward(tape, cache, spec, x, wgt, bias, dt)
            buf = ops.empty_act((b, hh, ww, c_up[i] + c), dt, img_a.device)
            x = self._resblocks(tape, x, "down_%d.conv_0.resnet_block" % i, c, 1, dt, out_t=buf[..., c_up[i]:])
            skips.append((x, buf))
            x = E.maxpool_forward(tape, x)
            cin = c
            hh, ww = hh // 2, ww // 2
        x = E.conv_forward(tape, cache, ConvSpec(cin, 2 * cin, 1, 1, 0, act=ACT_LRELU), x, *self._wb("c1.conv2d"), dt)
        x = self._resblocks(tape, x, "t", 2 * cin, 3, dt)
        x = E.conv_forward(tape, cache, ConvSpec(2 * cin, cin, 1, 1, 0, act=ACT_LRELU), x, *self._wb("c2.conv2d"), dt)
        n = len(self.NDF)
        for i, c in zip(range(n, 0, -1), self.NUF):
            s, buf = skips[i - 1]
            x = E.upsample_concat_forward(tape, x, s, buf=buf)
            x = E.conv_forward(tape, cache, ConvSpec(cin + self.NDF[i - 1], c, 3, 1, 1, act=ACT_LRELU), x,
                               *self._wb("up_%d.conv2d" % i), dt)
            cin = c
        x = self._resblocks(tape, x, "refine.0", cin, 1, dt)
        x = E.conv_forward(tape, cache, ConvSpec(cin, cin, 1, 1, 0, act=ACT_LRELU), x, *self._wb("refine.1.conv2d"), dt)
        x = E.conv_forward(tape, cache, ConvSpec(cin, 2, 3, 1, 1, use_bias=True, out_f32=True), x,
                           *self._wb("output.conv2d"), dt)

        def finish(in_acts):
            g, _ = E.take_grad(in_acts[0])
            if g is None:
                return [None, None]
            gv = _to_nchw_view(g)  # (B, 2, H, W) fp32, channel 0 = img_a, 1 = img_b
            return [gv[:, 0:1] if need_in[0] else None, gv[:, 1:2] if need_in[1] else None]
        return [x], [x_act], finish


# ----------------------------------------------------------------------------- small autograd nodes
class _AvgPoolFn(torch.autograd.Function):
    """F.avg_pool2d(x, full).view(B, -1) for a 1-channel fp32 map (Model/HdGan.py:145,279)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return ops.avgpool_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.avgpool_bwd(g, ctx.shape)


def global_avgpool(x: torch.Tensor) -> torch.Tensor:
    _require_cuda(x)
    if x.shape[1] != 1:
        raise NotImplementedError("global_avgpool: single-channel PatchGAN map expected")
    if x.dtype != torch.float32:
        x = x.float()
    return _AvgPoolFn.apply(x)


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, flow):
        ctx.save_for_backward(src, flow)
        return ops.warp_fwd(src, flow)

    @staticmethod
    def backward(ctx, g):
        src, flow = ctx.saved_tensors
        dsrc, dflow = ops.warp_bwd(src, flow, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dsrc, dflow


def warp(src, flow):
    _require_cuda(src)
    if src.shape[1] != 1 or flow.shape[1] != 2:
        raise NotImplementedError("Transformer_2D: 1-channel source and 2-channel flow expected")
    return _WarpFn.apply(src.float(), flow.float())


class _SmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, weight):
        ctx.save_for_backward(f)
        ctx.weight = weight
        return ops.smooth_fwd(f, weight)

    @staticmethod
    def backward(ctx, g):
        (f,) = ctx.saved_tensors
        return ops.smooth_bwd(f, g.contiguous(), ctx.weight), None


def smoothing_loss(flow, weight=1.0):
    """weight * smooothing_loss(flow): the loss weight rides the reduction's scale (no scalar multiply launch)."""
    _require_cuda(flow)
    return _SmoothFn.apply(flow.float(), float(weight))


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mask, weight):
        a, b = a.contiguous(), b.contiguous()
        mask = mask.contiguous() if mask is not None else None
        ctx.save_for_backward(a, b, mask)
        ctx.weight = weight
        return ops.l1_fwd(a, b, mask, weight)

    @staticmethod
    def backward(ctx, g):
        a, b, mask = ctx.saved_tensors
        return ops.l1_bwd(a, b, mask, g.contiguous(), ctx.weight), None, None, None


def l1_loss(a, b, weight=1.0):
    """weight * mean |a - b| (gradient w.r.t. `a` only: the targets on this path are data)."""
    _require_cuda(a)
    return _L1Fn.apply(a.float(), b.detach().float(), None, float(weight))


def masked_l1_loss(a, b, mask_src, weight=1.0):
    """weight * the stage-2 masked L1 of trainer/HdTrainer.py:726-735 in one pass (bb = mask_src >= 0.3)."""
    _require_cuda(a)
    return _L1Fn.apply(a.float(), b.detach().float(), mask_src.detach().float(), float(weight))


class _LsganFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nb, t0, s0, t1, s1, mode):
        loss, pooled = ops.lsgan_fwd(x, nb, t0, s0, t1, s1, mode)
        ctx.save_for_backward(pooled)
        ctx.args = (tuple(x.shape), nb, t0, s0, t1, s1, mode)
        return loss

    @staticmethod
    def backward(ctx, g):
        (pooled,) = ctx.saved_tensors
        shape, nb, t0, s0, t1, s1, mode = ctx.args
        return ops.lsgan_bwd(pooled, shape, nb, t0, s0, t1, s1, g.contiguous(), mode), None, None, None, None, None, None


def lsgan_loss(patch, target, weight=1.0, bce=False):
    """weight * MSELoss(avg_pool(patch), target) for a 1-channel PatchGAN map (B, 1, h, w): pooling, difference, square, batch
    mean and weight in one fused reduction (GANLoss, Model/HdGan.py:276-285; Discriminator + MSE, HdTrainer.py:211).
    bce=True: nn.BCELoss instead of nn.MSELoss (GANLoss(use_lsgan=False), :266-267; the map is a sigmoid output)."""
    _require_cuda(patch)
    if patch.shape[1] != 1:
        raise NotImplementedError("lsgan_loss: single-channel PatchGAN map expected")
    b = patch.shape[0]
    return _LsganFn.apply(patch.float().contiguous(), b, float(target), float(weight) / b, 0.0, 0.0, int(bool(bce)))


def lsgan_loss_pair(patch, nb, target_first, target_rest, weight=1.0, bce=False):
    """weight * (MSE(avg_pool(patch[:nb]), target_first) + MSE(avg_pool(patch[nb:]), target_rest)): the fake and the real half
    of ONE batched discriminator pass (HdTrainer.py:745-747) without slicing the map."""
    _require_cuda(patch)
    b = patch.shape[0]
    if patch.shape[1] != 1 or not 0 < nb < b:
        raise NotImplementedError("lsgan_loss_pair: (B, 1, h, w) map with 0 < nb < B expected")
    return _LsganFn.apply(patch.float().contiguous(), nb, float(target_first), float(weight) / nb, float(target_rest),
                          float(weight) / (b - nb), int(bool(bce)))


class _SumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *terms):
        return ops.sum_scalars([t.reshape(()) for t in terms])

    @staticmethod
    def backward(ctx, g):
        return tuple(g for _ in ctx.needs_input_grad)


def add_scalars(*terms):
    """Sum of the step's loss terms (HdTrainer.py:736) in one launch; every term receives the incoming gradient."""
    terms = [t for t in terms if t is not None]
    if len(terms) == 1:
        return terms[0]
    return _SumFn.apply(*[t.float() for t in terms])
