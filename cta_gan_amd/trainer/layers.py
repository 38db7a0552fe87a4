"""HIP-backed counterparts of the reference's `trainer/layers.py` building blocks (`Conv`, `DownBlock`,
`ResnetTransformer`, `ResnetBlock`; `:71-104,156-183,216-300`) as stand-alone modules with the reference's
constructor arguments and `state_dict` keys.  Inside `Reg` the whole U-Net runs as ONE autograd node
(cta_gan_amd/nets.py: RegNet); these classes exist for users who compose the blocks themselves.  Tensors cross
the boundary as logical (B, C, H, W); only the configurations the reference instantiates are implemented
(affine-free InstanceNorm, reflect padding in the res-blocks, ReLU / LeakyReLU(0.2) / no activation,
`use_norm=False` in `Conv`), anything else raises NotImplementedError.
"""
from __future__ import annotations

import torch.nn as nn

from .. import engine as E
from ..engine import ACT_LRELU, ACT_NONE, ACT_RELU, ConvSpec
from ..nets import HipNet, ResidualBlockNet, _Slot, _attach, _res_block, _to_nchw_view, _to_nhwc

_ACTS = {"leaky_relu": ACT_LRELU, "relu": ACT_RELU, None: ACT_NONE}


def _init_conv(slot, activation, init_func):
    """get_init_function (layers.py:23-53) for the cases the reference uses."""
    if init_func == "zeros":
        nn.init.normal_(slot.weight, mean=0.0, std=1e-5)
    elif init_func == "kaiming":
        act = "relu" if activation is None else activation
        a = 0.2 if activation == "leaky_relu" else 0.0
        nn.init.kaiming_normal_(slot.weight, a=a, nonlinearity=act, mode="fan_in")
    else:
        raise NotImplementedError("init_func=%r" % (init_func,))
    slot.bias.data.zero_()


class ResnetBlock(ResidualBlockNet):
    """reference: trainer/layers.py:243-300 (keys conv_block.{1,5}.{weight,bias})."""

    def __init__(self, dim, padding_type="reflect", norm_layer=None, use_dropout=False, use_bias=True):
        if padding_type != "reflect" or use_dropout or not use_bias:
            raise NotImplementedError("ResnetBlock: only reflect / no dropout / bias (layers.py:220-222)")
        super().__init__(dim)
        for p in (self._p("conv_block.1"), self._p("conv_block.5")):
            nn.init.kaiming_normal_(p.weight, a=0.0, nonlinearity="relu", mode="fan_in")
            p.bias.data.zero_()


class _BlockNet(HipNet):
    """Shared runner: conv (+act) -> n res-blocks -> optional 2x2 max-pool; one autograd node."""

    def _setup(self, cin, cout, k, stride, pad, activation, init_func, n_res, res_prefix, conv_key, pool):
        if activation not in _ACTS:
            raise NotImplementedError("activation=%r" % (activation,))
        if cin % 32 or cout % 32:
            raise NotImplementedError("stand-alone Conv needs channel counts that are multiples of 32")
        self.spec = ConvSpec(cin, cout, k, stride, pad, use_bias=True, act=_ACTS[activation]) if cin else None
        self.n_res, self.res_prefix, self.conv_key, self.pool, self.dim = n_res, res_prefix, conv_key, pool, cout
        if cin:
            slot = _Slot((cout, cin, k, k), (cout,))
            _attach(self, conv_key, slot)
            _init_conv(slot, activation, init_func)
        for i in range(n_res):
            for idx in (1, 5):
                slot = _Slot((cout, cout, 3, 3), (cout,))
                _attach(self, "%s.%d.conv_block.%d" % (res_prefix, i, idx), slot)
                nn.init.kaiming_normal_(slot.weight, a=0.0, nonlinearity="relu", mode="fan_in")
                slot.bias.data.zero_()

    def _run(self, tape, inputs, need_in):
        (x,) = inputs
        dt = self.dtype_
        xa = E.Act(_to_nhwc(x, dt), req=need_in[0])
        h = xa
        if self.spec is not None:
            s = self._p(self.conv_key)
            h = E.conv_forward(tape, self._cache, self.spec, h, s.weight, s.bias, dt)
        rspec = ConvSpec(self.dim, self.dim, 3, 1, 1, reflect=True, use_bias=False)
        for i in range(self.n_res):
            a, b = self._p("%s.%d.conv_block.1" % (self.res_prefix, i)), self._p("%s.%d.conv_block.5" % (self.res_prefix, i))
            h = _res_block(tape, self._cache, rspec, h, (a.weight, a.bias), (b.weight, b.bias), dt)
        outs = [h]
        if self.pool:
            outs = [E.maxpool_forward(tape, h), h]

        def finish(in_acts):
            g, _ = E.take_grad(in_acts[0])
            return [None if g is None else _to_nchw_view(g).to(x.dtype)]
        return outs, [xa], finish


class Conv(_BlockNet):
    """reference: trainer/layers.py:71-104.  conv -> activation -> (ResnetTransformer(out, 1) if use_resnet);
    keys `conv2d.{weight,bias}` and `resnet_block.model.0.conv_block.{1,5}.*`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias=True, activation="relu",
                 init_func="kaiming", use_norm=False, use_resnet=False, **kwargs):
        super().__init__()
        if use_norm or not bias:
            raise NotImplementedError("Conv: the reference only builds use_norm=False, bias=True on this path")
        self._setup(in_channels, out_channels, kernel_size, stride, padding, activation, init_func,
                    1 if use_resnet else 0, "resnet_block.model", "conv2d", False)

    def forward(self, x):
        return self._call(x)[0].to(x.dtype)


class DownBlock(_BlockNet):
    """reference: trainer/layers.py:156-183.  conv_0 (Conv) then MaxPool2d(2); returns (pooled, skip)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias=False, activation="relu",
                 init_func="kaiming", use_norm=False, use_resnet=False, skip=True, refine=False, pool=True,
                 pool_size=2, **kwargs):
        super().__init__()
        if use_norm or not bias or refine or not pool or pool_size != 2:
            raise NotImplementedError("DownBlock: only the configuration of trainer/reg.py:43-45 is implemented")
        self.skip = skip
        self._setup(in_channels, out_channels, kernel_size, stride, padding, activation, init_func,
                    1 if use_resnet else 0, "conv_0.resnet_block.model", "conv_0.conv2d", True)

    def forward(self, x):
        pooled, skip = self._call(x)
        pooled, skip = pooled.to(x.dtype), skip.to(x.dtype)
        return (pooled, skip) if self.skip else pooled


class ResnetTransformer(_BlockNet):
    """reference: trainer/layers.py:216-240.  n reflect-padded res-blocks; keys `model.{i}.conv_block.{1,5}.*`."""

    def __init__(self, dim, n_blocks, init_func="kaiming"):
        super().__init__()
        if init_func != "kaiming":
            raise NotImplementedError("init_func=%r" % (init_func,))
        self._setup(0, dim, 3, 1, 1, None, init_func, n_blocks, "model", None, False)

    def forward(self, x):
        return self._call(x)[0].to(x.dtype)
