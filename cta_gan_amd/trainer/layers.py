"""HIP-backed counterparts of the reference's `trainer/layers.py` building blocks that exist as
stand-alone modules on the hot path.  Inside `Reg` the whole U-Net runs as one autograd node
(cta_gan_amd/nets.py: RegNet), so only the pieces a user may construct directly are exposed here."""
from __future__ import annotations

import torch.nn as nn

from ..nets import ResidualBlockNet


class ResnetBlock(ResidualBlockNet):
    """reference: trainer/layers.py:243-300 -- reflect-padded conv-IN-ReLU-conv-IN + skip
    (keys conv_block.{1,5}.{weight,bias}).  Only the configuration the reference instantiates
    (padding_type='reflect', affine-free InstanceNorm, no dropout, bias) is implemented."""

    def __init__(self, dim, padding_type="reflect", norm_layer=None, use_dropout=False, use_bias=True):
        if padding_type != "reflect" or use_dropout or not use_bias:
            raise NotImplementedError("ResnetBlock: only reflect / no dropout / bias (layers.py:220-222)")
        super().__init__(dim)
        for p in (self._p("conv_block.1"), self._p("conv_block.5")):
            nn.init.kaiming_normal_(p.weight, a=0.0, nonlinearity="relu", mode="fan_in")
            p.bias.data.zero_()
