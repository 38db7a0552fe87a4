"""MI355X-native drop-in for the reference's `trainer/reg.py:101-132`."""
from __future__ import annotations

from ..nets import RegNet


class Reg(RegNet):
    """Registration network: forward(img_a, img_b) -> (B, 2, H, W) pixel displacement field
    (channel 0 = rows, channel 1 = columns).  Same ctor as the reference: Reg(height, width,
    in_channels_a, in_channels_b); state_dict keys `offset_map.*` identical."""

    def __init__(self, height, width, in_channels_a, in_channels_b):
        super().__init__(in_channels_a, in_channels_b)
        self.oh, self.ow = height, width
        self.in_channels_a, self.in_channels_b = in_channels_a, in_channels_b
