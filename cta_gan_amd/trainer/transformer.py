"""MI355X-native drop-in for the reference's `trainer/transformer.py:7-31`."""
from __future__ import annotations

import torch.nn as nn

from ..nets import warp


class Transformer_2D(nn.Module):
    """forward(src (B,1,H,W), flow (B,2,H,W)) -> src sampled at (row + flow[:,0], col + flow[:,1]) with
    bilinear interpolation and border clamping (grid_sample align_corners=True, padding_mode='border')."""

    def forward(self, src, flow):
        return warp(src, flow)
