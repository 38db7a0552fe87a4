"""Device-side arithmetic of the reference's trainer/datasets.py (hot-path neighbours only).

The DICOM readers themselves (SimpleITK / pydicom) are host I/O outside this build; what follows them is here:
`read_ori_w` turns one raw HU slice into the windowed and the full-range training image."""
from __future__ import annotations

import torch

from .. import ops


def read_ori_w(hu: torch.Tensor, center: float = 50.0, width: float = 400.0):
    """trainer/datasets.py:36-71 from the point where `data1` (raw HU, SimpleITK convention) is in memory:
    returns (image1, image2) = (CT window [center, width] -> [-1, 1], full 12-bit range -> [-1, 1]), fp32, on hu's GPU."""
    return ops.hu_to_inputs(hu, center, width)
