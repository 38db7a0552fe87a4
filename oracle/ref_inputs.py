"""ORACLE (test infrastructure only): numpy / stock-torch restatement of the input-pipeline arithmetic.

  read_ori_w_arith   trainer/datasets.py:36-71 after `data1 = np.squeeze(sitk.GetArrayFromImage(dicom))`
  resize_nearest     trainer/utils.py:13-32 (F.interpolate default mode)
Pinned by tests/golden/inputs_*.npz (oracle/make_golden_inputs.py runs the reference's own read_ori_w on synthetic HU
arrays through an in-memory stand-in for the SimpleITK reader)."""
import numpy as np
import torch
import torch.nn.functional as F


def read_ori_w_arith(data1, center=50, width=400):
    data = data1 + 1024
    win_min = (2 * center - width) / 2.0 + 0.5
    win_max = (2 * center + width) / 2.0 + 0.5
    d_factor = 255.0 / (win_max - win_min)
    image = data1 - win_min
    image1 = np.trunc(image * d_factor)
    image1[image1 > 255] = 255
    image1[image1 < 0] = 0
    image1 = image1 / 255
    image1 = (image1 - 0.5) / 0.5
    image2 = data
    image2[image2 < 0] = 0
    image2 = image2 / 4095
    image2 = (image2 - 0.5) / 0.5
    return image1, image2


def resize_nearest(x, size):
    """x: (C, H, W) float32 tensor."""
    return F.interpolate(x.unsqueeze(0), size=[size[0], size[1]]).squeeze(0)
