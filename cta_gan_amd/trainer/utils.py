"""Hot-path pieces of the reference's `trainer/utils.py`: `smooothing_loss` (:165-173), `ReplayBuffer`
(:120-140) and a PyYAML>=6-safe `get_config` (:161-163).  Logger/Visdom, Resize, ToTensor are out of scope."""
from __future__ import annotations

import random

import torch
import yaml

from ..nets import smoothing_loss as _smooth


def smooothing_loss(y_pred, weight=1.0):
    """mean(dx^2) + mean(dy^2) over forward differences of the (B, 2, H, W) flow.  `weight` (an extension of the reference's
    signature; default 1): the loss weight folded into the reduction instead of a scalar multiply behind it."""
    return _smooth(y_pred, weight)


class ReplayBuffer:
    """Host-side 50-image history pool driven by Python's global `random`, as in the reference."""

    def __init__(self, max_size=50):
        assert max_size > 0, "Empty buffer or trying to create a black hole. Be careful."
        self.max_size = max_size
        self.data = []

    def push_and_pop(self, data):
        to_return = []
        for element in data.data:
            element = torch.unsqueeze(element, 0)
            if len(self.data) < self.max_size:
                self.data.append(element)
                to_return.append(element)
            elif random.uniform(0, 1) > 0.5:
                i = random.randint(0, self.max_size - 1)
                to_return.append(self.data[i].clone())
                self.data[i] = element
            else:
                to_return.append(element)
        return torch.cat(to_return)


def get_config(config):
    with open(config, "r") as stream:
        return yaml.safe_load(stream)


class Resize:
    """trainer/utils.py:13-32 on device tensors: nearest-neighbour resize of a (C, H, W) tensor to `size_tuple`."""

    def __init__(self, size_tuple, use_cv=True):
        self.size_tuple = size_tuple
        self.use_cv = use_cv

    def __call__(self, tensor):
        from .. import ops
        return ops.resize_nearest(tensor.unsqueeze(0), self.size_tuple).squeeze(0)


class ToTensor:
    """trainer/utils.py:33-36: (H, W) array / tensor -> (1, H, W) tensor (kept on its device)."""

    def __call__(self, tensor):
        import numpy as np
        import torch
        if isinstance(tensor, np.ndarray):
            tensor = torch.from_numpy(tensor)
        return tensor.unsqueeze(0)
