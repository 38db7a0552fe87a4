"""Deterministic synthetic weights and inputs.

There is no network for checkpoints or DICOM data, so every parity test, the
smoke run and the benchmark use tensors made here.  The filler is keyed by the
state_dict *name* of each tensor (CRC32 of the name, mixed with a seed), so the
same weights come out on every box and for every implementation that shares
the reference's state_dict keys -- it does not depend on
torch's RNG stream, on construction order or on the device.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng_for(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode("utf-8")), seed & 0xFFFFFFFF])


def fill_tensor(name: str, shape, seed: int = 0, gain: float = 1.0) -> torch.Tensor:
    """Value of the parameter `name` with the given shape.

    Weights (ndim > 1) ~ N(0, gain^2 * 2 / fan_in) (He-style, so that
    activations keep O(1) scale through the 24-conv generator); biases
    (ndim == 1) ~ U(-0.1, 0.1).
    """
    rng = _rng_for(name, seed)
    shape = tuple(int(s) for s in shape)
    if len(shape) > 1:
        fan_in = int(np.prod(shape[1:]))
        std = gain * np.sqrt(2.0 / max(fan_in, 1))
        arr = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
    else:
        arr = rng.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))


@torch.no_grad()
def fill_module(module: torch.nn.Module, seed: int = 0, gains: dict | None = None) -> torch.nn.Module:
    """Overwrite every parameter of `module` in place with `fill_tensor(key, ...)`.

    `gains` maps a substring of the key to a gain override (first match wins);
    used e.g. to give Reg's flow head a visible, non-zero output.
    """
    sd = module.state_dict()
    for key, val in sd.items():
        if not torch.is_floating_point(val):
            continue
        gain = 1.0
        if gains:
            for sub, g in gains.items():
                if sub in key:
                    gain = g
                    break
        val.copy_(fill_tensor(key, val.shape, seed=seed, gain=gain).to(val.device, val.dtype))
    return module


def synth_images(name: str, batch: int, size: int, seed: int = 1234, channels: int = 1) -> torch.Tensor:
    """`U(-1, 1)` fp32 images `(batch, channels, size, size)`."""
    rng = _rng_for(name, seed)
    arr = rng.uniform(-1.0, 1.0, size=(batch, channels, size, size)).astype(np.float32)
    return torch.from_numpy(arr)


def synth_smooth_images(name: str, batch: int, size: int, seed: int = 1234) -> torch.Tensor:
    """Band-limited images in [-1, 1]: a few random low-frequency cosines.

    Uniform noise makes the registration flow gradient pure noise; these give
    the STN / smoothness terms something structured to chew on in parity tests.
    """
    rng = _rng_for(name, seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, size, dtype=np.float32),
                         np.linspace(0, 1, size, dtype=np.float32), indexing="ij")
    out = np.zeros((batch, 1, size, size), dtype=np.float32)
    for b in range(batch):
        acc = np.zeros((size, size), dtype=np.float32)
        for _ in range(6):
            fy, fx = rng.uniform(0.5, 6.0, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            acc += rng.uniform(0.2, 1.0) * np.cos(2 * np.pi * (fy * yy + fx * xx) + ph).astype(np.float32)
        acc /= max(np.abs(acc).max(), 1e-6)
        out[b, 0] = acc
    return torch.from_numpy(out)
