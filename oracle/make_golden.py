"""ORACLE tooling: generate `tests/golden/*.npz` from the IMPORTED REFERENCE.

Runs only in the build container (where `/root/reference` is mounted); the
reference never travels -- only the small result arrays written here do.

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden reg_256    # some cases

How the reference is imported: `Model.CycleGan` imports as is;
`Model.HdGan` needs an in-memory stub of the single torchvision symbol it uses
(`transforms.functional.center_crop`); `trainer.{layers,reg}` import through a
synthetic `trainer` package object that skips the package `__init__` (which
pulls visdom/pydicom/...).  `trainer.transformer.Transformer_2D` has a hard
`.cuda()` (transformer.py:21) and `trainer.utils` (home of `smooothing_loss`)
imports visdom: for the lifetime of THIS process only, `torch.Tensor.cuda` is
the identity and an empty in-memory `visdom` module stands in, so both run as
written on the CPU.  The trainer step bodies (unconditional Visdom connection,
DICOM list files) cannot be constructed here; for those the generator uses the
`oracle.ref_steps` restatement *on top of the imported reference networks,
Transformer_2D and smooothing_loss*.
"""
from __future__ import annotations

import os
import sys
import time
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = os.environ.get("CTAGAN_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    def center_crop(img, output_size):
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        h, w = img.shape[-2:]
        ch, cw = output_size
        top = int(round((h - ch) / 2.0))
        left = int(round((w - cw) / 2.0))
        return img[..., top:top + ch, left:left + cw]

    tvf.center_crop = center_crop
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    sys.path.insert(0, REF)
    import Model.HdGan as hd  # noqa: E402
    import Model.CycleGan as cyc  # noqa: E402
    pkg = types.ModuleType("trainer")
    pkg.__path__ = [os.path.join(REF, "trainer")]
    sys.modules["trainer"] = pkg
    import trainer.reg as reg  # noqa: E402
    vis = types.ModuleType("visdom")
    vis.Visdom = type("Visdom", (), {})          # never instantiated: only the import at trainer/utils.py:8 needs it
    sys.modules["visdom"] = vis
    torch.Tensor.cuda = lambda self, *a, **k: self   # trainer/transformer.py:21 on a CPU-only box (this process only)
    import trainer.transformer as transformer  # noqa: E402
    import trainer.utils as utils  # noqa: E402
    sys.path.remove(REF)
    return hd, cyc, reg, transformer, utils


def reference_namespace():
    hd, cyc, reg, transformer, utils = import_reference()
    assert all(m.__file__.startswith(REF) for m in (hd, cyc, reg, transformer, utils))
    return SimpleNamespace(Generator=hd.Generator, ResidualBlock=hd.ResidualBlock, Discriminator=hd.Discriminator,
                           NLayerDiscriminator=hd.NLayerDiscriminator, Discriminator_m=hd.Discriminator_m, GANLoss=hd.GANLoss, Reg=reg.Reg,
                           Transformer_2D=transformer.Transformer_2D, smooothing_loss=utils.smooothing_loss,
                           ReplayBuffer=utils.ReplayBuffer, device="cpu", cyc=cyc)


def main(argv):
    from oracle.golden_cases import CASES
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    ns = reference_namespace()
    # the two model files define the same G/D: check once that CycleGan's match HdGan's on the same weights
    from cta_gan_amd import synth
    x = synth.synth_images("samecheck", 1, 32)
    a = synth.fill_module(ns.Generator(1, 1), seed=9)(x)
    b = synth.fill_module(ns.cyc.Generator(1, 1), seed=9)(x)
    assert torch.equal(a, b), "Model/CycleGan.Generator != Model/HdGan.Generator"
    os.makedirs(OUT, exist_ok=True)
    names = argv or list(CASES)
    for name in names:
        t0 = time.time()
        res = CASES[name](ns)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **res)
        print("%-24s %6.1fs  %7.1f KB" % (name, time.time() - t0, os.path.getsize(path) / 1024), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
