"""tests/golden/inputs_*.npz from the REFERENCE's own read_ori_w (trainer/datasets.py:36-71) and Resize (trainer/utils.py:13-32).

The modules cannot be imported (SimpleITK, pydicom, torchvision, visdom absent), so only those two definitions are compiled
out of the files; `sitk` is an in-memory stand-in whose ReadImage hands back a synthetic int16 HU array.  Only seeds / arrays
of inputs and outputs are written to the repo."""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _extract(path, names, ns):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in body} == set(names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)


def main():
    store = {}
    sitk = types.SimpleNamespace(ReadImage=lambda p: store[p], GetArrayFromImage=lambda im: im)
    ns = {"np": np, "sitk": sitk}
    _extract("/root/reference/trainer/datasets.py", ["read_ori_w"], ns)
    ns2 = {"F": torch.nn.functional, "torch": torch, "np": np}
    _extract("/root/reference/trainer/utils.py", ["Resize"], ns2)
    rng = np.random.RandomState(7)
    gold = os.path.join(ROOT, "tests", "golden")
    for name, shape in (("a", (96, 80)), ("b", (64, 64))):
        # HU-like data: air (-2000 / -1024), soft tissue around 0..100, contrast / bone up to 3071, a few exact edges
        hu = rng.randint(-1100, 1500, size=shape).astype(np.int16)
        hu[:4] = -2000
        hu[4, :8] = [-150, -149, -1024, -1025, 250, 251, 3071, 0]
        store["k"] = hu.copy()[None]          # (1, H, W) as SimpleITK returns a single slice
        i1, i2 = ns["read_ori_w"]("k")
        size = (48, 56) if name == "a" else (128, 96)
        rz = ns2["Resize"](size_tuple=size)(torch.from_numpy(i1.astype(np.float32))[None])
        np.savez_compressed(os.path.join(gold, "inputs_%s.npz" % name), hu=hu, image1=i1.astype(np.float32),
                            image2=i2.astype(np.float32), size=np.array(size), resized=rz.numpy())
        print(name, i1.dtype, float(i1.mean()), float(i2.mean()), rz.shape)


if __name__ == "__main__":
    main()
