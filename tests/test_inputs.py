"""Input-pipeline arithmetic: read_ori_w after the DICOM read, Resize (nearest).

CPU: oracle/ref_inputs.py against fixtures made by the reference's own functions (oracle/make_golden_inputs.py).
GPU: csrc/metrics.hip through the C ABI against the same fixtures -- bit-exact (the float64 arithmetic of the reference is
reproduced in double on the device, then cast to float32 as the reference's transform does)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import ref_inputs

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "inputs_*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_reference_fixtures(path):
    z = np.load(path)
    assert len(GOLD) >= 2
    i1, i2 = ref_inputs.read_ori_w_arith(z["hu"].copy())
    assert np.array_equal(i1.astype(np.float32), z["image1"]) and np.array_equal(i2.astype(np.float32), z["image2"])
    rz = ref_inputs.resize_nearest(torch.from_numpy(z["image1"])[None], tuple(z["size"]))
    assert np.array_equal(rz.numpy(), z["resized"])


@pytest.mark.gpu
def test_hip_input_pipeline_matches_reference_fixtures():
    from cta_gan_amd.trainer.datasets import read_ori_w
    from cta_gan_amd.trainer.utils import Resize, ToTensor
    for path in GOLD:
        z = np.load(path)
        i1, i2 = read_ori_w(torch.from_numpy(z["hu"]).cuda())
        assert np.array_equal(i1.cpu().numpy(), z["image1"]), path
        assert np.array_equal(i2.cpu().numpy(), z["image2"]), path
        rz = Resize(size_tuple=tuple(int(v) for v in z["size"]))(ToTensor()(i1))
        assert np.array_equal(rz.cpu().numpy(), z["resized"]), path


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", [((3, 37, 53), (512, 512)), ((1, 512, 512), (256, 256)), ((2, 100, 100), (77, 131))])
def test_hip_resize_nearest_vs_torch(shape, size):
    import torch.nn.functional as F
    from cta_gan_amd import ops
    x = torch.randn(*shape)
    want = F.interpolate(x.unsqueeze(0), size=list(size)).squeeze(0)
    got = ops.resize_nearest(x.cuda(), size).cpu()
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        ops.resize_nearest(x, size)
