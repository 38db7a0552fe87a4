"""Host-side logic of the HIP path that needs no GPU: weight layouts made with torch ops, launch policies."""
import numpy as np
import torch


def test_kx_window_weight_layout():
    """ops.kxw_pack: w[n][8 ky + kx] (zero for kx >= k, ky >= k and padded rows) -- the operand of the kx-window first-layer conv
    (csrc/conv_small.hip KXW): column j of the packed row multiplies pixel (y + j // 8, x + j % 8) of the patch."""
    from cta_gan_amd import ops
    rng = np.random.default_rng(0)
    for n, k, npad in ((64, 7, 64), (40, 5, 64), (64, 8, 64)):
        w = torch.from_numpy(rng.standard_normal((n, k * k)).astype(np.float32))
        p = ops.kxw_pack(w, k, npad, torch.float32)
        assert p.shape == (1, npad, 64) and p.dtype == torch.float32
        ref = torch.zeros(npad, 64)
        for ky in range(k):
            for kx in range(k):
                ref[:n, 8 * ky + kx] = w[:, ky * k + kx]
        assert torch.equal(p[0], ref)
        pb = ops.kxw_pack(w, k, npad, torch.bfloat16)
        assert pb.dtype == torch.bfloat16 and torch.equal(pb[0].float(), ref.bfloat16().float())


def test_kx_window_policy():
    """Only the one-plane, stride-1, 5..8 wide windows with 33..64 output channels in a bf16 mode take the kx-window layout
    (the 7x7 head of the generator and the input gradient of its 7x7 tail: Model/HdGan.py:70,110)."""
    from cta_gan_amd import ops
    assert ops.kxw_ok(1, 64, 7, 1, torch.bfloat16)
    assert not ops.kxw_ok(2, 64, 7, 1, torch.bfloat16)       # two planes
    assert not ops.kxw_ok(1, 64, 4, 2, torch.bfloat16)       # the discriminator's stride-2 first layer
    assert not ops.kxw_ok(1, 32, 7, 1, torch.bfloat16)       # 32-channel tile
    assert not ops.kxw_ok(1, 64, 3, 1, torch.bfloat16)       # K fits 32: nothing to window
    assert not ops.kxw_ok(1, 64, 7, 1, torch.float32)
