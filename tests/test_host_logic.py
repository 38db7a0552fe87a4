"""Host-side logic of the HIP path that needs no GPU: weight layouts made with torch ops, launch policies."""
import pytest
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


def test_plain_backward_scope_is_thread_local():
    """"bf16x3f": `ops.plain_backward()` (entered by nets._NetFn.backward on autograd's thread) makes ops.PAIR read False on THAT
    thread only -- a forward running on another host thread at the same time keeps the split-pair mode (the neighbour stress test runs
    exactly that); outside the mode the scope is a no-op; the process-wide mode is restored by set_default_compute_dtype."""
    import threading
    import torch
    from cta_gan_amd import nets, ops
    try:
        nets.set_default_compute_dtype("bf16x3f")
        assert nets.compute_mode() == "bf16x3f" and ops.PAIR and ops.PAIR_BWD_PLAIN and not ops.PAIR_BWD_ACTIVE
        inside, go, seen = threading.Event(), threading.Event(), {}

        def backward_thread():
            with ops.plain_backward():
                seen["bwd"] = (ops.PAIR, ops.PAIR_BWD_ACTIVE, ops.DT_MIX)
                inside.set()
                go.wait(10)
            seen["bwd_after"] = (ops.PAIR, ops.PAIR_BWD_ACTIVE)

        th = threading.Thread(target=backward_thread)
        th.start()
        assert inside.wait(10)
        seen["fwd"] = (ops.PAIR, ops.PAIR_BWD_ACTIVE)        # this thread, while the other one is inside its backward
        go.set()
        th.join(10)
        assert seen == {"bwd": (False, True, 3), "bwd_after": (True, False), "fwd": (True, False)}, seen
        nets.set_default_compute_dtype("bf16x3")
        with ops.plain_backward():
            assert ops.PAIR and not ops.PAIR_BWD_ACTIVE       # bf16x3 proper: the backward stays split-pair
        assert nets.compute_mode() == "bf16x3"
        nets.set_default_compute_dtype(torch.bfloat16)
        with ops.plain_backward():
            assert not ops.PAIR and not ops.PAIR_BWD_ACTIVE and nets.compute_mode() == "bf16"
        with pytest.raises(ValueError):
            nets.set_default_compute_dtype("bf16x2")
    finally:
        nets.set_default_compute_dtype(torch.float32)
    assert nets.compute_mode() == "fp32" and not ops.PAIR
