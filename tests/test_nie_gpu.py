"""Conv + InstanceNorm (+ ReLU / skip) in ONE launch (csrc/conv_halo.h NIE, ctg_conv_epilogue.nie_*): the residual blocks of a
forward that keeps nothing for a backward pass (Model/HdGan.py:49-63 under torch.no_grad(); trainer/HdTrainer.py:742-743).  The
workgroups of a sample exchange tile moments through device memory and normalise their accumulators in registers."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm())


def _ref_block(x, w1, w5):
    h = F.relu(F.instance_norm(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w1)))
    return x + F.instance_norm(F.conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"), w5))


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("shape", [(3, 256, 64, 64), (2, 128, 72, 88), (1, 256, 16, 48), (5, 256, 128, 128), (16, 256, 128, 128)],
                         ids=["256x64", "128x72x88", "256x16x48", "256x128_b5", "256x128_b16_forced"])
def test_no_grad_residual_blocks_match_the_unfused_path_and_torch(shape, mode, dev):
    """Two chained residual blocks: the no-grad forward (one launch per conv + norm) against (a) the same modules with
    autograd recording (conv, finalize and in_apply launches: the path every golden pins) and (b) stock torch in fp32.
    bf16: the fused path normalises the UNROUNDED accumulators, the unfused one the stored bf16 conv result, so the two differ by
    storage rounding (rel-L2 1e-2 against each other, 1.5e-2 against fp32 torch); split pair: 2e-5 / 5e-5.  Ragged maps (72 x 88:
    tiles hanging over the edge), a one-tile-row map, 5 samples x 64 tiles x 2 channel tiles (more workgroups than the chip holds
    at once), and -- with the policy limit lifted -- the bench shape itself: 2048 workgroups in 32 groups, four rounds of the
    chip's workgroup slots, the case the dispatch-order argument of csrc/conv_halo.h is about."""
    from cta_gan_amd import nets, ops, synth
    from cta_gan_amd.Model.HdGan import ResidualBlock
    nets.set_default_compute_dtype(torch.bfloat16 if mode == "bf16" else mode)
    # the tile moments travel between workgroups INSIDE the launch: a read that overtakes its write must not find a plausible
    # value there (the allocator hands the buffer of the previous, identical call back) -- NaNs instead
    ops.NIE_POISON = True
    max_wgs, max_pair = ops.NIE_MAX_WGS, ops.NIE_MAX_WGS_PAIR
    if shape[0] >= 5:      # above the policy limit of one mode or both (ops.conv_in_fusable): lifted, the mechanism is what is tested
        ops.NIE_MAX_WGS = ops.NIE_MAX_WGS_PAIR = 1 << 20
    try:
        c = shape[1]
        blocks = [synth.fill_module(ResidualBlock(c), seed=70 + i).to(dev) for i in range(2)]
        x = torch.from_numpy(np.random.default_rng(9).standard_normal(shape).astype(np.float32)).to(dev)
        log, ops.OP_LOG = ops.OP_LOG, []
        with torch.no_grad():
            y_f = blocks[1](blocks[0](x))
            y_f2 = blocks[1](blocks[0](x))
        labels = [r[0] for r in ops.OP_LOG]
        ops.OP_LOG = log
        assert sum(l.startswith("conv+IN") for l in labels) == 8, labels      # 2 runs x 2 blocks x 2 convs, nothing else
        assert not any(l.startswith("conv ") for l in labels), labels
        assert ops.nie_failures() == 0
        assert torch.equal(y_f, y_f2)                                         # fixed summation order: bitwise repeatable
        xg = x.clone().requires_grad_(True)
        y_u = blocks[1](blocks[0](xg))                                        # recording: unfused
        ws = [[dict(m.named_parameters())[k].detach().float() for k in ("conv_block.1.weight", "conv_block.5.weight")]
              for m in blocks]
        y_r = _ref_block(_ref_block(x, *ws[0]), *ws[1])
        e_u, e_r = _rel(y_f, y_u), _rel(y_f, y_r)
        print(shape, mode, "fused vs unfused %.2e, fused vs torch fp32 %.2e, unfused vs torch %.2e" % (e_u, e_r, _rel(y_u, y_r)))
        assert torch.isfinite(y_f).all()
        if mode == "bf16":
            assert e_u < 1e-2 and e_r < 1.5e-2
        else:
            assert e_u < 2e-5 and e_r < 5e-5
    finally:
        ops.NIE_MAX_WGS, ops.NIE_MAX_WGS_PAIR = max_wgs, max_pair
        ops.NIE_POISON = False
        nets.set_default_compute_dtype(torch.float32)


def test_shapes_outside_the_fused_form_fall_back(dev):
    """64-channel blocks (no 128-channel tile), maps with more than 256 tiles per sample and launches above the policy limit
    (ops.NIE_MAX_WGS workgroups: the waiting costs more than the two saved launches there) take the conv + finalize + in_apply
    launches and give the recording path's result bit for bit."""
    from cta_gan_amd import nets, ops, synth
    from cta_gan_amd.Model.HdGan import ResidualBlock
    nets.set_default_compute_dtype(torch.bfloat16)
    try:
        for shape in [(1, 64, 32, 48), (1, 128, 272, 256), (9, 256, 128, 128)]:
            blk = synth.fill_module(ResidualBlock(shape[1]), seed=3).to(dev)
            x = torch.from_numpy(np.random.default_rng(2).standard_normal(shape).astype(np.float32)).to(dev)
            log, ops.OP_LOG = ops.OP_LOG, []
            with torch.no_grad():
                y0 = blk(x)
            labels = [r[0] for r in ops.OP_LOG]
            ops.OP_LOG = log
            assert not any(l.startswith("conv+IN") for l in labels), labels
            y1 = blk(x.clone().requires_grad_(True))
            assert torch.equal(y0, y1.detach())
    finally:
        nets.set_default_compute_dtype(torch.float32)
