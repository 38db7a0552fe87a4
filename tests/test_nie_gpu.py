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


def test_a_wait_that_runs_out_is_an_error_never_a_silent_nan(dev):
    """The bounded wait forced to run out (`ops.NIE_BUDGET = 1`: a workgroup that does not find its group complete at its first
    look gives up -- every workgroup of a group but the last one to arrive): the launch leaves NaN tiles and the sticky flag, the
    trainers' check (`ops.nie_check`, made wherever they synchronise: `train_step(sync_losses=True)`, end of epoch, test loop)
    raises RuntimeError, clears the flag and switches fusion off, and the same module then runs the unfused launches and gives the
    recording path's result bit for bit.  At the trainer: the step that hit the timeout raises, the next one trains unfused."""
    from cta_gan_amd import nets, ops, synth
    from cta_gan_amd.Model.HdGan import ResidualBlock
    from cta_gan_amd.trainer import Hd_Trainer_x2
    nets.set_default_compute_dtype(torch.bfloat16)
    saved = ops.NIE_BUDGET, ops._NO_NIE
    try:
        assert ops.nie_failures() == 0
        blk = synth.fill_module(ResidualBlock(256), seed=5).to(dev)
        x = torch.from_numpy(np.random.default_rng(4).standard_normal((3, 256, 64, 64)).astype(np.float32)).to(dev)
        ops.NIE_BUDGET = 1
        log, ops.OP_LOG = ops.OP_LOG, []
        with torch.no_grad():
            y = blk(x)
        labels = [r[0] for r in ops.OP_LOG]
        ops.OP_LOG = log
        assert sum(l.startswith("conv+IN") for l in labels) == 2, labels
        assert ops.nie_failures() > 0 and bool(torch.isnan(y).any())
        with pytest.raises(RuntimeError, match="gave up waiting"):
            ops.nie_check("forced")
        assert ops._NO_NIE and ops.nie_failures() == 0
        ops.nie_check("clean again")                       # flags were cleared: no second error
        ops.NIE_BUDGET = 0
        log, ops.OP_LOG = ops.OP_LOG, []
        with torch.no_grad():
            y0 = blk(x)
        labels = [r[0] for r in ops.OP_LOG]
        ops.OP_LOG = log
        assert not any(l.startswith("conv+IN") for l in labels), labels
        assert torch.equal(y0, blk(x.clone().requires_grad_(True)).detach()) and bool(torch.isfinite(y0).all())
        # ---- the trainer: the D step's no-grad generator forward (trainer/HdTrainer.py:742-743) is where the fused launches run
        ops._NO_NIE = False
        ops.NIE_BUDGET = 1
        cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=2, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
                   Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1)
        tr = Hd_Trainer_x2(cfg)
        batch = tr.synthetic_batch()
        tr._graph = ("a captured step graph stands here",)      # (its fused launches are baked in -- the check must drop it)
        with pytest.raises(RuntimeError, match="gave up waiting"):
            tr.train_step(batch, sync_losses=True)
        assert ops._NO_NIE and tr._graph is None
        # the epoch loop reads the flag every `nie_check_every` steps, not only at the end of the epoch
        from cta_gan_amd.trainer.HdTrainer import run_epoch_steps
        ops._NO_NIE = False
        tr3 = Hd_Trainer_x2(dict(cfg, nie_check_every=2))
        with pytest.raises(RuntimeError, match="step 2 of the epoch"):
            run_epoch_steps(tr3, (tr3.synthetic_batch(i) for i in range(5)))
        assert ops._NO_NIE
        ops.NIE_BUDGET = 0
        tr2 = Hd_Trainer_x2(cfg)                            # (the failed step trained D on NaNs: a fresh trainer, as a user would restart)
        losses = tr2.train_step(batch, sync_losses=True)
        assert all(np.isfinite(v) for v in losses.values()), losses
        assert bool(torch.isfinite(tr2.last["fake_B"]).all())
    finally:
        ops.NIE_BUDGET, ops._NO_NIE = saved
        for b in ops._NIE_SYNC.values():
            b[0].zero_()
        nets.set_default_compute_dtype(torch.float32)


def test_the_library_refuses_counters_it_was_not_given_and_oversized_samples(dev):
    """`ctg_conv_epilogue.nie_groups` is the capacity of the counter buffer: a launch whose B x channel tiles exceeds it is
    CTG_EINVAL (the kernel would index past the buffer).  A sample whose workgroups (spatial tiles x channel tiles) exceed the
    launch's share of the chip's slots (half of them by default: CTG_NIE_SHARE = 2) is answered with 2 = not served: 1024-channel
    blocks at 128^2 and 512-channel ones on the 8-row tiles of batch size 1 fall back to the unfused launches."""
    from cta_gan_amd import nets, ops, synth
    from cta_gan_amd.Model.HdGan import ResidualBlock
    nets.set_default_compute_dtype(torch.bfloat16)
    saved = ops.NIE_GROUPS
    try:
        blk = synth.fill_module(ResidualBlock(256), seed=5).to(dev)
        x = torch.from_numpy(np.random.default_rng(4).standard_normal((3, 256, 64, 64)).astype(np.float32)).to(dev)
        with torch.no_grad():
            blk(x)                       # allocates the counters at the real capacity
        # straight at the C ABI wrapper (ops.conv_in_fusable would not even ask): claim a smaller counter buffer than 3 samples x 2
        # channel tiles need
        from cta_gan_amd.ops import pack_tap
        xa = torch.randn(3, 64, 64, 256, device=dev).to(torch.bfloat16)
        wp = (torch.randn(9, 256, 256, device=dev) * 0.02).to(torch.bfloat16)
        ya = torch.empty_like(xa)
        taps = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
        assert ops.conv_igemm(xa, wp, 256, ya, None, 256, 64, 64, 0, 0, 1, 1, ops.PAD_REFLECT, ops.ACT_NONE, taps, want_stats=True,
                              in_fuse=ops.ACT_RELU) is True
        ops.NIE_GROUPS = 4
        with pytest.raises(RuntimeError, match="CTG_EINVAL"):
            ops.conv_igemm(xa, wp, 256, ya, None, 256, 64, 64, 0, 0, 1, 1, ops.PAD_REFLECT, ops.ACT_NONE, taps, want_stats=True,
                           in_fuse=ops.ACT_RELU)
        ops.NIE_GROUPS = saved
        # a sample too large for the launch's share of the slots: "not served" (None), nothing launched
        xb = torch.randn(1, 128, 128, 512, device=dev).to(torch.bfloat16)
        wb = (torch.randn(9, 512, 512, device=dev) * 0.02).to(torch.bfloat16)
        assert ops.conv_igemm(xb, wb, 512, torch.empty_like(xb), None, 512, 128, 128, 0, 0, 1, 1, ops.PAD_REFLECT, ops.ACT_NONE, taps,
                              want_stats=True, in_fuse=ops.ACT_RELU) is None
        # 1024 channels on a 128 x 128 map: 64 tiles x 8 channel tiles = 512 workgroups per sample > half of the slots (the policy
        # limit of 1024 workgroups per launch is met: the residency test is what refuses)
        blk5 = synth.fill_module(ResidualBlock(1024), seed=6).to(dev)
        x5 = torch.from_numpy(np.random.default_rng(5).standard_normal((2, 1024, 128, 128)).astype(np.float32)).to(dev)
        assert not ops.conv_in_fusable(torch.empty(2, 128, 128, 1024, dtype=torch.bfloat16, device=dev), 1024, 1024, 3, 1, 128, 128)
        log, ops.OP_LOG = ops.OP_LOG, []
        with torch.no_grad():
            y5 = blk5(x5)
        labels = [r[0] for r in ops.OP_LOG]
        ops.OP_LOG = log
        assert not any(l.startswith("conv+IN") for l in labels), labels
        assert torch.equal(y5, blk5(x5.clone().requires_grad_(True)).detach())
        assert ops.nie_failures() == 0
    finally:
        ops.NIE_GROUPS = saved
        nets.set_default_compute_dtype(torch.float32)


_TWO_PROC = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from cta_gan_amd import nets, ops, synth
from cta_gan_amd.Model.HdGan import Generator
nets.set_default_compute_dtype(torch.bfloat16)
g = synth.fill_module(Generator(1, 1), seed=0).cuda()
x = synth.synth_images("two_proc_%%s" %% sys.argv[1], 4, 512).cuda()
ops.OP_LOG = []
with torch.no_grad():
    y0 = g(x)
    fused = sum(r[0].startswith("conv+IN") for r in ops.OP_LOG)
    ops.OP_LOG = None
    print("READY", flush=True)
    sys.stdin.readline()                      # both processes start their loop together
    ok = True
    for _ in range(40):
        ok = ok and torch.equal(g(x), y0)     # bitwise repeatable, whoever shares the card
torch.cuda.synchronize()
print("RESULT fused=%%d failures=%%d finite=%%d same=%%d" %% (fused, ops.nie_failures(), int(torch.isfinite(y0).all()), int(ok)), flush=True)
"""


def test_two_processes_on_one_card_both_issue_fused_launches(dev):
    """Two processes share the card and both run no-grad generator forwards whose residual blocks are fused conv + InstanceNorm
    launches (B=4 at 512^2: 64 tiles x 2 channel tiles per sample, 512 workgroups per launch), 40 forwards each, started
    together: no bounded wait may run out (`nie_failures() == 0`), every forward reproduces the first bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_PROC % root, str(i)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for i in range(2)]
    try:
        for p in procs:
            line = p.stdout.readline()
            assert line.startswith("READY"), (line, p.stderr.read()[-2000:])
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-2000:]
            res = [ln for ln in out.splitlines() if ln.startswith("RESULT")][-1]
            kv = dict(t.split("=") for t in res.split()[1:])
            assert int(kv["fused"]) == 18 and kv["failures"] == "0" and kv["finite"] == "1" and kv["same"] == "1", res
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
