"""GPU, slow: the BENCHMARKED steps pinned numerically at their full size against the CPU oracle.

BASELINE.json configs[2] (`Hd_Trainer_x2.train_step`, B=16, 512x512; reference trainer/HdTrainer.py:705-751) and configs[3]
(`Cyc_Trainer.train_step`, B=8, 512x512; trainer/CycTrainer.py:138-197): the oracle (`oracle.ref_steps`, stock fp32 torch ops on the
host cores) takes the same optimiser step on the same slices and the same synthetic weights ONCE per module, the product trainers
then take it in every compute mode.

Stated tolerances (the same as at 256^2, tests/test_step_parity_gpu.py):
  fp32 and bf16x3:  every loss term <= 2e-3 relative, the first generator output <= 1e-3 rel-L2 (north_star)
  bf16:             every loss term <= 1e-2 relative (observed <= 1.5e-3), the first generator output <= 1.5e-1 rel-L2 -- bf16 STORAGE
                    is 2.3e-2 from fp32 on the bench's noise images and 8.8e-2 on these smooth ones (small outputs, same absolute
                    error); it is the throughput mode, not the parity mode, and its output after the step is not asserted (one
                    sign-like Adam step on bf16 gradients moves it by 0.3)

The CPU autograd tape of the whole batch is ~45 GB at B=16 (2.6 GB per slice, measured): where the host has less than 70 GB
available the oracle evaluates the SAME step over chunks of 4 slices with gradient accumulation (`micro_batch`; every loss term is a
batch mean and every normalisation per sample, so the step is the chunk-weighted mean -- oracle/ref_steps.py), otherwise the whole
batch at once exactly as the reference does.  CTG_FULLSIZE_MICRO=<n> forces a chunk size, CTG_SKIP_FULLSIZE=1 skips the module.
"""
import os
import random
import time

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

HD_KEYS = ("SM", "SR", "adv", "SR2", "total", "loss_D")
CYC_KEYS = ("GAN_A2B", "GAN_B2A", "cyc_ABA", "cyc_BAB", "total", "loss_D_A", "loss_D_B")
HD_CFG = dict(input_nc=1, output_nc=1, size=512, batchSize=16, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
              Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1)
CYC_CFG = dict(input_nc=1, output_nc=1, size=512, batchSize=8, lr=1e-4, Adv_lamda=1, Cyc_lamda=10, epoch=0, n_epochs=1,
               decay_epoch=1)
MODES = {"fp32": torch.float32, "bf16x3": "bf16x3", "bf16x3f": "bf16x3f", "bf16": torch.bfloat16}
LOSS_TOL = {"fp32": 2e-3, "bf16x3": 2e-3, "bf16x3f": 2e-3, "bf16": 1e-2}
FAKE_TOL = {"fp32": 1e-3, "bf16x3": 1e-3, "bf16x3f": 1e-3, "bf16": 1.5e-1}


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if os.environ.get("CTG_SKIP_FULLSIZE"):
        pytest.skip("CTG_SKIP_FULLSIZE")
    from cta_gan_amd import _lib, nets
    _lib.load()
    nets.set_default_compute_dtype(torch.float32)
    yield
    nets.set_default_compute_dtype(torch.float32)
    torch.cuda.empty_cache()


def _mem_available_gb():
    """What the host can still give this process: MemAvailable, capped by the cgroup limit when there is one."""
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) / 1e6
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                left = (int(v) - int(open(cur).read().strip())) / 1e9
                avail = left if avail is None else min(avail, left)
        except (OSError, ValueError):
            pass
    return avail if avail is not None else 0.0


def _oracle_threads():
    """The oracle's host threads: one GPU's share of the cores (bench.cpu_share) -- oneDNN on all 256 cores of a shared host crawls."""
    import bench
    n, _ = bench.cpu_share()
    torch.set_num_threads(max(1, min(n, 64)))
    print("CPU oracle starting on %d threads ..." % torch.get_num_threads(), flush=True)


def _micro(need_gb):
    forced = os.environ.get("CTG_FULLSIZE_MICRO")
    if forced:
        return int(forced)
    return None if _mem_available_gb() >= need_gb else 4


def _rel_l2(got, want):
    got, want = got.double(), want.double()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


def _close(got, want, tol):
    return abs(got - want) <= tol * max(abs(want), 1e-6) + 1e-6


def _hd_cpu_batch():
    from cta_gan_amd import synth
    return {k: synth.synth_smooth_images("full_hd_" + k, 16, 512) for k in ("A2", "B1", "B2")}


@pytest.fixture(scope="module")
def hd_oracle():
    """`oracle.ref_steps.hd_step` (stage 2) on the 16 slices, once."""
    from cta_gan_amd import synth
    from oracle import golden_cases, ref_steps
    from oracle.golden_cases import REG_GAINS
    ons = golden_cases.oracle_namespace()
    G = synth.fill_module(ons.Generator(1, 1), seed=0)
    D = synth.fill_module(ons.Discriminator_m(1), seed=1)
    R = synth.fill_module(ons.Reg(512, 512, 1, 1), seed=4, gains=REG_GAINS)
    nets_ = dict(G=G, D=D, R=R, T=ons.Transformer_2D())
    opts = dict(G=ref_steps.make_adam(G.parameters()), D=ref_steps.make_adam(D.parameters()), R=ref_steps.make_adam(R.parameters()))
    mb = _micro(70.0)
    _oracle_threads()
    t0 = time.time()
    want = ref_steps.hd_step(nets_, opts, _hd_cpu_batch(), stage=2, smooth_fn=ons.smooothing_loss, gan_loss=ons.GANLoss(),
                             micro_batch=mb)
    print("CPU oracle Hd step B=16 @ 512^2: %.0f s on %d threads (%s)" % (
        time.time() - t0, torch.get_num_threads(), "whole batch" if mb is None else "chunks of %d, accumulated" % mb))
    return {k: (v if isinstance(v, float) else v.clone()) for k, v in want.items()}


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16x3f", "bf16"])
def test_hd_step_b16_512_vs_the_cpu_oracle(mode, hd_oracle):
    """BASELINE.json configs[2] at full size: the six loss terms of `Hd_Trainer_x2.train_step` and the generator output of its G
    step against the oracle's step on the same 16 slices."""
    from cta_gan_amd import nets, ops, synth
    from cta_gan_amd.trainer import Hd_Trainer_x2
    from oracle.golden_cases import REG_GAINS
    nets.set_default_compute_dtype(MODES[mode])
    try:
        tr = Hd_Trainer_x2(dict(HD_CFG))
        synth.fill_module(tr.netG_A2B, seed=0)
        synth.fill_module(tr.netD_B, seed=1)
        synth.fill_module(tr.R_A, seed=4, gains=REG_GAINS)
        batch = {k: v.cuda() for k, v in _hd_cpu_batch().items()}
        with torch.no_grad():
            first = tr.netG_A2B(batch["A2"]).float().cpu()       # the G step's forward (same weights: nothing has stepped yet)
        losses = tr.train_step(batch, sync_losses=True)
        e_first = _rel_l2(first, hd_oracle["fake_B_first"])
        e_after = _rel_l2(tr.last["fake_B"].detach().float().cpu(), hd_oracle["fake_B"])
        print(mode, {k: (round(losses[k], 6), round(hd_oracle[k], 6)) for k in HD_KEYS}, "fake_B first %.2e after the step %.2e" % (
            e_first, e_after))
        for k in HD_KEYS:
            assert _close(losses[k], hd_oracle[k], LOSS_TOL[mode]), (mode, k, losses[k], hd_oracle[k])
        assert e_first <= FAKE_TOL[mode], (mode, e_first)
        # after one sign-like Adam step on every weight (2e-2 at 256^2 in the fp32 mode; 4e-2 split pair)
        if mode != "bf16":
            assert e_after <= {"fp32": 2e-2, "bf16x3": 4e-2, "bf16x3f": 4e-2}[mode], (mode, e_after)
        assert ops.nie_failures() == 0
        del tr
    finally:
        nets.set_default_compute_dtype(torch.float32)
        torch.cuda.empty_cache()


def _cyc_cpu_batch():
    from cta_gan_amd import synth
    return {k: synth.synth_smooth_images("full_cyc_" + k, 8, 512) for k in ("A", "B")}


@pytest.fixture(scope="module")
def cyc_oracle():
    """`oracle.ref_steps.cyc_step` on the 8 slices, once."""
    import itertools
    from cta_gan_amd import synth
    from oracle import golden_cases, ref_steps
    ons = golden_cases.oracle_namespace()
    nets_ = dict(G_A2B=synth.fill_module(ons.Generator(1, 1), seed=0), G_B2A=synth.fill_module(ons.Generator(1, 1), seed=5),
                 D_A=synth.fill_module(ons.Discriminator(1), seed=6), D_B=synth.fill_module(ons.Discriminator(1), seed=1))
    opts = dict(G=ref_steps.make_adam(itertools.chain(nets_["G_A2B"].parameters(), nets_["G_B2A"].parameters())),
                D_A=ref_steps.make_adam(nets_["D_A"].parameters()), D_B=ref_steps.make_adam(nets_["D_B"].parameters()))
    bufs = dict(A=ref_steps.ReplayBuffer(), B=ref_steps.ReplayBuffer())
    mb = _micro(80.0)
    _oracle_threads()
    random.seed(42)
    t0 = time.time()
    want = ref_steps.cyc_step(nets_, opts, bufs, _cyc_cpu_batch(), micro_batch=mb)
    print("CPU oracle CycleGan step B=8 @ 512^2: %.0f s on %d threads (%s)" % (
        time.time() - t0, torch.get_num_threads(), "whole batch" if mb is None else "chunks of %d, accumulated" % mb))
    return {k: (v if isinstance(v, float) else v.clone()) for k, v in want.items()}


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16x3f", "bf16"])
def test_cyc_step_b8_512_vs_the_cpu_oracle(mode, cyc_oracle):
    """BASELINE.json configs[3] at full size: the seven loss terms of `Cyc_Trainer.train_step` and both generators' outputs."""
    from cta_gan_amd import nets, ops, synth
    from cta_gan_amd.trainer import Cyc_Trainer
    nets.set_default_compute_dtype(MODES[mode])
    try:
        random.seed(42)
        tr = Cyc_Trainer(dict(CYC_CFG))
        synth.fill_module(tr.netG_A2B, seed=0)
        synth.fill_module(tr.netG_B2A, seed=5)
        synth.fill_module(tr.netD_A, seed=6)
        synth.fill_module(tr.netD_B, seed=1)
        batch = {k: v.cuda() for k, v in _cyc_cpu_batch().items()}
        losses = tr.train_step(batch, sync_losses=True)
        e_b = _rel_l2(tr.last["fake_B"].detach().float().cpu(), cyc_oracle["fake_B"])
        e_a = _rel_l2(tr.last["fake_A"].detach().float().cpu(), cyc_oracle["fake_A"])
        print(mode, {k: (round(losses[k], 6), round(cyc_oracle[k], 6)) for k in CYC_KEYS}, "fake_B %.2e fake_A %.2e" % (e_b, e_a))
        for k in CYC_KEYS:
            assert _close(losses[k], cyc_oracle[k], LOSS_TOL[mode]), (mode, k, losses[k], cyc_oracle[k])
        assert e_b <= FAKE_TOL[mode] and e_a <= FAKE_TOL[mode], (mode, e_b, e_a)
        assert ops.nie_failures() == 0
        del tr
    finally:
        nets.set_default_compute_dtype(torch.float32)
        torch.cuda.empty_cache()
