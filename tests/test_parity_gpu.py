"""GPU parity proper: the HIP path (through the C ABI) against (1) the committed golden vectors that
oracle/make_golden.py produced from the imported reference and (2) the CPU oracle on fresh seeded inputs.

Stated tolerances (fp32 compute mode -- exact-f32 MFMA, fp32 storage):
  * network outputs:           rel-L2 <= 1e-3  (BASELINE.json north_star), observed ~1e-6..1e-5
  * gradients / grad norms:    rel <= 5e-3 (reduction order over up to 2.6e5 pixels differs from oneDNN's)
  * loss scalars:              rel <= 1e-3
  * index/shape ops (crop, feature-map shapes, mask thresholds): exact
bf16 compute mode (bf16 storage + MFMA, fp32 accumulate/statistics): generator output rel-L2 <= 3e-2.
Biases in front of an affine-free InstanceNorm are mathematically dead: the HIP path returns
no gradient for them, the reference returns rounding noise; their grad-norm entries are skipped.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ns():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cta_gan_amd import _lib, nets
    _lib.load()
    nets.set_default_compute_dtype(torch.float32)
    from hip_ns import hip_namespace
    return hip_namespace()


def rel_l2(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.sqrt(((got - want) ** 2).sum()) / max(np.sqrt((want ** 2).sum()), 1e-30))


DEAD_BIAS_HINTS = ("model_head", "model_body", "model_tail.0", "model_tail.3", "model.2.", "model.5.", "model.8.",
                   "_layer1.", "_layer2.", "_layer3.", "conv_block", "model1.", "model2.",
                   "model3.", "layer0.2.", "layer0.5.", "layer0.8.", "layer1.2.", "layer1.5.", "layer1.8.")


def _is_dead_bias(key):
    return key.endswith(".bias") and any(h in key for h in DEAD_BIAS_HINTS)


def _compare(name, got, want, out_tol=1e-3, grad_tol=5e-3):
    report = {}
    for key in want.files:
        w = want[key]
        g = got[key]
        if w.dtype.kind in "US":
            assert list(g) == list(w)
            continue
        if key == "gradnorm_vals":
            keys = list(want["gradnorm_keys"])
            for k, gv, wv in zip(keys, np.asarray(g), w):
                if _is_dead_bias(str(k)):
                    continue
                assert abs(gv - wv) <= grad_tol * abs(wv) + 1e-7, (name, k, gv, wv)
            continue
        if key.startswith("shape_"):
            assert tuple(g) == tuple(w), (name, key)
            continue
        if np.ndim(w) == 0:
            tol = out_tol if key.startswith(("loss", "sum", "abs", "flow", "smooth")) else grad_tol
            assert abs(float(g) - float(w)) <= tol * max(abs(float(w)), 1e-6) + 1e-7, (name, key, float(g), float(w))
            continue
        tol = grad_tol if ("grad" in key or "delta" in key) else out_tol
        if "stats" in key:
            assert np.allclose(g, w, rtol=5e-3, atol=5e-4), (name, key, g, w)
            continue
        err = rel_l2(g, w)
        report[key] = err
        assert err <= tol, "%s[%s]: rel-L2 %.3e > %.1e" % (name, key, err, tol)
    return report


GOLDEN = ["generator_64", "resblock_256x12", "discriminator_64", "discriminator2_64", "discriminator_m1_64", "discriminator_m2_128",
          "nlayer_d_64", "nlayer_d_interm_64", "nlayer_d_sigmoid_64", "nlayer_d_interm_sigmoid_64",
          "nlayer_d_bn_64", "nlayer_d_bn_interm_64",       # NLayerDiscriminator's own default norm: nn.BatchNorm2d (train + eval)
          "discriminator_m_flat_128", "reg_256", "stn_smooth_48"]


@pytest.mark.parametrize("name", GOLDEN)
def test_hip_matches_reference_golden(name, ns, golden_dir):
    from oracle import golden_cases
    want = np.load(os.path.join(golden_dir, name + ".npz"))
    got = golden_cases.CASES[name](ns)
    rep = _compare(name, got, want)
    print(name, {k: "%.2e" % v for k, v in rep.items()})


@pytest.mark.parametrize("name", ["hd_step_stage1_256", "hd_step_stage2_256", "hd_step_stage2_256_b4", "cyc_step_128",
                                  "p2p_step_128", "reg_step_256"])
def test_hip_step_matches_reference_golden(name, ns, golden_dir):
    """One full optimiser step (oracle.ref_steps driving the HIP networks with torch's Adam) vs the reference run.
    After-step quantities pass through Adam's sign-like first step, so they get a looser bound."""
    from oracle import golden_cases
    want = np.load(os.path.join(golden_dir, name + ".npz"))
    got = golden_cases.CASES[name](ns)
    for key in want.files:
        w, g = want[key], got[key]
        if np.ndim(w) == 0:
            assert abs(float(g) - float(w)) <= 2e-3 * max(abs(float(w)), 1e-6) + 1e-6, (name, key, float(g), float(w))
        elif "stats" in key:
            assert np.allclose(g, w, rtol=2e-2, atol=2e-3), (name, key, g, w)
        elif "delta" in key:
            assert np.allclose(g, w, rtol=5e-2, atol=1e-6), (name, key, g, w)
        else:
            tol = 2e-2 if "after" in key else 1e-3
            assert rel_l2(g, w) <= tol, (name, key, rel_l2(g, w))


def _oracle_ns():
    from oracle import golden_cases
    return golden_cases.oracle_namespace()


def test_generator_vs_oracle_256_and_bf16(ns):
    """configs[1]-shaped check at a size the oracle finishes in seconds: rel-L2 of the generator output,
    fp32 mode <= 1e-3 (north_star), bf16 mode <= 3e-2."""
    from cta_gan_amd import nets, synth
    ons = _oracle_ns()
    x = synth.synth_images("gen256", 2, 256)
    ref = synth.fill_module(ons.Generator(1, 1), seed=21)
    with torch.no_grad():
        want = ref(x).numpy()
    hip = synth.fill_module(ns.Generator(1, 1), seed=21).to("cuda")
    with torch.no_grad():
        got32 = hip(x.cuda()).float().cpu().numpy()
        hip.compute_dtype = torch.bfloat16
        got16 = hip(x.cuda()).float().cpu().numpy()
    e32, e16 = rel_l2(got32, want), rel_l2(got16, want)
    print("generator 256^2 rel-L2: fp32 %.3e  bf16 %.3e" % (e32, e16))
    assert e32 <= 1e-3
    assert e16 <= 3e-2


def test_trainer_step_hip_adam_vs_golden(ns, golden_dir):
    """The fully-HIP trainer (HIP Adam, fused masked L1) reproduces the reference's stage-2 step scalars."""
    from cta_gan_amd import synth
    from cta_gan_amd.trainer import Hd_Trainer_x2
    from oracle.golden_cases import REG_GAINS
    want = np.load(os.path.join(golden_dir, "hd_step_stage2_256.npz"))
    cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=2, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
               Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1)
    tr = Hd_Trainer_x2(cfg)
    synth.fill_module(tr.netG_A2B, seed=0)
    synth.fill_module(tr.netD_B, seed=1)
    synth.fill_module(tr.R_A, seed=4, gains=REG_GAINS)
    batch = {k: synth.synth_smooth_images("hd_" + k, 2, 256).cuda() for k in ("A2", "B1", "B2")}
    losses = tr.train_step(batch, sync_losses=True)
    for k in ("SM", "SR", "adv", "SR2", "total", "loss_D"):
        w = float(want["loss_" + k])
        assert abs(losses[k] - w) <= 2e-3 * max(abs(w), 1e-6) + 1e-6, (k, losses[k], w)
    assert rel_l2(tr.last["fake_B"].cpu().numpy()[:, :, ::8, ::8], want["fake_after_sub"]) <= 2e-2


def test_p2p_and_reg_trainers_vs_golden(ns, golden_dir):
    """`P2p_Trainer` / `Reg_Trainer` (HIP Adam, batched D pass) reproduce the step scalars of
    the reference's step bodies (p2pTrainer.py:122-148, RegTrainer.py:170-198) run on the imported reference networks.
    Tolerance 2e-3 relative on every loss term, 2e-2 rel-L2 on the generator output after the step (fp32 mode)."""
    from cta_gan_amd import synth
    from cta_gan_amd.trainer import P2p_Trainer, Reg_Trainer
    from oracle.golden_cases import REG_GAINS
    want = np.load(os.path.join(golden_dir, "p2p_step_128.npz"))
    cfg = dict(input_nc=1, output_nc=1, size=128, batchSize=2, lr=1e-4, Adv_lamda=1, P2P_lamda=100, epoch=0, n_epochs=1,
               decay_epoch=1)
    tr = P2p_Trainer(cfg)
    synth.fill_module(tr.netG_A2B, seed=0)
    synth.fill_module(tr.netD_B, seed=7)
    batch = {k: synth.synth_smooth_images("p2p_" + k, 2, 128).cuda() for k in ("A", "B")}
    losses = tr.train_step(batch, sync_losses=True)
    for k in ("L1", "GAN_A2B", "total", "loss_D"):
        w = float(want["loss_" + k])
        assert abs(losses[k] - w) <= 2e-3 * max(abs(w), 1e-6) + 1e-6, (k, losses[k], w)
    assert rel_l2(tr.last["fake_B"].cpu().numpy()[:, :, ::4, ::4], want["fake_after_sub"]) <= 2e-2
    tr.update_learning_rate()
    assert tr.optimizer_G.param_groups[0]["lr"] == 0.0 and tr.optimizer_D_B.param_groups[0]["lr"] == 0.0

    want = np.load(os.path.join(golden_dir, "reg_step_256.npz"))
    cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=2, lr=1e-4, Adv_lamda=1, Corr_lamda=20, Smooth_lamda=10,
               epoch=0, n_epochs=1, decay_epoch=2)
    tr = Reg_Trainer(cfg)
    synth.fill_module(tr.netG_A2B, seed=0)
    synth.fill_module(tr.netD_B, seed=1)
    synth.fill_module(tr.R_A, seed=4, gains=REG_GAINS)
    batch = {k: synth.synth_smooth_images("reg_" + k, 2, 256).cuda() for k in ("A", "B")}
    losses = tr.train_step(batch, sync_losses=True)
    for k in ("SM", "SR", "adv", "total", "loss_D"):
        w = float(want["loss_" + k])
        assert abs(losses[k] - w) <= 2e-3 * max(abs(w), 1e-6) + 1e-6, (k, losses[k], w)
    assert rel_l2(tr.last["fake_B"].cpu().numpy()[:, :, ::8, ::8], want["fake_after_sub"]) <= 2e-2
    tr.update_learning_rate()
    assert all(abs(o.param_groups[0]["lr"] - 5e-5) < 1e-12 for o in (tr.optimizer_G, tr.optimizer_R_A, tr.optimizer_D_B))
    out = tr.test([dict(batch, WC=40.0, WW=400.0)])
    assert out["num"] == 2 and np.isfinite(out["PSNR"])


def test_hip_graph_step_matches_eager_step(ns):
    """config['hip_graph']: a step replayed from the captured hipGraph (device-side Adam step counter) equals the eager
    step from the same state. The graph trainer takes its 3 eager warm-up steps, is rewound IN PLACE to the initial
    weights / zero Adam moments, then captures and replays; two replayed steps are compared with two eager steps of a
    fresh trainer. Tolerance 1e-4 (first step) / 1e-3 (second step) relative on the losses and 2e-4 rel-L2 on the weights (two Adam steps move them
    by ~2e-3 rel-L2): the warp backward scatters with float atomics, so two eager runs differ at the 1e-4 level too --
    Adam's first steps are sign-like and flip on gradients that are pure rounding noise."""
    from cta_gan_amd import synth
    from cta_gan_amd.trainer import Hd_Trainer_x2

    def make(graph):
        cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=1, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
                   Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=graph)
        tr = Hd_Trainer_x2(cfg)
        synth.fill_module(tr.netG_A2B, seed=0)
        synth.fill_module(tr.netD_B, seed=1)
        synth.fill_module(tr.R_A, seed=4)
        return tr

    def batch(i):
        return {k: synth.synth_smooth_images("gr%d_%s" % (i, k), 1, 256).cuda() for k in ("A2", "B1", "B2")}

    def weights(tr):
        return torch.cat([p.detach().reshape(-1) for m in (tr.netG_A2B, tr.R_A, tr.netD_B)
                          for p in m.parameters()]).cpu().numpy()

    eager = make(False)
    want = [eager.train_step(batch(i), sync_losses=True) for i in (0, 1)]
    w_want = weights(eager)

    tr = make(True)
    for i in (7, 8, 9):
        tr.train_step(batch(i))
    assert tr._graph is None
    fresh = make(False)
    with torch.no_grad():
        for m, f in ((tr.netG_A2B, fresh.netG_A2B), (tr.R_A, fresh.R_A), (tr.netD_B, fresh.netD_B)):
            for p, q in zip(m.parameters(), f.parameters()):
                p.copy_(q)
                p._ctg_version = getattr(p, "_ctg_version", 0) + 1
    for o in (tr.optimizer_G, tr.optimizer_R_A, tr.optimizer_D_B):
        for st in o.state.values():
            st["step"] = 0
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
        o._dev_state.clear()
    got = [tr.train_step(batch(i), sync_losses=True) for i in (0, 1)]
    assert tr._graph is not None
    assert all(st["step"] == 2 for st in tr.optimizer_G.state.values())
    for a, b, tol in zip(got, want, (1e-4, 1e-3)):
        for k in ("SM", "SR", "adv", "SR2", "total", "loss_D"):
            assert abs(a[k] - b[k]) <= tol * max(abs(b[k]), 1e-6) + 1e-6, (k, a[k], b[k])
    assert rel_l2(weights(tr), w_want) <= 2e-4
    # update_learning_rate() invalidates the captured Adam constants: the next step re-captures
    g0 = tr._graph[0]
    tr.update_learning_rate()
    tr.train_step(batch(2))
    assert tr._graph[0] is not g0


def test_hip_graph_replays_stay_sane_at_bench_shape(ns):
    """Regression: at the bench shape (B=16, 512^2, bf16) the zero-fill of the warp backward's scatter target used to be a
    hipMemsetAsync; as a memset node of the captured hipGraph it raced with the scatter kernel on replay and the
    generator's gradients came out as garbage / inf from the third replay on (5 runs of 6).  Seven steps with
    config['hip_graph']: every generator gradient stays finite and the gradient norm of the replayed steps stays within
    3x of the last eager warm-up step's."""
    from cta_gan_amd import nets, synth
    from cta_gan_amd.trainer import Hd_Trainer_x2
    prev = nets.default_compute_dtype() if hasattr(nets, "default_compute_dtype") else None
    nets.set_default_compute_dtype(torch.bfloat16)
    try:
        cfg = dict(input_nc=1, output_nc=1, size=512, batchSize=16, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
                   Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=True)
        tr = Hd_Trainer_x2(cfg)
        synth.fill_module(tr.netG_A2B, seed=0)
        synth.fill_module(tr.netD_B, seed=1)
        synth.fill_module(tr.R_A, seed=4)
        norms = []
        for i in range(7):
            batch = {k: synth.synth_smooth_images("gve%d_%s" % (i % 4, k), 16, 512).cuda() for k in ("A2", "B1", "B2")}
            losses = tr.train_step(batch, sync_losses=True)
            assert all(v == v and abs(v) < 1e6 for v in losses.values()), (i, losses)
            g = torch.cat([p.grad.reshape(-1).float() for p in tr.netG_A2B.parameters() if p.grad is not None])
            assert bool(torch.isfinite(g).all()), "non-finite generator gradient at step %d" % i
            norms.append(float(g.norm()))
        assert max(norms[3:]) <= 3.0 * norms[2], norms      # steps 3.. are replays, steps 0-2 the eager warm-up
        del tr
    finally:
        nets.set_default_compute_dtype(prev or torch.float32)
        torch.cuda.empty_cache()


def test_full_size_properties_512(ns):
    """BASELINE full size (512^2): size-independent properties instead of an oracle run.
    (a) per-sample independence (InstanceNorm has no cross-sample state): G(x)[i] == G(x[i:i+1]) exactly-ish;
    (b) tanh range; (c) feature-map shapes of the multi-scale discriminator (index/shape ops: exact)."""
    from cta_gan_amd import synth
    G = synth.fill_module(ns.Generator(1, 1), seed=2).cuda()
    x = synth.synth_images("p512", 2, 512).cuda()
    with torch.no_grad():
        y = G(x)
        y0 = G(x[:1])
    assert y.shape == (2, 1, 512, 512) and float(y.abs().max()) <= 1.0
    assert rel_l2(y[:1].cpu().numpy(), y0.cpu().numpy()) < 1e-6
    D = synth.fill_module(ns.Discriminator_m(1), seed=3).cuda()
    with torch.no_grad():
        feats = D(y)
    assert [tuple(f.shape[1:]) for f in feats[0]] == [(64, 256, 256), (128, 128, 128), (256, 64, 64), (512, 63, 63),
                                                      (1, 62, 62)]


_BENCH_REF = {}


def _bench_shape_reference():
    """The CPU oracle's generator (seed-0 synthetic weights) on slices 0 and 15 of the 16-slice 512^2 batch `bshape`, run at
    B=1 each (per-sample InstanceNorm: a sample's result does not depend on its batch); ~3 s of CPU, computed once."""
    if not _BENCH_REF:
        from cta_gan_amd import synth
        ons = _oracle_ns()
        ref = synth.fill_module(ons.Generator(1, 1), seed=0)
        x = synth.synth_images("bshape", 16, 512)
        with torch.no_grad():
            _BENCH_REF["x"] = x
            _BENCH_REF["want"] = {i: ref(x[i:i + 1]).numpy() for i in (0, 15)}
    return _BENCH_REF["x"], _BENCH_REF["want"]


@pytest.mark.parametrize("mode,batch,tol", [("fp32", 8, 1e-3), ("fp32", 16, 1e-3), ("bf16x3", 16, 1e-3), ("bf16", 16, 3e-2)],
                         ids=["fp32_B8_configs1", "fp32_B16", "bf16x3_B16", "bf16_B16"])
def test_generator_at_the_bench_shape_vs_the_cpu_oracle(ns, mode, batch, tol):
    """BASELINE.json configs[1] (generator forward, B=8, 512^2, fp32) and the batch the benchmarked step runs on (B=16,
    512^2) in every compute mode, pinned NUMERICALLY at full size: the first and the last slice of the batch against the CPU
    oracle's result for those slices -- fp32 and bf16x3 inside the north_star's 1e-3 rel-L2, bf16 inside 3e-2 -- plus the
    size-independent properties (tanh range, finite, and per-sample independence: slice 0 inside the batch == slice 0 alone)."""
    from cta_gan_amd import nets, synth
    x, want = _bench_shape_reference()
    last = batch - 1
    xs = torch.cat([x[:last], x[15:16]]).cuda()           # slices 0 .. batch-2 and slice 15 as the last one
    prev = nets.compute_mode()
    nets.set_default_compute_dtype({"fp32": torch.float32, "bf16": torch.bfloat16}.get(mode, mode))
    try:
        G = synth.fill_module(ns.Generator(1, 1), seed=0).cuda()
        with torch.no_grad():
            y = G(xs).float()
            y0 = G(xs[:1]).float()
        assert tuple(y.shape) == (batch, 1, 512, 512) and bool(torch.isfinite(y).all()) and float(y.abs().max()) <= 1.0
        e0 = rel_l2(y[0:1].cpu().numpy(), want[0])
        e15 = rel_l2(y[last:].cpu().numpy(), want[15])
        print("generator %s B=%d @ 512^2 vs CPU oracle: slice 0 %.3e, last slice %.3e" % (mode, batch, e0, e15))
        assert e0 <= tol and e15 <= tol, (mode, batch, e0, e15)
        # (bf16: a lone slice runs the 8-row tile variant of the wide convs -- other summation order of the InstanceNorm moments,
        #  and bf16 storage rounds the 1e-7 difference to other neighbours: observed 1.4e-2, the mode's own distance from fp32;
        #  bf16x3: a lone slice's residual blocks run the fused conv + InstanceNorm launches (ops.conv_in_fusable), which normalise
        #  the unrounded accumulators instead of the stored split pair: 2.4e-5, the mode's storage rounding)
        assert rel_l2(y[:1].cpu().numpy(), y0.cpu().numpy()) < {"fp32": 1e-6, "bf16x3": 1e-4, "bf16": 3e-2}[mode]
    finally:
        nets.set_default_compute_dtype({"fp32": torch.float32, "bf16": torch.bfloat16}.get(prev, prev))
        torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", [(1, 64, 96), (3, 128, 64)], ids=["B1_64x96", "B3_128x64"])
def test_generator_ragged_shapes_vs_oracle(ns, shape):
    """Non-square images and odd batch sizes (tiles hang over the grid in both directions), fwd + input grad."""
    from cta_gan_amd import synth
    b, h, w = shape
    ons = _oracle_ns()
    rng = np.random.default_rng(b * 1000 + h)
    x = torch.from_numpy(rng.uniform(-1, 1, (b, 1, h, w)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((b, 1, h, w)).astype(np.float32))
    ref = synth.fill_module(ons.Generator(1, 1), seed=31)
    hip = synth.fill_module(ns.Generator(1, 1), seed=31).to("cuda")
    xr = x.clone().requires_grad_(True)
    xh = x.cuda().requires_grad_(True)
    yr = ref(xr); yr.backward(g)
    yh = hip(xh); yh.backward(g.cuda())
    assert rel_l2(yh.detach().cpu().numpy(), yr.detach().numpy()) <= 1e-3
    assert rel_l2(xh.grad.cpu().numpy(), xr.grad.numpy()) <= 5e-3
    wr = dict(ref.named_parameters())["model_body.3.conv_block.5.weight"].grad
    wh = dict(hip.named_parameters())["model_body.3.conv_block.5.weight"].grad
    assert rel_l2(wh.cpu().numpy(), wr.numpy()) <= 5e-3


@pytest.mark.parametrize("shape", [(1, 72, 80), (2, 50, 66)], ids=["B1_72x80", "B2_50x66"])
def test_discriminator_odd_shapes_vs_oracle(ns, shape):
    """Odd spatial sizes through the stride-2 4x4 stack (floor division at every level) and the parity-class
    backward of those layers."""
    from cta_gan_amd import synth
    b, h, w = shape
    ons = _oracle_ns()
    rng = np.random.default_rng(h * 7 + w)
    x = torch.from_numpy(rng.uniform(-1, 1, (b, 1, h, w)).astype(np.float32))
    ref = synth.fill_module(ons.Discriminator(1), seed=32)
    hip = synth.fill_module(ns.Discriminator(1), seed=32).to("cuda")
    xr = x.clone().requires_grad_(True)
    xh = x.cuda().requires_grad_(True)
    yr = ref(xr); ((yr - 1) ** 2).mean().backward()
    yh = hip(xh); ((yh - 1) ** 2).mean().backward()
    assert rel_l2(yh.detach().cpu().numpy(), yr.detach().numpy()) <= 1e-3
    assert rel_l2(xh.grad.cpu().numpy(), xr.grad.numpy()) <= 5e-3
    for k in ("model.0.weight", "model.5.weight", "model.11.weight", "model.11.bias"):
        wr = dict(ref.named_parameters())[k].grad
        wh = dict(hip.named_parameters())[k].grad
        assert rel_l2(wh.cpu().numpy(), wr.numpy()) <= 5e-3, k


def test_cpu_tensors_and_bad_shapes_fail_loudly(ns):
    """No CPU fallback; illegal geometries raise instead of computing something else."""
    G = ns.Generator(1, 1).to("cuda")
    with pytest.raises(RuntimeError):
        G(torch.zeros(1, 1, 64, 64))                      # CPU tensor
    with pytest.raises(RuntimeError):
        G(torch.zeros(1, 1, 66, 64, device="cuda"))       # not a multiple of 4
    R = ns.Reg(128, 128, 1, 1).to("cuda")
    with pytest.raises(RuntimeError):
        R(torch.zeros(1, 1, 128, 128, device="cuda"), torch.zeros(1, 1, 128, 128, device="cuda"))


@pytest.mark.parametrize("which", ["gen", "reg", "disc"])
def test_network_forward_backward_bitwise_repeatable_at_bench_shape(ns, which):
    """Race screen at the bench shape (B=16, 512^2, bf16): a network's forward + backward run three times on the same inputs
    gives bit-identical outputs, input gradients and parameter gradients (no atomics inside the networks; split-K partials
    are reduced in a fixed order).  Only the STN scatter of the full step is order-dependent."""
    from cta_gan_amd import nets, synth
    from cta_gan_amd.Model.HdGan import Discriminator_m, Generator
    from cta_gan_amd.trainer.reg import Reg
    nets.set_default_compute_dtype(torch.bfloat16)
    try:
        B, S = 16, 512
        a = synth.synth_smooth_images("det_a", B, S).cuda()
        b = synth.synth_smooth_images("det_b", B, S).cuda()
        if which == "reg":
            net = synth.fill_module(Reg(S, S, 1, 1), seed=4).cuda()
            run = lambda x: net(x, b)                               # noqa: E731
        elif which == "gen":
            net = synth.fill_module(Generator(1, 1), seed=0).cuda()
            run = lambda x: net(x)                                  # noqa: E731
        else:
            net = synth.fill_module(Discriminator_m(1), seed=1).cuda()
            run = lambda x: net(x)[0][-1]                           # noqa: E731
        ref = None
        for rep in range(3):
            for p in net.parameters():
                p.grad = None
            x = a.clone().requires_grad_(True)
            y = run(x)
            (y.float() * torch.linspace(0.5, 1.5, y.numel(), device=y.device).view_as(y)).sum().backward()
            cur = (y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
            if ref is None:
                ref = cur
                continue
            assert torch.equal(ref[0], cur[0]), "output differs between runs"
            assert torch.equal(ref[1], cur[1]), "input gradient differs between runs"
            for k in ref[2]:
                assert torch.equal(ref[2][k], cur[2][k]), "gradient of %s differs between runs" % k
        del net, ref, cur
    finally:
        nets.set_default_compute_dtype(torch.float32)
        torch.cuda.empty_cache()
