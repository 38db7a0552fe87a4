"""GPU: (1) the stand-alone `trainer.layers` blocks against the oracle's; (2) checkpoint interchange: a
`state_dict` written by the CPU oracle (= the reference's keys and shapes) loads into the HIP classes with
`strict=True` and reproduces the oracle's outputs; the HIP classes' own `state_dict` round-trips through
`torch.save` / `torch.load` into the oracle."""
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cta_gan_amd import nets
    nets.set_default_compute_dtype(torch.float32)
    return torch.device("cuda:0")


def test_standalone_layers_match_oracle(dev):
    from cta_gan_amd import synth
    from cta_gan_amd.trainer import layers as L
    from oracle import ref_models as R
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 32, 24, 40)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((2, 64, 12, 20)).astype(np.float32))
    hip = synth.fill_module(L.DownBlock(32, 64, 3, 1, 1, activation="leaky_relu", init_func="kaiming", bias=True,
                                        use_resnet=True, use_norm=False), seed=11).to(dev)
    ref = synth.fill_module(R.DownBlock(32, 64), seed=11)
    assert list(hip.state_dict()) == list(ref.state_dict())
    xh = x.to(dev).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ph, sh = hip(xh)
    pr, sr = ref(xr)
    assert _rel(ph, pr) < 1e-4 and _rel(sh, sr) < 1e-4
    (ph * g.to(dev)).sum().backward()
    (pr * g).sum().backward()
    assert _rel(xh.grad, xr.grad) < 2e-3
    hip_t = synth.fill_module(L.ResnetTransformer(64, 2, "kaiming"), seed=12).to(dev)
    ref_t = synth.fill_module(R.ResnetTransformer(64, 2), seed=12)
    assert list(hip_t.state_dict()) == list(ref_t.state_dict())
    y = torch.from_numpy(rng.standard_normal((1, 64, 20, 18)).astype(np.float32))
    assert _rel(hip_t(y.to(dev)), ref_t(y)) < 1e-4
    hip_c = synth.fill_module(L.Conv(64, 32, 1, 1, 0, activation="leaky_relu", init_func="kaiming"), seed=13).to(dev)
    ref_c = synth.fill_module(R.Conv(64, 32, 1, 1, 0), seed=13)
    assert list(hip_c.state_dict()) == list(ref_c.state_dict())
    assert _rel(hip_c(y.to(dev)), ref_c(y)) < 1e-4


def test_checkpoint_interchange(dev):
    from cta_gan_amd import synth
    from cta_gan_amd.Model import HdGan as H
    from cta_gan_amd.trainer.reg import Reg
    from oracle import ref_models as R
    x = synth.synth_images("ckpt_x", 1, 256)
    pairs = [(R.Generator(1, 1), H.Generator(1, 1), lambda m, t: m(t)),
             (R.Discriminator(1), H.Discriminator(1), lambda m, t: m(t)),
             (R.Discriminator_m(1, num_D=2), H.Discriminator_m(1, num_D=2), lambda m, t: m(t)[1][-1]),
             (R.Reg(256, 256, 1, 1), Reg(256, 256, 1, 1), lambda m, t: m(t, t.flip(-1)))]
    for seed, (ref, hip, call) in enumerate(pairs):
        synth.fill_module(ref, seed=40 + seed, gains={"output.conv2d.weight": 0.25})
        buf = io.BytesIO()
        torch.save(ref.state_dict(), buf)                       # what the reference trainers write (.pth)
        buf.seek(0)
        missing = hip.load_state_dict(torch.load(buf), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        hip = hip.to(dev)
        with torch.no_grad():
            want = call(ref, x)
            got = call(hip, x.to(dev))
        assert _rel(got, want) < 1e-3, type(ref).__name__
        buf2 = io.BytesIO()
        torch.save(hip.state_dict(), buf2)                      # ... and back
        buf2.seek(0)
        back = type(ref)(*([1, 1] if isinstance(ref, R.Generator) else [256, 256, 1, 1] if isinstance(ref, R.Reg)
                           else [1])) if not isinstance(ref, R.Discriminator_m) else R.Discriminator_m(1, num_D=2)
        back.load_state_dict({k: v.cpu() for k, v in torch.load(buf2).items()}, strict=True)
        with torch.no_grad():
            assert _rel(call(back, x), want) < 1e-6


def test_trainer_epoch_checkpoints_and_resume(dev, tmp_path):
    """`train()` ends every epoch with the reference's checkpoint files (p2pTrainer.py:179-184 names; state_dict keys the
    reference classes load strictly) plus `train_state_<epoch>.pth`; `resume(epoch)` on a fresh trainer restores weights,
    Adam moments / step counts and rates, so its next step equals the original trainer's next step (1e-5 relative on the
    losses: same kernels, same inputs, same state)."""
    import os
    from cta_gan_amd import synth
    from cta_gan_amd.trainer import P2p_Trainer
    from oracle import ref_models as R
    root = str(tmp_path) + "/"

    def cfg():
        return dict(input_nc=1, output_nc=1, size=64, batchSize=2, lr=1e-4, Adv_lamda=1, P2P_lamda=100, epoch=0, n_epochs=1,
                    decay_epoch=1, synthetic_steps=2, save_root=root)
    tr = P2p_Trainer(cfg())
    synth.fill_module(tr.netG_A2B, seed=0)
    synth.fill_module(tr.netD_B, seed=7)
    tr.train()                                               # epochs 1 and 2 (the second at the decayed rate 0)
    for stem in ("netG_A2B_", "netD_B_", "train_state_"):
        assert os.path.exists(root + stem + "1.pth") and os.path.exists(root + stem + "2.pth"), stem
    R.Generator(1, 1).load_state_dict(torch.load(root + "netG_A2B_1.pth", map_location="cpu"), strict=True)
    R.Discriminator(2).load_state_dict(torch.load(root + "netD_B_1.pth", map_location="cpu"), strict=True)

    a, b = P2p_Trainer(cfg()), P2p_Trainer(cfg())
    for t in (a, b):
        t.resume(1)
        assert t.config["epoch"] == 1 and t.optimizer_G.state_dict()["state"][0]["step"] == 2
    batch = a.synthetic_batch(99)
    la = a.train_step(batch, sync_losses=True)
    # b continues from the same files: identical next step
    lb = b.train_step(batch, sync_losses=True)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * max(abs(la[k]), 1e-6), (k, la[k], lb[k])
    # and the resumed state is not the initial one: a fresh, un-resumed trainer with the initial weights gives another loss
    c = P2p_Trainer(cfg())
    synth.fill_module(c.netG_A2B, seed=0)
    synth.fill_module(c.netD_B, seed=7)
    lc = c.train_step(batch, sync_losses=True)
    assert abs(lc["total"] - la["total"]) > 1e-4 * abs(la["total"])


def test_validation_pass_every_fifth_epoch_names_the_checkpoints(dev, tmp_path, capsys):
    """The in-training validation pass (p2pTrainer.py:153-174; HdTrainer.py:765-790 is the same loop with the `b.pth` suffix): every
    fifth epoch the generator runs over the validation batches, PSNR(fake, real) and compare_ssim(fake, real) are averaged -- on the
    device (ops.val_psnr / ops.ssim) -- and spliced into the checkpoint names as
    `str(epoch) + '_' + str(round(PSNR, 4)) + '_' + str(round(SSIM, 4))`; the other epochs keep the plain names.  The numbers are the
    oracle's on the same generator outputs, and `resume(5)` finds the renamed files."""
    import os
    from cta_gan_amd import synth
    from cta_gan_amd.trainer import P2p_Trainer
    from cta_gan_amd.trainer.HdTrainer import run_validation
    from oracle import ref_metrics
    root = str(tmp_path) + "/"
    cfg = dict(input_nc=1, output_nc=1, size=64, batchSize=2, lr=1e-4, Adv_lamda=1, P2P_lamda=100, epoch=3, n_epochs=5,
               decay_epoch=0, synthetic_steps=1, save_root=root)
    tr = P2p_Trainer(cfg)
    synth.fill_module(tr.netG_A2B, seed=0)
    synth.fill_module(tr.netD_B, seed=7)
    val = [{"A": synth.synth_smooth_images("va%d" % i, 2, 64), "B": synth.synth_smooth_images("vb%d" % i, 2, 64)} for i in range(2)]
    val[1]["B"][:, :, :9] = -1          # background pixels are left out of the PSNR
    tr.train(val_dataloader=val)        # epochs 4 and 5 (synthetic training batches)
    out = capsys.readouterr().out
    assert out.count("PSNR:") == 1 and out.count("SSIM:") == 1
    assert os.path.exists(root + "netG_A2B_4.pth") and os.path.exists(root + "netD_B_4.pth")
    # what the pass must have computed, from the weights epoch 5 saved
    psnr = ssim = 0.0
    with torch.no_grad():
        for bt in val:
            fk = tr.netG_A2B(bt["A"].cuda()).float().cpu().numpy()
            for i in range(2):
                psnr += ref_metrics.psnr(fk[i, 0], bt["B"][i, 0].numpy())
                ssim += ref_metrics.ssim(fk[i, 0], bt["B"][i, 0].numpy())
    psnr, ssim = psnr / 4, ssim / 4
    got = run_validation(tr, val, ("A", "B"))
    assert got[2] == 4 and abs(got[0] - psnr) < 1e-4 * abs(psnr) and abs(got[1] - ssim) < 1e-6
    st = "5_" + str(round(got[0], 4)) + "_" + str(round(got[1], 4))
    for stem in ("netG_A2B_", "netD_B_"):
        assert os.path.exists(root + stem + st + ".pth"), sorted(os.listdir(root))
        assert not os.path.exists(root + stem + "5.pth")
    state = torch.load(root + "train_state_5.pth", map_location="cpu")
    assert state["files"]["netG_A2B_"] == "netG_A2B_" + st + ".pth" and abs(state["val"]["SSIM"] - got[1]) < 1e-12
    fresh = P2p_Trainer(dict(cfg, epoch=0))
    fresh.resume(5)
    for k, v in tr.netG_A2B.state_dict().items():
        assert torch.equal(v, fresh.netG_A2B.state_dict()[k]), k
    # Hd trainers: same pass, `b.pth` suffix (HdTrainer.py:785-790)
    from cta_gan_amd.trainer import Hd_Trainer_x2
    assert Hd_Trainer_x2._val_suffix == "b.pth" and Hd_Trainer_x2._val_keys == ("A2", "B2")
