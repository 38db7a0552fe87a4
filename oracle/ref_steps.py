"""ORACLE (test infrastructure only): the reference's per-batch step bodies.

Behavioural restatement of `trainer/HdTrainer.py:192-228` (stage 1),
`:705-751` (stage 2), `trainer/CycTrainer.py:138-197`, `trainer/p2pTrainer.py:122-148` and
`trainer/RegTrainer.py:170-198`, written as plain
functions over any set of modules with the reference's call signatures (the
imported reference classes in `make_golden.py`, `oracle.ref_models` on CPU, or
the HIP-backed `Model.*` classes in the GPU parity tests).  Device-agnostic:
tensors stay wherever the modules and the batch live.
"""
from __future__ import annotations

import copy
import random

import torch
import torch.nn.functional as F

HD_LAMBDAS = dict(Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2, Smooth_lamda=10)  # Yaml/HdGan.yaml:10-15
CYC_LAMBDAS = dict(Adv_lamda=1, Cyc_lamda=10)  # Yaml/CycleGan.yaml:9-12
P2P_LAMBDAS = dict(Adv_lamda=1, P2P_lamda=100)  # Yaml/P2p.yaml:8-9
REG_LAMBDAS = dict(Adv_lamda=1, Corr_lamda=20, Smooth_lamda=10)  # Yaml/CycleGan.yaml:9-12 (the keys Reg_Trainer reads)


def make_adam(params, lr=1e-4):
    return torch.optim.Adam(params, lr=lr, betas=(0.5, 0.999))  # HdTrainer.py:612-616


def hd_step(nets, opts, batch, cfg=HD_LAMBDAS, stage=2, smooth_fn=None, gan_loss=None, micro_batch=None):
    """One G+R step then one D step.  `nets` = dict(G, D, R, T); `opts` = dict(G, D, R).

    stage=1: `Discriminator` + plain MSE (HdTrainer.py:192-228).
    stage=2: `Discriminator_m` + GANLoss + masked L1 term (HdTrainer.py:705-751).
    Returns a dict of the scalar loss terms (as Python floats) and fake_B of the D step.

    micro_batch (None = the whole batch at once, the reference's own form): the SAME optimiser step evaluated over chunks of that
    many samples with gradient accumulation, for full-size batches whose CPU autograd tape would not fit the host (B=16 at 512^2:
    ~45 GB).  Exact in real arithmetic: every loss term of the step is a mean over the batch (L1 / MSE means, smooothing_loss's
    means, GANLoss over per-sample pooled maps) and every normalisation is per sample (InstanceNorm), so the step's loss is the
    chunk-size-weighted mean of the chunk losses and its gradient the same mean of the chunk gradients.
    """
    if micro_batch is not None and micro_batch < batch["A2"].shape[0]:
        return _hd_step_chunked(nets, opts, batch, cfg, stage, smooth_fn, gan_loss, int(micro_batch))
    G, D, R, T = nets["G"], nets["D"], nets["R"], nets["T"]
    real_A2 = batch["A2"]
    real_B1 = batch["B1"].clone()  # the reference binarises its input buffer in place (:726-728)
    real_B2 = batch["B2"]
    real_BB2 = copy.deepcopy(real_B2)
    dev = real_A2.device
    one = torch.ones(1, 1, device=dev)
    zero = torch.zeros(1, 1, device=dev)

    opts["R"].zero_grad()
    opts["G"].zero_grad()
    fake_B, flow, warped, terms = _hd_g_terms(G, D, R, T, real_A2, real_B1, real_B2, cfg, stage, smooth_fn, gan_loss, one)
    sm, sr, adv, sr2, total = terms
    total.backward()
    opts["R"].step()
    opts["G"].step()

    opts["D"].zero_grad()
    with torch.no_grad():
        fake_B2 = G(real_A2)
    loss_d = _hd_d_loss(D, fake_B2, real_BB2, cfg, stage, gan_loss, one, zero)
    loss_d.backward()
    opts["D"].step()
    return dict(SM=float(sm), SR=float(sr), adv=float(adv), SR2=float(sr2), total=float(total),
                loss_D=float(loss_d), fake_B_first=fake_B.detach(), fake_B=fake_B2.detach(),
                flow=flow.detach(), warped=warped.detach())


def _hd_g_terms(G, D, R, T, real_A2, real_B1, real_B2, cfg, stage, smooth_fn, gan_loss, one):
    """The G+R half's forward and loss terms (HdTrainer.py:712-736; stage 1 :199-214); real_B1 is binarised in place."""
    fake_B = G(real_A2)
    flow = R(fake_B, real_B2)
    warped = T(fake_B, flow)
    sm = cfg["Smooth_lamda"] * smooth_fn(flow)
    sr = cfg["Corr_lamda1"] * F.l1_loss(warped, real_B2)
    pred_fake = D(fake_B)
    if stage == 1:
        adv = cfg["Adv_lamda1"] * F.mse_loss(pred_fake, one.expand_as(pred_fake))
        total = sm + adv + sr
        sr2 = torch.zeros(())
    else:
        adv = cfg["Adv_lamda1"] * gan_loss(pred_fake, True)
        bb = real_B1
        bb[bb < 0.3] = 0
        bb[bb >= 0.3] = 1
        real_B2m = real_B2 * bb
        real_B2m[real_B2m == 0] = -1
        warped_m = warped * bb
        warped_m[warped_m == 0] = -1  # in-place on a graph tensor, as the reference does (:733-734)
        sr2 = cfg["Corr_lamda2"] * F.l1_loss(warped_m, real_B2m)
        total = sm + adv + sr + sr2
    return fake_B, flow, warped, (sm, sr, adv, sr2, total)


def _hd_d_loss(D, fake_B2, real_BB2, cfg, stage, gan_loss, one, zero):
    """The D half's loss (HdTrainer.py:744-749; stage 1 :222-226)."""
    pf = D(fake_B2)
    pr = D(real_BB2)
    if stage == 1:
        return cfg["Adv_lamda1"] * F.mse_loss(pf, zero.expand_as(pf)) + \
            cfg["Adv_lamda1"] * F.mse_loss(pr, one.expand_as(pr))
    return cfg["Adv_lamda1"] * (gan_loss(pf, False) + gan_loss(pr, True)) / 2


def _hd_step_chunked(nets, opts, batch, cfg, stage, smooth_fn, gan_loss, mb):
    """`hd_step` over chunks of `mb` samples with gradient accumulation (see its docstring)."""
    G, D, R, T = nets["G"], nets["D"], nets["R"], nets["T"]
    nb = batch["A2"].shape[0]
    dev = batch["A2"].device
    one = torch.ones(1, 1, device=dev)
    zero = torch.zeros(1, 1, device=dev)
    acc = dict(SM=0.0, SR=0.0, adv=0.0, SR2=0.0, total=0.0, loss_D=0.0)
    keep = dict(fake_B_first=[], flow=[], warped=[], fake_B=[])
    opts["R"].zero_grad()
    opts["G"].zero_grad()
    for s in range(0, nb, mb):
        sl = slice(s, min(s + mb, nb))
        w = (sl.stop - sl.start) / nb
        fake_B, flow, warped, terms = _hd_g_terms(G, D, R, T, batch["A2"][sl], batch["B1"][sl].clone(), batch["B2"][sl], cfg, stage,
                                                  smooth_fn, gan_loss, one)
        (terms[4] * w).backward()
        for k, v in zip(("SM", "SR", "adv", "SR2", "total"), terms):
            acc[k] += w * float(v)
        keep["fake_B_first"].append(fake_B.detach()); keep["flow"].append(flow.detach()); keep["warped"].append(warped.detach())
        del fake_B, flow, warped, terms
    opts["R"].step()
    opts["G"].step()
    opts["D"].zero_grad()
    for s in range(0, nb, mb):
        sl = slice(s, min(s + mb, nb))
        w = (sl.stop - sl.start) / nb
        with torch.no_grad():
            fake_B2 = G(batch["A2"][sl])
        loss_d = _hd_d_loss(D, fake_B2, copy.deepcopy(batch["B2"][sl]), cfg, stage, gan_loss, one, zero)
        (loss_d * w).backward()
        acc["loss_D"] += w * float(loss_d)
        keep["fake_B"].append(fake_B2.detach())
    opts["D"].step()
    out = dict(acc)
    out.update({k: torch.cat(v) for k, v in keep.items()})
    return out


def reg_step(nets, opts, batch, cfg=REG_LAMBDAS, smooth_fn=None):
    """`Reg_Trainer.train` body (trainer/RegTrainer.py:170-198): batch keys A, B; `nets` = dict(G, D, R, T) with the
    CycleGan `Discriminator`.  Term for term it is the stage-1 CTA-GAN step (HdTrainer.py:192-228) under the config keys
    Corr_lamda / Adv_lamda / Smooth_lamda, so it is evaluated by `hd_step(stage=1)`."""
    hd_cfg = dict(Adv_lamda1=cfg["Adv_lamda"], Corr_lamda1=cfg["Corr_lamda"], Smooth_lamda=cfg["Smooth_lamda"])
    hd_batch = dict(A2=batch["A"], B1=batch["B"], B2=batch["B"])
    return hd_step(nets, opts, hd_batch, cfg=hd_cfg, stage=1, smooth_fn=smooth_fn)


def p2p_step(nets, opts, batch, cfg=P2P_LAMBDAS):
    """`P2p_Trainer.train` body (trainer/p2pTrainer.py:122-148): `nets` = dict(G, D) with D = `Discriminator(2 input_nc)`
    judging the channel-concatenated (input, output) pair; `opts` = dict(G, D)."""
    G, D = nets["G"], nets["D"]
    real_A, real_B = batch["A"], batch["B"]
    dev = real_A.device
    one = torch.ones(1, 1, device=dev)
    zero = torch.zeros(1, 1, device=dev)

    opts["G"].zero_grad()
    fake_B = G(real_A)
    loss_l1 = F.l1_loss(fake_B, real_B) * cfg["P2P_lamda"]                       # :129
    pred_fake = D(torch.cat((real_A, fake_B), 1))                                # :131-132
    loss_gan = F.mse_loss(pred_fake, one.expand_as(pred_fake)) * cfg["Adv_lamda"]  # :133
    total = loss_l1 + loss_gan
    total.backward()
    opts["G"].step()

    opts["D"].zero_grad()
    with torch.no_grad():
        fake_B2 = G(real_A)
    pf = D(torch.cat((real_A, fake_B2), 1)) * cfg["Adv_lamda"]                    # :143 (the weight scales the PREDICTION)
    pr = D(torch.cat((real_A, real_B), 1)) * cfg["Adv_lamda"]
    loss_d = F.mse_loss(pf, zero.expand_as(pf)) + F.mse_loss(pr, one.expand_as(pr))
    loss_d.backward()
    opts["D"].step()
    return dict(L1=float(loss_l1), GAN_A2B=float(loss_gan), total=float(total), loss_D=float(loss_d),
                fake_B_first=fake_B.detach(), fake_B=fake_B2.detach())


class ReplayBuffer:
    """50-image history pool using Python's global `random` -- trainer/utils.py:120-140."""

    def __init__(self, max_size=50):
        self.max_size = max_size
        self.data = []

    def push_and_pop(self, data):
        out = []
        for element in data.data:
            element = torch.unsqueeze(element, 0)
            if len(self.data) < self.max_size:
                self.data.append(element)
                out.append(element)
            elif random.uniform(0, 1) > 0.5:
                i = random.randint(0, self.max_size - 1)
                out.append(self.data[i].clone())
                self.data[i] = element
            else:
                out.append(element)
        return torch.cat(out)


def cyc_step(nets, opts, bufs, batch, cfg=CYC_LAMBDAS, micro_batch=None):
    """CycleGAN step: G (both directions), D_A, D_B -- trainer/CycTrainer.py:138-197.

    `nets` = dict(G_A2B, G_B2A, D_A, D_B); `opts` = dict(G, D_A, D_B); `bufs` = dict(A, B).
    micro_batch: as for `hd_step` -- the same step over chunks of samples with gradient accumulation (every term is a batch
    mean, the networks are per-sample; the history pools see the samples in the same order).
    """
    GA, GB, DA, DB = nets["G_A2B"], nets["G_B2A"], nets["D_A"], nets["D_B"]
    real_A, real_B = batch["A"], batch["B"]
    dev = real_A.device
    one = torch.ones(1, 1, device=dev)
    zero = torch.zeros(1, 1, device=dev)

    def mse(p, t):
        return F.mse_loss(p, t.expand_as(p))

    if micro_batch is not None and micro_batch < real_A.shape[0]:
        nb, mb = real_A.shape[0], int(micro_batch)
        chunks = [slice(s, min(s + mb, nb)) for s in range(0, nb, mb)]
        acc = dict(GAN_A2B=0.0, GAN_B2A=0.0, cyc_ABA=0.0, cyc_BAB=0.0, total=0.0, loss_D_A=0.0, loss_D_B=0.0)
        fakes_A, fakes_B = [], []
        opts["G"].zero_grad()
        for sl in chunks:
            w = (sl.stop - sl.start) / nb
            fake_B = GA(real_A[sl])
            l_gan_ab = cfg["Adv_lamda"] * mse(DB(fake_B), one)
            fake_A = GB(real_B[sl])
            l_gan_ba = cfg["Adv_lamda"] * mse(DA(fake_A), one)
            l_cyc_a = cfg["Cyc_lamda"] * F.l1_loss(GB(fake_B), real_A[sl])
            l_cyc_b = cfg["Cyc_lamda"] * F.l1_loss(GA(fake_A), real_B[sl])
            total = l_gan_ab + l_gan_ba + l_cyc_a + l_cyc_b
            (total * w).backward()
            for k, v in zip(("GAN_A2B", "GAN_B2A", "cyc_ABA", "cyc_BAB", "total"), (l_gan_ab, l_gan_ba, l_cyc_a, l_cyc_b, total)):
                acc[k] += w * float(v)
            fakes_A.append(fake_A.detach()); fakes_B.append(fake_B.detach())
            del fake_A, fake_B, total
        opts["G"].step()
        for key, Dn, real, fakes, buf in (("loss_D_A", DA, real_A, fakes_A, bufs["A"]), ("loss_D_B", DB, real_B, fakes_B, bufs["B"])):
            opts["D_A" if key == "loss_D_A" else "D_B"].zero_grad()
            for sl, fk in zip(chunks, fakes):
                w = (sl.stop - sl.start) / nb
                l_d = cfg["Adv_lamda"] * mse(Dn(real[sl]), one) + cfg["Adv_lamda"] * mse(Dn(buf.push_and_pop(fk).detach()), zero)
                (l_d * w).backward()
                acc[key] += w * float(l_d)
            opts["D_A" if key == "loss_D_A" else "D_B"].step()
        out = dict(acc)
        out.update(fake_B=torch.cat(fakes_B), fake_A=torch.cat(fakes_A))
        return out

    opts["G"].zero_grad()
    fake_B = GA(real_A)
    l_gan_ab = cfg["Adv_lamda"] * mse(DB(fake_B), one)
    fake_A = GB(real_B)
    l_gan_ba = cfg["Adv_lamda"] * mse(DA(fake_A), one)
    rec_A = GB(fake_B)
    l_cyc_a = cfg["Cyc_lamda"] * F.l1_loss(rec_A, real_A)
    rec_B = GA(fake_A)
    l_cyc_b = cfg["Cyc_lamda"] * F.l1_loss(rec_B, real_B)
    total = l_gan_ab + l_gan_ba + l_cyc_a + l_cyc_b
    total.backward()
    opts["G"].step()

    opts["D_A"].zero_grad()
    l_da = cfg["Adv_lamda"] * mse(DA(real_A), one) + \
        cfg["Adv_lamda"] * mse(DA(bufs["A"].push_and_pop(fake_A).detach()), zero)
    l_da.backward()
    opts["D_A"].step()

    opts["D_B"].zero_grad()
    l_db = cfg["Adv_lamda"] * mse(DB(real_B), one) + \
        cfg["Adv_lamda"] * mse(DB(bufs["B"].push_and_pop(fake_B).detach()), zero)
    l_db.backward()
    opts["D_B"].step()
    return dict(GAN_A2B=float(l_gan_ab), GAN_B2A=float(l_gan_ba), cyc_ABA=float(l_cyc_a),
                cyc_BAB=float(l_cyc_b), total=float(total), loss_D_A=float(l_da), loss_D_B=float(l_db),
                fake_B=fake_B.detach(), fake_A=fake_A.detach())
