"""MI355X-native drop-in for the reference's `Model/HdGan.py`.

Same class names, constructor signatures, forward return structures and
`state_dict` keys as yml-bit/CTA-GAN `Model/HdGan.py:11-293`; every layer
underneath is a hand-written gfx950 kernel reached through libctagan_hip.so
(cta_gan_amd/nets.py, engine.py).  Inputs must live on the GPU: there is no CPU
fallback in the product path.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn

from .. import engine as E
from .. import nets
from ..nets import GeneratorNet, HipNet, PatchStack, ResidualBlockNet, global_avgpool


class ResidualBlock(ResidualBlockNet):
    """reference: Model/HdGan.py:49-63 (keys conv_block.{1,5}.{weight,bias})."""


class Generator(GeneratorNet):
    """reference: Model/HdGan.py:65-113.  forward(x: (B, input_nc, H, W)) -> (B, output_nc, H, W) in (-1, 1)."""


class Discriminator(HipNet):
    """reference: Model/HdGan.py:115-145.  forward(x) -> (B, 1): PatchGAN map, globally average-pooled."""

    def __init__(self, input_nc):
        super().__init__()
        self.stack = PatchStack(self, input_nc, ["model.0", "model.2", "model.5", "model.8", "model.11"])

    def _run(self, tape, inputs, need_in):
        feats, x_act = self.stack.run_stack(tape, self._cache, inputs[0], need_in[0], self.dtype_)
        return [feats[-1]], [x_act], nets._image_grad_finish(1)

    def forward(self, x):
        patch = self._call(x)[0]           # (B, 1, h, w) fp32
        return global_avgpool(patch)       # HdGan.py:145


class NLayerDiscriminator(HipNet):
    """reference: Model/HdGan.py:148-205 (InstanceNorm variant; the BatchNorm default is never used on this path)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False,
                 getIntermFeat=False):
        super().__init__()
        norm = _check_norm(norm_layer)
        self.getIntermFeat, self.n_layers = getIntermFeat, n_layers
        if getIntermFeat:
            keys = ["model%d.0" % j for j in range(n_layers + 2)]
            norm_keys = ["model%d.1" % j for j in range(1, n_layers + 1)]
        else:
            keys = _sequential_conv_keys("model", n_layers)
            norm_keys = ["model.%d" % (int(k.split(".")[1]) + 1) for k in keys[1:n_layers + 1]]
        # use_sigmoid appends nn.Sigmoid() as its own group (:177-178).  With getIntermFeat the reference's forward only walks
        # model0 .. model{n_layers+1} (:185-187), so the sigmoid is never applied there: reproduced.
        self.stack = PatchStack(self, input_nc, keys, ndf, n_layers, sigmoid=bool(use_sigmoid) and not getIntermFeat,
                                norm=norm, norm_keys=norm_keys if norm == "batch" else ())

    def _run(self, tape, inputs, need_in):
        feats, x_act = self.stack.run_stack(tape, self._cache, inputs[0], need_in[0], self.dtype_)
        outs = feats if self.getIntermFeat else [feats[-1]]
        return outs, [x_act], nets._image_grad_finish(1)

    def forward(self, input):
        res = self._call(input)
        return list(res) if self.getIntermFeat else res[0]


def _sequential_conv_keys(prefix, n_layers):
    """state_dict prefixes of the convs inside NLayerDiscriminator's flat nn.Sequential `model` (getIntermFeat=False,
    Model/HdGan.py:183-187): [conv, lrelu] [conv, norm, lrelu] x n [conv] -> indices 0, 2, 5, 8, 11 for n_layers = 3."""
    idx, keys = 0, []
    for j in range(n_layers + 2):
        keys.append("%s.%d" % (prefix, idx))
        idx += 2 if j == 0 else 3
    return keys


def _check_norm(norm_layer):
    """"instance" for an affine-free nn.InstanceNorm2d (every trainer of the reference passes it, Model/HdGan.py:208), "batch"
    for a default nn.BatchNorm2d (NLayerDiscriminator's own default, :149); anything else is not implemented in HIP."""
    probe = norm_layer(4)
    if isinstance(probe, nn.InstanceNorm2d) and not probe.affine and not probe.track_running_stats:
        return "instance"
    if type(probe) is nn.BatchNorm2d and probe.affine and probe.track_running_stats and probe.momentum == 0.1 and probe.eps == 1e-5:
        return "batch"
    raise NotImplementedError("only affine-free nn.InstanceNorm2d (the reference's norm on this path) and the default "
                              "nn.BatchNorm2d are implemented in HIP; got %r" % (probe,))


def center_crop(img: torch.Tensor, size: int) -> torch.Tensor:
    """torchvision.transforms.functional.center_crop for tensors (an index op: bit-exact by construction)."""
    h, w = img.shape[-2:]
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return img[..., top:top + size, left:left + size]


class _ScaleNet(HipNet):
    """One scale of Discriminator_m as its own autograd node; parameters live on the owner."""

    def __init__(self, owner, stack):
        super().__init__()
        self._owner = [owner]
        self.stack = stack
        self._cache = owner._cache      # ONE pack cache per discriminator: _NetFn refreshes the packs the stack uses

    def parameters(self, recurse=True):
        ps = [p for s in self.stack._slots() for p in (s.weight, s.bias)]
        if self.stack.norm == "batch":
            ps += [p for j in range(len(self.stack.norm_keys)) for p in (self.stack._norm_slot(j).weight, self.stack._norm_slot(j).bias)]
        return iter(ps)

    def _run(self, tape, inputs, need_in):
        feats, x_act = self.stack.run_stack(tape, self._owner[0]._cache, inputs[0], need_in[0],
                                            self._owner[0].dtype_)
        if self._owner[0].patch_only:
            feats = feats[-1:]
        return feats, [x_act], nets._image_grad_finish(1)


class Discriminator_m(HipNet):
    """reference: Model/HdGan.py:207-256.  forward(x) -> list[num_D] of list[n_layers+2] feature maps; scale i
    runs sub-net `scale{num_D-1-i}` and the input is centre-cropped to half size between scales (:251)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=functools.partial(nn.InstanceNorm2d, affine=False),
                 use_sigmoid=False, num_D=1, getIntermFeat=True):
        super().__init__()
        norm = _check_norm(norm_layer)
        self.num_D, self.n_layers, self.getIntermFeat = num_D, n_layers, getIntermFeat
        # An extension the trainers of this package switch on: a caller that only reads the PatchGAN map of each scale
        # (GANLoss does: feats[-1]) gets None in place of the intermediate feature maps, which then never leave the network
        # (in the split-pair "bf16x3" mode every returned wide map costs a pair -> fp32 pass).  Default: the reference's result.
        self.patch_only = False
        self._scales = []
        for i in range(num_D):
            if getIntermFeat:     # scale{i}_layer{j} = netD.model{j} (:218-219); the sigmoid group is never copied (j < n_layers + 2)
                keys = ["scale%d_layer%d.0" % (i, j) for j in range(n_layers + 2)]
                norm_keys = ["scale%d_layer%d.1" % (i, j) for j in range(1, n_layers + 1)]
            else:                 # layer{i} = netD.model, the flat Sequential (:221), sigmoid included
                keys = _sequential_conv_keys("layer%d" % i, n_layers)
                norm_keys = ["layer%d.%d" % (i, int(k.split(".")[1]) + 1) for k in keys[1:n_layers + 1]]
            stack = PatchStack(self, input_nc, keys, ndf, n_layers, sigmoid=bool(use_sigmoid) and not getIntermFeat,
                               norm=norm, norm_keys=norm_keys if norm == "batch" else ())
            object.__setattr__(self, "_scale%d" % i, _ScaleNet(self, stack))  # not a registered submodule
            self._scales.append(getattr(self, "_scale%d" % i))

    def forward(self, input):
        result = []
        cur = input
        for i in range(self.num_D):
            s = cur.size(2)
            net = self._scales[self.num_D - 1 - i]
            feats = list(net._call(cur))
            if self.patch_only and self.getIntermFeat:
                feats = [None] * (self.n_layers + 1) + feats
            result.append(feats if self.getIntermFeat else feats[-1:])      # singleD_forward (:226-234): [x] without the maps
            if i != self.num_D - 1:
                cur = center_crop(cur, int(s / 2))
        return result


class GANLoss(nn.Module):
    """reference: Model/HdGan.py:258-293.  LSGAN on the globally pooled last feature map, scale weights
    w = [1.8, 0.2] (:273).  The (B, 1) MSE against the broadcast target is scalar glue on B floats."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=torch.Tensor):
        super().__init__()
        # the reference fills its target tensors with 1.0 / 0.0 whatever the two label arguments say (:262-263): reproduced
        self.real_label, self.fake_label = 1.0, 0.0
        self.bce = not use_lsgan        # nn.BCELoss instead of nn.MSELoss (:264-267): expects a sigmoid discriminator

    W = (1.8, 0.2)      # per-scale weights of the multi-scale branch (Model/HdGan.py:273)

    def _one(self, x, target_is_real, weight):
        tgt = self.real_label if target_is_real else self.fake_label
        if self.bce and x.shape[0] != 1:
            # nn.BCELoss refuses a (1, 1) target against a (B, 1) input: the reference's BCE branch only runs at batch size 1
            raise ValueError("Using a target size (torch.Size([1, 1])) that is different to the input size (torch.Size([%d, 1])) "
                             "is deprecated. Please ensure they have the same size." % x.shape[0])
        # pooling, (. - target)^2 or BCE, batch mean and weight: one fused reduction
        return nets.lsgan_loss(x, tgt, weight, bce=self.bce)

    def __call__(self, input, target_is_real, weight=1.0):
        """`weight` (an extension of the reference's signature; default 1) multiplies the loss inside the fused reduction."""
        if isinstance(input[0], list):
            terms = [self._one(feats[-1], target_is_real, weight * self.W[i]) for i, feats in enumerate(input)]
            return nets.add_scalars(*terms)
        return self._one(input[-1], target_is_real, weight)

    def pair(self, input, nb, weight=1.0):
        """GANLoss(input[:nb], False) + GANLoss(input[nb:], True) for ONE batched discriminator pass over [fake | real]
        (the D step, HdTrainer.py:745-747): no slicing of the feature maps, one fused reduction per scale."""
        if isinstance(input[0], list):
            terms = [nets.lsgan_loss_pair(feats[-1], nb, self.fake_label, self.real_label, weight * self.W[i], bce=self.bce)
                     for i, feats in enumerate(input)]
            return nets.add_scalars(*terms)
        return nets.lsgan_loss_pair(input[-1], nb, self.fake_label, self.real_label, weight, bce=self.bce)


class DataPrefetcher:
    """reference: Model/HdGan.py:11-47 -- side-stream host->device prefetch of dict batches (`next()` returns the batch
    whose copy was started one call earlier and starts the next one; `None` at the end; the 'meta' entry stays on the
    host).  The reference defines it and never uses it: its loop does three synchronous copies per step
    (trainer/HdTrainer.py:708-711).  Here the trainers' `train()` consume it, and the copy really is asynchronous: every
    tensor goes through one of TWO page-locked staging buffers per key (a pageable source makes `non_blocking=True` a
    synchronous staged copy), so the H2D of batch i+1 runs on the copy stream while batch i trains."""

    def __init__(self, loader, device="cuda:0"):
        self.loader = iter(loader)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._stage = [{}, {}]          # key -> pinned tensor, per slot
        self._done = [None, None]       # event: the slot's last H2D has drained (its buffers may be rewritten)
        self._slot = 0
        self.copy_events = None         # (start, end) of the most recent H2D on the copy stream (tests / profiling)
        self.preload()

    def _pinned(self, slot, key, t):
        buf = self._stage[slot].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:     # first use, or a ragged trailing batch
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._stage[slot][key] = buf
        return buf

    def preload(self):
        try:
            self.batch = next(self.loader)
        except StopIteration:
            self.batch = None
            return
        slot = self._slot
        self._slot ^= 1
        if self._done[slot] is not None:
            self._done[slot].synchronize()      # two batches ago: long finished in steady state
        host = {}
        for k, v in self.batch.items():
            if k == "meta" or not torch.is_tensor(v) or v.is_cuda:
                continue
            if v.is_pinned():
                host[k] = v
            else:
                host[k] = self._pinned(slot, k, v)
                host[k].copy_(v)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            start.record()
            for k, h in host.items():
                self.batch[k] = h.to(device=self.device, non_blocking=True)
            end.record()
        self._done[slot] = end
        self.copy_events = (start, end)

    def next(self):
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        batch = self.batch
        if batch is not None:
            for k, v in batch.items():
                if k != "meta" and torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)        # allocated on the copy stream, consumed on this one
        self.preload()
        return batch

    def __iter__(self):
        while True:
            b = self.next()
            if b is None:
                return
            yield b
