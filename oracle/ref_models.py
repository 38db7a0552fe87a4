"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in stock fp32 `torch.nn` ops, of the reference networks on the
hot path.  Each class cites the reference lines it follows; `state_dict` keys
and shapes are identical to the reference's, which is what
`oracle/make_golden.py` relies on to load the same deterministic weights into
both and compare.  Pinned against the real reference by
`tests/test_oracle_golden.py` (fixtures under `tests/golden/`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# Generator / Discriminator  (Model/HdGan.py:49-145 == Model/CycleGan.py:6-103)
# --------------------------------------------------------------------------
class ResidualBlock(nn.Module):
    """x + IN(conv3(rpad(relu(IN(conv3(rpad(x)))))))  -- Model/HdGan.py:49-63."""

    def __init__(self, in_features: int):
        super().__init__()
        c = in_features
        # indices 1 and 5 hold the convs => keys conv_block.{1,5}.{weight,bias}
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(c, c, 3), nn.InstanceNorm2d(c), nn.ReLU(inplace=True),
            nn.ReflectionPad2d(1), nn.Conv2d(c, c, 3), nn.InstanceNorm2d(c))

    def forward(self, x):
        return x + self.conv_block(x)


class Generator(nn.Module):
    """9-block ResNet generator -- Model/HdGan.py:65-113."""

    def __init__(self, input_nc: int, output_nc: int, n_residual_blocks: int = 9):
        super().__init__()
        head = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, 64, 7), nn.InstanceNorm2d(64), nn.ReLU(inplace=True)]
        ch = 64
        for _ in range(2):  # two stride-2 zero-padded downsamplers (HdGan.py:77-82)
            head += [nn.Conv2d(ch, ch * 2, 3, stride=2, padding=1), nn.InstanceNorm2d(ch * 2), nn.ReLU(inplace=True)]
            ch *= 2
        body = [ResidualBlock(ch) for _ in range(n_residual_blocks)]
        tail = []
        for _ in range(2):  # two transposed-conv upsamplers (HdGan.py:92-97)
            tail += [nn.ConvTranspose2d(ch, ch // 2, 3, stride=2, padding=1, output_padding=1),
                     nn.InstanceNorm2d(ch // 2), nn.ReLU(inplace=True)]
            ch //= 2
        tail += [nn.ReflectionPad2d(3), nn.Conv2d(64, output_nc, 7), nn.Tanh()]
        self.model_head = nn.Sequential(*head)
        self.model_body = nn.Sequential(*body)
        self.model_tail = nn.Sequential(*tail)

    def forward(self, x):
        return self.model_tail(self.model_body(self.model_head(x)))


def _patch_stack(input_nc: int, ndf: int, n_layers: int, norm):
    """The 4x4 PatchGAN conv stack as a list of per-layer lists.

    Model/HdGan.py:154-175 (NLayerDiscriminator) and :120-136 (Discriminator)
    build the same thing for ndf=64, n_layers=3.
    """
    pad = 1  # int(ceil((4-1)/4)), HdGan.py:155
    seq = [[nn.Conv2d(input_nc, ndf, 4, stride=2, padding=pad), nn.LeakyReLU(0.2, True)]]
    nf = ndf
    for _ in range(1, n_layers):
        prev, nf = nf, min(nf * 2, 512)
        seq.append([nn.Conv2d(prev, nf, 4, stride=2, padding=pad), norm(nf), nn.LeakyReLU(0.2, True)])
    prev, nf = nf, min(nf * 2, 512)
    seq.append([nn.Conv2d(prev, nf, 4, stride=1, padding=pad), norm(nf), nn.LeakyReLU(0.2, True)])
    seq.append([nn.Conv2d(nf, 1, 4, stride=1, padding=pad)])
    return seq


class Discriminator(nn.Module):
    """PatchGAN + global average pool -> (B, 1)  -- Model/HdGan.py:115-145."""

    def __init__(self, input_nc: int):
        super().__init__()
        layers = [m for group in _patch_stack(input_nc, 64, 3, nn.InstanceNorm2d) for m in group]
        self.model = nn.Sequential(*layers)  # convs land on indices 0,2,5,8,11

    def forward(self, x):
        x = self.model(x)
        return F.avg_pool2d(x, x.size()[2:]).view(x.size(0), -1)


class NLayerDiscriminator(nn.Module):
    """Model/HdGan.py:148-205."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False,
                 getIntermFeat=False):
        super().__init__()
        self.getIntermFeat = getIntermFeat
        self.n_layers = n_layers
        seq = _patch_stack(input_nc, ndf, n_layers, norm_layer)
        if use_sigmoid:
            seq.append([nn.Sigmoid()])
        if getIntermFeat:
            for n, group in enumerate(seq):
                setattr(self, "model" + str(n), nn.Sequential(*group))
        else:
            self.model = nn.Sequential(*[m for group in seq for m in group])

    def forward(self, x):
        if not self.getIntermFeat:
            return self.model(x)
        feats = []
        for n in range(self.n_layers + 2):
            x = getattr(self, "model" + str(n))(x)
            feats.append(x)
        return feats


def center_crop(img: torch.Tensor, size: int) -> torch.Tensor:
    """torchvision.transforms.functional.center_crop for tensors (crop <= image)."""
    h, w = img.shape[-2:]
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return img[..., top:top + size, left:left + size]


class Discriminator_m(nn.Module):
    """Multi-scale discriminator on centre crops -- Model/HdGan.py:207-256."""

    def __init__(self, input_nc, ndf=64, n_layers=3,
                 norm_layer=functools.partial(nn.InstanceNorm2d, affine=False),
                 use_sigmoid=False, num_D=1, getIntermFeat=True):
        super().__init__()
        self.num_D, self.n_layers, self.getIntermFeat = num_D, n_layers, getIntermFeat
        for i in range(num_D):
            net = NLayerDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid, getIntermFeat)
            if getIntermFeat:
                for j in range(n_layers + 2):
                    setattr(self, "scale%d_layer%d" % (i, j), getattr(net, "model" + str(j)))
            else:
                setattr(self, "layer" + str(i), net.model)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)  # unused (:250)

    def forward(self, x):
        out = []
        cur = x
        for i in range(self.num_D):
            s = cur.size(2)
            k = self.num_D - 1 - i  # scale index used for this input (:243)
            if self.getIntermFeat:
                feats, h = [], cur
                for j in range(self.n_layers + 2):
                    h = getattr(self, "scale%d_layer%d" % (k, j))(h)
                    feats.append(h)
                out.append(feats)
            else:
                out.append([getattr(self, "layer" + str(k))(cur)])
            if i != self.num_D - 1:
                cur = center_crop(cur, int(s / 2))  # :251
        return out


class GANLoss(nn.Module):
    """LSGAN loss on the globally pooled last feature map -- Model/HdGan.py:258-293."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=torch.Tensor):
        super().__init__()
        self.target_real = tensor(1, 1).fill_(1.0)
        self.target_fake = tensor(1, 1).fill_(0.0)
        self.loss = nn.MSELoss() if use_lsgan else nn.BCELoss()

    def __call__(self, inp, target_is_real):
        tgt = self.target_real if target_is_real else self.target_fake

        def pooled(x):
            return F.avg_pool2d(x, x.size()[2:]).view(x.size(0), -1)

        if isinstance(inp[0], list):
            w = [1.8, 0.2]  # :273
            total = 0
            for i, feats in enumerate(inp):
                total = total + self.loss(pooled(feats[-1]), tgt) * w[i]
            return total
        return self.loss(pooled(inp[-1]), tgt)


# --------------------------------------------------------------------------
# Registration U-Net (trainer/layers.py:71-300, trainer/reg.py:31-132)
# --------------------------------------------------------------------------
_IN = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)  # layers.py:14


def _kaiming(w, a):
    nn.init.kaiming_normal_(w, a=a, nonlinearity="leaky_relu" if a else "relu", mode="fan_in")


class ResnetBlock(nn.Module):
    """reflect-pad resnet block -- trainer/layers.py:243-300 (padding_type='reflect')."""

    def __init__(self, dim: int):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0, bias=True), _IN(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, padding=0, bias=True), _IN(dim))

    def forward(self, x):
        return x + self.conv_block(x)


class ResnetTransformer(nn.Module):
    """n resnet blocks, kaiming(relu) weights, zero bias -- trainer/layers.py:216-240."""

    def __init__(self, dim: int, n_blocks: int):
        super().__init__()
        self.model = nn.Sequential(*[ResnetBlock(dim) for _ in range(n_blocks)])
        for m in self.model.modules():
            if isinstance(m, nn.Conv2d):
                _kaiming(m.weight, 0.0)
                m.bias.data.zero_()

    def forward(self, x):
        return self.model(x)


class Conv(nn.Module):
    """conv -> (leaky_relu|none) -> (resnet block)  -- trainer/layers.py:71-104 (use_norm is always False in Reg)."""

    def __init__(self, cin, cout, k, stride, pad, activation="leaky_relu", use_resnet=False, init="kaiming"):
        super().__init__()
        self.conv2d = nn.Conv2d(cin, cout, k, stride, pad, bias=True)
        self.resnet_block = ResnetTransformer(cout, 1) if use_resnet else None
        self.slope = 0.2 if activation == "leaky_relu" else None
        if init == "zeros":
            nn.init.normal_(self.conv2d.weight, mean=0.0, std=1e-5)  # layers.py:44-45
        else:
            a = 0.2 if activation == "leaky_relu" else 0.0
            nn.init.kaiming_normal_(self.conv2d.weight, a=a,
                                    nonlinearity=activation if activation else "relu", mode="fan_in")
        self.conv2d.bias.data.zero_()

    def forward(self, x):
        x = self.conv2d(x)
        if self.slope is not None:
            x = F.leaky_relu(x, self.slope)
        if self.resnet_block is not None:
            x = self.resnet_block(x)
        return x


class DownBlock(nn.Module):
    """Conv(+resblock) then 2x2 max-pool; returns (pooled, skip) -- trainer/layers.py:156-183."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv_0 = Conv(cin, cout, 3, 1, 1, use_resnet=True)

    def forward(self, x):
        skip = self.conv_0(x)
        return F.max_pool2d(skip, 2), skip


class ResUnet(nn.Module):
    """cfg 'A' of trainer/reg.py:15-99."""

    NDF = [32, 64, 64, 64, 64, 64, 64]
    NUF = [64, 64, 64, 64, 64, 64, 32]

    def __init__(self, nc_a, nc_b):
        super().__init__()
        cin = nc_a + nc_b
        for i, c in enumerate(self.NDF, start=1):
            setattr(self, "down_%d" % i, DownBlock(cin, c))
            cin = c
        self.c1 = Conv(cin, 2 * cin, 1, 1, 0)
        self.t = ResnetTransformer(2 * cin, 3)
        self.c2 = Conv(2 * cin, cin, 1, 1, 0)
        n = len(self.NDF)
        for i, c in zip(range(n, 0, -1), self.NUF):
            setattr(self, "up_%d" % i, Conv(cin + self.NDF[i - 1], c, 3, 1, 1))
            cin = c
        self.refine = nn.Sequential(ResnetTransformer(cin, 1), Conv(cin, cin, 1, 1, 0))
        self.output = Conv(cin, 2, 3, 1, 1, activation=None, init="zeros")

    def forward(self, a, b):
        x = torch.cat([a, b], 1)
        skips = []
        n = len(self.NDF)
        for i in range(1, n + 1):
            x, s = getattr(self, "down_%d" % i)(x)
            skips.append(s)
        x = self.c2(self.t(self.c1(x)))
        for i in range(n, 0, -1):
            s = skips[i - 1]
            x = F.interpolate(x, (s.size(2), s.size(3)), mode="bilinear")  # align_corners=False (reg.py:93)
            x = getattr(self, "up_%d" % i)(torch.cat([x, s], 1))
        return self.output(self.refine(x))


class Reg(nn.Module):
    """trainer/reg.py:101-132: forward(img_a, img_b) -> 2-channel displacement field."""

    def __init__(self, height, width, in_channels_a, in_channels_b):
        super().__init__()
        self.oh, self.ow = height, width
        self.offset_map = ResUnet(in_channels_a, in_channels_b)

    def forward(self, img_a, img_b):
        return self.offset_map(img_a, img_b)


class Transformer_2D(nn.Module):
    """Dense warp by a pixel displacement field -- trainer/transformer.py:11-31 (minus the hard .cuda())."""

    def forward(self, src, flow):
        b, _, h, w = flow.shape
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=flow.device),
                                torch.arange(w, dtype=torch.float32, device=flow.device), indexing="ij")
        ny = 2.0 * ((gy + flow[:, 0]) / (h - 1) - 0.5)  # channel 0 moves rows (:24)
        nx = 2.0 * ((gx + flow[:, 1]) / (w - 1) - 0.5)  # channel 1 moves columns
        grid = torch.stack([nx, ny], dim=-1)  # grid_sample wants (x, y) (:27)
        return F.grid_sample(src, grid, align_corners=True, padding_mode="border")


def smooothing_loss(y_pred):
    """mean(dx^2) + mean(dy^2) of forward differences -- trainer/utils.py:165-173."""
    dy = y_pred[:, :, 1:, :] - y_pred[:, :, :-1, :]
    dx = y_pred[:, :, :, 1:] - y_pred[:, :, :, :-1]
    return torch.mean(dx * dx) + torch.mean(dy * dy)
