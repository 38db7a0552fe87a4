"""GPU: every HIP kernel family against a plain fp32 torch CPU reference of the same op, through the C ABI.

Tolerances:
  fp32 path (exact-f32 MFMA, fp32 storage):  2e-4 of the reference tensor's max |value| (summation order differs from
      oneDNN's), 4e-4 for gradients
  bf16 path (bf16 storage + MFMA, fp32 accumulate): rel-L2 5e-3, forward AND gradients, against an fp32 reference that
      sees what the kernels see: inputs, weights and incoming gradients rounded to bf16, and every tensor the HIP path
      STORES in bf16 (conv outputs, normalised activations, the gradients flowing back through them) rounded at the same
      point (`_Q`).  What is left is accumulation order, the 1-ulp flips it causes at rounding ties, and activation
      masks of pre-activations within rounding of 0 -- a bug confined to a tile edge or to one k-chunk does not fit
      under that bound (the bound was 4e-2 / 8e-2 against an unrounded reference in round 1).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 5e-3}
GTOL = {torch.float32: 4e-4, torch.bfloat16: 5e-3}


class _Q(torch.autograd.Function):
    """A tensor the bf16 path keeps in bf16: value rounded on the way forward, its gradient rounded on the way back."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _q(dtype):
    return _Q.apply if dtype == torch.bfloat16 else (lambda t: t)


def _rnd(t, dtype):
    return t.to(dtype).float()


def _in_ref(y, q):
    """InstanceNorm as the HIP path evaluates it: statistics of the conv's fp32 accumulators (the fused epilogue moments),
    applied to the STORED (rounded) tensor.  With q = identity this is F.instance_norm(y, eps=1e-5).  Matching the
    statistics matters for the gradients: an activation mask that differs on a fraction f of the elements is a rel-L2
    error of sqrt(f), and statistics of the rounded tensor move xhat by ~1e-4, i.e. f ~ 1e-4."""
    mu = y.mean((2, 3), keepdim=True)
    var = y.var((2, 3), unbiased=False, keepdim=True)
    return (q(y) - mu) * torch.rsqrt(var + 1e-5)


def _rel(got, want, l2=False):
    """max |diff| / max |want| (fp32 checks) or rel-L2 (bf16 checks: single flipped activation masks dominate a max)."""
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    if l2:
        return float((got - want).norm() / want.norm().clamp_min(1e-20))
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-20))


def _db_err(got, want, gout):
    """Bias gradient = a plain sum over batch and pixels: its rounding error scales with the norm of the summed gradient,
    not with the (possibly cancelling) sum itself: max |diff| relative to the per-channel L2 norm of the gradient."""
    scale = gout.float().pow(2).sum((0, 2, 3)).sqrt().clamp_min(1e-20)
    return float(((got.detach().float().cpu() - want.detach().float().cpu()).abs() / scale).max())


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cta_gan_amd import _lib
    _lib.load()  # fail loudly if the HIP library is absent
    return torch.device("cuda:0")


def _make_probe(spec, with_norm_act=None):
    from cta_gan_amd import engine as E
    from cta_gan_amd import nets

    class Probe(nets.HipNet):
        def __init__(self):
            super().__init__()
            wshape = (spec.cin, spec.cout, spec.k, spec.k) if spec.transposed else (spec.cout, spec.cin, spec.k, spec.k)
            self.slot = nets._Slot(wshape, (spec.cout,))
            torch.manual_seed(5)
            torch.nn.init.normal_(self.slot.weight, std=(2.0 / (spec.cin * spec.k * spec.k)) ** 0.5)
            torch.nn.init.uniform_(self.slot.bias, -0.5, 0.5)

        def forward(self, x):
            return self._call(x)[0]

        def _run(self, tape, inputs, need_in):
            (x,) = inputs
            dt = self.dtype_
            if spec.cin <= 2:
                b, c, h, w = x.shape
                xa = E.Act(torch.zeros(1, device=x.device).expand(b, h, w, c), req=need_in[0])
                srcs = (nets._img_plane(x, 0), nets._img_plane(x, 1) if c == 2 else None)
                y = E.conv_forward(tape, self._cache, spec, xa, self.slot.weight, self.slot.bias, dt, img_sources=srcs)
            else:
                xa = E.Act(nets._to_nhwc(x, dt), req=need_in[0])
                y = E.conv_forward(tape, self._cache, spec, xa, self.slot.weight, self.slot.bias, dt)
            if with_norm_act is not None:
                y = E.inorm_forward(tape, y, with_norm_act)

            def finish(in_acts):
                g, _ = E.take_grad(in_acts[0])
                return [None if g is None else nets._to_nchw_view(g).float()]
            return [y], [xa], finish
    return Probe()


def _ref_conv(spec, x, w, b, act, norm_act=None, q=lambda t: t):
    """`q`: storage rounding of the compute dtype (identity for fp32), applied where the HIP path stores a tensor."""
    from cta_gan_amd.engine import ACT_LRELU, ACT_RELU, ACT_TANH
    if spec.transposed:
        y = F.conv_transpose2d(x, w, b if spec.use_bias else None, stride=2, padding=spec.pad, output_padding=1)
    else:
        xp = F.pad(x, (spec.pad,) * 4, mode="reflect") if spec.reflect else x
        y = F.conv2d(xp, w, b if spec.use_bias else None, stride=spec.stride, padding=0 if spec.reflect else spec.pad)

    def a(t, code):
        if code == ACT_RELU:
            return F.relu(t)
        if code == ACT_LRELU:
            return F.leaky_relu(t, 0.2)
        if code == ACT_TANH:
            return torch.tanh(t)
        return t
    y = a(y, act)
    if norm_act is not None:
        return q(a(_in_ref(y, q), norm_act))
    return y if spec.out_f32 else q(y)


def _conv_specs():
    from cta_gan_amd.engine import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, ConvSpec
    return {
        "res3x3_reflect_64": (ConvSpec(64, 64, 3, 1, 1, reflect=True, use_bias=True), (2, 64, 12, 20), None),
        "res3x3_reflect_256": (ConvSpec(256, 256, 3, 1, 1, reflect=True, use_bias=False), (1, 256, 9, 11), ACT_RELU),
        "down3x3_s2": (ConvSpec(64, 128, 3, 2, 1, use_bias=True), (2, 64, 16, 24), None),
        "up_convT": (ConvSpec(128, 64, 3, 2, 1, transposed=True, use_bias=True), (2, 128, 7, 9), None),
        "d_4x4_s2": (ConvSpec(64, 128, 4, 2, 1, use_bias=False), (2, 64, 16, 16), ACT_LRELU),
        "d_4x4_s1": (ConvSpec(256, 512, 4, 1, 1, use_bias=False), (1, 256, 9, 9), ACT_LRELU),
        "d_last_512to1": (ConvSpec(512, 1, 4, 1, 1, use_bias=True, out_f32=True), (2, 512, 7, 7), None),
        "g_tail_7x7_tanh": (ConvSpec(64, 1, 7, 1, 3, reflect=True, use_bias=True, act=ACT_TANH, out_f32=True),
                            (2, 64, 12, 16), None),
        "g_head_7x7_cin1": (ConvSpec(1, 64, 7, 1, 3, reflect=True, use_bias=True), (2, 1, 20, 24), None),
        "d_first_cin1_lrelu": (ConvSpec(1, 64, 4, 2, 1, use_bias=True, act=ACT_LRELU), (2, 1, 16, 24), None),
        "reg_first_cin2_lrelu": (ConvSpec(2, 32, 3, 1, 1, use_bias=True, act=ACT_LRELU), (2, 2, 12, 16), None),
        "reg_3x3_lrelu_32": (ConvSpec(32, 32, 3, 1, 1, use_bias=True, act=ACT_LRELU), (2, 32, 10, 14), None),
        "reg_up_96to32": (ConvSpec(96, 32, 3, 1, 1, use_bias=True, act=ACT_LRELU), (1, 96, 8, 8), None),
        "reg_1x1_64to128": (ConvSpec(64, 128, 1, 1, 0, use_bias=True, act=ACT_LRELU), (2, 64, 4, 4), None),
        "reg_out_32to2": (ConvSpec(32, 2, 3, 1, 1, use_bias=True, out_f32=True), (2, 32, 12, 12), None),
        "big_tile_partial": (ConvSpec(128, 256, 3, 1, 1, use_bias=True), (1, 128, 23, 19), None),
        # >= 16x16 outputs, stride 1, full tap window: the halo-resident kernel (conv_halo.h), incl. ragged tiles
        "halo_tail_7x7_tanh": (ConvSpec(64, 1, 7, 1, 3, reflect=True, use_bias=True, act=ACT_TANH, out_f32=True),
                               (2, 64, 32, 48), None),
        "halo_reg_3x3_lrelu_32_ragged": (ConvSpec(32, 32, 3, 1, 1, use_bias=True, act=ACT_LRELU), (2, 32, 40, 24), None),
        "halo_reflect_64_ragged": (ConvSpec(64, 64, 3, 1, 1, reflect=True, use_bias=True), (1, 64, 33, 17), None),
        "halo_reg_up_96to32": (ConvSpec(96, 32, 3, 1, 1, use_bias=True, act=ACT_LRELU), (1, 96, 32, 32), None),
        "halo_d_4x4_s1_256to512": (ConvSpec(256, 512, 4, 1, 1, use_bias=False), (1, 256, 20, 20), ACT_LRELU),
        "halo_d_last_512to1": (ConvSpec(512, 1, 4, 1, 1, use_bias=True, out_f32=True), (2, 512, 19, 19), None),
        "halo_reg_out_32to2": (ConvSpec(32, 2, 3, 1, 1, use_bias=True, out_f32=True), (2, 32, 32, 16), None),
        "halo_128to256": (ConvSpec(128, 256, 3, 1, 1, use_bias=True), (1, 128, 23, 19), None),
        "halo_up_convT_classes": (ConvSpec(128, 64, 3, 2, 1, transposed=True, use_bias=True), (1, 128, 24, 20), None),
        "halo_down_s2_bwd_classes": (ConvSpec(64, 128, 3, 2, 1, use_bias=True), (1, 64, 48, 40), None),
        # 64 -> 1 tail (conv_tail.hip): tiles of 20 x 32 outputs, ragged in both directions, several tiles
        "tail7_ragged": (ConvSpec(64, 1, 7, 1, 3, reflect=True, use_bias=True, act=ACT_TANH, out_f32=True),
                         (2, 64, 50, 70), None),
        # reflect convs on 16-aligned maps: backward-data = tile-aligned interior (halo kernel) + 1-pixel frame (gather)
        "frame_reflect_64": (ConvSpec(64, 64, 3, 1, 1, reflect=True, use_bias=True), (2, 64, 32, 48), None),
        "frame_reflect_256": (ConvSpec(256, 256, 3, 1, 1, reflect=True, use_bias=True), (1, 256, 48, 32), None),
        # (padded grid 146 x 162: every edge of the frame launch spans two 128-slot tiles, the second one mostly pad slots)
        "frame_reflect_64_long_edges": (ConvSpec(64, 64, 3, 1, 1, reflect=True, use_bias=True), (1, 64, 144, 160), None),
        # first layers straight from image planes (conv_small.hip): several / ragged 16x16 tiles, fused IN moments
        "small_head_7x7_cin1_in_relu": (ConvSpec(1, 64, 7, 1, 3, reflect=True, use_bias=False), (2, 1, 40, 56), ACT_RELU),
        "small_reg_first_cin2_in_lrelu": (ConvSpec(2, 32, 3, 1, 1, use_bias=False), (2, 2, 36, 20), ACT_LRELU),
        "small_d_first_cin1_s2": (ConvSpec(1, 64, 4, 2, 1, use_bias=True, act=ACT_LRELU), (2, 1, 70, 66), None),
        "small_head_cin2": (ConvSpec(2, 64, 5, 1, 2, reflect=True, use_bias=True), (1, 2, 33, 47), None),
        # >= 4096 output pixels and Cout > 64: the 256x128 / 8-wave / 3-stage-ring configuration (bf16)
        "ring_res3x3_reflect_256": (ConvSpec(256, 256, 3, 1, 1, reflect=True, use_bias=True), (2, 256, 64, 64), None),
        "ring_d_4x4_s2_tail": (ConvSpec(128, 256, 4, 2, 1, use_bias=True), (1, 128, 130, 134), None),
        # stride-2 convs with >= 16 x 16 outputs: polyphase slices on the halo-resident kernel (ConvArgs::s2d) -- the PatchGAN's
        # 4x4 layers in bf16 (even and odd input sizes, InstanceNorm moments from the epilogue), every stride-2 conv in bf16x3
        "s2d_d_4x4_64to128_in_lrelu": (ConvSpec(64, 128, 4, 2, 1, use_bias=False), (2, 64, 70, 66), ACT_LRELU),
        "s2d_d_4x4_128to256_odd": (ConvSpec(128, 256, 4, 2, 1, use_bias=False), (1, 128, 37, 51), ACT_LRELU),
        "s2d_down3x3_64to128": (ConvSpec(64, 128, 3, 2, 1, use_bias=False), (2, 64, 48, 40), ACT_RELU),
        "s2d_down3x3_128to256_odd": (ConvSpec(128, 256, 3, 2, 1, use_bias=True), (1, 128, 45, 33), None),
    }


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("name", list(_conv_specs()))
def test_conv_family(name, dtype, dev):
    spec, shape, norm_act = _conv_specs()[name]
    probe = _make_probe(spec, norm_act).to(dev)
    probe.compute_dtype = dtype
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    need_dx = True
    xg = x.to(dev).requires_grad_(need_dx)
    y = probe(xg)
    gout = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(gout.to(dev).to(y.dtype))
    w = _rnd(probe.slot.weight.detach().cpu(), dtype).requires_grad_(True)
    b = probe.slot.bias.detach().cpu().clone().requires_grad_(True)
    xr = _rnd(x, dtype).requires_grad_(True)
    yr = _ref_conv(spec, xr, w, b, spec.act, norm_act, _q(dtype))
    yr.backward(_rnd(gout, y.dtype))
    tol, gtol = TOL[dtype], GTOL[dtype]
    l2 = dtype == torch.bfloat16
    errs = {"fwd": _rel(y, yr, l2), "dx": _rel(xg.grad, xr.grad, l2), "dw": _rel(probe.slot.weight.grad, w.grad, l2)}
    if spec.use_bias and norm_act is None:
        errs["db"] = _db_err(probe.slot.bias.grad, b.grad, gout)
    print(name, str(dtype), {k: "%.2e" % v for k, v in errs.items()})
    assert errs["fwd"] < tol, ("fwd", errs)
    assert all(v < gtol for k, v in errs.items() if k != "fwd"), errs


def _run_probe_case(spec, shape, norm_act, dtype, dev, seed):
    probe = _make_probe(spec, norm_act).to(dev)
    probe.compute_dtype = dtype
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    xg = x.to(dev).requires_grad_(True)
    y = probe(xg)
    gout = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(gout.to(dev).to(y.dtype))
    w = _rnd(probe.slot.weight.detach().cpu(), dtype).requires_grad_(True)
    b = probe.slot.bias.detach().cpu().clone().requires_grad_(True)
    xr = _rnd(x, dtype).requires_grad_(True)
    yr = _ref_conv(spec, xr, w, b, spec.act, norm_act, _q(dtype))
    yr.backward(_rnd(gout, y.dtype))
    tol, gtol, l2 = TOL[dtype], GTOL[dtype], dtype == torch.bfloat16
    assert _rel(y, yr, l2) < tol, ("fwd", shape, _rel(y, yr, l2))
    assert _rel(xg.grad, xr.grad, l2) < gtol, ("input grad", shape, _rel(xg.grad, xr.grad, l2))
    assert _rel(probe.slot.weight.grad, w.grad, l2) < gtol, ("weight grad", shape, _rel(probe.slot.weight.grad, w.grad, l2))
    if spec.use_bias and norm_act is None:
        assert _db_err(probe.slot.bias.grad, b.grad, gout) < gtol, ("bias grad", shape)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_first_and_last_layer_kernels_random_shapes(dtype, dev):
    """conv_small / corr_small / conv_tail on a sweep of awkward shapes (tiles of 16x16 resp. 20x32 outputs: below one
    tile, one pixel over, prime sizes), batch 1-3, against stock torch."""
    from cta_gan_amd.engine import ACT_LRELU, ACT_TANH, ConvSpec
    rng = np.random.default_rng(2024)
    families = [
        lambda: ConvSpec(1, 64, 7, 1, 3, reflect=True, use_bias=True),                                   # G head
        lambda: ConvSpec(64, 1, 7, 1, 3, reflect=True, use_bias=True, act=ACT_TANH, out_f32=True),       # G tail
        lambda: ConvSpec(2, 32, 3, 1, 1, use_bias=True, act=ACT_LRELU),                                  # Reg first
        lambda: ConvSpec(1, 64, 4, 2, 1, use_bias=True, act=ACT_LRELU),                                  # D first
    ]
    sizes = [(4, 4), (7, 9), (16, 16), (17, 33), (21, 31), (40, 65), (53, 19)]
    for fi, fam in enumerate(families):
        for (h, w) in sizes:
            spec = fam()
            if spec.stride == 2 and (h < 4 or w < 4):
                continue
            bsz = int(rng.integers(1, 4))
            _run_probe_case(spec, (bsz, spec.cin, h, w), None, dtype, dev, seed=100 * fi + h + w)


def test_instance_norm_residual_and_fold(dev):
    """ResidualBlock-like composite incl. the padded-grid gradient fold, checked on odd sizes."""
    from cta_gan_amd.Model.HdGan import ResidualBlock
    from oracle import ref_models
    from cta_gan_amd import synth
    for ch, hh, ww in [(32, 5, 7), (64, 9, 4), (128, 2, 3)]:
        hip = synth.fill_module(ResidualBlock(ch), seed=8).to(dev)
        ref = synth.fill_module(ref_models.ResidualBlock(ch), seed=8)
        rng = np.random.default_rng(ch)
        x = torch.from_numpy(rng.standard_normal((2, ch, hh, ww)).astype(np.float32))
        g = torch.from_numpy(rng.standard_normal((2, ch, hh, ww)).astype(np.float32))
        xh = x.to(dev).requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        oh = hip(xh); oh.backward(g.to(dev))
        orf = ref(xr); orf.backward(g)
        assert _rel(oh, orf) < 3e-4
        assert _rel(xh.grad, xr.grad) < 1e-3
        for k, p in ref.named_parameters():
            if k.endswith("weight"):
                assert _rel(dict(hip.named_parameters())[k].grad, p.grad) < 1e-3, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 256, 40, 24), (2, 64, 33, 17), (1, 32, 128, 64), (5, 8, 16, 16), (2, 128, 7, 5)],
                         ids=["256", "64odd", "32", "8", "128tiny"])
def test_instance_norm_finalize_fused_into_elementwise_kernels(shape, dtype, dev, monkeypatch):
    """ctg_in_apply_part / ctg_in_bwd_stats (finalize in the kernel prologue; round-3 experiment, opt-in with CTG_FIN_FUSE)
    against (a) the separate finalize launch + plain elementwise kernels (the same partials: equal up to the last float bit
    of mean / rstd) and (b) stock torch."""
    from cta_gan_amd import ops
    b, c, h, w = shape
    rng = np.random.default_rng(c * h + w)
    x = torch.from_numpy((rng.standard_normal((b, h, w, c)) * 1.7 + 0.3).astype(np.float32)).to(dev).to(dtype)
    r = torch.from_numpy(rng.standard_normal((b, h, w, c)).astype(np.float32)).to(dev).to(dtype)
    g = torch.from_numpy(rng.standard_normal((b, h, w, c)).astype(np.float32)).to(dev).to(dtype)
    for act, res in ((ops.ACT_RELU, None), (ops.ACT_NONE, r), (ops.ACT_LRELU, None)):
        part, ns = ops.in_partial(x)
        assert ns <= ops.FUSED_MAX_SLABS
        o1 = torch.empty_like(x)
        mean1, rstd1 = ops.in_apply_part(x, part, act, res, o1)
        mean0, rstd0 = ops.in_finalize(part, ns, h * w)
        o0 = torch.empty_like(x)
        ops.in_apply(x, mean0, rstd0, act, res, o0)
        torch.cuda.synchronize()
        assert torch.allclose(mean1, mean0, rtol=1e-6, atol=1e-7) and torch.allclose(rstd1, rstd0, rtol=1e-6, atol=0)
        assert _rel(o1, o0) < (1e-6 if dtype == torch.float32 else 8e-3)      # bf16: a 1-ulp flip at a rounding tie
        xf = x.float().permute(0, 3, 1, 2)
        ref = F.instance_norm(xf, eps=1e-5)
        ref = F.relu(ref) if act == ops.ACT_RELU else F.leaky_relu(ref, 0.2) if act == ops.ACT_LRELU else ref
        if res is not None:
            ref = ref + res.float().permute(0, 3, 1, 2)
        assert _rel(o1.permute(0, 3, 1, 2), ref, l2=True) < (1e-5 if dtype == torch.float32 else 4e-3)
        # backward: in_bwd = statistics pass + fused elementwise pass; CTG-independent check against autograd
        d0 = torch.empty_like(x)
        ops.in_bwd(x, g, 0, mean1, rstd1, act, d0)          # statistics pass, finalize launch, elementwise pass
        monkeypatch.setattr(ops, "_FIN_FUSE", True)
        d1 = torch.empty_like(x)
        ops.in_bwd(x, g, 0, mean1, rstd1, act, d1)          # statistics pass, ONE fused launch
        monkeypatch.setattr(ops, "_FIN_FUSE", False)
        assert _rel(d1, d0) < (1e-6 if dtype == torch.float32 else 8e-3)
        xa = xf.clone().requires_grad_(True)
        ya = F.instance_norm(xa, eps=1e-5)
        ya = F.relu(ya) if act == ops.ACT_RELU else F.leaky_relu(ya, 0.2) if act == ops.ACT_LRELU else ya
        ya.backward(g.float().permute(0, 3, 1, 2))
        assert _rel(d1.permute(0, 3, 1, 2), xa.grad, l2=True) < (2e-5 if dtype == torch.float32 else 6e-3)
    # many slabs (a conv epilogue's per-tile partials of a large map): the finalize launch stays; same entry point
    if c >= 32:
        ns_big = 200
        part = (torch.from_numpy(rng.standard_normal((b, ns_big, c, 2)).astype(np.float32)).to(dev))
        part[..., 1] = part[..., 1].abs() * 4
        mean, rstd = ops.in_finalize(part, ns_big, h * w)
        assert not ops.fin_fusable(ns_big)
        d_sep = torch.empty_like(x)
        ops.in_bwd_stats(x, g, mean, rstd, ops.ACT_RELU, d_sep, part)
        s = part.double().sum(1) / (h * w)
        xf32 = x.float()
        xh = (xf32 - mean[:, None, None, :]) * rstd[:, None, None, :]
        gg = torch.where(xh > 0, g.float(), torch.zeros_like(xh))
        want = rstd[:, None, None, :] * (gg - s[:, None, None, :, 0].float() - xh * s[:, None, None, :, 1].float())
        assert _rel(d_sep, want, l2=True) < (1e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_maxpool_bilinear_concat(dtype, dev):
    from cta_gan_amd import ops
    rng = np.random.default_rng(3)
    b, h, w, c = 2, 8, 12, 32
    x = torch.from_numpy(rng.standard_normal((b, c, h, w)).astype(np.float32))
    xq = x.to(dtype).float()  # what the kernel really sees
    xn = xq.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype)
    # max pool fwd / bwd
    out = torch.empty((b, h // 2, w // 2, c), dtype=dtype, device=dev)
    ops.maxpool2_fwd(xn, out)
    xr = xq.clone().requires_grad_(True)
    pr = F.max_pool2d(xr, 2)
    assert _rel(out.permute(0, 3, 1, 2), pr) < 1e-6
    g = torch.from_numpy(rng.standard_normal(tuple(pr.shape)).astype(np.float32)).to(dtype).float()
    pr.backward(g)
    dx = torch.empty_like(xn)
    ops.maxpool2_bwd(xn, g.permute(0, 2, 3, 1).contiguous().to(dev).to(dtype), dx, False)
    assert _rel(dx.permute(0, 3, 1, 2), xr.grad) < 1e-6
    # bilinear x2 fwd / bwd into a concat buffer slice
    buf = torch.zeros((b, 2 * h, 2 * w, c + 32), dtype=dtype, device=dev)
    ops.bilinear_fwd(xn, buf[..., :c])
    xr2 = xq.clone().requires_grad_(True)
    ur = F.interpolate(xr2, (2 * h, 2 * w), mode="bilinear")
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert _rel(buf[..., :c].permute(0, 3, 1, 2), ur) < tol
    assert float(buf[..., c:].abs().max()) == 0.0
    g2 = torch.from_numpy(rng.standard_normal(tuple(ur.shape)).astype(np.float32)).to(dtype).float()
    ur.backward(g2)
    gbuf = torch.zeros_like(buf)
    gbuf[..., :c] = g2.permute(0, 2, 3, 1).to(dev).to(dtype)
    dxx = torch.empty_like(xn)
    ops.bilinear_bwd(gbuf[..., :c], dxx)
    assert _rel(dxx.permute(0, 3, 1, 2), xr2.grad) < tol


def test_stn_smooth_l1_avgpool_against_torch(dev):
    from cta_gan_amd import nets, synth
    from oracle import ref_models
    size = 40
    src = synth.synth_smooth_images("t_src", 2, size)
    flow = 2.5 * synth.synth_images("t_flow", 2, size, channels=2)
    flow[0, :, :3] = 30.0   # drive some samples out of the image: border clamp and zero flow-gradient
    flow[1, :, -2:] = -40.0
    w = synth.synth_images("t_w", 2, size)
    sh, fh = src.to(dev).requires_grad_(True), flow.to(dev).requires_grad_(True)
    sr, fr = src.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    oh = nets.warp(sh, fh)
    orf = ref_models.Transformer_2D()(sr, fr)
    assert _rel(oh, orf) < 1e-5
    (oh * w.to(dev)).sum().backward()
    (orf * w).sum().backward()
    assert _rel(sh.grad, sr.grad) < 1e-4
    assert _rel(fh.grad, fr.grad) < 1e-4
    # channels-last flow (the layout Reg produces) gives the same result
    fcl = flow.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    assert _rel(nets.warp(src.to(dev), fcl), orf) < 1e-5
    # smoothness
    f2h = flow.to(dev).requires_grad_(True)
    f2r = flow.clone().requires_grad_(True)
    lh, lr = nets.smoothing_loss(f2h), ref_models.smooothing_loss(f2r)
    assert abs(float(lh) - float(lr)) < 1e-5 * abs(float(lr))
    (3.0 * lh).backward(); (3.0 * lr).backward()
    assert _rel(f2h.grad, f2r.grad) < 1e-4
    # L1 and masked L1
    a, b_, m = synth.synth_images("l1a", 2, size), synth.synth_images("l1b", 2, size), synth.synth_images("l1m", 2, size)
    ah = a.to(dev).requires_grad_(True)
    ar = a.clone().requires_grad_(True)
    l1h, l1r = nets.l1_loss(ah, b_.to(dev)), F.l1_loss(ar, b_)
    assert abs(float(l1h) - float(l1r)) < 1e-5
    (2.0 * l1h).backward(); (2.0 * l1r).backward()
    assert _rel(ah.grad, ar.grad) < 1e-5
    ah2 = a.to(dev).requires_grad_(True)
    ar2 = a.clone().requires_grad_(True)
    lmh = nets.masked_l1_loss(ah2, b_.to(dev), m.to(dev))
    bb = m.clone(); bb[bb < 0.3] = 0; bb[bb >= 0.3] = 1
    rb = b_ * bb; rb[rb == 0] = -1
    wm = ar2 * bb; wm[wm == 0] = -1
    lmr = F.l1_loss(wm, rb)
    assert abs(float(lmh) - float(lmr)) < 1e-5
    lmh.backward(); lmr.backward()
    assert _rel(ah2.grad, ar2.grad) < 1e-5
    # global average pool
    p = synth.synth_images("pool", 3, 31)
    ph = p.to(dev).requires_grad_(True)
    pr = p.clone().requires_grad_(True)
    oh, orf = nets.global_avgpool(ph), F.avg_pool2d(pr, pr.shape[2:]).view(3, -1)
    assert _rel(oh, orf) < 1e-5
    (oh ** 2).sum().backward(); (orf ** 2).sum().backward()
    assert _rel(ph.grad, pr.grad) < 1e-5


def test_fused_lsgan_weighted_losses_and_scalar_sum_against_torch(dev):
    """Round 3: the loss glue of the step as fused HIP reductions -- `lsgan_loss` (avg-pool + (. - target)^2 + batch mean +
    weight: GANLoss, Model/HdGan.py:276-285), `lsgan_loss_pair` (fake and real halves of one batched D pass), the loss weight
    folded into `smoothing_loss` / `l1_loss` / `masked_l1_loss`, `add_scalars`, and the Tanh backward + bias-gradient sum of
    the generator's tail -- forward values and gradients against the stock torch expressions they replace."""
    from cta_gan_amd import nets, ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, 1, 30, 31, generator=g)
    xh = x.to(dev).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    loss_h = nets.lsgan_loss(xh, 1.0, weight=1.8 * 3.0)
    loss_r = 3.0 * 1.8 * F.mse_loss(F.avg_pool2d(xr, (30, 31)).view(6, -1), torch.ones(1, 1).expand(6, 1))
    pair_h = nets.lsgan_loss_pair(xh, 2, 0.0, 1.0, weight=0.5)
    pr = F.avg_pool2d(xr, (30, 31)).view(6, -1)
    pair_r = 0.5 * ((pr[:2] ** 2).mean() + ((pr[2:] - 1.0) ** 2).mean())
    f = torch.randn(2, 2, 20, 24, generator=g)
    fh, fr = f.to(dev).requires_grad_(True), f.clone().requires_grad_(True)
    sm_h = nets.smoothing_loss(fh, weight=10.0)
    dy, dx = fr[:, :, 1:, :] - fr[:, :, :-1, :], fr[:, :, :, 1:] - fr[:, :, :, :-1]
    sm_r = 10.0 * ((dx * dx).mean() + (dy * dy).mean())
    a, b, m = (torch.rand(2, 1, 16, 16, generator=g) * 2 - 1 for _ in range(3))
    ah, ar = a.to(dev).requires_grad_(True), a.clone().requires_grad_(True)
    l1_h = nets.l1_loss(ah, b.to(dev), weight=20.0)
    l1_r = 20.0 * (ar - b).abs().mean()
    bb = (m >= 0.3).float()
    bm = b * bb
    bm = torch.where(bm == 0, -torch.ones_like(bm), bm)
    am = ar * bb
    am = torch.where(am == 0, -torch.ones_like(am), am)
    ml_h = nets.masked_l1_loss(ah, b.to(dev), m.to(dev), weight=2.0)
    ml_r = 2.0 * (am - bm).abs().mean()
    tot_h = nets.add_scalars(loss_h, pair_h, sm_h, l1_h, ml_h)
    tot_r = loss_r + pair_r + sm_r + l1_r + ml_r
    for got, want in ((loss_h, loss_r), (pair_h, pair_r), (sm_h, sm_r), (l1_h, l1_r), (ml_h, ml_r), (tot_h, tot_r)):
        assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want))), (float(got), float(want))
    tot_h.backward()
    tot_r.backward()
    assert _rel(xh.grad, xr.grad) < 1e-5 and _rel(fh.grad, fr.grad) < 1e-5 and _rel(ah.grad, ar.grad) < 1e-5
    # Tanh backward + bias sum in one pass
    y = torch.tanh(torch.randn(3, 50, 50, 1, generator=g)).to(dev)
    gy = torch.randn(3, 50, 50, 1, generator=g).to(dev)
    db = torch.full((1,), 7.0, device=dev)
    out = ops.act_bwd_sum_f32(gy, y, ops.ACT_TANH, db)
    want = gy * (1 - y * y)
    assert _rel(out, want) < 1e-6 and abs(float(db) - float(want.double().sum())) < 1e-3
    ops.act_bwd_sum_f32(gy, y, ops.ACT_TANH, db, accumulate=True)
    assert abs(float(db) - 2 * float(want.double().sum())) < 2e-3


def test_adam_matches_torch(dev):
    from cta_gan_amd import optim
    torch.manual_seed(0)
    shapes = [(64, 1, 7, 7), (64,), (256, 256, 3, 3), (1,), (5000,)] + [(7, 3)] * 40
    ps_h = [torch.nn.Parameter(torch.randn(s).to(dev)) for s in shapes]
    ps_r = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps_h]
    oh = optim.Adam(ps_h, lr=1e-4, betas=(0.5, 0.999))
    orf = torch.optim.Adam(ps_r, lr=1e-4, betas=(0.5, 0.999))
    for step in range(3):
        for ph, pr in zip(ps_h, ps_r):
            g = torch.randn(pr.shape) * (10.0 ** (step - 1))
            pr.grad = g.clone()
            ph.grad = g.to(dev)
        oh.step(); orf.step()
    for ph, pr in zip(ps_h, ps_r):
        assert _rel(ph, pr) < 1e-6


@pytest.mark.parametrize("shape", [(256, 9, 32, 32), (300, 1, 64, 32), (100, 9, 64, 64), (7, 4, 32, 32), (64, 49, 16, 64)],
                         ids=lambda s: "x".join(map(str, s)))
def test_wgrad_reduce_all_slab_counts(shape, dev):
    """ctg_wgrad_reduce: the slab-parallel variant (few elements, >= 64 slabs) and the plain one give the sum over
    slabs, honour Mreal/Nreal cropping, the destination strides and `accumulate`."""
    from cta_gan_amd import _lib
    z, nt, mc, nc = shape
    lib = _lib.load()
    part = torch.randn(z, nt, mc, nc, device=dev)
    mreal, nreal = mc - 3, nc - 5
    dst = torch.full((mreal, nreal, nt), 0.5, device=dev)          # weight layout (Cout, Cin, taps)
    want = part.double().sum(0)[:, :mreal, :nreal].permute(1, 2, 0)
    for acc in (0, 1):
        rc = lib.ctg_wgrad_reduce(part.data_ptr(), z, nt, mc, nc, dst.data_ptr(), mreal, nreal, nreal * nt, nt, 1, acc,
                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    # first call overwrote the 0.5 fill, second accumulated the same sum on top
    assert _rel(dst, 2 * want.float()) < 1e-5


def test_conv_kernels_are_bitwise_repeatable(dev):
    """Race screen: the LDS-DMA conv / weight-gradient kernels use fixed summation orders, so 12 repetitions of the same
    residual-block forward + backward (256 channels, 64x64, bf16) must agree bit for bit -- a synchronisation bug in the
    staging pipelines shows up as run-to-run differences long before it shows up against a tolerance."""
    from cta_gan_amd import synth
    from cta_gan_amd.Model.HdGan import ResidualBlock
    blk = synth.fill_module(ResidualBlock(256), seed=3).to(dev)
    blk.compute_dtype = torch.bfloat16
    x = synth.synth_images("rep_x", 2, 64).to(dev).expand(2, 256, 64, 64).contiguous()
    x = (x + torch.linspace(-1, 1, 256, device=dev).view(1, 256, 1, 1)).requires_grad_(True)
    g = torch.randn(2, 256, 64, 64, device=dev)
    ref = None
    for _ in range(12):
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        y = blk(x)
        y.backward(g.to(y.dtype))
        cur = [y.detach().float().clone(), x.grad.float().clone()] + \
              [p.grad.clone() for p in blk.parameters() if p.grad is not None]
        if ref is None:
            ref = cur
        else:
            for a, b in zip(cur, ref):
                assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 32, 48), (1, 256, 48, 32), (1, 32, 64, 32)], ids=["64", "256", "32"])
def test_residual_block_backward_fused_into_conv_epilogue(shape, dtype, dev):
    """Residual block (Model/HdGan.py:49-63) on 16-aligned maps >= 32 pixels: the backward-data pass of each reflect-padded
    3x3 conv emits the unpadded gradient (frame folded in from the interior launch's epilogue) and the first conv's also
    adds the skip gradient there (`res` / `fold` of ctg_conv_igemm) -- against the oracle's block in stock torch.
    Two chained blocks, so the second block's fused output feeds the first block's InstanceNorm backward.
    fp32: max-error 2e-4 / 4e-4 relative to the tensor's max; bf16: rel-L2 5e-3 forward, 1.2e-2 gradients against the
    blocks restated with bf16 rounding at the HIP path's storage points (`_ref_resblock`) -- four InstanceNorms deep, the
    accumulation-order differences that cross a rounding tie in one layer move the next layer's ReLU masks (single
    layers, where that cannot happen, hold 5e-3: test_conv_family)."""
    from cta_gan_amd import synth
    from cta_gan_amd.Model.HdGan import ResidualBlock
    c = shape[1]
    hip = [synth.fill_module(ResidualBlock(c), seed=60 + i).to(dev) for i in range(2)]
    for m in hip:
        m.compute_dtype = dtype
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    xg = x.to(dev).requires_grad_(True)
    y = hip[1](hip[0](xg))
    gout = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    y.float().backward(gout.to(dev))
    ws = [[_rnd(dict(m.named_parameters())[k].detach().cpu(), dtype).requires_grad_(True)
           for k in ("conv_block.1.weight", "conv_block.5.weight")] for m in hip]
    xr = _rnd(x, dtype).requires_grad_(True)
    q = _q(dtype)
    yr = _ref_resblock(_ref_resblock(xr, ws[0][0], ws[0][1], q), ws[1][0], ws[1][1], q)
    yr.backward(_rnd(gout, dtype))
    tol, gtol, l2 = TOL[dtype], GTOL[dtype], dtype == torch.bfloat16
    errs = {"fwd": _rel(y, yr, l2), "dx": _rel(xg.grad, xr.grad, l2)}
    for i in range(2):
        for j, k in enumerate(("conv_block.1.weight", "conv_block.5.weight")):
            errs["dw%d%d" % (i, j)] = _rel(dict(hip[i].named_parameters())[k].grad, ws[i][j].grad, l2)
    print(shape, str(dtype), {k: "%.2e" % v for k, v in errs.items()})
    if l2:
        gtol = 1.2e-2
    assert errs["fwd"] < tol, errs
    assert all(v < gtol for k, v in errs.items() if k != "fwd"), errs


def _ref_resblock(x, w1, w5, q):
    """x + IN(conv(rpad(relu(IN(conv(rpad(x))))))) (Model/HdGan.py:49-63; the biases cancel in the affine-free IN) with
    the storage rounding `q` after each conv and each normalisation, as the HIP path stores them."""
    h = q(F.relu(_in_ref(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w1), q)))
    return q(x + _in_ref(F.conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"), w5), q))


def test_stride2_weight_gradient_polyphase_shapes(dev):
    """Stride-2 convs / transposed convs whose weight gradient runs as one halo launch per polyphase component of the
    input (bf16, output grid >= 8 x 16): odd and even input sizes (the last input row / column is or is not read), 3x3 and
    4x4 windows, 64- and 128-wide channel tiles, batch 1-2 -- forward, input gradient and weight gradient against stock
    torch on bf16-rounded operands; rel-L2 5e-3 (bf16)."""
    from cta_gan_amd.engine import ACT_LRELU, ConvSpec
    cases = [
        (ConvSpec(64, 128, 3, 2, 1, use_bias=True), (2, 64, 33, 47)),        # odd x odd input -> 17 x 24 outputs
        (ConvSpec(64, 128, 3, 2, 1, use_bias=True), (1, 64, 32, 64)),        # even x even
        (ConvSpec(128, 256, 3, 2, 1, use_bias=True), (1, 128, 37, 32)),      # 2 x 4 channel tiles
        (ConvSpec(64, 128, 4, 2, 1, use_bias=False), (2, 64, 34, 38)),       # D's 4x4 window: four 2x2 components
        (ConvSpec(128, 256, 4, 2, 1, use_bias=True, act=ACT_LRELU), (1, 128, 41, 35)),
        (ConvSpec(128, 64, 3, 2, 1, transposed=True, use_bias=True), (2, 128, 17, 21)),   # roles swapped: G on the small grid
        (ConvSpec(256, 128, 3, 2, 1, transposed=True, use_bias=True), (1, 256, 16, 16)),
    ]
    for i, (spec, shape) in enumerate(cases):
        _run_probe_case(spec, shape, None, torch.bfloat16, dev, seed=700 + i)


@pytest.mark.parametrize("case", [(2, 128, 64, 24, 20, 3), (8, 256, 128, 112, 100, 3), (4, 128, 256, 104, 112, 4),
                                  (1, 256, 128, 128, 128, 3)],
                         ids=["convT_128to64", "convT_256to128", "s2_bwd_4x4", "convT_256to128_B1"])
def test_merged_parity_classes_equal_four_class_launches(case, dev):
    """`ctg_conv_igemm_classes` (the four parity classes of a transposed conv / stride-2 backward-data pass in ONE launch)
    writes bit for bit what four `ctg_conv_igemm` class launches write (ragged tiles included); the InstanceNorm moments agree
    after finalisation (the partials are ordered differently)."""
    from cta_gan_amd import ops
    from cta_gan_amd.engine import _convT_classes
    b, cin, cout, h, w, k = case
    g = torch.Generator().manual_seed(cin + h + k)
    x = torch.randn(b, h, w, cin, generator=g).to(dev).to(torch.bfloat16)
    wp = (torch.randn(k * k, max(cout, 128) if cout > 64 else (64 if cout > 32 else 32), cin, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    npad = wp.shape[1]
    classes = _convT_classes(k, 1)
    y4 = torch.zeros(b, 2 * h, 2 * w, cout, dtype=torch.bfloat16, device=dev)
    parts = [ops.conv_igemm(x, wp, npad, y4, None, cout, h, w, c[0], c[1], 2, 1, ops.PAD_ZERO, 0, c[2], want_stats=True)
             for c in classes]
    y1 = torch.zeros_like(y4)
    merged = ops.conv_igemm_classes(x, wp, npad, y1, None, cout, h, w, classes, ops.PAD_ZERO, 0, want_stats=True)
    # (B=1 at 128^2, the reference's shipped batch size: the small-grid 8-row tiles must not pre-empt the merged launch)
    assert merged is not None, "shape should be served by the merged launch"
    assert torch.equal(y1, y4)
    m4 = ops.in_finalize(torch.cat([p[0] for p in parts], dim=1), sum(p[1] for p in parts), 4 * h * w)
    m1 = ops.in_finalize(merged[0], merged[1], 4 * h * w)
    assert torch.allclose(m1[0], m4[0], rtol=1e-5, atol=1e-6) and torch.allclose(m1[1], m4[1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(4, 512, 512), (5, 500, 430), (9, 352, 344)], ids=["4x512x512", "5x500x430_ragged", "9x352x344"])
def test_sliding_window_conv32_equals_the_halo_kernel(shape, dev):
    """`conv_strip32_kernel` (round 3: one WAVE per 16-pixel column strip sliding down a band of rows, weights in registers,
    one new input row per output row by LDS-DMA into a private ring, no workgroup barrier) serves the 32 -> 32 channel 3x3
    launches with >= 2^20 output pixels; the same launch restricted to one sample runs `conv_halo_kernel`.  Both accumulate the
    taps in list order with the same MFMA operand roles, so the batched launch must equal the per-sample launches BIT FOR BIT:
    reflect and zero padding, forward and flipped (backward-data) tap order, bias + LeakyReLU / ReLU epilogues, ragged strips
    and bands; the InstanceNorm moments (other partial layout) agree after finalisation."""
    from cta_gan_amd import ops
    b, h, w = shape
    assert b * h * w >= (1 << 20) and h * w < (1 << 20)
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(b, h, w, 32, generator=g).to(dev).to(torch.bfloat16)
    wp = (torch.randn(9, 32, 32, generator=g) * 0.08).to(dev).to(torch.bfloat16)
    bias = torch.randn(32, generator=g).to(dev)
    fwd = [ops.pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    flip = [ops.pack_tap(1 - ky, 1 - kx, ky * 3 + kx) for ky in range(3) for kx in range(3)]

    def run(xs, taps, pad_mode, bias_, act, want_stats):
        y = torch.empty(xs.shape[0], h, w, 32, dtype=torch.bfloat16, device=dev)
        part = ops.conv_igemm(xs, wp, 32, y, bias_, 32, h, w, 0, 0, 1, 1, pad_mode, act, taps, want_stats=want_stats)
        return y, part

    for taps, pad_mode, bias_, act, stats in ((fwd, ops.PAD_REFLECT, None, ops.ACT_NONE, True),
                                              (flip, ops.PAD_ZERO, None, ops.ACT_NONE, False),
                                              (fwd, ops.PAD_ZERO, bias, ops.ACT_LRELU, False),
                                              (flip, ops.PAD_REFLECT, bias, ops.ACT_RELU, False)):
        y_all, (p_all, n_all) = run(x, taps, pad_mode, bias_, act, stats)
        for i in sorted({0, b // 2, b - 1}):
            y_i, (p_i, n_i) = run(x[i:i + 1], taps, pad_mode, bias_, act, stats)
            assert torch.equal(y_all[i:i + 1], y_i), (i, pad_mode, act)
            if stats:
                assert n_all > 0 and n_i > 0 and n_all != n_i      # two kernels, two partial layouts
                m_all = ops.in_finalize(p_all[i:i + 1].contiguous(), n_all, h * w)
                m_i = ops.in_finalize(p_i, n_i, h * w)
                assert torch.allclose(m_all[0], m_i[0], rtol=1e-5, atol=1e-6) and torch.allclose(m_all[1], m_i[1], rtol=1e-5)
    # and against stock torch on a crop-free sample (bf16 operands, fp32 accumulate)
    xr = x[:1].float().permute(0, 3, 1, 2)
    wr = wp.float().reshape(3, 3, 32, 32).permute(2, 3, 0, 1)
    ref = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr)
    y_all, _ = run(x, fwd, ops.PAD_REFLECT, None, ops.ACT_NONE, False)
    assert _rel(y_all[:1].permute(0, 3, 1, 2), ref, l2=True) < 5e-3


@pytest.mark.parametrize("shape", [(4, 256, 256), (5, 250, 230), (20, 120, 112)], ids=["4x256x256", "5x250x230_ragged", "20x120x112"])
def test_sliding_window_transposed_conv_equals_the_class_kernels(shape, dev):
    """`conv_stript_128_64_kernel` (round 3: the 128 -> 64 channel stride-2 transposed 3x3 conv as a wave-autonomous sliding window
    -- all four parity classes of a 16-pixel input strip per step, a wave's share of the weights in registers, no barrier) serves
    `ctg_conv_igemm_classes` launches with >= 2^18 input pixels; the same launch restricted to one sample runs the merged parity
    classes on `conv_halo_kernel`.  Per class both accumulate (64-channel half, tap, k-step) in the same order with the same MFMA
    operand roles: the batched launch must equal the per-sample launches BIT FOR BIT, ragged strips and bands included; the
    InstanceNorm moments (another partial layout) agree after finalisation; and the result matches F.conv_transpose2d."""
    from cta_gan_amd import ops
    from cta_gan_amd.engine import _convT_classes
    b, h, w = shape
    assert b * h * w >= (1 << 18) and h * w < (1 << 18)
    g = torch.Generator().manual_seed(h * 3 + w)
    x = torch.randn(b, h, w, 128, generator=g).to(dev).to(torch.bfloat16)
    wp = (torch.randn(9, 64, 128, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    classes = _convT_classes(3, 1)

    def run(xs, want_stats):
        y = torch.zeros(xs.shape[0], 2 * h, 2 * w, 64, dtype=torch.bfloat16, device=dev)
        r = ops.conv_igemm_classes(xs, wp, 64, y, None, 64, h, w, classes, ops.PAD_ZERO, ops.ACT_NONE, want_stats=want_stats)
        assert r is not None
        return y, r

    y_all, (p_all, n_all) = run(x, True)
    for i in sorted({0, b // 2, b - 1}):
        y_i, (p_i, n_i) = run(x[i:i + 1], True)
        assert torch.equal(y_all[i:i + 1], y_i), i
        assert n_all > 0 and n_i > 0 and n_all != n_i
        m_all = ops.in_finalize(p_all[i:i + 1].contiguous(), n_all, 4 * h * w)
        m_i = ops.in_finalize(p_i, n_i, 4 * h * w)
        assert torch.allclose(m_all[0], m_i[0], rtol=1e-5, atol=1e-6) and torch.allclose(m_all[1], m_i[1], rtol=1e-5)
    y_ns, _ = run(x, False)
    assert torch.equal(y_ns, y_all)
    # against stock torch: W_packed[t = ky*3+kx][co][ci] is ConvTranspose2d's weight[ci][co][ky][kx]
    wt = wp.float().reshape(3, 3, 64, 128).permute(3, 2, 0, 1).contiguous()
    ref = F.conv_transpose2d(x[:1].float().permute(0, 3, 1, 2), wt, stride=2, padding=1, output_padding=1)
    assert _rel(y_all[:1].permute(0, 3, 1, 2), ref, l2=True) < 5e-3


@pytest.mark.parametrize("shape", [(4, 256, 256), (5, 250, 224), (20, 120, 112)], ids=["4x256x256", "5x250x224", "20x120x112"])
def test_sliding_window_stride2_conv_equals_the_gather_kernel(shape, dev):
    """`conv_strips2_64_128_kernel` (round 3: the 64 -> 128 channel stride-2 3x3 conv as a sliding window over input row pairs, a
    wave's share of the weights in registers, whole-pixel stores through a staging tile) serves `ctg_conv_igemm` launches with
    >= 2^18 output pixels; the same launch restricted to one sample runs conv_igemm_kernel.  Both accumulate (tap, k-step) in the
    same order with the same MFMA operand roles: the batched launch must equal the per-sample launches BIT FOR BIT; the
    InstanceNorm moments (another partial layout) agree after finalisation; and the result matches F.conv2d."""
    from cta_gan_amd import ops
    from cta_gan_amd.ops import pack_tap
    b, ho, wo = shape
    assert b * ho * wo >= (1 << 18) and ho * wo < (1 << 18)
    g = torch.Generator().manual_seed(ho * 3 + wo)
    x = torch.randn(b, 2 * ho, 2 * wo, 64, generator=g).to(dev).to(torch.bfloat16)
    wp = (torch.randn(9, 128, 64, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    taps = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]

    def run(xs, want_stats):
        y = torch.zeros(xs.shape[0], ho, wo, 128, dtype=torch.bfloat16, device=dev)
        r = ops.conv_igemm(xs, wp, 128, y, None, 128, ho, wo, 0, 0, 1, 2, ops.PAD_ZERO, ops.ACT_NONE, taps, want_stats=want_stats)
        return y, r

    y_all, (p_all, n_all) = run(x, True)
    for i in sorted({0, b // 2, b - 1}):
        y_i, (p_i, n_i) = run(x[i:i + 1], True)
        assert torch.equal(y_all[i:i + 1], y_i), i
        assert n_all > 0 and n_i > 0 and n_all != n_i
        m_all = ops.in_finalize(p_all[i:i + 1].contiguous(), n_all, ho * wo)
        m_i = ops.in_finalize(p_i, n_i, ho * wo)
        assert torch.allclose(m_all[0], m_i[0], rtol=1e-5, atol=1e-6) and torch.allclose(m_all[1], m_i[1], rtol=1e-5)
    y_ns, _ = run(x, False)
    assert torch.equal(y_ns, y_all)
    # against stock torch: W_packed[t = ky*3+kx][co][ci] is Conv2d's weight[co][ci][ky][kx]
    wt = wp.float().reshape(3, 3, 128, 64).permute(2, 3, 0, 1).contiguous()
    ref = F.conv2d(x[:1].float().permute(0, 3, 1, 2), wt, stride=2, padding=1)
    assert _rel(y_all[:1].permute(0, 3, 1, 2), ref, l2=True) < 5e-3


def test_sliding_window_stride2_kernels_take_channel_slices(dev):
    """Both stride-2 sliding-window kernels address their operands through x_ld / y_ld: an input that is a channel slice of a wider
    buffer and an output that lands in a slice of a concat buffer give the same bits as dense tensors, and the bytes around the
    output slice stay untouched."""
    from cta_gan_amd import ops
    from cta_gan_amd.engine import _convT_classes
    from cta_gan_amd.ops import pack_tap
    g = torch.Generator().manual_seed(5)
    # transposed conv 128 -> 64
    b, h, w = 4, 256, 256
    xw = torch.randn(b, h, w, 160, generator=g).to(dev).to(torch.bfloat16)
    x = xw[..., 16:144]
    wp = (torch.randn(9, 64, 128, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    classes = _convT_classes(3, 1)
    y_dense = torch.zeros(b, 2 * h, 2 * w, 64, dtype=torch.bfloat16, device=dev)
    assert ops.conv_igemm_classes(x.contiguous(), wp, 64, y_dense, None, 64, h, w, classes, ops.PAD_ZERO, ops.ACT_NONE) is not None
    yw = torch.full((b, 2 * h, 2 * w, 96), 7.0, dtype=torch.bfloat16, device=dev)
    assert ops.conv_igemm_classes(x, wp, 64, yw[..., 8:72], None, 64, h, w, classes, ops.PAD_ZERO, ops.ACT_NONE) is not None
    assert torch.equal(yw[..., 8:72], y_dense) and bool((yw[..., :8] == 7.0).all()) and bool((yw[..., 72:] == 7.0).all())
    # stride-2 conv 64 -> 128
    ho = wo = 256
    xw = torch.randn(b, 2 * ho, 2 * wo, 72, generator=g).to(dev).to(torch.bfloat16)
    x = xw[..., 8:72]
    wp = (torch.randn(9, 128, 64, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    taps = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
    y_dense = torch.zeros(b, ho, wo, 128, dtype=torch.bfloat16, device=dev)
    ops.conv_igemm(x.contiguous(), wp, 128, y_dense, None, 128, ho, wo, 0, 0, 1, 2, ops.PAD_ZERO, ops.ACT_NONE, taps)
    yw = torch.full((b, ho, wo, 160), 7.0, dtype=torch.bfloat16, device=dev)
    ops.conv_igemm(x, wp, 128, yw[..., 16:144], None, 128, ho, wo, 0, 0, 1, 2, ops.PAD_ZERO, ops.ACT_NONE, taps)
    assert torch.equal(yw[..., 16:144], y_dense) and bool((yw[..., :16] == 7.0).all()) and bool((yw[..., 144:] == 7.0).all())


@pytest.mark.parametrize("shape", [(8, 128, 128), (12, 120, 112)], ids=["8x128x128", "12x120x112"])
def test_sliding_window_wide_stride2_conv_equals_the_gather_kernel(shape, dev):
    """`conv_strips2_128_256_kernel` (the 128 -> 256 channel stride-2 3x3 conv: eight waves, half of the output channels per
    workgroup, each wave's 36 weight fragments in registers) serves `ctg_conv_igemm` launches with >= 2^17 output pixels; the same
    launch restricted to one sample runs conv_igemm_kernel.  Same (tap, k-step) order: BIT-identical outputs; moments agree after
    finalisation; and the result matches F.conv2d."""
    from cta_gan_amd import ops
    from cta_gan_amd.ops import pack_tap
    b, ho, wo = shape
    assert b * ho * wo >= (1 << 17) and ho * wo < (1 << 17)
    g = torch.Generator().manual_seed(ho * 5 + wo)
    x = torch.randn(b, 2 * ho, 2 * wo, 128, generator=g).to(dev).to(torch.bfloat16)
    wp = (torch.randn(9, 256, 128, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    taps = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]

    def run(xs, want_stats):
        y = torch.zeros(xs.shape[0], ho, wo, 256, dtype=torch.bfloat16, device=dev)
        r = ops.conv_igemm(xs, wp, 256, y, None, 256, ho, wo, 0, 0, 1, 2, ops.PAD_ZERO, ops.ACT_NONE, taps, want_stats=want_stats)
        return y, r

    y_all, (p_all, n_all) = run(x, True)
    for i in sorted({0, b // 2, b - 1}):
        y_i, (p_i, n_i) = run(x[i:i + 1], True)
        assert torch.equal(y_all[i:i + 1], y_i), i
        assert n_all > 0 and n_i > 0 and n_all != n_i
        m_all = ops.in_finalize(p_all[i:i + 1].contiguous(), n_all, ho * wo)
        m_i = ops.in_finalize(p_i, n_i, ho * wo)
        assert torch.allclose(m_all[0], m_i[0], rtol=1e-5, atol=1e-6) and torch.allclose(m_all[1], m_i[1], rtol=1e-5)
    y_ns, _ = run(x, False)
    assert torch.equal(y_ns, y_all)
    wt = wp.float().reshape(3, 3, 256, 128).permute(2, 3, 0, 1).contiguous()
    ref = F.conv2d(x[:1].float().permute(0, 3, 1, 2), wt, stride=2, padding=1)
    assert _rel(y_all[:1].permute(0, 3, 1, 2), ref, l2=True) < 5e-3


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("case", [(3, 64, 128, 64, 80, False), (2, 128, 256, 46, 66, False), (2, 128, 64, 40, 48, True)],
                         ids=["conv64to128", "conv128to256_ragged", "convT128to64"])
def test_merged_polyphase_weight_gradient_of_stride2_convs(case, mode, dev):
    """`conv_wgrad_s2m_kernel` (round 5): the weight gradient of a stride-2 3x3 conv -- and of a stride-2 transposed conv, the same
    contraction with the tensors' roles swapped -- with all four polyphase components of the strided operand in ONE launch (the G
    tile fetched once per pixel tile instead of once per component launch), bf16 and split pair, several pixel slabs per sample,
    ragged tiles, odd input sizes: against torch's fp32 conv weight gradient on the values the kernel sees (bf16: operands rounded
    to bf16, rel-L2 5e-3 -- accumulation order only; split pair: 2e-5)."""
    import torch.nn.functional as F
    from cta_gan_amd import nets, ops
    from cta_gan_amd.ops import pack_tap
    b, cin, cout, hi, wi, transposed = case
    nets.set_default_compute_dtype(torch.bfloat16 if mode == "bf16" else mode)
    try:
        g_ = torch.Generator().manual_seed(hi * 7 + wi)
        taps = [pack_tap(ky - 1, kx - 1, ky * 3 + kx) for ky in range(3) for kx in range(3)]
        if not transposed:
            ho, wo = (hi + 2 - 3) // 2 + 1, (wi + 2 - 3) // 2 + 1
            x32 = torch.randn(b, hi, wi, cin, generator=g_).to(dev)
            dy32 = torch.randn(b, ho, wo, cout, generator=g_).to(dev)
            if mode == "bf16":
                xs, gs = x32.to(torch.bfloat16), dy32.to(torch.bfloat16)
                xr, gr = xs.float(), gs.float()
            else:
                xs, gs = ops.to_pair(x32), ops.to_pair(dy32)
                xr, gr = ops.from_pair(xs), ops.from_pair(gs)
            dw = torch.zeros(cout, cin, 3, 3, device=dev)
            ops.conv_wgrad(gs, xs, taps, 2, ops.PAD_ZERO, dw, cout, cin, cin * 9, 9, 1, target_blocks=512)
            want = torch.nn.grad.conv2d_weight(xr.permute(0, 3, 1, 2), (cout, cin, 3, 3), gr.permute(0, 3, 1, 2), stride=2, padding=1)
        else:
            # ConvTranspose2d(cin, cout, 3, s2, p1, op1): G = the layer input on its own grid, X = dL/dy read at 2 i - 1 + k
            ho, wo = 2 * hi, 2 * wi
            x32 = torch.randn(b, hi, wi, cin, generator=g_).to(dev)
            dy32 = torch.randn(b, ho, wo, cout, generator=g_).to(dev)
            if mode == "bf16":
                xs, gs = x32.to(torch.bfloat16), dy32.to(torch.bfloat16)
                xr, gr = xs.float(), gs.float()
            else:
                xs, gs = ops.to_pair(x32), ops.to_pair(dy32)
                xr, gr = ops.from_pair(xs), ops.from_pair(gs)
            dw = torch.zeros(cin, cout, 3, 3, device=dev)
            ops.conv_wgrad(xs, gs, taps, 2, ops.PAD_ZERO, dw, cin, cout, cout * 9, 9, 1, target_blocks=512)
            xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
            w = torch.zeros(cin, cout, 3, 3, device=dev, requires_grad=True)
            y = F.conv_transpose2d(xin, w, stride=2, padding=1, output_padding=1)
            (y * gr.permute(0, 3, 1, 2)).sum().backward()
            want = w.grad
        torch.cuda.synchronize()
        err = _rel(dw, want, l2=True)
        print(case, mode, "rel-L2 %.2e" % err)
        assert err < (5e-3 if mode == "bf16" else 2e-5), err
    finally:
        nets.set_default_compute_dtype(torch.float32)


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 1), (1, 5, 7, 1), (3, 17, 33, 1), (2, 9, 21, 2), (1, 4, 4, 0)],
                         ids=["64x64_pad1", "5x7_pad1", "17x33_pad1", "9x21_pad2", "4x4_pad0"])
def test_patchgan_last_layer_on_the_vector_alus(mode, shape):
    """csrc/conv_cout1.hip: Conv2d(512, 1, 4, padding) forward, input gradient and weight gradient (Model/HdGan.py:136-137,
    Model/CycleGan.py:104) against fp64 torch on the operands the kernels see (the activation as stored, the fp32 master weights):
    they compute in fp32 from exact operands, so 1e-5 relative in both storage modes; ragged runs (widths that are not multiples
    of 16), maps smaller than a run, one output pixel, B = 1."""
    from cta_gan_amd import nets, ops
    b, h, w, pad = shape
    nets.set_default_compute_dtype("bf16x3" if mode == "bf16x3" else torch.bfloat16)
    try:
        gen = torch.Generator().manual_seed(11)
        x32 = torch.randn(b, h, w, 512, generator=gen).cuda()
        wt = (torch.randn(1, 512, 4, 4, generator=gen) * 0.05).cuda()
        bias = torch.randn(1, generator=gen).cuda()
        if mode == "bf16x3":
            x = ops.to_pair(x32)
            xs = ops.from_pair(x).double().cpu()        # what the kernels read: hi + lo
        else:
            x = x32.bfloat16()
            xs = x.double().cpu()
        ho, wo = h + 2 * pad - 3, w + 2 * pad - 3
        w16 = ops.cout1_pack(wt)
        y = torch.empty(b, ho, wo, dtype=torch.float32, device="cuda")
        ops.conv_cout1_fwd(x, w16, bias, y, 0, pad)
        xr = xs.permute(0, 3, 1, 2).clone().requires_grad_(True)
        wr = wt.double().cpu().clone().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, bias.double().cpu(), padding=pad)
        g = torch.randn(b, ho, wo, generator=gen)
        yr.backward(g.double()[:, None])
        scale = float(yr.detach().abs().max())
        assert float((y.double().cpu() - yr[:, 0]).abs().max()) <= 1e-5 * scale
        dx = ops.empty_act((b, h, w, 512), torch.bfloat16, x.device)
        ops.conv_cout1_bwd(g.cuda(), w16, dx, pad)
        dxf = (ops.from_pair(dx) if mode == "bf16x3" else dx.float()).double().cpu()
        want_dx = xr.grad.permute(0, 2, 3, 1)
        tol_dx = 2.0 ** -16 if mode == "bf16x3" else 2.0 ** -8          # the storage rounding of the result
        assert float((dxf - want_dx).abs().max()) <= tol_dx * float(want_dx.abs().max())
        dw = torch.zeros(1, 512, 4, 4, device="cuda")
        ops.conv_cout1_wgrad(g.cuda(), x, dw, pad)
        torch.cuda.synchronize()
        assert float((dw.double().cpu() - wr.grad).abs().max()) <= 1e-5 * float(wr.grad.abs().max())
    finally:
        nets.set_default_compute_dtype(torch.float32)


def test_patchgan_last_layer_kernels_beside_narrow_halo_convs_on_another_stream():
    """Regression for the round-5 find: the first conv_cout1 input-gradient / weight-gradient kernels multiplied by a broadcast
    gradient value through v_pk_fma_f32's op_sel modifiers and returned wrong products in lanes 48-63 -- only while their waves shared
    a CU with workgroups of conv_halo_kernel<BN <= 32> running on ANOTHER stream (the CycleGan step's adversarial branch beside the
    generators' backward; found by that step's bit-repeatability test).  Here the three kernels run in a loop on a second stream while
    the main stream runs such convolutions; every result must be the solo result (fp64-checked).  scripts/lds_neighbour_stress.py is the
    long form."""
    from cta_gan_amd import nets, ops
    from cta_gan_amd.engine import ConvSpec
    nets.set_default_compute_dtype(torch.bfloat16)
    os.environ["CTG_NO_COUT1"] = "1"          # the main stream's 512 -> 1 probe must be the narrow MFMA launch, not the kernels under test
    try:
        gen = torch.Generator().manual_seed(1)
        b, h, w = 8, 63, 63
        wt = (torch.randn(1, 512, 4, 4, generator=gen) * 0.05).cuda()
        w16 = ops.cout1_pack(wt)
        g = torch.randn(b, h - 1, w - 1, generator=gen).cuda()
        xin = torch.randn(b, h, w, 512, generator=gen).cuda().bfloat16()
        xd = xin.double().permute(0, 3, 1, 2)
        dx_ref = torch.nn.functional.conv_transpose2d(g.double()[:, None], wt.double(), padding=1).permute(0, 2, 3, 1)
        y_ref = torch.nn.functional.conv2d(xd, wt.double(), padding=1)[:, 0]
        dw_ref = torch.nn.grad.conv2d_weight(xd, (1, 512, 4, 4), g.double()[:, None], padding=1)
        probes = []
        for (cin, cout, k, size, f32) in ((64, 32, 3, 128, False), (512, 1, 4, 63, True), (32, 2, 3, 256, True)):
            probes.append((_make_probe(ConvSpec(cin, cout, k, 1, (k - 1) // 2, use_bias=True, out_f32=f32), None).cuda(),
                           torch.randn(16, cin, size, size, device="cuda")))
        side = torch.cuda.Stream()
        for rep in range(2):
            dxs = [ops.empty_act((b, h, w, 512), torch.bfloat16, g.device) for _ in range(60)]
            ys = [torch.empty(b, h - 1, w - 1, device="cuda") for _ in range(30)]
            dws = [torch.zeros(1, 512, 4, 4, device="cuda") for _ in range(20)]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for t in dxs:
                    ops.conv_cout1_bwd(g, w16, t, 1)
                for t in ys:
                    ops.conv_cout1_fwd(xin, w16, None, t, 0, 1)
                for t in dws:
                    ops.conv_cout1_wgrad(g, xin, t, 1)
            with torch.no_grad():
                for _ in range(12):
                    for p, px in probes:
                        p(px)
            torch.cuda.synchronize()
            lim = 0.02 * float(dx_ref.abs().max())
            assert sum(int(((t.double() - dx_ref).abs() > lim).sum()) for t in dxs) == 0
            assert sum(int(((t.double() - y_ref).abs() > 1e-4 * float(y_ref.abs().max())).sum()) for t in ys) == 0
            assert sum(int(((t.double() - dw_ref).abs() > 1e-4 * float(dw_ref.abs().max())).sum()) for t in dws) == 0
            assert all(torch.equal(t, dxs[0]) for t in dxs) and all(torch.equal(t, ys[0]) for t in ys) and all(torch.equal(t, dws[0]) for t in dws)
    finally:
        os.environ.pop("CTG_NO_COUT1", None)
        nets.set_default_compute_dtype(torch.float32)


def test_lds_canary_beside_a_training_step():
    """csrc/lds_canary.hip: workgroups that fill 8 KB of LDS with a pattern and keep re-reading it, on a second stream beside one
    Hd training step (bf16 and bf16x3): nothing else may have written their LDS -- every LDS-DMA destination of the conv kernels is
    computed by hand, and an address past a workgroup's allocation would land in a neighbour's."""
    from cta_gan_amd import nets, ops, synth
    from cta_gan_amd.trainer import Hd_Trainer_x2
    for mode in (torch.bfloat16, "bf16x3"):
        nets.set_default_compute_dtype(mode)
        try:
            cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=4, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
                       Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1)
            tr = Hd_Trainer_x2(cfg)
            batch = {k: synth.synth_smooth_images("can_" + k, 4, 256).cuda() for k in ("A2", "B1", "B2")}
            tr.train_step(batch)
            torch.cuda.synchronize()
            canary_stream = torch.cuda.Stream()
            rep = ops.lds_canary(blocks=512, spins=20000, cap=16, stream=canary_stream)
            for _ in range(3):
                tr.train_step(batch)
            torch.cuda.synchronize()
            r = rep.cpu()
            assert int(r[0]) == 0 and int(r[1]) == 0, (str(mode), r[:24].tolist())
        finally:
            nets.set_default_compute_dtype(torch.float32)
