"""Which kernel disturbs a small static-LDS neighbour?  conv_cout1_bwd (1.2 KB of LDS per workgroup: a patch written once, read once
behind a barrier) runs in a loop on a second stream while the main stream runs a workload; every result is checked against an fp64
reference.  Wrong elements mean something else wrote this kernel's LDS while it was resident.
    python scripts/lds_neighbour_stress.py WORKLOAD   (gen_fwd | gen_fwdbwd | disc_fwdbwd | none)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, ops, synth
from cta_gan_amd.Model.CycleGan import Generator, Discriminator
nets.set_default_compute_dtype(torch.bfloat16)
what = sys.argv[1] if len(sys.argv) > 1 else "gen_fwdbwd"
B, S = 8, 512
gen = torch.Generator().manual_seed(1)
b, h, w = 8, 63, 63
wt = (torch.randn(1, 512, 4, 4, generator=gen) * 0.05).cuda()
w16 = ops.cout1_pack(wt)
g = torch.randn(b, h - 1, w - 1, generator=gen).cuda()
ref = torch.nn.functional.conv_transpose2d(g.double()[:, None], wt.double(), padding=1).permute(0, 2, 3, 1)
lim = 0.02 * ref.abs().max()
side = torch.cuda.Stream()
x = synth.synth_images("ls_a", B, S).cuda().requires_grad_(True)
G = Generator(1, 1).cuda()
D = Discriminator(1).cuda()
synth.fill_module(G, seed=0)
synth.fill_module(D, seed=6)


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
PROBE = None
if what.startswith("probe"):
    # one conv layer on the main stream: probe_<cin>_<cout>_<k>_<size>[_f32out][_fwd]
    import test_kernels_gpu as K
    from cta_gan_amd.engine import ConvSpec
    parts = what.split("_")
    cin, cout, k, size = int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4])
    spec = ConvSpec(cin, cout, k, 1, (k - 1) // 2, use_bias=True, out_f32="f32out" in parts)
    PROBE = K._make_probe(spec, None).cuda()
    px = torch.randn(16, cin, size, size, device="cuda").requires_grad_(True)


def main_work():
    if PROBE is not None:
        for _ in range(40):
            if "fwd" in what.split("_"):
                with torch.no_grad():
                    PROBE(px)
            else:
                PROBE(px).float().mean().backward()
        return
    if what == "gen_fwd":
        with torch.no_grad():
            G(x)
    elif what == "gen_fwdbwd":
        G(x).float().mean().backward()
    elif what == "disc_fwdbwd":
        D(x).float().mean().backward()


bufs = [ops.empty_act((b, h, w, 512), torch.bfloat16, g.device) for _ in range(240)]
# the forward and the weight gradient as victims too
xin = torch.randn(b, h, w, 512, generator=gen).cuda().bfloat16()
yref = torch.nn.functional.conv2d(xin.double().permute(0, 3, 1, 2), wt.double(), padding=1)[:, 0]
dwref = torch.nn.grad.conv2d_weight(xin.double().permute(0, 3, 1, 2), (1, 512, 4, 4), g.double()[:, None], padding=1)
ybufs = [torch.empty(b, h - 1, w - 1, device="cuda") for _ in range(120)]
dwbufs = [torch.zeros(1, 512, 4, 4, device="cuda") for _ in range(60)]
total_wrong = 0
for rep in range(6):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for dq in bufs:
            ops.conv_cout1_bwd(g, w16, dq, 1)
        for yq in ybufs:
            ops.conv_cout1_fwd(xin, w16, None, yq, 0, 1)
        for dq in dwbufs:
            ops.conv_cout1_wgrad(g, xin, dq, 1)
    main_work()
    torch.cuda.synchronize()
    wy = sum(int(((yq.double() - yref).abs() > 1e-4 * yref.abs().max()).sum()) for yq in ybufs)
    ww = sum(int(((dq.double() - dwref).abs() > 1e-4 * dwref.abs().max()).sum()) for dq in dwbufs)
    total_wrong += wy + ww
    print("rep", rep, what, "forward: wrong elements", wy, "| weight gradient: wrong elements", ww)
    wrong = [int(((dq.double() - ref).abs() > lim).sum()) for dq in bufs]
    worst = max(range(len(bufs)), key=lambda i: wrong[i])
    if wrong[worst] and not getattr(sys, "_shown", False):
        sys._shown = True
        idx = ((bufs[worst].double() - ref).abs() > lim).nonzero()
        import collections
        print("   worst launch %d: %d wrong; by pixel-in-run %s" % (worst, wrong[worst], sorted(collections.Counter((idx[:, 2] % 16).tolist()).items())))
        print("   by lane %s" % sorted(collections.Counter((idx[:, 3] // 8).tolist()).items()))
        print("   by row-run (b, iy, seg): %d distinct of %d; by wave-in-workgroup (= seg) %s" % (
            len(set((int(a), int(b_), int(c) // 16) for a, b_, c in idx[:, :3].tolist())), 8 * 63 * 4,
            sorted(collections.Counter((idx[:, 2] // 16).tolist()).items())))
    total_wrong += sum(wrong)
    print("rep", rep, what, "launches with wrong elements:", sum(1 for v in wrong if v), "of", len(wrong), "first at", next((i for i, v in enumerate(wrong) if v), None))
print("TOTAL", what, total_wrong)
