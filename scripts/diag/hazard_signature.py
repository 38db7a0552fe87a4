"""What do the wrong elements of the failing input-gradient kernel look like?  (run with CTG_LIB = a failing variant of
scripts/diag/hazard_variants.py)  The kernel runs in a loop on one stream while a second HOST thread keeps narrow halo convs running on
another (the recipe of tests/test_neighbour_stress_gpu.py); every wrong element is classified by lane, channel parity inside the
lane's chunk (even = low half of a packed result, odd = high half) and pixel-in-run, and the error is compared with what the
high half would get if the instruction took the pair's HIGH register where op_sel_hi says low (g[c + 1] instead of g[c])."""
import collections
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from cta_gan_amd import nets, ops  # noqa: E402
from cta_gan_amd.engine import ConvSpec  # noqa: E402
import test_kernels_gpu as K  # noqa: E402

nets.set_default_compute_dtype(torch.bfloat16)
gen = torch.Generator().manual_seed(1)
b, h, w = 8, 63, 63
wt = (torch.randn(1, 512, 4, 4, generator=gen) * 0.05).cuda()
w16 = ops.cout1_pack(wt)
g = torch.randn(b, h - 1, w - 1, generator=gen).cuda()
ref = torch.nn.functional.conv_transpose2d(g.double()[:, None], wt.double(), padding=1).permute(0, 2, 3, 1)      # [b, h, w, 512]
lim = 1e-2 * float(ref.abs().max())
probes = []
for (cin, cout, k, size, f32) in ((64, 32, 3, 128, False), (512, 2, 4, 63, True), (32, 2, 3, 256, True)):
    probes.append((K._make_probe(ConvSpec(cin, cout, k, 1, (k - 1) // 2, use_bias=True, out_f32=f32), None).cuda(),
                   torch.randn(16, cin, size, size, device="cuda")))
stop = threading.Event()
nb_stream, side = torch.cuda.Stream(), torch.cuda.Stream()
rounds = [0]


def neighbours():
    torch.cuda.set_device(0)
    with torch.cuda.stream(nb_stream), torch.no_grad():
        while not stop.is_set():
            for p, px in probes:
                p(px)
            rounds[0] += 1
            if rounds[0] % 8 == 0:
                nb_stream.synchronize()


th = threading.Thread(target=neighbours, daemon=True)
th.start()
while rounds[0] < 16:
    stop.wait(0.001)
by_lane, by_parity, by_pix, n_wrong, n_launch, n_bad_launch = collections.Counter(), collections.Counter(), collections.Counter(), 0, 0, 0
shown = 0
for rep in range(20):
    with torch.cuda.stream(side):
        bufs = [ops.empty_act((b, h, w, 512), torch.bfloat16, g.device) for _ in range(40)]
        for dq in bufs:
            ops.conv_cout1_bwd(g, w16, dq, 1)
    side.synchronize()
    for dq in bufs:
        n_launch += 1
        err = dq.double() - ref
        idx = (err.abs() > lim).nonzero()
        if idx.numel() == 0:
            continue
        n_bad_launch += 1
        n_wrong += idx.shape[0]
        by_lane.update((idx[:, 3] // 8).tolist())
        by_parity.update((idx[:, 3] % 2).tolist())
        by_pix.update((idx[:, 2] % 16).tolist())
        if shown < 6:
            shown += 1
            bb, yy, xx, cc = [int(v) for v in idx[0]]
            # the taps of this input pixel: dx[y, x, c] = sum_{ky, kx} g[y - ky + 1, x - kx + 1] * w[c, ky, kx]; the hypothetical
            # "high register instead of low" error of ONE tap = (g[.., x - kx + 2] - g[.., x - kx + 1]) * w (the patch's next column)
            e = float(err[bb, yy, xx, cc])
            cands = []
            for ky in range(4):
                for kx in range(4):
                    oy, ox = yy - ky + 1, xx - kx + 1
                    gv = lambda a, c_: float(g[bb, a, c_]) if 0 <= a < h - 1 and 0 <= c_ < w - 1 else 0.0      # noqa: E731
                    cands.append(((ky, kx), (gv(oy, ox + 1) - gv(oy, ox)) * float(wt[0, cc, ky, kx]), -gv(oy, ox) * float(wt[0, cc, ky, kx])))
            best_next = min(cands, key=lambda c_: abs(c_[1] - e))
            best_drop = min(cands, key=lambda c_: abs(c_[2] - e))
            print("wrong element (b %d, y %d, x %d, c %d = lane %d, channel-in-chunk %d): error %+.5f; closest single-tap 'next column' "
                  "error %+.5f at tap %s; closest single-tap 'dropped' error %+.5f at tap %s; wrong elements in this launch: %d"
                  % (bb, yy, xx, cc, cc // 8, cc % 8, e, best_next[1], best_next[0], best_drop[2], best_drop[0], idx.shape[0]))
stop.set()
th.join(30)
torch.cuda.synchronize()
print("launches %d, with wrong elements %d, wrong elements %d" % (n_launch, n_bad_launch, n_wrong))
print("by lane:", sorted(by_lane.items()))
print("by channel parity (0 = low half of a packed result, 1 = high half):", sorted(by_parity.items()))
print("by pixel-in-run:", sorted(by_pix.items()))
