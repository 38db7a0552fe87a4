"""u2 / u1 transposed-conv forward (4 parity-class launches): sequential on one stream vs concurrently on 4 streams
(a proxy for a merged launch whose classes share the input halo through L2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import engine as E, ops, nets
from cta_gan_amd.engine import ConvSpec, PackCache
dt = torch.bfloat16
B = 16
for (cin, cout, hi) in ((128, 64, 256), (256, 128, 128)):
    spec = ConvSpec(cin, cout, 3, 2, 1, transposed=True, use_bias=False)
    w = torch.randn(cin, cout, 3, 3, device="cuda") * 0.05
    x = torch.randn(B, hi, hi, cin, device="cuda").to(dt)
    y = torch.empty(B, 2 * hi, 2 * hi, cout, device="cuda", dtype=dt)
    cache = PackCache()
    wp, npad = E._pack_fwd(cache, spec, w, dt)
    classes = E._convT_classes(3, 1)
    streams = [torch.cuda.Stream() for _ in range(4)]
    def seq():
        for py, px, taps in classes:
            ops.conv_igemm(x, wp, npad, y, None, cout, hi, hi, py, px, 2, 1, ops.PAD_ZERO, 0, taps, want_stats=True)
    def par():
        cur = torch.cuda.current_stream()
        for s_, (py, px, taps) in zip(streams, classes):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                ops.conv_igemm(x, wp, npad, y, None, cout, hi, hi, py, px, 2, 1, ops.PAD_ZERO, 0, taps, want_stats=True)
        for s_ in streams:
            cur.wait_stream(s_)
    for name, fn in (("sequential", seq), ("4 streams", par)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        byts = (x.numel() + y.numel()) * 2
        print("convT %d->%d @%d^2  %-10s %.1f us per layer   (%.2f TB/s of in-once + out-once bytes)" % (cin, cout, hi, name, t * 1e3, byts / t / 1e9))

# ---- where does a class launch spend its time?  one 4-tap class of the 128 -> 64 layer, variants
cin, cout, hi = 128, 64, 256
spec = ConvSpec(cin, cout, 3, 2, 1, transposed=True, use_bias=False)
w = torch.randn(cin, cout, 3, 3, device="cuda") * 0.05
x = torch.randn(B, hi, hi, cin, device="cuda").to(dt)
y2 = torch.empty(B, 2 * hi, 2 * hi, cout, device="cuda", dtype=dt)
yd = torch.empty(B, hi, hi, cout, device="cuda", dtype=dt)
cache = PackCache()
wp, npad = E._pack_fwd(cache, spec, w, dt)
py, px, taps = [c for c in E._convT_classes(3, 1) if len(c[2]) == 4][0]
def t_(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("4-tap class, strided output + moments : %.1f us" % t_(lambda: ops.conv_igemm(x, wp, npad, y2, None, cout, hi, hi, py, px, 2, 1, ops.PAD_ZERO, 0, taps, want_stats=True)))
print("4-tap class, strided output, no moments: %.1f us" % t_(lambda: ops.conv_igemm(x, wp, npad, y2, None, cout, hi, hi, py, px, 2, 1, ops.PAD_ZERO, 0, taps, want_stats=False)))
print("same taps, DENSE output (os=1)        : %.1f us" % t_(lambda: ops.conv_igemm(x, wp, npad, yd, None, cout, hi, hi, 0, 0, 1, 1, ops.PAD_ZERO, 0, taps, want_stats=False)))
one = [taps[0]]
print("1 tap, dense output                   : %.1f us" % t_(lambda: ops.conv_igemm(x, wp, npad, yd, None, cout, hi, hi, 0, 0, 2, 1, ops.PAD_ZERO, 0, one, want_stats=False) if False else ops.conv_igemm(x, wp, npad, y2, None, cout, hi, hi, py, px, 2, 1, ops.PAD_ZERO, 0, one, want_stats=False)))
xin = torch.empty_like(yd)
print("elementwise copy of the same bytes (268 MB in, 134 MB out): %.1f us" % t_(lambda: yd.copy_(x[..., :cout])))
classes = E._convT_classes(3, 1)
print("4 class launches (moments)            : %.1f us" % t_(lambda: [ops.conv_igemm(x, wp, npad, y2, None, cout, hi, hi, c[0], c[1], 2, 1, ops.PAD_ZERO, 0, c[2], want_stats=True) for c in classes]))
r = ops.conv_igemm_classes(x, wp, npad, y2, None, cout, hi, hi, classes, ops.PAD_ZERO, 0, want_stats=True)
print("merged served:", r is not None)
print("ONE merged launch (moments)           : %.1f us" % t_(lambda: ops.conv_igemm_classes(x, wp, npad, y2, None, cout, hi, hi, classes, ops.PAD_ZERO, 0, want_stats=True)))
print("ONE merged launch (no moments)        : %.1f us" % t_(lambda: ops.conv_igemm_classes(x, wp, npad, y2, None, cout, hi, hi, classes, ops.PAD_ZERO, 0, want_stats=False)))
