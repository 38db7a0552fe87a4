"""Do other kernels whose packed-fp32 instructions carry broadcast modifiers (the bilinear resampling kernels: 113 of them) change their
results next to conv_halo_kernel<BN <= 32> on another stream, as the first conv_cout1 kernels did?  Bitwise comparison with a solo run.
    python scripts/pk_neighbour_stress.py [bf16|x3]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cta_gan_amd import nets, ops
from cta_gan_amd.engine import ConvSpec
import test_kernels_gpu as K
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
nets.set_default_compute_dtype("bf16x3" if mode == "x3" else torch.bfloat16)
probes = []
for (cin, cout, k, size, f32) in ((64, 32, 3, 128, False), (512, 1, 4, 63, True), (32, 2, 3, 256, True)):
    os.environ["CTG_NO_COUT1"] = "1"
    p = K._make_probe(ConvSpec(cin, cout, k, 1, (k - 1) // 2, use_bias=True, out_f32=f32), None).cuda()
    probes.append((p, torch.randn(16, cin, size, size, device="cuda")))
x32 = torch.randn(16, 128, 128, 64, device="cuda")
x = ops.to_pair(x32) if mode == "x3" else x32.bfloat16()
g32 = torch.randn(16, 256, 256, 64, device="cuda")
g = ops.to_pair(g32) if mode == "x3" else g32.bfloat16()


def victims():
    outs = []
    for _ in range(40):
        up = ops.empty_act((16, 256, 256, 64), torch.bfloat16, x.device)
        ops.bilinear_fwd(x, up)
        dn = ops.empty_act((16, 128, 128, 64), torch.bfloat16, x.device)
        ops.bilinear_bwd(g, dn)
        outs.append((up, dn))
    return outs


def raw(t):
    return t.as_strided((t.numel() * (2 if mode == "x3" else 1),), (1,), t.storage_offset()).view(torch.int16)


solo = victims()
torch.cuda.synchronize()
ref_up, ref_dn = raw(solo[0][0]).clone(), raw(solo[0][1]).clone()
assert all(torch.equal(raw(u), ref_up) and torch.equal(raw(d), ref_dn) for u, d in solo)
del solo
side = torch.cuda.Stream()
bad = 0
for rep in range(4):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        outs = victims()
    with torch.no_grad():
        for _ in range(30):
            for p, px in probes:
                p(px)
    torch.cuda.synchronize()
    n = sum(int(not torch.equal(raw(u), ref_up)) + int(not torch.equal(raw(d), ref_dn)) for u, d in outs)
    bad += n
    print("rep", rep, mode, "bilinear launches whose bits differ from the solo run:", n, "of", 2 * len(outs))
    del outs
print("TOTAL", mode, bad)
