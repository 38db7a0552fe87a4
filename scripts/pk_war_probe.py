"""Build scripts/diag/pk_war_canary.hip (hipcc, on the GPU box) and run it on a second stream beside one conv layer on the main stream.
    python scripts/pk_war_probe.py [probe_<cin>_<cout>_<k>_<size>[_f32out] | none]"""
import sys, os, ctypes, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
so = os.path.join(tempfile.gettempdir(), "pk_war_canary.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(ROOT, "scripts", "diag", "pk_war_canary.hip"), "-o", so])
from cta_gan_amd import nets, ops, _lib
_lib.load()
lib = ctypes.CDLL(so)
lib.pk_war_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
nets.set_default_compute_dtype(torch.bfloat16)
os.environ["CTG_NO_COUT1"] = "1"
what = sys.argv[1] if len(sys.argv) > 1 else "probe_64_32_3_128"
net = None
if what != "none":
    from cta_gan_amd.engine import ConvSpec
    import test_kernels_gpu as K
    parts = what.split("_")
    cin, cout, k, size = int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4])
    net = K._make_probe(ConvSpec(cin, cout, k, 1, (k - 1) // 2, use_bias=True, out_f32="f32out" in parts), None).cuda()
    x = torch.randn(16, cin, size, size, device="cuda")
gsrc = torch.randn(128, device="cuda")
side = torch.cuda.Stream()
for form, name in ((0, "op_sel form, its pair overwritten by the next LDS read"), (1, "plain form on a copy, same LDS read behind it")):
    rep = torch.zeros(8, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = lib.pk_war_launch(form, 1024, 3000, gsrc.data_ptr(), rep.data_ptr(), side.cuda_stream)
    assert rc == 0, rc
    if net is not None:
        with torch.no_grad():
            for _ in range(100):
                net(x)
    torch.cuda.synchronize()
    r = rep.cpu()
    print("%-24s form %d (%s): %d lane-iterations differ; by lane quarter %s" % (what, form, name, int(r[0]), r[4:8].tolist()))
