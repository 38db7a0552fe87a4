"""Minimal reproducer attempt of the packed-fp32 `op_sel` hazard: scripts/diag/opsel_min.hip beside the narrow halo convs launched by
a second host thread (the recipe of tests/test_neighbour_stress_gpu.py).
    python scripts/diag/opsel_min.py build      # CPU: compiles cta_gan_amd/_build/diag/libopsel_min.so (packed fp32 ON)
    python scripts/diag/opsel_min.py            # GPU: per instruction form, lanes whose packed and scalar chains differ"""
import ctypes
import os
import subprocess
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LIB = os.path.join(ROOT, "cta_gan_amd", "_build", "diag", "libopsel_min.so")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(ROOT, "scripts", "diag", "opsel_min.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", "--save-temps=obj", src, "-o", LIB],
                       capture_output=True, text=True, cwd=os.path.dirname(LIB))
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    print("built", LIB)


def main():
    import torch
    from cta_gan_amd import nets
    from cta_gan_amd.engine import ConvSpec
    import test_kernels_gpu as K
    nets.set_default_compute_dtype(torch.bfloat16)
    lib = ctypes.CDLL(LIB)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.opsel_launch.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp]
    lib.opsel_launch.restype = ci
    g = torch.Generator().manual_seed(3)
    patch = torch.randn(64 * 76, generator=g).cuda()
    w = (torch.randn(64 * 32 * 2, generator=g) * 0.05).cuda()
    sink = torch.zeros(4 + 2048 * 256 * 4, device="cuda")
    probes = []
    for (cin, cout, k, size, f32) in ((64, 32, 3, 128, False), (512, 2, 4, 63, True), (32, 2, 3, 256, True)):
        probes.append((K._make_probe(ConvSpec(cin, cout, k, 1, (k - 1) // 2, use_bias=True, out_f32=f32), None).cuda(),
                       torch.randn(16, cin, size, size, device="cuda")))
    stop, rounds = threading.Event(), [0]
    nb_stream, side = torch.cuda.Stream(), torch.cuda.Stream()

    def neighbours():
        torch.cuda.set_device(0)
        with torch.cuda.stream(nb_stream), torch.no_grad():
            while not stop.is_set():
                for p, px in probes:
                    p(px)
                rounds[0] += 1
                if rounds[0] % 8 == 0:
                    nb_stream.synchronize()

    def run(form, with_neighbours):
        bad = torch.zeros(64, dtype=torch.int32, device="cuda")
        with torch.cuda.stream(side):
            for _ in range(40):
                rc = lib.opsel_launch(form, 2048, 6, patch.data_ptr(), w.data_ptr(), bad.data_ptr(), sink.data_ptr(), side.cuda_stream)
                assert rc == 0, rc
        side.synchronize()
        b = bad.cpu().tolist()
        return sum(b), [i for i, v in enumerate(b) if v]

    for form, name in ((0, "op_sel:[1,0,0] (low half <- high register)"), (1, "op_sel_hi:[0,1,1] (high half <- low register)"),
                       (2, "materialised pair, no modifier")):
        print("alone       form %d %-48s mismatching halves %d, lanes %s" % ((form, name) + run(form, False)))
    th = threading.Thread(target=neighbours, daemon=True)
    th.start()
    while rounds[0] < 16:
        stop.wait(0.001)
    for rep in range(3):
        for form, name in ((0, "op_sel:[1,0,0] (low half <- high register)"), (1, "op_sel_hi:[0,1,1] (high half <- low register)"),
                           (2, "materialised pair, no modifier")):
            n, lanes = run(form, True)
            print("neighbours  form %d %-48s mismatching halves %d, lanes %s" % (form, name, n, lanes if len(lanes) < 24 else "%d..%d (%d lanes)" % (lanes[0], lanes[-1], len(lanes))))
    stop.set()
    th.join(30)
    torch.cuda.synchronize()


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else main()
