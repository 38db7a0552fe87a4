"""Does the benchmarked precision TRAIN like the reference's fp32?  The same 120 HdGan stage-2 steps (B=4, 256^2, the same batches,
the same initial weights) in the four compute modes; windowed medians of the loss terms side by side.  Single steps are chaotic
(tests/test_step_parity_gpu.py), the windowed medians are not: bf16x3, bf16x3f and bf16 must stay within 15 % of the fp32 run.
python scripts/precision_soak.py [steps]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B, S = 4, 256
runs = {}
_batches = {}
MODES = ("fp32", "bf16x3", "bf16x3f", "bf16")
for mode in MODES:
    nets.set_default_compute_dtype({"fp32": torch.float32, "bf16": torch.bfloat16}.get(mode, mode))
    cfg = dict(input_nc=1, output_nc=1, size=S, batchSize=B, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
               Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=False)
    tr = Hd_Trainer_x2(cfg)
    synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
    hist = []
    for i in range(N):
        if i % 16 not in _batches:
            _batches[i % 16] = {k: synth.synth_smooth_images("ps%d_%s" % (i % 16, k), B, S).cuda() for k in ("A2", "B1", "B2")}
        batch = _batches[i % 16]
        out = tr.train_step(batch, sync_losses=True)
        assert all(v == v and abs(v) < 1e6 for v in out.values()), (mode, i, out)
        hist.append(out)
    runs[mode] = hist
    del tr
    torch.cuda.empty_cache()
print("window      " + "".join("%-44s" % k for k in ("SR (fp32 / bf16x3 / bf16x3f / bf16)", "total", "SM", "loss_D")))
for w0 in range(0, N - 19, 20):
    row = "%3d-%3d   " % (w0, w0 + 19)
    for k in ("SR", "total", "SM", "loss_D"):
        m = [statistics.median(h[k] for h in runs[mode][w0:w0 + 20]) for mode in MODES]
        row += "%-44s" % ("%.4f / %.4f / %.4f / %.4f" % tuple(m))
        if k in ("SR", "total"):
            assert all(abs(v - m[0]) <= 0.15 * abs(m[0]) for v in m[1:]), (w0, k, m)
    print(row)
print("precision soak ok: %d steps, windowed medians of SR / total of bf16x3, bf16x3f and bf16 within 15 %% of fp32" % N)
