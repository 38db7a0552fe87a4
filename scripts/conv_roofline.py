"""Per-launch roofline of every conv-class launch of one training step (HIP events around each launch, cta_gan_amd.ops.OP_LOG):

    python scripts/conv_roofline.py [--dtype bf16|bf16x3] [--min-us 100] > profiles/rNN_conv_roofline_<dtype>.md

For every (launch label) of the benchmarked Hd step (B=16, 512^2): launches per step, FLOP and algorithmic HBM bytes of ONE
launch (every operand tensor once; computed by the wrappers in cta_gan_amd/ops.py from the call's own shapes), average launch
time, and the fraction of its roofline: max(FLOP / MFMA peak, bytes / 8 TB/s) / time."""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# Per-launch times are only meaningful for launches that have the chip to themselves: the step runs its adversarial branch on a
# second stream beside Reg (trainer/HdTrainer.py side_branch), and launches that overlap there read up to 2x long (the PatchGAN's
# 256 -> 512 layer: 210 us alone, 442 us beside Reg's kernels).  This table therefore runs the step on ONE stream unless
# --two-streams is given; the step itself is ~2 % slower that way.
if "--two-streams" not in sys.argv:
    os.environ["CTG_NO_SIDE_STREAM"] = "1"
import torch
import bench
from cta_gan_amd import nets, ops, synth, build
from cta_gan_amd.trainer import Hd_Trainer_x2

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--min-us", type=float, default=100.0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--two-streams", action="store_true", help="keep the adversarial branch on the second stream (overlapped launches read long)")
args = ap.parse_args()
nets.set_default_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else args.dtype)
peak = bench.PEAK_TFLOPS[args.dtype] * 1e12
torch.manual_seed(42)
tr = Hd_Trainer_x2(dict(bench.YAML_HD, size=512, batchSize=16))
batch = {k: synth.synth_images("bench_%s_r0" % k, 16, 512).cuda() for k in ("A2", "B1", "B2")}
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
ops.OP_LOG = []
for _ in range(args.steps):
    tr.train_step(batch)
torch.cuda.synchronize()
log, ops.OP_LOG = ops.OP_LOG, None
agg = collections.OrderedDict()
for label, flop, nbytes, e0, e1 in log:
    a = agg.setdefault(label, [0, 0.0, flop, nbytes])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
print("# Per-launch conv roofline, Hd step B=16 @ 512^2, %s (build %s)\n" % (args.dtype, build._digest()[:16]))
print("%s.  " % ("Adversarial branch on the second stream: launches of the PatchGAN and of Reg overlap and read long" if args.two_streams
                 else "The step on ONE stream (CTG_NO_SIDE_STREAM=1), so that every launch has the chip to itself"), end="")
print("HIP events around every conv-class launch inside %d training steps (`scripts/conv_roofline.py`); bound = the larger of FLOP / "
      "%.0f TFLOP/s and algorithmic bytes / 8 TB/s; launches >= %.0f us.  (An event pair costs the stream a few us: short "
      "launches read slightly long.)\n" % (args.steps, peak / 1e12, args.min_us))
print("| launch | n/step | GFLOP | MB | us | TFLOP/s | TB/s | bound | fraction of its bound |\n|---|---|---|---|---|---|---|---|---|")
rows = []
for label, (n, ms, flop, nbytes) in agg.items():
    us = 1e3 * ms / n
    t_m, t_h = flop / peak, nbytes / 8e12
    rows.append((ms / args.steps, label, n / args.steps, flop, nbytes, us, "mfma" if t_m >= t_h else "hbm", max(t_m, t_h) / (us * 1e-6)))
tot = sum(r[0] for r in rows)
for msstep, label, n, flop, nbytes, us, bound, frac in sorted(rows, reverse=True):
    if us >= args.min_us:
        print("| %s | %.4g | %.1f | %.1f | %.1f | %.0f | %.2f | %s | %.2f |" % (label, n, flop / 1e9, nbytes / 1e6, us, flop / us / 1e6,
                                                                           nbytes / us / 1e6, bound, frac))
print("\nConv-class launches: %.2f ms per step in total (all sizes); rows under %.0f us: %.2f ms." % (
    tot, args.min_us, sum(r[0] for r in rows if r[5] < args.min_us)))
