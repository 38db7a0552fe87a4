"""Three Hd stage-2 steps (correlation weights 0: no float atomics, so the step is bit-reproducible) -> sha256 of the losses
and of every weight.  Run once plainly and once with CTG_DP_FORCE=1 (a ONE-rank RCCL process group: buckets written in place,
all-reduced with ReduceOp.AVG from inside the backward on RCCL's stream): the two digests must be equal.
tests/test_bench_launcher.py runs both as child processes."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cta_gan_amd import dp, nets, synth
from cta_gan_amd.trainer import Hd_Trainer_x2
dp.init_from_env()
torch.cuda.set_device(0)
MODE = os.environ.get("DPCHECK_MODE", "bf16")      # bf16 | bf16x3 | bf16x3f
nets.set_default_compute_dtype(torch.bfloat16 if MODE == "bf16" else MODE)
cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=2, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=0, Corr_lamda2=0,
           Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1, hip_graph=False)
tr = Hd_Trainer_x2(cfg)
synth.fill_module(tr.netG_A2B, seed=0); synth.fill_module(tr.netD_B, seed=1); synth.fill_module(tr.R_A, seed=4)
h = hashlib.sha256()
for i in range(3):
    batch = {k: synth.synth_smooth_images("dpf%d_%s" % (i, k), 2, 256).cuda() for k in ("A2", "B1", "B2")}
    out = tr.train_step(batch, sync_losses=True)
    h.update(repr(sorted(out.items())).encode())
for m in (tr.netG_A2B, tr.R_A, tr.netD_B):
    for p in m.parameters():
        h.update(p.detach().cpu().numpy().tobytes())
sync = tr._grad_sync()
print("DPCHECK backend=%s enabled=%s buckets=%s stray=%s digest=%s" % (
    dp.backend_name(), dp.enabled(), None if sync is None else len(sync["G"].buckets),
    None if sync is None else sync["G"].last_stray + sync["D"].last_stray, h.hexdigest()))
if dp.enabled():
    torch.distributed.destroy_process_group()
