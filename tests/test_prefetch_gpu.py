"""GPU: `Model.HdGan.DataPrefetcher` (reference Model/HdGan.py:11-47) -- pinned, double-buffered
H2D one batch ahead, consumed by the trainers' `train()`."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _batches(n, b, s, ragged_last=True):
    out = []
    for i in range(n):
        bb = 1 if (ragged_last and i == n - 1) else b
        g = torch.Generator().manual_seed(100 + i)
        out.append({"A2": torch.rand(bb, 1, s, s, generator=g), "B1": torch.rand(bb, 1, s, s, generator=g),
                    "B2": torch.rand(bb, 1, s, s, generator=g), "meta": {"idx": i}})
    return out


def test_batches_arrive_intact_in_order_with_a_ragged_tail():
    from cta_gan_amd.Model.HdGan import DataPrefetcher
    src = _batches(5, 3, 64)
    keep = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for b in src]
    pf = DataPrefetcher(src)
    seen = 0
    for i, batch in enumerate(pf):
        assert batch["meta"] == {"idx": i}
        for k in ("A2", "B1", "B2"):
            assert batch[k].is_cuda and torch.equal(batch[k].cpu(), keep[i][k]), (i, k)
        seen += 1
    assert seen == 5 and pf.next() is None
    # the staging buffers are page-locked and there are exactly two slots
    assert all(t.is_pinned() for slot in pf._stage for t in slot.values()) and len(pf._stage) == 2


def test_h2d_of_the_next_batch_overlaps_compute():
    """The loop `batch = pf.next(); step(batch)` as the trainers run it: while the GPU computes on batch i, `next()` stages
    batch i+1 into the page-locked slot and enqueues its H2D on the copy stream.  Checked on the GPU timeline with events:
    every copy runs inside the loop's compute (the host is ahead of the device, so it may be the previous step's), the loop
    takes no longer than its compute alone (+10 %), and `next()` returns in a fraction of a step (the host never blocks on
    the device: with a pageable source the copy would be synchronous)."""
    from cta_gan_amd.Model.HdGan import DataPrefetcher
    n = 4 * 1024 * 1024                     # 3 x 16 MiB per batch: a B=16 batch of 512x512 fp32 slices
    src = [{"A2": torch.rand(n), "B1": torch.rand(n), "B2": torch.rand(n)} for _ in range(5)]
    x = torch.rand(16 * 1024 * 1024, device="cuda")

    def compute(reps):
        y = x
        for _ in range(reps):
            y = y * 1.0001 + 0.5
        return y

    compute(20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); compute(200); e1.record(); torch.cuda.synchronize()
    reps = int(200 * 60.0 / e0.elapsed_time(e1))        # ~60 ms of GPU work per "step" (the bench step is 53 ms)
    pf = DataPrefetcher(src)
    batch = pf.next()
    compute(reps)                           # warm the allocator for the loop's temporaries
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record(); compute(3 * reps); g1.record(); torch.cuda.synchronize()
    t_compute_only = g0.elapsed_time(g1)
    rec = []
    first, last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    first.record()
    for i in range(3):
        compute(reps)
        t0 = time.perf_counter()
        batch = pf.next()
        rec.append(pf.copy_events + ((time.perf_counter() - t0) * 1e3,))
    last.record()
    torch.cuda.synchronize()
    t_loop = first.elapsed_time(last)
    print("3 steps of compute alone %.1f ms; with the prefetcher feeding them %.1f ms" % (t_compute_only, t_loop))
    for cs, ce, t_next in rec:
        lead, tail, t_copy = first.elapsed_time(cs), ce.elapsed_time(last), cs.elapsed_time(ce)
        print("copy of the next batch ran %.1f..%.1f ms into the %.1f ms loop (%.2f ms, %.1f GB/s); next() took %.1f ms"
              % (lead, lead + t_copy, t_loop, t_copy, 3 * n * 4 / t_copy / 1e6, t_next))
        assert lead > 0 and tail > 0, (lead, tail)          # every H2D ran while the compute of the loop was in flight
        assert t_next < 0.5 * t_compute_only / 3, (t_next, t_compute_only)      # the host never waited for the device
    assert t_loop < 1.1 * t_compute_only, (t_loop, t_compute_only)              # ... and the copies cost the loop nothing
    assert batch["A2"].is_cuda


def test_trainer_train_consumes_host_batches_through_the_prefetcher():
    from cta_gan_amd import nets
    from cta_gan_amd.trainer import Hd_Trainer_x2
    nets.set_default_compute_dtype(torch.bfloat16)
    try:
        cfg = dict(input_nc=1, output_nc=1, size=256, batchSize=2, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20,
                   Corr_lamda2=2, Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=0)
        tr = Hd_Trainer_x2(cfg)
        loader = [{k: v * 2 - 1 for k, v in b.items() if k != "meta"} for b in _batches(3, 2, 256)]
        tr.train(loader)
        assert tr.last["fake_B"].shape[0] == 1          # the ragged trailing batch was the last one trained
        assert all(torch.isfinite(p).all() for p in tr.netG_A2B.parameters())
    finally:
        nets.set_default_compute_dtype(torch.float32)
