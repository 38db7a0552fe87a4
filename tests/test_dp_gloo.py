"""CPU, world_size 2, gloo: the data-parallel exchange of the G+D step (cta_gan_amd/dp.py).

(0) `GradSync`: gradients written by a backward pass straight into the persistent flat buckets (`grad_buffer`), buckets
    all-reduced from `fire_mark` in the middle of that backward, `.grad` adopted in place (no stray copies), a network
    traversed twice in one backward handled through the slow path -- all with the right averages;
(1) `allreduce_grads` averages `.grad` across ranks through ONE persistent flat bucket and leaves views behind;
(2) DP == big batch: with per-sample InstanceNorm and batch-mean losses, the average of the two half-batch
    gradients equals the full-batch gradient (checked on the oracle discriminator, CPU fp32);
(3) `broadcast_params` makes differently-initialised replicas identical (rank 0's weights)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _ToyNet:
    """Stands in for a HipNet: its backward writes every parameter gradient where `dp.grad_buffer` says (as engine.py's
    weight-gradient launches do) and fires the "mid" / "done" marks."""

    def __init__(self, shapes):
        self.params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        self.launched_at = []


class _ToyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, scale, x, *params):
        ctx.net, ctx.scale = net, scale
        return x.sum() * 0 + sum((p * 0).sum() for p in params) + x.sum()

    @staticmethod
    def backward(ctx, gout):
        from cta_gan_amd import dp
        net = ctx.net
        grads = []
        half = len(net.params) // 2
        for i, p in reversed(list(enumerate(net.params))):      # backward order: last parameter first
            buf = dp.grad_buffer(p)
            g = buf if buf is not None else torch.empty_like(p)
            g.copy_(torch.full_like(p, ctx.scale * (i + 1)) * gout)
            grads.append((i, g))
            if i == half:
                dp.fire_mark(net, "mid")
        dp.fire_mark(net, "done")
        grads = dict(grads)
        return (None, None, torch.ones(()).expand(1) * 0 + gout.expand(1)) + tuple(grads[i] for i in range(len(net.params)))


def _grad_sync_checks(rank, world):
    from cta_gan_amd import dp
    net = _ToyNet([(3, 4), (5,), (2, 2), (7,)])
    late, early = net.params[2:], net.params[:2]
    sync = dp.GradSync([(late, (net, "mid")), (early, (net, "done"))])
    flats = [b.flat for b in sync.buckets]
    for step in range(2):
        for p in net.params:
            p.grad = None
        sync.begin()
        x = torch.ones(1, requires_grad=True)
        _ToyFn.apply(net, float(rank + 1 + step), x, *net.params).backward()
        # both collectives were launched from inside the backward, the late bucket first
        assert all(b.work is not None for b in sync.buckets)
        assert sync.finish() == 0                   # every gradient was adopted in its slot: nothing to pack or re-send
        for i, p in enumerate(net.params):
            want = (i + 1) * (sum(r + 1 + step for r in range(world)) / world)
            assert torch.allclose(p.grad, torch.full_like(p, want)), (step, i, p.grad, want)
        for b in sync.buckets:
            for i, p in enumerate(b.params):
                assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * b.offsets[i]
        assert [b.flat.data_ptr() for b in sync.buckets] == [f.data_ptr() for f in flats]     # persistent
    # a network traversed twice in one backward (the CycleGAN generators): autograd sums the two contributions itself, so
    # the bucket has no trigger (a mark of the first traversal would be premature) and is reduced in finish()
    sync = dp.GradSync([(net.params, None)])
    for p in net.params:
        p.grad = None
    sync.begin()
    x = torch.ones(1, requires_grad=True)
    (_ToyFn.apply(net, 1.0 + rank, x, *net.params) + _ToyFn.apply(net, 10.0, x, *net.params)).backward()
    sync.finish()
    for i, p in enumerate(net.params):
        want = (i + 1) * (sum(r + 1.0 for r in range(world)) / world + 10.0)
        assert torch.allclose(p.grad, torch.full_like(p, want)), (i, p.grad, want)


def _grad_accumulation_checks(rank, world):
    """`.grad` NOT reset between two backward passes (zero_grad(set_to_none=False) / gradient accumulation): after step one
    every `.grad` is a view of its bucket slot; handing that slot to the kernels again would make autograd add the slot to
    itself (a doubled gradient), and a triggered bucket's in-place accumulate would race with its all-reduce.  Expected:
    no slot is handed out, no collective starts inside the backward, finish() reduces the accumulated sum."""
    import gc
    from cta_gan_amd import dp
    net = _ToyNet([(3, 4), (5,), (2, 2), (7,)])
    sync = dp.GradSync([(net.params[2:], (net, "mid")), (net.params[:2], (net, "done"))])
    sync.begin()
    x = torch.ones(1, requires_grad=True)
    _ToyFn.apply(net, float(rank + 1), x, *net.params).backward()
    assert sync.finish() == 0
    mean1 = sum(r + 1 for r in range(world)) / world
    # second backward WITHOUT clearing .grad: accumulate (i + 1) * 5 on every rank on top of the averaged first step
    sync.begin()
    assert not any(b.live_trigger for b in sync.buckets)
    assert all(dp.grad_buffer(p) is None for p in net.params)
    _ToyFn.apply(net, 5.0, x, *net.params).backward()
    assert all(b.work is None for b in sync.buckets)          # nothing was launched from inside the backward
    sync.finish()
    for i, p in enumerate(net.params):
        want = (i + 1) * (mean1 + 5.0)
        assert torch.allclose(p.grad, torch.full_like(p, want)), (i, p.grad, want)
    # a dropped exchange leaves no entry (and no bucket memory) behind
    ids = [id(p) for p in net.params]
    assert all(i in dp._SLOTS for i in ids)
    del sync
    gc.collect()
    assert not any(i in dp._SLOTS for i in ids)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from cta_gan_amd import dp, synth
    from oracle import ref_models
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dp.world_size() == world
    _grad_sync_checks(rank, world)
    _grad_accumulation_checks(rank, world)
    # (1) plain averaging through the flat bucket, None grads skipped
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
    ps[0].grad = torch.full((3, 4), float(rank + 1))
    ps[1].grad = torch.arange(5.0) * (rank + 1)
    dp.allreduce_grads(ps)
    assert torch.allclose(ps[0].grad, torch.full((3, 4), 1.5)) and torch.allclose(ps[1].grad, torch.arange(5.0) * 1.5)
    assert ps[2].grad is None
    # second step through the same persistent bucket
    ps[0].grad = torch.full((3, 4), 2.0 * (rank + 1))
    ps[1].grad = None
    dp.allreduce_grads(ps)
    assert torch.allclose(ps[0].grad, torch.full((3, 4), 3.0)) and ps[1].grad is None
    # (2) DP == big batch on the oracle discriminator
    D = synth.fill_module(ref_models.Discriminator(1), seed=1)
    x = synth.synth_images("dp_x", 4, 32)
    loss = ((D(x[2 * rank:2 * rank + 2]) - 1.0) ** 2).mean()
    loss.backward()
    dp.allreduce_grads(D.parameters())
    if rank == 0:
        Dfull = synth.fill_module(ref_models.Discriminator(1), seed=1)
        ((Dfull(x) - 1.0) ** 2).mean().backward()
        for (k, p), q in zip(D.named_parameters(), Dfull.parameters()):
            if k in ("model.2.bias", "model.5.bias", "model.8.bias"):
                continue  # dead biases: rounding noise on both sides
            assert torch.allclose(p.grad, q.grad, rtol=2e-4, atol=1e-7), k
    # (3) replicas start identical: rank-dependent init is overwritten by rank 0's weights
    torch.manual_seed(100 + rank)
    M = ref_models.Discriminator(1)
    dp.broadcast_params(M)
    flat = torch.cat([p.detach().reshape(-1) for p in M.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    dp.barrier()
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_dp_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
