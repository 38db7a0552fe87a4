"""Data parallelism for the G+D step: one process per GPU, gradients all-reduced over RCCL/xGMI.

The reference is single-GPU (train.py:14).  Samples are independent on this path (InstanceNorm statistics
are per sample, every loss is a batch mean), so with equal per-rank batches the averaged gradient IS the
global-batch gradient.

Exchange: every optimiser group owns PERSISTENT flat fp32 buckets (`GradSync`).  The weight-gradient kernels write
straight into views of a bucket (`grad_buffer`, called by engine.py where it used to allocate `empty_like(weight)`),
autograd adopts those views as `.grad`, and a bucket is all-reduced IN PLACE the moment its last gradient has been
enqueued -- `torch.distributed`'s RCCL backend runs the collective on its own stream behind an event, so it overlaps
the rest of the backward pass.  HdGan G step, in backward order: {Reg} (8.2 MB; reduced while the whole generator
backward runs), {G: tail + residual blocks 5-8} (reduced behind blocks 0-4 + head), {G: blocks 0-4 + head};
D step: {D} (11 MB).  Nothing is concatenated or copied per step; the mean is taken by the collective (AVG) on RCCL.
xGMI is point-to-point (7 links per GPU): these messages are far below the step's compute time (65 MB against
~54 ms), so few large buckets beat many small ones.
"""
from __future__ import annotations

import os
import time
import weakref
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the launcher's environment (torch.distributed.run or bench.py's own spawner);
    initialises the default process group.  CTG_DP_BACKEND overrides the backend (tests: `gloo` with several ranks on
    one card; RCCL needs one GPU per rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("CTG_DP_BACKEND")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if torch.cuda.is_available():
            if backend == "nccl" and local >= torch.cuda.device_count():
                raise RuntimeError("RCCL needs one GPU per rank: LOCAL_RANK %d but %d device(s) visible"
                                   % (local, torch.cuda.device_count()))
            torch.cuda.set_device(local % torch.cuda.device_count())
        if backend == "nccl":
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


# CTG_DP_FORCE=1: run the whole exchange (process group, buckets, in-backward all-reduce) even with ONE rank -- the only way to
# execute the RCCL code path on a single-GPU box (tests/test_bench_launcher.py); results must equal the plain run bit for bit.
_FORCE = bool(os.environ.get("CTG_DP_FORCE"))


_DRYRUN = bool(os.environ.get("CTG_DP_DRYRUN"))


class _NoWork:
    def wait(self):
        return True


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def enabled() -> bool:
    """Is a gradient exchange to be run?  (more than one rank, or CTG_DP_FORCE with an initialised group)"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def backend_name() -> str:
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else "none"


# ----------------------------------------------------------------------------------------------- gradient buckets
# id(param) -> (weak reference to its bucket, index in it): where that parameter's gradient lives.  Weak, and pruned when the
# owning GradSync dies: a dropped trainer frees its buckets (65 MB for the Hd step) and leaves no entry behind.
_SLOTS: Dict[int, tuple] = {}


def _prune_slots(entries):
    """entries: (id(param), the weak bucket reference registered for it) of a GradSync that is being collected."""
    for i, bref in entries:
        ent = _SLOTS.get(i)
        if ent is not None and ent[0] is bref:      # not re-registered by a younger exchange meanwhile
            del _SLOTS[i]


class _Bucket:
    def __init__(self, params: Sequence[torch.nn.Parameter], trigger):
        self.params = list(params)
        self.trigger = trigger          # (net, tag) mark of engine.fire_mark that completes this bucket, or None
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)     # dead-bias slots stay zero for ever
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.handed = set()             # parameters whose slot was given out in this step
        self.work = None                # in-flight collective
        self.owner = None               # weak reference to the GradSync this bucket belongs to
        self.live_trigger = False       # set per step by GradSync.begin(): may this bucket be launched from its mark?

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)


class GradSync:
    """The gradient exchange of ONE optimiser step: an ordered list of buckets (backward order) over the step's
    parameters.  `begin()` before the backward pass, `finish()` before `optimizer.step()`.

    buckets: list of (parameter list, trigger) with trigger = (network, tag) -- the `engine.fire_mark` event after which
    every gradient of the bucket has been enqueued ("done" = end of that network's backward) -- or None for "reduce in
    finish()".  A trigger is only valid for a network traversed ONCE per backward (a second traversal adds to gradients
    whose bucket would already be in flight): the CycleGAN generators use None.  Parameters must be contiguous fp32.

    What the in-place fast path relies on, and what happens otherwise: a slot is handed to the kernels only while the
    parameter's `.grad` is None (`zero_grad(set_to_none=True)`, the trainers' setting), so autograd ADOPTS the view.  A
    parameter that still carries a `.grad` at `begin()` (gradient accumulation, `set_to_none=False`) gets a fresh tensor from
    the kernels, autograd accumulates into the old `.grad`, and -- because that in-place add would race with an all-reduce
    already in flight -- the bucket's trigger is ignored for that step: it is packed and reduced in `finish()`."""

    def __init__(self, buckets):
        self.buckets: List[_Bucket] = []
        for params, trigger in buckets:
            params = [p for p in params if p.requires_grad]
            if params:
                self.buckets.append(_Bucket(params, trigger))
        entries = []
        for b in self.buckets:
            b.owner = weakref.ref(self)
            bref = weakref.ref(b)
            for i, p in enumerate(b.params):
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("GradSync: parameters must be contiguous fp32")
                _SLOTS[id(p)] = (bref, i)
                entries.append((id(p), bref))
        weakref.finalize(self, _prune_slots, entries)
        self.active = False
        self.last_stray = 0          # gradients of the last finish() that were NOT exchanged in their bucket slot

    def begin(self):
        for b in self.buckets:
            b.handed.clear()
            b.work = None
            # a bucket may start its all-reduce from inside the backward only if every gradient of it will be WRITTEN
            # into its slot by the kernels and adopted by autograd, i.e. no parameter still carries a .grad
            b.live_trigger = b.trigger is not None and all(p.grad is None for p in b.params)
        self.active = True
        _ACTIVE.add(self)

    def _launch(self, b: _Bucket):
        if b.work is not None:
            return
        if _DRYRUN:      # measurement aid: everything but the collective itself
            b.work = _NoWork()
            return
        if dist.get_backend() == "nccl":
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, async_op=True)

    def on_mark(self, net, tag):
        """engine.fire_mark: the gradients produced so far by `net`'s backward are enqueued on the current stream."""
        if not self.active:
            return
        for b in self.buckets:
            if b.live_trigger and b.trigger[0] is net and b.trigger[1] == tag:
                self._launch(b)

    def finish(self):
        """Wait for every bucket (launching the ones without a trigger); afterwards each `.grad` holds the mean over
        ranks.  A gradient autograd did NOT leave in its bucket slot (it cloned instead of adopting the view, or summed
        two uses of a shared network) is detected by address and exchanged on a slow path, so the result never depends
        on autograd's buffer-stealing rules."""
        self.active = False
        _ACTIVE.discard(self)
        world = world_size()
        stray = []
        for b in self.buckets:
            late = b.work is None
            if late:
                # no overlap asked for: pack what is not in place (one multi-tensor copy), then reduce
                views, srcs = [], []
                for i, p in enumerate(b.params):
                    if p.grad is not None and p.grad.data_ptr() != b.flat.data_ptr() + 4 * b.offsets[i]:
                        v = b.view(i)
                        views.append(v)
                        srcs.append(p.grad if p.grad.dtype == torch.float32 else p.grad.float())
                        p.grad = v
                if views:
                    torch._foreach_copy_(views, srcs)
                self._launch(b)
            if WAIT_LOG is not None and b.flat.is_cuda:
                # bench.py (N > 1): how long the optimiser's stream (HIP events) and the host stall for this bucket's collective
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
                b.work.wait()
                e1.record()
                WAIT_LOG.append((e0, e1, time.perf_counter() - t0))
            else:
                b.work.wait()
            b.work = None
            if dist.get_backend() != "nccl":
                b.flat.mul_(1.0 / world)
            if not late:
                for i, p in enumerate(b.params):
                    if p.grad is not None and p.grad.data_ptr() != b.flat.data_ptr() + 4 * b.offsets[i]:
                        stray.append(p)
        if stray:
            flat = torch.cat([p.grad.reshape(-1).float() for p in stray])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / world)
            off = 0
            for p in stray:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self.last_stray = len(stray)
        return len(stray)


_ACTIVE: set = set()
# bench.py sets this to a list for the timed steps of a multi-rank run: (start event, end event, host seconds) around every
# bucket's wait in finish() -- the time a step loses to its gradient exchange, per rank, so that a scaling loss is attributable
WAIT_LOG = None


def grad_buffer(param: torch.Tensor) -> Optional[torch.Tensor]:
    """A fresh view of `param`'s bucket slot for the kernels to write its gradient into (once per step), or None when the
    parameter belongs to no active exchange / its slot is taken (a second use of a shared network in the same backward)."""
    slot = _SLOTS.get(id(param))
    if slot is None:
        return None
    bref, i = slot
    b = bref()
    if b is None:                       # the exchange this entry belonged to is gone (id() of a dead parameter re-used)
        del _SLOTS[id(param)]
        return None
    owner = b.owner() if b.owner is not None else None
    if b.params[i] is not param or i in b.handed or b.work is not None or owner is None or not owner.active:
        return None
    if param.grad is not None:
        # autograd would ADD the kernels' result to the old .grad instead of adopting it: if that .grad is last step's view
        # of this very slot, the slot would be added to itself.  Hand out nothing; finish() packs and reduces the sum.
        return None
    b.handed.add(i)
    return b.view(i)


def wants_mark(net, tag) -> bool:
    """Does an active exchange have a bucket that this mark completes?  (engine.fire_mark only flushes the queued split-K
    reductions in the middle of a backward when somebody is waiting for them)"""
    return any(b.live_trigger and b.trigger[0] is net and b.trigger[1] == tag and b.work is None
               for s in _ACTIVE for b in s.buckets)


def fire_mark(net, tag):
    for s in list(_ACTIVE):
        s.on_mark(net, tag)


_LEGACY: Dict[tuple, GradSync] = {}


def allreduce_grads(params: Iterable[torch.nn.Parameter]) -> None:
    """Average `.grad` of `params` across ranks through one persistent flat bucket (created on first use; from the
    next step on the kernels write into it directly); no-op for world size 1.  For steps without overlap (the CycleGAN /
    pix2pix trainers, whose generators are traversed twice per backward)."""
    if not enabled():
        return
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return
    key = tuple(id(p) for p in ps)
    sync = _LEGACY.get(key)
    if sync is None:
        sync = _LEGACY[key] = GradSync([(ps, None)])
    sync.finish()
    sync.begin()        # slots are handed out again in the next backward


def broadcast_params(*modules) -> None:
    """Make every rank start from rank 0's weights (replicas of a data-parallel job must be identical; the reference,
    being single-GPU, never needed this).  One flat broadcast per call; no-op for world size 1."""
    if not enabled():
        return
    ps = [p for m in modules for p in m.parameters()]
    if not ps:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
        dist.broadcast(flat, src=0)
        off = 0
        for p in ps:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            p._ctg_version = getattr(p, "_ctg_version", 0) + 1
            off += n


def barrier():
    if enabled():
        dist.barrier()
