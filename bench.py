#!/usr/bin/env python3
"""Benchmark of the CTA-GAN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU, RCCL gradient all-reduce.  Either launched by `python -m torch.distributed.run
     --nproc-per-node N bench.py --gpus N ...`, or run directly: the parent process then starts the N ranks itself as
     fresh child processes BEFORE touching the GPU and exits with their status.)

Workload (BASELINE.json configs[2] / [4]): one full HdGan stage-2 G+D training step
(trainer/HdTrainer.py:705-751: G fwd, Reg fwd, STN, D fwd, all losses, backward through D/STN/Reg/G,
Adam on R and G, second G fwd, 2x D fwd+bwd, Adam on D) on synthetic paired 512x512 slices already
resident in HBM, 16 slices per GPU, bf16 storage/MFMA with fp32 accumulation (weak scaling over N).
Prints ONE JSON line on rank 0: paired slices/s (whole job), the roofline of the dominant kernel family
(the three kernels of the 256->256 3x3 convs of the residual blocks -- forward, backward-data with its fused
epilogue, weight gradient -- each timed live with HIP events on its launch stream inside the timed steps), the HBM side of
the roofline (`roofline.hbm`: the InstanceNorm elementwise kernels on those 256-channel maps and Reg's 32-channel
full-resolution convs, timed the same way, GB/s against 8 TB/s), the CPU baseline (the oracle's torch-CPU restatement of
the same step on a bounded sample, timed on this box's host cores), the second half of BASELINE.json's metric -- the
generator output's rel-L2 against that CPU oracle in the timed precision (`gen_rel_l2`) -- and, at N=1, a `parity_mode`
leg: the same step timed again in the split-bf16 mode ("bf16x3"), the fastest mode inside the north_star's 1e-3.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3,   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
               "bf16x3": 2500.0 / 3,              # split-bf16: three bf16 MFMAs per fp32-grade product
               "bf16x3f": 1500.0}                 # split forward (3 MFMAs per product), bf16 backward (1): with a third of the products in the
                                                  # forward, 3 units of work take 3 + 1 + 1 = 5 MFMA units -> 2.5 PF x 3 / 5 (step-level only:
                                                  # a run with --dtype bf16x3f prices its per-kernel rows against this one figure)
# what the build EXECUTES where it differs from the survey's count: the G step's backward through the frozen discriminator owes
# no weight gradient (HdTrainer.py:242-248 zeroes optimizer_D_B afterwards; the HIP path never computes it): -25.434 GF per slice
GFLOP_EXECUTED = {"hd": 1982.6 - 25.434, "reg": 1982.6 - 25.434}
GFLOP_PER_SLICE = {"hd": 1982.6, "gen": 389.835, "cyc": 5135.8,   # SURVEY.md §8d / BASELINE.md §2 at 512x512
                   # SURVEY.md §8f rank 4, same per-network figures: P2p = 4 G + 8 D traversals (G fwd/bwd-data/bwd-weight +
                   # no-grad fwd; D fwd + bwd-data in the G step, 2 x (fwd + bwd-data + bwd-weight) in the D step);
                   # Reg = the stage-1 CTA-GAN step = the Hd figure (3 (G + Reg + D) + G + 6 D)
                   "p2p": 4 * 389.835 + 8 * 25.434, "reg": 1982.6}
KERNEL_NAMES = {"fwd": "conv_halo_kernel<bf16,BN=128,FUSE=0,KWC=3> (forward)",
                "fwd_in": "conv_halo_kernel<bf16,BN=128,KWC=3,NIE=1> (no-grad forward: conv + InstanceNorm + ReLU / skip in one launch)",
                "bwd_data": "conv_halo_kernel<bf16,BN=128,FUSE=1,KWC=3> (backward-data + fold/residual/IN-sum epilogue)",
                "wgrad": "conv_wgrad_halo_kernel<64,64,9> (weight gradient)"}
PEAK_HBM_GBS = 8000.0                                 # HBM3E, MI355X_MICROARCH.md
HBM_KERNELS = {   # ops.KERNEL_EVENTS key -> what it times (cta_gan_amd/ops.py brackets these shapes)
    "in_apply": "in_apply_kernel: ReLU(InstanceNorm(z)) on a residual block's [B,128,128,256] map (read z, write h)",
    "in_apply_res": "in_apply_kernel: InstanceNorm(z) + skip on a residual block's map (read z and skip, write x)",
    "in_bwd_apply": "in_bwd_apply_kernel: InstanceNorm backward, elementwise pass on that map (read z and g, write dz)",
    "conv32": "conv_strip32_kernel: the 32->32 channel 3x3 reflect convs of Reg's full-resolution residual blocks "
              "(read x, write y: 64 B per pixel each way)",
    "convt64": "conv_stript_128_64_kernel: ConvTranspose2d(128, 64, 3, s2) [B,256,256,128] -> [B,512,512,64] (u2 forward, d1 "
               "backward-data; 155 GFLOP per launch at B=16: read x, write y once)",
    "convs2": "conv_strips2_64_128_kernel: Conv2d(64, 128, 3, s2) [B,512,512,64] -> [B,256,256,128] (d1 forward, u2 "
              "backward-data; 155 GFLOP per launch at B=16: read x, write y once)"}
UNSHAPED_HBM_KEYS = ("conv32", "convt64", "convs2")
YAML_HD = dict(input_nc=1, output_nc=1, lr=1e-4, lrd=1e-4, Adv_lamda1=1, Corr_lamda1=20, Corr_lamda2=2,
               Smooth_lamda=10, epoch=0, n_epochs=1, decay_epoch=1)
YAML_P2P = dict(input_nc=1, output_nc=1, lr=1e-4, Adv_lamda=1, P2P_lamda=100, epoch=0, n_epochs=1, decay_epoch=1)
YAML_REG = dict(input_nc=1, output_nc=1, lr=1e-4, Adv_lamda=1, Corr_lamda=20, Smooth_lamda=10, epoch=0, n_epochs=1,
                decay_epoch=1)
YAML_CYC = dict(input_nc=1, output_nc=1, lr=1e-4, Adv_lamda=1, Cyc_lamda=10, epoch=0, n_epochs=1, decay_epoch=1)


def host_gpu_count(sysfs="/sys"):
    """GPUs of this HOST, which may be more than this process may see: the GPU nodes of the KFD topology, where a node whose
    properties this user may not read (another tenant's card on a shared 8-GPU host) still counts as a GPU; never less than the
    visible device count (counting devices does not initialise the GPU)."""
    try:
        visible = max(1, torch.cuda.device_count())
    except Exception:      # noqa: BLE001
        visible = 1
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    n = 0
    try:
        for d in os.listdir(base):
            if not d.isdigit():
                continue
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(base, d, "properties")) if len(ln.split()) >= 2)
                n += int(props.get("simd_count", "0")) > 0
            except OSError:
                n += 1
    except OSError:
        pass
    return max(visible, n)


def cpu_share():
    """Host threads of the CPU leg: ONE GPU's share of the cores this process may use -- len(sched_getaffinity) // GPUs of the host
    (`host_gpu_count`: on a shared 8-GPU host a one-GPU box still sees all 256 cores; 256 oneDNN threads on a 2-slice batch do not
    finish) -- so that the figure is what a rank of an N-GPU job has beside its card.  Returns (threads, how it was derived)."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gpus = host_gpu_count()
    n = max(1, allowed // gpus)
    return n, "%d allowed cores / %d GPU(s) on this host" % (allowed, gpus)


def cpu_baseline(workload: str, size: int):
    """The oracle (CPU restatement of the reference step in stock torch ops) timed on the host cores, one step on
    ONE paired slice (bounded sample: ~10-30 s of CPU work), after a tiny warm-up that spins up the thread pool."""
    from cta_gan_amd import synth
    from oracle import golden_cases, ref_steps
    ncpu, how = cpu_share()
    torch.set_num_threads(ncpu)
    ons = golden_cases.oracle_namespace()
    with torch.no_grad():
        ons.Generator(1, 1)(torch.zeros(1, 1, 32, 32))
    if workload == "gen":
        G = synth.fill_module(ons.Generator(1, 1), seed=0)
        x = synth.synth_images("cpu_x", 1, size)
        t0 = time.perf_counter()
        with torch.no_grad():
            G(x)
        dt = time.perf_counter() - t0
        sample = "1 generator forward, B=1 @ %dx%d" % (size, size)
    elif workload == "cyc":
        import itertools
        nets_ = dict(G_A2B=ons.Generator(1, 1), G_B2A=ons.Generator(1, 1), D_A=ons.Discriminator(1),
                     D_B=ons.Discriminator(1))
        opts = dict(G=ref_steps.make_adam(itertools.chain(nets_["G_A2B"].parameters(), nets_["G_B2A"].parameters())),
                    D_A=ref_steps.make_adam(nets_["D_A"].parameters()), D_B=ref_steps.make_adam(nets_["D_B"].parameters()))
        bufs = dict(A=ref_steps.ReplayBuffer(), B=ref_steps.ReplayBuffer())
        batch = dict(A=synth.synth_images("cpu_A", 1, size), B=synth.synth_images("cpu_B", 1, size))
        t0 = time.perf_counter()
        ref_steps.cyc_step(nets_, opts, bufs, batch)
        dt = time.perf_counter() - t0
        sample = "1 CycleGan step, B=1 @ %dx%d" % (size, size)
    elif workload in ("p2p", "reg"):
        if workload == "p2p":
            nets_ = dict(G=ons.Generator(1, 1), D=ons.Discriminator(2))
        else:
            nets_ = dict(G=ons.Generator(1, 1), D=ons.Discriminator(1), R=ons.Reg(size, size, 1, 1), T=ons.Transformer_2D())
        opts = {k: ref_steps.make_adam(nets_[k].parameters()) for k in nets_ if k != "T"}
        batch = dict(A=synth.synth_images("cpu_A", 2, size), B=synth.synth_images("cpu_B", 2, size))
        t0 = time.perf_counter()
        if workload == "p2p":
            ref_steps.p2p_step(nets_, opts, batch)
        else:
            ref_steps.reg_step(nets_, opts, batch, smooth_fn=ons.smooothing_loss)
        dt = time.perf_counter() - t0
        return {"value": round(2 / dt, 5), "unit": "paired slices/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "1 %s step, B=2 @ %dx%d (fp32, oneDNN)" % (workload, size, size), "seconds": round(dt, 2)}
    else:
        nets_ = dict(G=ons.Generator(1, 1), D=ons.Discriminator_m(1), R=ons.Reg(size, size, 1, 1), T=ons.Transformer_2D())
        opts = dict(G=ref_steps.make_adam(nets_["G"].parameters()), D=ref_steps.make_adam(nets_["D"].parameters()),
                    R=ref_steps.make_adam(nets_["R"].parameters()))
        nb, timed = 2, 2   # 1 warm-up + 2 timed steps of B=2: ~20 s on 16 cores, inside the 10-30 s the sample is meant to take
        batch = {k: synth.synth_images("cpu_" + k, nb, size) for k in ("A2", "B1", "B2")}
        times = []
        for _ in range(1 + timed):
            t0 = time.perf_counter()
            ref_steps.hd_step(nets_, opts, batch, stage=2, smooth_fn=ons.smooothing_loss, gan_loss=ons.GANLoss())
            times.append(time.perf_counter() - t0)
        dt = sum(times[1:]) / timed
        sample = "HdGan stage-2 G+D step, B=%d @ %dx%d (fp32, oneDNN): 1 warm-up step (%.1f s) + mean of %d timed steps" % (
            nb, size, size, times[0], timed)
        return {"value": round(nb / dt, 5), "unit": "paired slices/s", "cores": torch.get_num_threads(), "cores_how": how,
                "cores_present": os.cpu_count(), "kind": "port", "sample": sample, "seconds": round(sum(times), 2)}
    return {"value": round(1.0 / dt, 5), "unit": "paired slices/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": sample, "seconds": round(dt, 2)}


def gen_reference(size: int, nb: int = 2):
    """The checker half of `gen_rel_l2` (part of the CPU-baseline leg): the oracle's generator (stock fp32 torch ops on the host)
    with the deterministic synthetic weights, on `nb` synthetic slices.  Returns (input, reference output)."""
    from cta_gan_amd import synth
    from oracle import golden_cases
    ons = golden_cases.oracle_namespace()
    G = synth.fill_module(ons.Generator(1, 1), seed=0)
    x = synth.synth_images("bench_gen_l2", nb, size)
    with torch.no_grad():
        want = G(x)
    return x, want


def gen_rel_l2(ref, dev):
    """rel-L2 of the HIP generator (current compute mode, same synthetic weights) against the CPU oracle's output."""
    from cta_gan_amd import synth
    from cta_gan_amd.Model.HdGan import Generator
    x, want = ref
    G = synth.fill_module(Generator(1, 1), seed=0).to(dev)
    with torch.no_grad():
        got = G(x.to(dev)).float().cpu()
    from cta_gan_amd import ops
    # (a two-slice no-grad forward runs its residual blocks as fused conv + InstanceNorm launches, the B=16 step does not --
    #  ops.conv_in_fusable; the two forms differ by less than the last printed digit here.  No bounded wait may have run out.)
    assert ops.nie_failures() == 0
    return float((got - want).norm() / want.norm())


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent has made no
    GPU call and never execs), one per GPU, rendezvous on 127.0.0.1; rank 0's JSON line goes to our stdout.  Returns the
    first non-zero child status (the others are stopped), else 0."""
    import socket
    import subprocess
    from cta_gan_amd import _lib
    _lib.ensure_built()       # a stale library is rebuilt HERE, once, not by N ranks behind a file lock with the group up
    rc = 0
    for attempt in range(3):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    code = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:       # a rank died: the others would wait in a collective for ever
                        q.terminate()
        if rc != EXIT_PORT_TAKEN:           # somebody took the probed port before rank 0 bound it: pick another one
            break
    return rc


EXIT_PORT_TAKEN = 98      # a rank's exit status when the rendezvous port was already in use (errno EADDRINUSE)


def _cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every HIP device, in HIP's device order, read from the KFD topology WITHOUT touching the GPU: the GPU nodes of
    /sys/class/kfd/kfd/topology/nodes in node order (ROCr's agent order = HIP's), each mapped through its `drm_render_minor` to
    /sys/class/drm/renderD<minor>/device/numa_node; HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES (integer lists) select and reorder.
    An entry is None where the node is unknown (-1: a single-node host); [] when there is no KFD topology (no GPU driver)."""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        names = sorted((d for d in os.listdir(base) if d.isdigit()), key=int)
    except OSError:
        return []
    nodes = []
    for d in names:
        try:
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, d, "properties")) if len(ln.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) <= 0:      # a CPU node
            continue
        numa = None
        try:
            minor = int(props.get("drm_render_minor", "-1"))
            v = int(open(os.path.join(sysfs, "class/drm/renderD%d/device/numa_node" % minor)).read().strip())
            numa = v if v >= 0 else None
        except (OSError, ValueError):
            pass
        nodes.append(numa)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        vis = os.environ.get(var)
        if vis:
            try:
                nodes = [nodes[int(i)] for i in vis.split(",") if i.strip() != "" and int(i) < len(nodes)]
            except ValueError:
                pass          # UUID lists: keep the topology order
            break
    return nodes


def core_order(allowed, sysfs="/sys"):
    """The allowed logical CPUs ordered (NUMA node, physical core, sibling): SMT siblings are adjacent and a node's cores form one
    run, whatever the machine's numbering (on the usual two-socket numbering 0-63 | 64-127 | siblings 128-191 | 192-255 a plain
    ascending deal would hand two ranks the hyperthread siblings of the same physical cores).  Returns [(node, cpu), ...]."""
    node_of = {}
    try:
        for d in os.listdir(os.path.join(sysfs, "devices/system/node")):
            if d.startswith("node") and d[4:].isdigit():
                try:
                    for c in _cpulist(open(os.path.join(sysfs, "devices/system/node", d, "cpulist")).read()):
                        node_of[c] = int(d[4:])
                except OSError:
                    pass
    except OSError:
        pass
    key = {}
    for c in allowed:
        first = c
        try:
            first = min(_cpulist(open(os.path.join(sysfs, "devices/system/cpu/cpu%d/topology/thread_siblings_list" % c)).read()))
        except (OSError, ValueError):
            pass
        key[c] = (node_of.get(c, 0), first, c)
    return [(key[c][0], c) for c in sorted(allowed, key=lambda c: key[c])]


def plan_rank_cores(local, world, allowed, gpu_nodes, order):
    """Cores of local rank `local` of `world`: whole physical cores on ITS GPU's NUMA node (the ranks whose GPUs share a node split
    that node's cores evenly, in rank order); when the GPU's node is unknown, or its node has no allowed core, the ranks split the
    (node, core)-ordered list into equal contiguous blocks.  Pure function of its arguments (tests feed it synthetic topologies).
    Returns (cores, how)."""
    per = len(allowed) // world
    if per < 1:
        return None, "fewer allowed cores than ranks"
    node = gpu_nodes[local] if local < len(gpu_nodes) else None
    if node is not None and len(gpu_nodes) >= world and all(n is not None for n in gpu_nodes[:world]):
        mates = [r for r in range(world) if gpu_nodes[r] == node]
        cores = [c for n, c in order if n == node]
        share = len(cores) // len(mates)
        if share >= 1:
            share -= share % 2 if share > 1 else 0        # whole physical cores (pairs of siblings) when there is more than one
            k = mates.index(local)
            return cores[k * share:(k + 1) * share], "NUMA node %d of GPU %d (%d rank(s) on that node)" % (node, local, len(mates))
    flat = [c for _, c in order]
    return flat[local * per:(local + 1) * per], "block %d of %d of the (node, core)-ordered allowed cores (GPU NUMA node unknown)" % (
        local, world)


PIN_HOW = None


def pin_rank_to_cores():
    """One process per GPU issues ~800 kernel launches per 50 ms step: give every rank of a multi-rank run its own physical cores
    on the NUMA node of ITS GPU (`gpu_numa_nodes`: KFD topology, read before anything touches the GPU; `plan_rank_cores`), so that
    eight launch threads (plus RCCL's proxy threads) neither migrate over each other nor cross the socket interconnect to reach
    their card.  Falls back to equal blocks of the (node, core)-ordered allowed cores where the topology says nothing.  Called in
    the child BEFORE anything touches the GPU; no exec.  CTG_NO_PIN=1 leaves the affinity alone.  Returns the core list (None when
    not pinned); `PIN_HOW` says which rule applied."""
    global PIN_HOW
    world = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
    if world <= 0:
        # no LOCAL_WORLD_SIZE (a multi-node launcher that does not export it): the ranks of THIS host are at most its GPUs
        world = int(os.environ.get("WORLD_SIZE", "1"))
        try:
            ngpu = torch.cuda.device_count()
        except Exception:      # noqa: BLE001
            ngpu = 0
        if ngpu > 0:
            world = min(world, ngpu)
    if world <= 1 or os.environ.get("CTG_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    local = int(os.environ.get("LOCAL_RANK", "0")) % world
    allowed = sorted(os.sched_getaffinity(0))
    mine, how = plan_rank_cores(local, world, allowed, gpu_numa_nodes(), core_order(allowed))
    if not mine:
        return None
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    PIN_HOW = how
    return sorted(mine)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["hd", "gen", "cyc", "p2p", "reg"], default="hd")
    ap.add_argument("--batch", type=int, default=None, help="paired slices per GPU (default 16; 8 for gen/cyc)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", choices=["bf16", "fp32", "bf16x3", "bf16x3f"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the bf16x3 leg of the default Hd run")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    cores = pin_rank_to_cores()

    from cta_gan_amd import _lib, dp, nets, ops, synth
    from cta_gan_amd.trainer import Cyc_Trainer, Hd_Trainer_x2, P2p_Trainer, Reg_Trainer
    _lib.load()   # no HIP library -> no benchmark
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X")
    try:
        rank, world, local = dp.init_from_env()
    except Exception as e:      # noqa: BLE001
        if "in use" in str(e).lower() or "eaddrinuse" in str(e).lower():
            sys.exit(EXIT_PORT_TAKEN)      # spawn_ranks() retries with another port
        raise
    if world != args.gpus or dp.world_size() != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s) (WORLD_SIZE=%s)"
                         % (args.gpus, dp.world_size(), os.environ.get("WORLD_SIZE")))
    local = local % torch.cuda.device_count()     # several ranks on one card only under CTG_DP_BACKEND=gloo (tests)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype_name = args.dtype or ("fp32" if args.workload == "gen" else "bf16")
    per_gpu = args.batch or (16 if args.workload in ("hd", "p2p", "reg") else 8)
    size = args.size
    MODES = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": "bf16x3", "bf16x3f": "bf16x3f"}

    # ---- CPU leg first (rank 0, N=1 only), so the GPU timing is not disturbed afterwards: the oracle's step timed on the host
    # cores (`cpu_baseline`) and its generator output on two synthetic slices (the reference of `gen_rel_l2`)
    cpu = gen_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, size)
        gen_ref = gen_reference(size)

    def run_leg(mode, steps, warmup):
        """`warmup` untimed + `steps` timed steps of the workload in compute mode `mode`; returns (elapsed seconds of the timed
        steps (max over ranks), kernel events, workload description, generator rel-L2 vs the CPU oracle or None)."""
        nets.set_default_compute_dtype(MODES[mode])
        l2 = gen_rel_l2(gen_ref, dev) if gen_ref is not None else None
        torch.manual_seed(42)      # identical replicas (the trainers also broadcast rank 0's weights); data differs per rank
        if args.workload == "hd":
            cfg = dict(YAML_HD, size=size, batchSize=per_gpu)
            tr = Hd_Trainer_x2(cfg)
            batch = {k: synth.synth_images("bench_%s_r%d" % (k, rank), per_gpu, size).to(dev) for k in ("A2", "B1", "B2")}
            step = lambda: tr.train_step(batch)                      # noqa: E731
            wl = "HdGan stage-2 full G+D train step (G, Reg, STN, D_m, losses, Adam x3), %d paired %dx%d slices/GPU" % (
                per_gpu, size, size)
        elif args.workload == "cyc":
            cfg = dict(YAML_CYC, size=size, batchSize=per_gpu)
            tr = Cyc_Trainer(cfg)
            batch = {k: synth.synth_images("bench_%s_r%d" % (k, rank), per_gpu, size).to(dev) for k in ("A", "B")}
            step = lambda: tr.train_step(batch)                      # noqa: E731
            wl = "CycleGan G/D_A/D_B train step, %d paired %dx%d slices/GPU" % (per_gpu, size, size)
        elif args.workload in ("p2p", "reg"):
            cfg = dict(YAML_P2P if args.workload == "p2p" else YAML_REG, size=size, batchSize=per_gpu)
            tr = (P2p_Trainer if args.workload == "p2p" else Reg_Trainer)(cfg)
            batch = {k: synth.synth_images("bench_%s_r%d" % (k, rank), per_gpu, size).to(dev) for k in ("A", "B")}
            step = lambda: tr.train_step(batch)                      # noqa: E731
            wl = "%s train step, %d paired %dx%d slices/GPU" % (type(tr).__name__, per_gpu, size, size)
        else:
            from cta_gan_amd.Model.HdGan import Generator
            G = Generator(1, 1).to(dev)
            x = synth.synth_images("bench_x_r%d" % rank, per_gpu, size).to(dev)

            def step():
                with torch.no_grad():
                    G(x)
            wl = "HdGan generator-only forward, %d %dx%d slices/GPU" % (per_gpu, size, size)

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        if not args.no_kernel_events and rank == 0:
            ops.KERNEL_EVENTS = {}            # kernel -> (start, end) HIP events around sampled launches of the timed kernels
            ops.KERNEL_BYTES.clear()
            ops._event_count.clear()
        if dp.enabled():
            dp.WAIT_LOG = []                  # (start, end, host s) around every gradient bucket's wait (GradSync.finish)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0        # this rank's own time, before it waits for the others
        dp.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
        waits, dp.WAIT_LOG = dp.WAIT_LOG, None
        per_rank = None
        if dp.enabled():
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
            # per-rank spread and the time lost to the gradient exchange, so that a scaling loss can be attributed: every rank's
            # own ms/step (before the closing barrier), the ms/step its optimiser stream waited for collectives (HIP events) and
            # the ms/step its host thread blocked in them
            mine = torch.tensor([1e3 * own / steps, sum(a.elapsed_time(b) for a, b, _ in waits) / steps,
                                 1e3 * sum(h for _, _, h in waits) / steps], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine)
            rows = torch.stack(allr).cpu()
            per_rank = {"ms_per_step": [round(float(v), 3) for v in rows[:, 0]],
                        "ms_per_step_min": round(float(rows[:, 0].min()), 3), "ms_per_step_max": round(float(rows[:, 0].max()), 3),
                        "gradsync_stream_wait_ms_per_step": [round(float(v), 3) for v in rows[:, 1]],
                        "gradsync_host_wait_ms_per_step": [round(float(v), 3) for v in rows[:, 2]],
                        "collectives_per_step": len(waits) // max(steps, 1),
                        "what": "per rank, over the timed steps: its own wall time per step up to its own device sync (the closing "
                                "barrier excluded); `gradsync_stream_wait`: HIP events around every bucket's wait() in "
                                "GradSync.finish() on the optimiser's stream = what the step loses to collectives that did not "
                                "finish under the backward; `gradsync_host_wait`: the host thread's time in those waits"}
        return elapsed, events, dict(ops.KERNEL_BYTES), wl, l2, per_rank

    def mfma_roofline(events, mode):
        """The residual blocks' 256 -> 256 3x3 convs: forward, backward-data (fused fold / residual / IN-sum epilogue) and
        weight gradient are three kernels doing the same 2*B*128^2*256^2*9 flop per launch."""
        if not events or not any(k in events for k in KERNEL_NAMES):
            return None
        flop = 2.0 * per_gpu * (size // 4) * (size // 4) * 256 * 256 * 9
        peak = PEAK_TFLOPS[mode]
        # HBM bytes per launch from separate rocprofv3 --pmc passes (scripts/pmc_step.sh -> scripts/pmc_tables.py -> profiles/), only when
        # they were taken on THIS kernel build and shape
        pmc_file = "pmc_dominant.json" if mode == "bf16" else "pmc_dominant_%s.json" % mode
        pmc = _pmc_table(pmc_file, per_gpu, size, mode)
        kernels, t_all, n_all = [], 0.0, 0
        for name in ("fwd", "fwd_in", "bwd_data", "wgrad"):
            evs = events.get(name)
            if not evs:
                continue
            ms = [s.elapsed_time(e) for s, e in evs]
            avg = sum(ms) / len(ms)
            if name != "fwd_in":      # listed below, but not part of the time-weighted conv figure: that launch also normalises
                t_all += sum(ms)
                n_all += len(ms)
            k = {"name": KERNEL_NAMES[name].replace("<bf16", "<" + mode), "launches": len(ms),
                 "avg_ms": round(avg, 4), "achieved": round(flop / (avg * 1e-3) / 1e12, 1),
                 "frac": round(flop / (avg * 1e-3) / 1e12 / peak, 4)}
            if pmc is not None and name in pmc.get("kernels", {}):
                k["traffic"] = pmc["kernels"][name]["traffic_bytes_per_launch"]
                k["algorithmic_bytes"] = pmc["kernels"][name]["algorithmic_bytes_per_launch"]
            kernels.append(k)
        if n_all == 0:      # a forward-only workload below the fusion limit: every launch is the fused one
            ms = [s.elapsed_time(e) for s, e in events["fwd_in"]]
            t_all, n_all = sum(ms), len(ms)
        avg_ms = t_all / n_all
        achieved = flop / (avg_ms * 1e-3) / 1e12          # time-weighted over the three kernels
        return {"bound": "mfma", "kernel": "the 256->256 3x3 reflect convs of the residual blocks (%s): forward, "
                "backward-data and weight-gradient kernels, time-weighted (the fused conv + InstanceNorm launches of forwards that "
                "keep nothing, `fwd_in`, are listed in `kernels` but not averaged in: their time includes the normalisation)" % mode,
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": pmc["traffic_bytes_per_launch"] if pmc is not None else None,     # launch-weighted, like avg_ms
                "traffic_source": None if pmc is None else "profiles/%s -- the builder's own PMC run replayed, not "
                "measured in this run (build digest %s = this library): %s" % (pmc_file, pmc.get("build"), pmc.get("source")),
                "mfma_busy": None if pmc is None else pmc.get("mfma_busy"),
                "launches_timed": n_all,
                "launches_sampled": "every %d. launch of each kernel" % ops.KERNEL_EVENT_STRIDE,
                "avg_launch_ms": round(avg_ms, 4), "flop_per_launch": flop, "kernels": kernels}

    def hbm_roofline(events, nbytes, mode):
        """The HBM-bound side of the step, each kernel timed like the MFMA ones: algorithmic bytes of one launch (every
        operand tensor read or written exactly once) / its average duration, against the 8 TB/s of MI355X_MICROARCH.md."""
        if not events:
            return None
        pmc = _pmc_table("pmc_hbm.json" if mode == "bf16" else "pmc_hbm_%s.json" % mode, per_gpu, size, mode)
        pmc_key = {"convt64": "convt64p", "convs2": "convs2p"} if mode == "bf16x3" else {}      # the split-pair kernels' rows
        rows = []
        shape = "|%dx%dx%dx256" % (per_gpu, size // 4, size // 4)      # the residual blocks' maps
        for key, label in HBM_KERNELS.items():
            full = key if key in UNSHAPED_HBM_KEYS else key + shape
            evs = events.get(full)
            if not evs or full not in nbytes:
                continue
            nb = nbytes[full]
            ms = [s.elapsed_time(e) for s, e in evs]
            avg = sum(ms) / len(ms)
            gbs = nb / (avg * 1e-3) / 1e9
            row = {"key": key, "kernel": label, "launches": len(ms), "avg_ms": round(avg, 4),
                   "algorithmic_bytes": int(nb), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                   "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None}
            if pmc is not None and pmc_key.get(key, key) in pmc.get("kernels", {}):
                row["traffic"] = pmc["kernels"][pmc_key.get(key, key)]["traffic_bytes_per_launch"]
            if mode == "bf16x3":
                row["kernel"] = label.replace("conv_stript_128_64_kernel", "conv_striptp_128_64_kernel").replace(
                    "conv_strips2_64_128_kernel", "conv_strips2p_64_128_kernel").replace("conv_strip32_kernel", "conv_strip32p_kernel") \
                    + " [split pair: 4 bytes per value]"
            rows.append(row)
        return rows or None

    def _pmc_table(fname, b, sz, mode):
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            return None
        from cta_gan_amd import build as _build
        cand = json.load(open(path))
        if (cand["per_gpu_batch"], cand["size"], cand["dtype"]) == (b, sz, mode) and \
                cand.get("build") == _build._digest()[:16]:
            return cand
        return None

    def leg_numbers(mode, steps, elapsed):
        """(slices/s, TFLOP/s by the survey's count, TFLOP/s of the work the build executes)"""
        value = per_gpu * world * steps / elapsed
        scale = (size / 512.0) ** 2 / 1e3
        return value, value * GFLOP_PER_SLICE[args.workload] * scale, \
            value * GFLOP_EXECUTED.get(args.workload, GFLOP_PER_SLICE[args.workload]) * scale

    TOL = 1e-3      # north_star: generator output within 1e-3 rel-L2 of the CPU reference

    elapsed, events, nbytes, wl, l2, per_rank = run_leg(dtype_name, args.steps, args.warmup)
    line = None
    if rank == 0:
        value, step_tflops, step_tflops_exec = leg_numbers(dtype_name, args.steps, elapsed)
        roof = mfma_roofline(events, dtype_name)
        if roof is not None:
            roof["hbm"] = hbm_roofline(events, nbytes, dtype_name)
        line = {"metric": "paired 512x512 slices/sec (G+D step); generator rel-L2 vs CPU reference" if args.workload != "gen"
                else "512x512 slices/sec (generator forward); generator rel-L2 vs CPU reference",
                "value": round(value, 3), "unit": "slices/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
                "gen_rel_l2": None if l2 is None else float("%.3e" % l2),
                "tolerance": TOL, "tolerance_met": None if l2 is None else bool(l2 <= TOL),
                "gen_rel_l2_sample": None if l2 is None else "generator forward, synthetic weights, B=2 @ %dx%d, vs the fp32 CPU "
                "oracle (north_star: <= 1e-3: `tolerance_met`; the bf16 leg is outside it, parity_mode inside)" % (size, size),
                "config": {"workload": wl, "per_gpu_batch": per_gpu, "global_batch": per_gpu * world, "size": size,
                           "parallelism": "dp%d" % world if world > 1 else "single",
                           "step_tflops_algorithmic": round(step_tflops, 2),
                           "step_tflops_executed": round(step_tflops_exec, 2)},
                "step_frac": round(step_tflops_exec / PEAK_TFLOPS[dtype_name], 4),      # executed work / dense peak
                "rccl_ranks": dp.world_size(), "dp_backend": dp.backend_name(),
                "rank_cores": None if cores is None else "rank 0 pinned to %d cores (%d..%d): %s" % (len(cores), cores[0], cores[-1], PIN_HOW),
                "per_rank": per_rank,
                "roofline": roof, "cpu_baseline": cpu}

    # ---- parity-mode leg (N=1, the Hd step in bf16): the same step again in the split-bf16 mode, the fastest one whose
    # generator output is inside the north_star's 1e-3 of the fp32 CPU reference
    if world == 1 and args.workload == "hd" and dtype_name == "bf16" and not args.no_parity_mode:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        p_steps = max(10, min(args.steps, 12))
        elapsed, events, nbytes, _, l2, _ = run_leg("bf16x3", p_steps, 2)
        value, step_tflops, step_tflops_exec = leg_numbers("bf16x3", p_steps, elapsed)
        roof = mfma_roofline(events, "bf16x3")
        line["parity_mode"] = {
            "dtype": "bf16x3", "what": "split-pair storage ([hi | lo] bf16 planes per pixel row, 4 bytes per value; fp32 statistics / "
            "parameters / losses); every conv contraction as three bf16 MFMAs (hi.hi + hi.lo + lo.hi) straight from the planes: "
            "peak = 2.5 PF / 3",
            "value": round(value, 3), "unit": "slices/s", "steps": p_steps, "warmup": 2,
            "ms_per_step": round(1e3 * elapsed / p_steps, 3), "gen_rel_l2": None if l2 is None else float("%.3e" % l2),
            "tolerance": TOL, "tolerance_met": None if l2 is None else bool(l2 <= TOL),
            "step_frac": round(step_tflops_exec / PEAK_TFLOPS["bf16x3"], 4),
            "roofline": None if roof is None else dict({k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                                                             "traffic_source", "mfma_busy", "launches_timed",
                                                                             "avg_launch_ms", "kernels")},
                                                       hbm=hbm_roofline(events, nbytes, "bf16x3"))}
        # ---- third leg: the same forward with the backward in plain bf16 ("bf16x3f"): output and losses are the parity leg's bit for
        # bit, the gradients are 1.5e-2 rel-L2 from the reference's (tests/test_bf16x3f_gpu.py) -- reported beside, never instead of,
        # the parity leg
        gc.collect()
        torch.cuda.empty_cache()

        def generator_gradients(mode):
            """every weight gradient and the input gradient of the generator (B = 2, 256^2: every fused path of the step is live)"""
            nets.set_default_compute_dtype(MODES[mode])
            from cta_gan_amd.Model.HdGan import Generator
            g = synth.fill_module(Generator(1, 1), seed=0).to(dev)
            x = synth.synth_smooth_images("bench_gx", 2, 256).to(dev).requires_grad_(True)
            y = g(x)
            (y.float() * torch.linspace(0.5, 1.5, y.numel(), device=dev).view_as(y)).sum().backward()
            return [x.grad.double()] + [p.grad.double() for k, p in g.named_parameters() if p.grad is not None and k.endswith("weight")]

        gx3, gx3f = generator_gradients("bf16x3"), generator_gradients("bf16x3f")
        errs = [float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(gx3f, gx3)]
        grad_vs_x3 = {"input_gradient": float("%.3e" % errs[0]), "weight_gradients_median": float("%.3e" % sorted(errs[1:])[len(errs[1:]) // 2]),
                      "weight_gradients_worst": float("%.3e" % max(errs[1:]))}
        del gx3, gx3f
        gc.collect()
        torch.cuda.empty_cache()
        elapsed, events, nbytes, _, l2, _ = run_leg("bf16x3f", p_steps, 2)
        value, step_tflops, step_tflops_exec = leg_numbers("bf16x3f", p_steps, elapsed)
        line["parity_mode_bf16_backward"] = {
            "dtype": "bf16x3f", "what": "the parity leg's forward (same kernels, same generator output and losses) with the backward in "
            "plain bf16: gradient tensors bf16, one bf16 MFMA per product on the hi planes of the saved activations; activation masks, "
            "max-pool argmax and the InstanceNorm backward's xhat still from hi + lo.  Gradients: 1.5e-2 rel-L2 from the fp32 reference's "
            "on the generator (bf16x3 8.8e-3, bf16 0.24; tests/test_bf16x3f_gpu.py)",
            "value": round(value, 3), "unit": "slices/s", "steps": p_steps, "warmup": 2,
            "ms_per_step": round(1e3 * elapsed / p_steps, 3), "gen_rel_l2": None if l2 is None else float("%.3e" % l2),
            "tolerance": TOL, "tolerance_met": None if l2 is None else bool(l2 <= TOL),
            "generator_gradient_rel_l2": 1.45e-2, "generator_gradient_source": "tests/test_bf16x3f_gpu.py::test_goldens_x3f[generator_64]",
            "generator_gradient_vs_bf16x3_rel_l2": grad_vs_x3,
            "generator_gradient_vs_bf16x3_sample": "measured in this run: rel-L2 per tensor of the generator's input gradient (the path "
            "through every layer) and of its 24 conv weight gradients, B=2 @ 256x256, bf16x3f against bf16x3 (itself 8.8e-3 from the fp32 oracle)"}
    if rank == 0:
        # The driver keeps `config` verbatim and drops unknown top-level keys: what a reader of BENCH_rNN.json needs to judge the
        # line -- is the headline inside the north_star's tolerance, what does the leg that IS inside it run at, which ranks
        # exchanged gradients over what and how long each waited -- is therefore repeated inside `config`.
        cfgx = line["config"]
        cfgx["gen_rel_l2"], cfgx["tolerance"], cfgx["tolerance_met"] = line["gen_rel_l2"], TOL, line["tolerance_met"]
        pm = line.get("parity_mode")
        if pm is not None:
            cfgx["parity_mode"] = {"dtype": pm["dtype"], "value": pm["value"], "unit": pm["unit"], "ms_per_step": pm["ms_per_step"],
                                   "steps": pm["steps"], "gen_rel_l2": pm["gen_rel_l2"], "tolerance_met": pm["tolerance_met"],
                                   "roofline_frac": None if pm["roofline"] is None else pm["roofline"]["frac"],
                                   "step_frac": pm["step_frac"]}
        pf = line.get("parity_mode_bf16_backward")
        if pf is not None:
            cfgx["parity_mode_bf16_backward"] = {k: pf[k] for k in ("dtype", "value", "unit", "ms_per_step", "steps", "gen_rel_l2",
                                                                    "tolerance_met", "generator_gradient_rel_l2", "generator_gradient_vs_bf16x3_rel_l2")}
        cfgx["rccl_ranks"], cfgx["dp_backend"] = line["rccl_ranks"], line["dp_backend"]
        if per_rank:        # N > 1: the first real multi-GPU run must be attributable from the driver record alone
            cfgx["per_rank"] = {k: per_rank[k] for k in ("ms_per_step", "ms_per_step_min", "ms_per_step_max",
                                                         "gradsync_stream_wait_ms_per_step", "gradsync_host_wait_ms_per_step",
                                                         "collectives_per_step")}
        print(json.dumps(line), flush=True)
    if dp.enabled():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
