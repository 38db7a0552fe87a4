"""`python bench.py --gpus N` run directly (no torch.distributed.run around it): the parent starts the N ranks itself.

CPU part: the spawner starts N children with the rendezvous environment and hands their failure back (no GPU here, so
every rank stops at "needs an MI355X" and the parent must exit non-zero, quickly, without hanging in a collective).
GPU part: two ranks on ONE card with CTG_DP_BACKEND=gloo (RCCL needs one GPU per rank; the driver's multi-GPU run is the
RCCL check) run two data-parallel HdGan steps through the bucketed gradient exchange and print one JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_spawner_propagates_rank_failure_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {}, 300)
    assert r.returncode != 0
    assert "MI355X" in r.stderr or "libctagan_hip" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "bf16x3f"])
def test_bench_gpus2_self_spawned_gloo_on_one_card(dtype):
    """(bf16x3f: the data-parallel exchange with the backward that switches the thread-local storage mode inside autograd's thread)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "256", "--batch", "2", "--no-cpu-baseline", "--dtype", dtype],
             {"CTG_DP_BACKEND": "gloo"}, 900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dp_backend"] == "gloo"
    assert line["config"]["global_batch"] == 4 and line["scaling"] == "weak" and line["value"] > 0
    assert line["roofline"] is None or line["roofline"]["frac"] > 0
    assert line["dtype"] == dtype
    # what the driver keeps verbatim carries the attribution of a multi-rank run (the top-level copies are dropped by its parser)
    cfg = line["config"]
    assert cfg["rccl_ranks"] == 2 and cfg["dp_backend"] == "gloo" and len(cfg["per_rank"]["ms_per_step"]) == 2
    assert cfg["per_rank"]["collectives_per_step"] == 4 and "tolerance_met" in cfg


def test_eight_rank_spawn_pins_every_rank_and_fails_cleanly_without_a_gpu():
    """BASELINE.json configs[4]'s launcher path with all EIGHT ranks (CPU box: each rank pins itself to its own block of cores,
    then stops at "needs an MI355X"; the parent must hand a non-zero status back without hanging)."""
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {}, 600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "MI355X" in r.stderr or "libctagan_hip" in r.stderr, r.stderr[-2000:]


def test_rank_pinning_gives_disjoint_core_blocks():
    """`pin_rank_to_cores` (bench.py) on THIS host: N ranks get N disjoint, equally sized core sets out of the allowed cores."""
    code = ("import os, sys; sys.path.insert(0, %r); import bench; c = bench.pin_rank_to_cores(); "
            "print('CORES', sorted(os.sched_getaffinity(0)) == c, c)" % ROOT)
    allowed = sorted(os.sched_getaffinity(0))
    n = min(4, len(allowed))
    seen = []
    for rk in range(n):
        env = dict(os.environ, LOCAL_RANK=str(rk), LOCAL_WORLD_SIZE=str(n), WORLD_SIZE=str(n), RANK=str(rk))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("CORES")][-1]
        assert line.split()[1] == "True", line
        seen.append(eval(line.split(" ", 2)[2]))
    flat = [c for blk in seen for c in blk]
    assert len(flat) == len(set(flat)) and set(flat) <= set(allowed)
    assert len({len(b) for b in seen}) == 1 and len(seen[0]) >= 1


def _fake_sysfs(tmp_path, sockets=2, cores_per_socket=8, gpu_nodes=(0, 0, 0, 0, 1, 1, 1, 1)):
    """A sysfs tree of the usual two-socket SMT numbering: cpus [0, S*C) are the first threads socket by socket, [S*C, 2*S*C) their
    hyperthread siblings in the same order; one KFD CPU node per socket and one GPU node per entry of gpu_nodes."""
    n = sockets * cores_per_socket
    for c in range(2 * n):
        d = tmp_path / ("devices/system/cpu/cpu%d/topology" % c)
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text("%d,%d\n" % (c % n, c % n + n))
    for s_ in range(sockets):
        d = tmp_path / ("devices/system/node/node%d" % s_)
        d.mkdir(parents=True)
        lo = s_ * cores_per_socket
        (d / "cpulist").write_text("%d-%d,%d-%d\n" % (lo, lo + cores_per_socket - 1, lo + n, lo + n + cores_per_socket - 1))
    k = 0
    for s_ in range(sockets):
        d = tmp_path / ("class/kfd/kfd/topology/nodes/%d" % k)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count 0\ndrm_render_minor -1\n" % cores_per_socket)
        k += 1
    for g, node in enumerate(gpu_nodes):
        d = tmp_path / ("class/kfd/kfd/topology/nodes/%d" % k)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % (128 + g))
        r = tmp_path / ("class/drm/renderD%d/device" % (128 + g))
        r.mkdir(parents=True)
        (r / "numa_node").write_text("%d\n" % node)
        k += 1
    return 2 * n


def test_ranks_are_pinned_to_the_numa_node_of_their_gpu(tmp_path, monkeypatch):
    """BASELINE.json configs[4] readiness without the node: on a synthetic two-socket SMT topology (bench.gpu_numa_nodes /
    core_order / plan_rank_cores take the sysfs root) eight ranks get disjoint sets of WHOLE physical cores, each on the NUMA node
    its own GPU hangs off -- including when the GPUs are not attached in rank order -- and the plain ascending deal's mistakes
    (ranks sharing hyperthread siblings, ranks on the remote socket) do not occur.  Unknown GPU nodes: equal contiguous blocks of
    the (node, core)-ordered list, still whole cores on one node."""
    import bench
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    for gpu_nodes in [(0, 0, 0, 0, 1, 1, 1, 1), (1, 1, 0, 0, 0, 0, 1, 1)]:
        root = tmp_path / ("t" + "".join(map(str, gpu_nodes)))
        ncpu = _fake_sysfs(root, gpu_nodes=gpu_nodes)
        sysfs = str(root)
        assert bench.gpu_numa_nodes(sysfs) == list(gpu_nodes)
        allowed = list(range(ncpu))
        order = bench.core_order(allowed, sysfs)
        assert [c for _, c in order][:4] == [0, 16, 1, 17]                    # siblings adjacent, node 0 first
        plans = [bench.plan_rank_cores(r, 8, allowed, list(gpu_nodes), order) for r in range(8)]
        flat = [c for cores, _ in plans for c in cores]
        assert len(flat) == len(set(flat)) == 32
        for r, (cores, how) in enumerate(plans):
            assert len(cores) == 4 and "NUMA node %d" % gpu_nodes[r] in how
            socket = {(c % 16) // 8 for c in cores}
            assert socket == {gpu_nodes[r]}, (r, cores)                        # on its GPU's socket
            assert {c % 16 for c in cores} == {c % 16 for c in cores if (c % 16 + 16 * (c < 16)) in cores}   # sibling pairs complete
    # a visible-devices list reorders the topology like HIP does
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,0,1")
    assert bench.gpu_numa_nodes(sysfs) == [0, 0, 1, 1]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # no GPU topology at all (this container): contiguous blocks of the (node, core) order
    plans = [bench.plan_rank_cores(r, 4, allowed, [], order) for r in range(4)]
    assert [len(c) for c, _ in plans] == [8] * 4 and plans[0][0] == [0, 16, 1, 17, 2, 18, 3, 19]
    assert all("unknown" in how for _, how in plans)


@pytest.mark.gpu
def test_bench_four_ranks_on_one_card_over_gloo():
    """The many-rank launcher path on hardware: four ranks on ONE GPU with CTG_DP_BACKEND=gloo (this pool lets at most six
    processes hold a card, and the test runner is one of them; eight ranks is the driver's multi-GPU run, its spawn path is
    covered on the CPU above), two data-parallel steps, every rank pinned to its own cores, one JSON line, rc 0."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _run(["--gpus", "4", "--steps", "2", "--warmup", "1", "--size", "256", "--batch", "1", "--no-cpu-baseline"],
             {"CTG_DP_BACKEND": "gloo"}, 1200)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["rccl_ranks"] == 4 and line["dp_backend"] == "gloo"
    assert line["config"]["global_batch"] == 4 and line["value"] > 0 and line["rank_cores"] is not None
    # the N > 1 line says where a scaling loss would come from: every rank's own ms/step and what its optimiser stream and host
    # waited for the gradient exchange (four buckets per step: {Reg}, two generator halves, {D})
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == 4 and 0 < pr["ms_per_step_min"] <= pr["ms_per_step_max"] <= line["ms_per_step"] * 1.05 + 1.0
    assert len(pr["gradsync_stream_wait_ms_per_step"]) == 4 and all(v >= 0 for v in pr["gradsync_stream_wait_ms_per_step"])
    assert pr["collectives_per_step"] == 4 and all(v >= 0 for v in pr["gradsync_host_wait_ms_per_step"])
    assert "pinned" in line["rank_cores"]


@pytest.mark.gpu
def test_bench_gpus2_over_rccl_when_two_gpus_are_visible():
    """BASELINE.json configs[4] in miniature, over the real backend: `bench.py --gpus 2` self-spawns one rank per GPU, the
    gradient buckets are all-reduced by RCCL ("nccl") from inside the backward.  Needs two visible GPUs (the one-GPU test box
    skips; any multi-GPU box exercises RCCL with more than one rank without further work)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL: one GPU per rank)")
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "256", "--batch", "2", "--no-cpu-baseline"],
             {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, 900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dp_backend"] == "nccl"
    assert line["config"]["global_batch"] == 4 and line["config"]["per_gpu_batch"] == 2 and line["value"] > 0


@pytest.mark.gpu
def test_bench_refuses_a_world_size_mismatch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "process group has 1 rank" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16", "bf16x3f"])
def test_rccl_code_path_with_one_rank_is_bit_identical_to_the_plain_run(mode):
    """The only way to execute RCCL on a one-GPU box: CTG_DP_FORCE=1 runs the whole exchange -- `init_process_group("nccl")`,
    persistent buckets written by the kernels, `all_reduce(AVG, async_op=True)` launched from inside the backward on RCCL's
    own stream, `wait()` before Adam -- with a single rank, where the all-reduce is the identity: three bf16 Hd steps must then
    reproduce the plain run bit for bit (losses and every weight).  A missing stream dependency between the weight-gradient
    kernels, the collective and the optimiser would show here."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = os.path.join(ROOT, "scripts", "dp_force_check.py")

    def run(extra):
        env = dict(os.environ, DPCHECK_MODE=mode, **extra)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-4000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DPCHECK")][-1]
        return dict(kv.split("=") for kv in line.split()[1:])
    plain = run({})
    forced = run({"CTG_DP_FORCE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert plain["enabled"] == "False" and forced["enabled"] == "True" and forced["backend"] == "nccl"
    assert forced["buckets"] == "3" and forced["stray"] == "0"
    assert forced["digest"] == plain["digest"], (plain, forced)
