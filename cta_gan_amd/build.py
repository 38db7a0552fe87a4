"""Build the gfx950 kernel library in-tree with hipcc (cross-compiles without a GPU).

    python -m cta_gan_amd.build [--force]

Output: cta_gan_amd/_build/libctagan_hip.so (git-ignored; travels to the GPU box
with the gpurun snapshot).  One object per .hip source, compiled in parallel.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libctagan_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-Wno-unused-result"]
# Packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) is switched OFF as a target feature for every source: the
# compiler folds splats, half-selects and negations into those instructions' op_sel / op_sel_hi / neg modifiers, and the
# broadcast form returned wrong products beside certain neighbours on the card ("the packed-fp32 modifier
# hazard"; cause unknown).  With the feature off neither instruction selection nor the SLP vectoriser can form them; a
# kernel that wants packed arithmetic writes it as inline assembly on whole register pairs (csrc/conv_cout1.hip:
# c1_pk_fma -- the assembler still accepts the mnemonic), and tests/test_isa_gate.py disassembles the library and fails on
# any v_pk_*_f32 that carries a modifier.  (clang applies -target-feature to the host pass too, which prints "not a
# recognized feature for this target (ignoring feature)": harmless.)  CTG_BUILD_PK_F32=all (or a comma list of sources)
# leaves the feature on: the developer A/B of what the switch costs.
NO_PK_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
PK_F32_SOURCES = frozenset(x for x in os.environ.get("CTG_BUILD_PK_F32", "").split(",") if x)


def flags_for(src: str) -> list:
    return FLAGS + ([] if (src in PK_F32_SOURCES or "all" in PK_F32_SOURCES) else NO_PK_F32)


def _hipcc() -> str:
    import shutil
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.exists(cand) if os.path.sep in cand else shutil.which(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    # the published C ABI header is compiled into every object (common.h includes it): a header-only change is a rebuild too
    with open(os.path.join(os.path.dirname(HERE), "include", "ctagan_hip.h"), "rb") as fh:
        h.update(fh.read())
    for f in sources():
        h.update((f + " " + " ".join(flags_for(f))).encode())
    return h.hexdigest()


def is_current() -> bool:
    stamp = os.path.join(OUT, "stamp")
    return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == _digest()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and is_current():
        return LIB
    os.makedirs(OUT, exist_ok=True)
    hipcc = _hipcc()
    # one builder at a time (the ranks of a multi-process run may all find the library stale)
    import fcntl
    lock = open(os.path.join(OUT, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and is_current():
            return LIB
        return _build_locked(hipcc, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(hipcc: str, verbose: bool) -> str:

    def compile_one(src):
        obj = os.path.join(OUT, src[:-4] + ".o")
        cmd = [hipcc, *flags_for(src), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose:
            print("compiled", src, flush=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(os.path.join(OUT, "stamp"), "w") as fh:
        fh.write(_digest())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
