"""Disassemble every gfx950 code object of libctagan_hip.so and list the packed-fp32 instructions that carry
source modifiers (op_sel / op_sel_hi / neg_lo / neg_hi).

Why: round 5 found `v_pk_fma_f32 ... op_sel_hi:[0,1,1]` (one register of a pair broadcast to both halves) returning wrong
products in lanes 48-63 beside narrow halo convs on another stream ("the packed-fp32 modifier hazard").  The cause
is not known, so the form is banned from the library: cta_gan_amd/build.py compiles every source with the packed-fp32 target
feature off, kernels that want packed arithmetic write it as inline assembly on whole register pairs, and
`tests/test_isa_gate.py` fails on any modifier-carrying packed-fp32 instruction outside the allow-list below.

    python -m cta_gan_amd.isa_gate [path/to/lib.so] [--all]     # per-kernel table; --all also lists the plain packed forms
"""
from __future__ import annotations

import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libctagan_hip.so")

PK_F32 = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
MODIFIER = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[[01,]+\]")
LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")

# kernel-name substring -> reason (must name the stress test that covers it).  Empty: nothing is exempt.
ALLOW: dict[str, str] = {}


def code_objects(lib: str, work: str) -> list[str]:
    """Unbundle the gfx950 code objects of `lib` into `work` (llvm-objdump --offloading writes next to its input)."""
    local = os.path.join(work, os.path.basename(lib))
    shutil.copy(lib, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=work)
    return sorted(os.path.join(work, f) for f in os.listdir(work) if "amdgcn" in f and "gfx950" in f)


def scan(lib: str = LIB):
    """-> {kernel: Counter{(mnemonic, modifiers) -> n}}, {kernel: n plain packed fp32}"""
    flagged: dict[str, collections.Counter] = collections.defaultdict(collections.Counter)
    plain: collections.Counter = collections.Counter()
    with tempfile.TemporaryDirectory() as work:
        objs = code_objects(lib, work)
        if not objs:
            raise RuntimeError("no gfx950 code object found in %s" % lib)
        for obj in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", obj], check=True, capture_output=True, text=True).stdout
            kernel = "?"
            for line in dis.splitlines():
                m = LABEL.match(line)
                if m:
                    kernel = m.group(1)
                    continue
                pm = PK_F32.search(line)
                if not pm:
                    continue
                mods = " ".join(x.group(0) for x in MODIFIER.finditer(line))
                if mods:
                    flagged[kernel][(pm.group(0), mods)] += 1
                else:
                    plain[kernel] += 1
    return flagged, plain


def demangle(names):
    filt = shutil.which("c++filt")
    if not filt or not names:
        return {n: n for n in names}
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.splitlines() if r.returncode == 0 else list(names)
    return dict(zip(names, out)) if len(out) == len(names) else {n: n for n in names}


def violations(flagged):
    return {k: v for k, v in flagged.items() if not any(a in k for a in ALLOW)}


def main(argv):
    lib = next((a for a in argv if not a.startswith("-")), LIB)
    flagged, plain = scan(lib)
    names = demangle(sorted(set(flagged) | set(plain)))
    total = 0
    for k in sorted(flagged):
        n = sum(flagged[k].values())
        total += n
        print("%5d  %s" % (n, names[k][:150]))
        for (mn, mods), c in sorted(flagged[k].items()):
            print("         %4d x %s %s" % (c, mn, mods))
    print("modifier-form packed fp32 instructions: %d in %d kernels; plain packed fp32: %d in %d kernels"
          % (total, len(flagged), sum(plain.values()), len(plain)))
    if "--all" in argv:
        for k in sorted(plain):
            print("  plain %5d  %s" % (plain[k], names[k][:150]))
    return 1 if violations(flagged) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
