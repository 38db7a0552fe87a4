"""CPU: the shipped gfx950 code objects hold no packed-fp32 instruction with source modifiers.

Round 5 found `v_pk_fma_f32 ... op_sel_hi:[0,1,1]` (one register of a pair broadcast to both halves -- what the compiler
makes of `float2{g, g} * w + acc`) returning wrong products in lanes 48-63 while the wave shared a CU with certain
workgroups of another stream; the cause is not understood.  The library is therefore built with the packed-fp32
target feature OFF (cta_gan_amd/build.py: NO_PK_F32) and the few kernels that want packed arithmetic write it as inline
assembly on whole register pairs.  This test disassembles every code object of the built library (llvm-objdump ships
with ROCm) and fails on any v_pk_{fma,mul,add}_f32 that carries op_sel / op_sel_hi / neg_lo / neg_hi, so that neither a
compiler fold nor a new kernel can bring the form back unnoticed.  An allow-list entry (cta_gan_amd/isa_gate.py: ALLOW)
must name the neighbour-stress GPU test that covers the kernel."""
import os

import pytest

from cta_gan_amd import build, isa_gate

pytestmark = pytest.mark.skipif(not os.path.exists(isa_gate.OBJDUMP), reason="llvm-objdump of the ROCm image not found")


@pytest.fixture(scope="module")
def scanned():
    build.build()
    return isa_gate.scan(build.LIB)


def test_no_packed_fp32_instruction_carries_a_modifier(scanned):
    flagged, _ = scanned
    bad = isa_gate.violations(flagged)
    lines = ["%s: %s" % (k, ", ".join("%d x %s %s" % (n, mn, mods) for (mn, mods), n in sorted(v.items()))) for k, v in sorted(bad.items())]
    assert not bad, "modifier-form packed fp32 instructions in the library:\n" + "\n".join(lines[:40])


def test_packed_fp32_only_where_it_is_written_by_hand(scanned):
    """With the feature off the compiler forms no packed fp32 at all: what is left is the inline assembly of the kernels that
    re-enable the feature for themselves (csrc/conv_cout1.hip).  A new name here is a kernel someone opted in: it must keep
    to the rule above, and this list is where the opt-in is recorded."""
    _, plain = scanned
    names = isa_gate.demangle(sorted(plain))
    opted_in = ("cout1_fwd_kernel", "cout1_bwd_kernel", "cout1_wgrad_kernel")
    others = sorted(names[k] for k in plain if not any(o in k for o in opted_in))
    assert not others, others
    assert sum(plain.values()) > 1000       # the scan really saw the library's code (the three kernels hold thousands)


def test_every_allow_list_entry_names_its_stress_test():
    for kernel, reason in isa_gate.ALLOW.items():
        assert "test_" in reason, (kernel, reason)


def test_the_scanner_sees_the_form_it_bans(tmp_path):
    """The gate is only worth something if the pattern matches what the disassembler prints: compile a two-line kernel WITH
    the feature on that must fold a splat into op_sel_hi, and find it."""
    src = tmp_path / "splat.hip"
    src.write_text('#include <hip/hip_runtime.h>\n'
                   'typedef float f2 __attribute__((ext_vector_type(2)));\n'
                   '__global__ void k(const float* a, const f2* w, f2* o) {\n'
                   '    const int i = threadIdx.x; const float g = a[i]; f2 acc = o[i];\n'
                   '    acc = f2{g, g} * w[i] + acc; acc = f2{g, g} * w[i + 64] + acc; o[i] = acc; }\n')
    obj = tmp_path / "splat.o"
    import subprocess
    r = subprocess.run([build._hipcc(), "-O3", "--offload-arch=gfx950", "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    flagged, plain = isa_gate.scan(str(obj))
    n = sum(sum(v.values()) for v in flagged.values())
    if n == 0:
        # this compiler materialised the pair instead of folding: then the plain form must be there, and the pattern is
        # checked on the disassembler's own syntax from a hand-written instruction
        assert sum(plain.values()) >= 1
        asm = tmp_path / "mod.hip"
        asm.write_text('#include <hip/hip_runtime.h>\n'
                       'typedef float f2 __attribute__((ext_vector_type(2)));\n'
                       '__global__ void k(const f2* w, f2* o) { f2 a = w[threadIdx.x], acc = o[threadIdx.x];\n'
                       '    asm("v_pk_fma_f32 %0, %1, %1, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a)); o[threadIdx.x] = acc; }\n')
        r = subprocess.run([build._hipcc(), "-O3", "--offload-arch=gfx950", "-c", str(asm), "-o", str(tmp_path / "mod.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        flagged, _ = isa_gate.scan(str(tmp_path / "mod.o"))
        n = sum(sum(v.values()) for v in flagged.values())
    assert n >= 1
