"""Discriminating experiments on the round-5 packed-fp32 hazard, made ON THE FAILING BUILD ITSELF at the assembly level.

The failing kernels (conv_cout1.hip of commit 2ebab5e: `c1_f2{g, g} * w + acc` folded into v_pk_fma_f32 op_sel_hi:[0,1,1])
are compiled to gfx950 assembly with the packed-fp32 feature ON, the text of the input-gradient / weight-gradient kernels
is edited per variant, and each variant is assembled, linked and bundled back into a full library next to feature-ON
objects of every other source (the neighbours as round 5 had them):

    python scripts/diag/hazard_variants.py            # builds cta_gan_amd/_build/diag/libctagan_hip_<variant>.so
    (on the GPU box)  bash scripts/diag/hazard_run.sh  # scripts/lds_neighbour_stress.py per variant, CTG_LIB=...

Variants (all = the failing ISA plus ONE change):
    a0_control     nothing changed (the round trip through text assembly must still fail)
    a1_nop         `s_nop 7` x2 after every s_waitcnt that waits on lgkmcnt (LDS data landed late?)
    a2_init        every VGPR the kernel READS as the unselected half of a broadcast pair but never WRITES is zeroed at
                   entry (does the hardware look at the unselected half?)
    a3_nop_pk      `s_nop 0` in front of every modifier-form v_pk_fma_f32 (an issue-slot / forwarding hazard?)
    a3b_nop7_pk    `s_nop 7` in front of every modifier-form v_pk_fma_f32
    a4_setprio     s_setprio 3 at entry (arbitration against the neighbour's waves)
    a5_lgkm0       every `s_waitcnt lgkmcnt(N)` waits for ALL LDS data (no LDS return in flight while packed instructions read)
    a6_allcnt0     every s_waitcnt becomes vmcnt(0) lgkmcnt(0)
    a7_no_opsel_lo only the instructions whose LOW half selects the pair's HIGH register (op_sel:[1,0,0] / [0,1,0]; 16 of the 1024
                   modifier forms of the input-gradient kernel) are rewritten through a free register
    s1_sgpr        source level: the broadcast value through v_readfirstlane (an SGPR operand instead of a VGPR pair)
    s2_occ1        source level: __launch_bounds__(256, 1) on the two kernels
    s3_scalar      source level: two v_fma_f32 instead of the packed form (round 5's finding: must NOT fail)
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cta_gan_amd import build as B  # noqa: E402

FAIL_REV = "2ebab5e"
LLVM = "/opt/rocm/lib/llvm/bin"
OUT = os.path.join(B.OUT, "diag")
CSRC = B.CSRC
FLAGS = B.FLAGS  # feature ON: no NO_PK_F32


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("%s\n%s" % (" ".join(cmd), r.stderr[-3000:]))
    return r.stdout


def failing_source() -> str:
    src = run(["git", "show", "%s:cta_gan_amd/csrc/conv_cout1.hip" % FAIL_REV], cwd=ROOT)
    return src.replace('#include "common.h"', '#include "%s/common.h"' % CSRC)


def kernel_spans(lines, name_part):
    """(start, end) line index ranges of every kernel whose label contains name_part"""
    spans, start = [], None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*%s\w*:" % name_part, l):
            start = i
        elif start is not None and l.strip().startswith("s_endpgm"):
            spans.append((start, i))
            start = None
    return spans


PK_MOD = re.compile(r"^\s*v_pk_(fma|mul|add)_f32 .*op_sel")
VREG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def edit(lines, variant):
    out = list(lines)
    for name in ("cout1_bwd_kernel", "cout1_wgrad_kernel"):
        for (a, b) in reversed(kernel_spans(out, name)):
            body = out[a:b + 1]
            if variant == "a1_nop":
                nb = []
                for l in body:
                    nb.append(l)
                    if l.strip().startswith("s_waitcnt") and "lgkmcnt" in l:
                        nb += ["\ts_nop 7\n", "\ts_nop 7\n"]
                body = nb
            elif variant == "a3_nop_pk":
                nb = []
                for l in body:
                    if PK_MOD.match(l):
                        nb.append("\ts_nop 0\n")
                    nb.append(l)
                body = nb
            elif variant == "a3b_nop7_pk":
                nb = []
                for l in body:
                    if PK_MOD.match(l):
                        nb.append("\ts_nop 7\n")
                    nb.append(l)
                body = nb
            elif variant in ("a5_lgkm0", "a6_allcnt0"):
                # every wait on LDS data becomes a wait for ALL of it (a6: and for every global load): no LDS return is in flight
                # while packed instructions read their operands
                nb = []
                for l in body:
                    if l.strip().startswith("s_waitcnt"):
                        if variant == "a6_allcnt0":
                            l = "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n"
                        elif "lgkmcnt" in l:
                            l = re.sub(r"lgkmcnt\(\d+\)", "lgkmcnt(0)", l)
                    nb.append(l)
                body = nb
            elif variant == "a7_no_opsel_lo":
                # ONLY the instructions whose LOW half selects the pair's HIGH register (op_sel:[1,0,0] / [0,1,0]: the splat of
                # an odd register) are rewritten: the register is copied into a free even register T and broadcast from there with
                # the op_sel_hi form that the other ~4000 instructions of the kernel use
                m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", "".join(out[a:a + 40000]))
                nfree = int(m.group(1))
                top = (nfree + 7) // 8 * 8
                T = top - 2
                if T < nfree or top > 256:
                    print("   %s: no free register pair (next_free_vgpr %d): left as it is" % (name, nfree))
                else:
                    nb, n = [], 0
                    for l in body:
                        mm = re.match(r"^(\s*v_pk_fma_f32 )(v\[\d+:\d+\]), (v\[\d+:\d+\]), (v\[\d+:\d+\]), (\S+) op_sel:\[(\d),(\d),0\](?: op_sel_hi:\[1,1,0\])?\s*$", l)
                        if mm and PK_MOD.match(l):
                            d, s0, s1, s2 = mm.group(2), mm.group(3), mm.group(4), mm.group(5)
                            which = 0 if mm.group(6) == "1" else 1
                            src = (s0, s1)[which]
                            hi = int(re.match(r"v\[\d+:(\d+)\]", src).group(1))
                            nb.append("\tv_mov_b32_e32 v%d, v%d\n" % (T, hi))
                            ops = [s0, s1]
                            ops[which] = "v[%d:%d]" % (T, T + 1)
                            hi_mod = ["1", "1", "0" if s2 == "0" else "1"]
                            hi_mod[which] = "0"
                            nb.append("%s%s, %s, %s, %s op_sel_hi:[%s]\n" % (mm.group(1), d, ops[0], ops[1], s2, ",".join(hi_mod)))
                            n += 1
                        else:
                            assert not re.search(r"op_sel:\[", l), l
                            nb.append(l)
                    print("   %s: %d low-half-selects-high-register instructions rewritten through v[%d:%d]" % (name, n, T, T + 1))
                    body = nb
            elif variant == "a4_setprio":
                body = [body[0], "\ts_setprio 3\n"] + body[1:]
            elif variant == "a2_init":
                written, read_hi = set(), set()
                for l in body:
                    s = l.strip()
                    if not s or s.startswith((";", ".", "s_")) or s.endswith(":"):
                        continue
                    ops = s.split(None, 1)
                    if len(ops) < 2:
                        continue
                    args = [x.strip() for x in ops[1].split(",")]
                    # destination(s): the first operand of VALU / DS / global loads (stores have none that matter here)
                    if not ops[0].startswith(("global_store", "ds_write", "buffer_store", "scratch_store")):
                        m = VREG.match(args[0])
                        if m:
                            if m.group(3):
                                written.add(int(m.group(3)))
                            else:
                                written.update(range(int(m.group(1)), int(m.group(2)) + 1))
                    if PK_MOD.match(l):
                        for x in args[1:]:
                            m = re.match(r"v\[(\d+):(\d+)\]", x)
                            if m:
                                read_hi.add(int(m.group(2)))
                never = sorted(read_hi - written)
                print("   %s: VGPRs read as a pair's upper half and never written: %s" % (name, never))
                body = [body[0]] + ["\tv_mov_b32_e32 v%d, 0\n" % r for r in never] + body[1:]
            out[a:b + 1] = body
    return out


def source_variant(src, variant):
    if variant == "s1_sgpr":
        src = src.replace("const float g1 = pr[ky][i - kx + KS - 1];",
                          "const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pr[ky][i - kx + KS - 1])));")
        src = src.replace("const float g1 = sp[ky * PW + i - kx + KS - 1];",
                          "const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sp[ky * PW + i - kx + KS - 1])));")
    elif variant == "s2_occ1":
        src = src.replace("__launch_bounds__(256, 2) void cout1_bwd_kernel", "__launch_bounds__(256, 1) void cout1_bwd_kernel")
    elif variant == "s3_scalar":
        src = src.replace("for (int q = 0; q < 4; ++q) d[q] = gv * wr[ky * KS + kx][q] + d[q];",
                          "for (int q = 0; q < 4; ++q) { d[q][0] = __builtin_fmaf(g1, wr[ky * KS + kx][q][0], d[q][0]); d[q][1] = __builtin_fmaf(g1, wr[ky * KS + kx][q][1], d[q][1]); }")
        src = src.replace("for (int q = 0; q < 4; ++q) acc[ky * KS + kx][q] = gv * v[q] + acc[ky * KS + kx][q];",
                          "for (int q = 0; q < 4; ++q) { acc[ky * KS + kx][q][0] = __builtin_fmaf(g1, v[q][0], acc[ky * KS + kx][q][0]); acc[ky * KS + kx][q][1] = __builtin_fmaf(g1, v[q][1], acc[ky * KS + kx][q][1]); }")
    return src


def build_variant(variant, src, others):
    d = os.path.join(OUT, variant)
    os.makedirs(d, exist_ok=True)
    hip = os.path.join(d, "conv_cout1.hip")
    with open(hip, "w") as fh:
        fh.write(source_variant(src, variant))
    hipcc = B._hipcc()
    extra = ["-fno-slp-vectorize"] if variant == "s3_scalar" else []
    asm = os.path.join(d, "dev.s")
    run([hipcc, *FLAGS, *extra, "--cuda-device-only", "-S", hip, "-o", asm])
    lines = open(asm).readlines()
    if variant.startswith("a"):
        lines = edit(lines, variant)
        with open(asm, "w") as fh:
            fh.writelines(lines)
    n_mod = sum(1 for l in lines if PK_MOD.match(l))
    run([LLVM + "/clang", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", os.path.join(d, "dev.o")])
    run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", os.path.join(d, "dev.out"),
         os.path.join(d, "dev.o")])
    run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
         "-input=" + os.path.join(d, "dev.out"), "-output=" + os.path.join(d, "dev.hipfb")])
    obj = os.path.join(d, "conv_cout1.o")
    run([hipcc, *FLAGS, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", os.path.join(d, "dev.hipfb"), "-c", hip,
         "-o", obj])
    lib = os.path.join(OUT, "libctagan_hip_%s.so" % variant)
    run([hipcc, "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", lib, *others, obj])
    print("built %s (%d modifier-form packed fp32 instructions in its conv_cout1 code object)" % (lib, n_mod), flush=True)
    return lib


def main():
    os.makedirs(OUT, exist_ok=True)
    hipcc = B._hipcc()

    def compile_pk(src):
        obj = os.path.join(OUT, "pk_" + src[:-4] + ".o")
        if not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(os.path.join(CSRC, src)):
            run([hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        others = list(ex.map(compile_pk, [s for s in B.sources() if s != "conv_cout1.hip"]))
    src = failing_source()
    variants = sys.argv[1:] or ["a0_control", "a1_nop", "a2_init", "a3_nop_pk", "a4_setprio", "s1_sgpr", "s2_occ1", "s3_scalar"]
    for v in variants:
        build_variant(v, src, others)


if __name__ == "__main__":
    main()
