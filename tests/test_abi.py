"""CPU: the C-ABI library loads, exports every symbol include/ctagan_hip.h declares, and the ctypes
signature table in cta_gan_amd/_lib.py matches the header parameter for parameter.  No compute calls."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_header(name="ctagan_hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(ctg_\w+)\s*\((.*?)\)\s*;", text, flags=re.S):
        sig = ""
        for p in m.group(2).split(","):
            p = p.strip()
            if p == "void":
                continue
            if "*" in p:
                sig += "p"
            elif re.search(r"\blong\b", p):
                sig += "l"
            elif re.search(r"\bfloat\b", p):
                sig += "f"
            elif re.search(r"\bdouble\b", p):
                sig += "d"
            elif re.search(r"\bint\b", p):
                sig += "i"
            else:
                raise AssertionError("unparsed parameter %r in %s" % (p, m.group(1)))
        decls[m.group(1)] = sig
    return decls


def test_header_and_ctypes_table_agree():
    from cta_gan_amd import _lib
    decls = parse_header()
    assert len(decls) >= 25
    assert set(decls) == set(_lib.SIGNATURES)
    for name, sig in decls.items():
        assert _lib.SIGNATURES[name] == sig, name


def test_library_builds_loads_and_exports_all_symbols():
    from cta_gan_amd import _lib, build
    build.build()
    lib = _lib.load()
    for name in parse_header():
        assert hasattr(lib, name), name


def test_diagnostics_are_declared_apart_from_the_product_abi():
    """include/ctagan_hip_diag.h: entry points the tree's own standing checks call (the LDS canary) ship in the library but are
    declared apart from the product ABI -- no reference call site corresponds to them and an integration never binds them."""
    from cta_gan_amd import _lib
    diag = parse_header("ctagan_hip_diag.h")
    assert diag == _lib.DIAG_SIGNATURES and diag
    assert not (set(diag) & set(parse_header())) and not (set(diag) & set(_lib.SIGNATURES))
    lib = _lib.load()
    for name in diag:
        assert hasattr(lib, name), name


def test_hip_sources_define_exactly_the_declared_symbols():
    decls = parse_header()
    found = set()
    csrc = os.path.join(ROOT, "cta_gan_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith(".hip"):
            found |= set(re.findall(r'extern "C" int (ctg_\w+)', open(os.path.join(csrc, f)).read()))
    assert found == set(decls) | set(parse_header("ctagan_hip_diag.h"))


def test_epilogue_struct_layout_matches_ctypes(tmp_path):
    """`ctg_conv_epilogue` is the one aggregate in the ABI: its C layout (gcc on the published header, which is plain C)
    equals the ctypes.Structure the binding passes."""
    import ctypes
    import subprocess
    from cta_gan_amd import ops
    fields = [f[0] for f in ops.ConvEpilogue._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ctagan_hip.h"\nint main(void) {\n'
                   + '  printf("%zu\\n", sizeof(ctg_conv_epilogue));\n'
                   + "".join('  printf("%%zu\\n", offsetof(ctg_conv_epilogue, %s));\n' % f for f in fields)
                   + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(ops.ConvEpilogue)
    for f, off in zip(fields, out[1:]):
        assert getattr(ops.ConvEpilogue, f).offset == off, f
