"""CPU: the C-ABI library loads and exports every symbol include/unidom_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "unidom_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ud_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from unidom_amd import _lib
    so = _lib.build()
    L = ctypes.CDLL(so)
    declared = _header_symbols()
    assert declared, "no symbols parsed from the header"
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"declared in include/unidom_hip.h but not exported: {missing}"
    assert sorted(_lib.SYMBOLS) == declared, (sorted(_lib.SYMBOLS), declared)
    L.ud_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.ud_version()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under unidom_amd/ may reference it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "unidom_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+\"[^\"]*oracle", txt, flags=re.M):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from unidom_amd import _lib
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    try:
        _lib.lib()
    except _lib.UnidomError as e:
        assert "no fallback" in str(e).lower()
    else:
        raise AssertionError("expected UnidomError")


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """include/unidom_hip.h is the contract; unidom_amd/_lib.py re-declares its structs for ctypes.  A field added on one side
    only would corrupt the conf silently: compile the header with gcc and compare sizeof / offsetof of every field."""
    import ctypes as C
    import subprocess

    from unidom_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"ud_cloth_conf": _lib.ud_cloth_conf, "ud_mpm_conf": _lib.ud_mpm_conf, "ud_plb_conf": _lib.ud_plb_conf}
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "unidom_hip.h"', "int main(void) {"]
    for name, st in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in st._fields_:
            lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, st in structs.items():
        assert int(got[name]) == C.sizeof(st), (name, got[name], C.sizeof(st))
        for field, _ in st._fields_:
            assert int(got[f"{name}.{field}"]) == getattr(st, field).offset, (name, field)


def test_integration_stub_and_lib_structs_are_the_header(tmp_path):
    """INTEGRATION.md shows the reference-side ctypes binding; its struct block is generated from include/unidom_hip.h
    (tools/gen_binding_stub.py) and must be current, and unidom_amd/_lib.py's hand-written structs must list the same
    fields with the same ctypes."""
    import ctypes as C
    import importlib.util

    from unidom_amd import _lib
    spec = importlib.util.spec_from_file_location("gen_binding_stub", os.path.join(ROOT, "tools", "gen_binding_stub.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.current_block() == gen.render(), "run `python tools/gen_binding_stub.py --write`"
    parsed = dict(gen.parse_structs())
    assert set(parsed) == {"ud_cloth_conf", "ud_mpm_conf", "ud_plb_conf"}
    for name, fields in parsed.items():
        mine = [(f, t) for f, t in getattr(_lib, name)._fields_]
        theirs = [(f, eval(ct, {"C": C})) for f, ct, _ in fields]
        assert [f for f, _ in mine] == [f for f, _ in theirs], name
        for (f, a), (_, b) in zip(mine, theirs):
            assert C.sizeof(a) == C.sizeof(b) and a._type_ == b._type_ if hasattr(a, "_type_") else a is b, (name, f)
