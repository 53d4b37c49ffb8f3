"""CPU: the C-ABI library loads and exports every symbol include/gridmm.h declares (no compute calls)."""
import os
import re

import pytest

from conftest import ROOT


def _header(debug=False):
    txt = open(os.path.join(ROOT, "include", "gridmm.h")).read()
    a, b = txt.index("#ifdef GRIDMM_DEBUG_HOOKS"), txt.index("#endif", txt.index("#ifdef GRIDMM_DEBUG_HOOKS"))
    return txt[a:b] if debug else txt[:a] + txt[b:]


def _declared(debug=False):
    return sorted(set(re.findall(r"^(?:int|size_t) (gridmm_\w+)\(", _header(debug), flags=re.M)))


def test_header_declares_the_hot_path_entry_points():
    names = _declared()
    for must in ("gridmm_grid_project", "gridmm_grid_bin", "gridmm_grid_aggregate", "gridmm_linear",
                 "gridmm_attention", "gridmm_layernorm", "gridmm_fuse_logits"):
        assert must in names


def test_library_loads_and_exports_every_declared_symbol():
    from gridmm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES or name.endswith("_workspace") or name.endswith("_bytes"), \
            "ctypes prototype missing for %s" % name
    assert lib.gridmm_abi_version() == _lib.ABI_VERSION


def test_shipping_library_has_no_debug_hooks_and_no_experimental_entry_points():
    """re-entrant, no global state.  The tuning overrides (process-global tables) live in the development build
    only (`make debug`, -DGRIDMM_DEBUG_HOOKS): the shipping .so exports no gridmm_debug_* symbol, the debug section of the
    header is the only place that declares one, and the two GEMM + LayerNorm experiments of round 4 are gone."""
    import subprocess
    from gridmm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\b(gridmm_\w+)\b", out)))
    assert exported, "nm found no gridmm_* symbol"
    bad = [n for n in exported if n.startswith("gridmm_debug") or "planes_ln" in n or "lnx" in n]
    assert not bad, bad
    assert sorted(exported) == sorted(_declared()), (set(exported) ^ set(_declared()))
    assert all(n.startswith("gridmm_debug_") for n in _declared(debug=True)) and _declared(debug=True)
    assert not [n for n in _declared() if n.startswith("gridmm_debug")]
    if os.path.exists(_lib.DEBUG_LIB_PATH) and os.path.getmtime(_lib.DEBUG_LIB_PATH) >= os.path.getmtime(_lib.LIB_PATH) - 600:
        # (a fresh development build: it exports the shipping surface + the hooks)
        dbg = subprocess.run(["nm", "-D", "--defined-only", _lib.DEBUG_LIB_PATH], capture_output=True, text=True, check=True).stdout
        dbg = set(re.findall(r"\b(gridmm_\w+)\b", dbg))
        assert set(_declared()) <= dbg and set(_declared(debug=True)) <= dbg


def _integration_snippet():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = txt[txt.index("## 3. C-ABI binding"):]
    return sec[sec.index("```python") + len("```python"):sec.index("```", sec.index("```python") + 10)]


def test_integration_snippet_names_the_current_abi():
    from gridmm_amd import _lib
    m = re.search(r"gridmm_abi_version\(\) == (\d+)", _integration_snippet())
    assert m and int(m.group(1)) == _lib.ABI_VERSION


@pytest.mark.gpu
def test_integration_snippet_runs_verbatim():
    """INTEGRATION.md section 3, executed as written (cwd = repo root): raw ctypes binding of gridmm_linear on device tensors."""
    import torch
    from gridmm_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(0)
    M, N, K = 48, 64, 96
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.1
    bias = torch.randn(N, device=dev)
    pw = ops.PackedLinear(W, bias)
    ns = dict(A=A, lda=K, w_hi=pw.hi, w_lo=pw.lo, Kp=pw.Kp, bias=bias, C=torch.empty(M, N, device=dev), ldc=N, M=M, N=N, K=K)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(_integration_snippet(), "INTEGRATION.md#3", "exec"), ns)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    want = (A.double() @ W.double().t() + bias.double()).float()
    assert float((ns["C"] - want).abs().max()) < 1e-4


def test_ops_refuse_cpu_tensors_loudly():
    import torch
    from gridmm_amd import ops, _lib
    pw = None
    with pytest.raises(_lib.GridmmLibraryError):
        ops.PackedLinear(torch.zeros(8, 8))           # CPU weight -> no silent fallback
    with pytest.raises(_lib.GridmmLibraryError):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-5)


# ---- header <-> ctypes contract ------------------------------------------------------------------------------------
def _prototypes(debug=False):
    """Every `int|size_t gridmm_*( ... );` prototype of include/gridmm.h as (name, return C type, [argument C types])."""
    txt = _header(debug)
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    out = []
    for m in re.finditer(r"^\s*(int|size_t)\s+(gridmm_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        ctypes_args = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                assert "(" not in a and "[" not in a, "unparsed declarator in %s: %r" % (name, a)
                if "*" in a:
                    ctypes_args.append("ptr")
                    continue
                toks = [t for t in a.split() if t not in ("const", "volatile")]
                assert len(toks) >= 2, "argument without a name in %s: %r" % (name, a)
                ctypes_args.append(" ".join(toks[:-1]))
        out.append((name, ret, ctypes_args))
    return out


def _ctype_of(c_type):
    import ctypes
    table = {"ptr": ctypes.c_void_p, "gridmm_stream_t": ctypes.c_void_p, "int": ctypes.c_int, "int32_t": ctypes.c_int,
             "float": ctypes.c_float, "int64_t": ctypes.c_int64, "long long": ctypes.c_int64, "size_t": ctypes.c_size_t,
             "uint64_t": ctypes.c_uint64, "unsigned long long": ctypes.c_uint64, "unsigned": ctypes.c_uint, "unsigned int": ctypes.c_uint, "uint32_t": ctypes.c_uint,
             "double": ctypes.c_double}
    assert c_type in table, "no ctypes mapping for C type %r" % c_type
    return table[c_type]


def _assert_binding_matches(protos, lib):
    import ctypes
    mismatches = []
    for name, ret, args in protos:
        fn = getattr(lib, name)
        want = [_ctype_of(a) for a in args]
        got = list(fn.argtypes) if fn.argtypes is not None else None
        if got is None:
            mismatches.append("%s: no argtypes bound" % name)
        elif len(got) != len(want):
            mismatches.append("%s: header has %d arguments, ctypes binds %d" % (name, len(want), len(got)))
        else:
            for i, (w, g) in enumerate(zip(want, got)):
                if w is not g:
                    mismatches.append("%s: argument %d is %s in the header, %s in _lib" % (name, i, args[i], g.__name__))
        want_ret = ctypes.c_size_t if ret == "size_t" else ctypes.c_int
        if fn.restype is not want_ret:
            mismatches.append("%s: returns %s in the header, restype is %s" % (name, ret, getattr(fn.restype, "__name__", fn.restype)))
    return mismatches


def test_ctypes_prototypes_equal_the_header_argument_by_argument():
    """include/gridmm.h is the contract; _lib.SIGNATURES (+ the explicit restype / argtypes lines of _lib.load) is a
    hand-written copy of it.  Parse every prototype, map the C types to ctypes and compare position by position, so an
    `int` that becomes `int64_t` (or an argument added on one side only) fails here instead of corrupting a call."""
    from gridmm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    protos = _prototypes()
    assert sorted(p[0] for p in protos) == _declared(), "the prototype parser and the name scan disagree"
    assert len(protos) >= 89
    assert not _assert_binding_matches(protos, lib)
    # every table entry is a declared prototype (no stale binding survives a removed entry point)
    assert set(_lib.SIGNATURES) <= set(p[0] for p in protos), set(_lib.SIGNATURES) - set(p[0] for p in protos)
    dbg = _prototypes(debug=True)
    assert sorted(p[0] for p in dbg) == sorted(_lib.DEBUG_SIGNATURES)
    for name, ret, args in dbg:
        assert ret == "int" and [_ctype_of(a) for a in args] == list(_lib.DEBUG_SIGNATURES[name]), name


def test_contract_test_sees_a_changed_argument_type(tmp_path, monkeypatch):
    """The check above must fail when ONE `int` of a prototype becomes `int64_t` in the header."""
    from gridmm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    real = open(os.path.join(ROOT, "include", "gridmm.h")).read()
    needle = "int M, int N, int K, int act, gridmm_stream_t stream);"
    assert needle in real
    mutated = real.replace(needle, "int M, int64_t N, int K, int act, gridmm_stream_t stream);", 1)
    import builtins
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if str(path).endswith(os.path.join("include", "gridmm.h")):
            import io
            return io.StringIO(mutated)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    bad = _assert_binding_matches(_prototypes(), lib)
    assert len(bad) == 1 and "int64_t in the header" in bad[0], bad
    # ... and a dropped argument
    monkeypatch.setattr(builtins, "open", real_open)
    mutated2 = real.replace(needle, "int M, int N, int K, gridmm_stream_t stream);", 1)

    def fake_open2(path, *a, **k):
        if str(path).endswith(os.path.join("include", "gridmm.h")):
            import io
            return io.StringIO(mutated2)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open2)
    bad = _assert_binding_matches(_prototypes(), lib)
    assert len(bad) == 1 and "arguments" in bad[0], bad
