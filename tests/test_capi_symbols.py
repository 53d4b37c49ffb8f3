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
