"""CPU: the C-ABI library loads and exports every symbol include/gridmm.h declares (no compute calls)."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "gridmm.h")).read()
    return sorted(set(re.findall(r"^int (gridmm_\w+)\(", txt, flags=re.M)))


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
        assert name in _lib.SIGNATURES, "ctypes prototype missing for %s" % name
    assert lib.gridmm_abi_version() == _lib.ABI_VERSION


def test_ops_refuse_cpu_tensors_loudly():
    import torch
    from gridmm_amd import ops, _lib
    pw = None
    with pytest.raises(_lib.GridmmLibraryError):
        ops.PackedLinear(torch.zeros(8, 8))           # CPU weight -> no silent fallback
    with pytest.raises(_lib.GridmmLibraryError):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-5)
