"""The C-ABI library loads on a GPU-less host, exports every symbol include/ipm_hip.h declares,
and the product path fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import pytest

import interiorpointmethod_amd as ipm
from interiorpointmethod_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ipm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ipm_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_header_symbols_all_exported(built_lib):
    lib = C.CDLL(built_lib)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(lib, sym), "libipm_hip.so does not export %s" % sym
    assert sorted(_lib.EXPORTS) == declared          # the ctypes binding covers the whole header


def test_abi_version_and_host_only_calls(built_lib):
    lib = ipm.load_library()
    assert lib.ipm_abi_version() == _lib.ABI_VERSION == 4
    opts = _lib.Options()
    lib.ipm_default_options(C.byref(opts))
    assert opts.eta == 0.91 and opts.pivot_guard_big == 1e64 and opts.check_every >= 1   # main.py:607
    nbytes = C.c_size_t(0)
    assert lib.ipm_workspace_bytes(4096, 8192, C.byref(nbytes)) == 0
    # A (4096x8192) + B (4096^2) + 32 inverse diagonal blocks, plus vectors
    assert nbytes.value >= 8 * (4096 * 8192 + 4096 * 4096 + 32 * 128 * 128)
    assert nbytes.value < 8 * (4096 * 8192 + 4096 * 4096) * 1.2
    assert lib.ipm_workspace_bytes(0, 5, C.byref(nbytes)) == -1
    assert b"bad arguments" in lib.ipm_last_error(None)


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Options) == 3 * 8 + 2 * 4 + 8 + 8
    assert C.sizeof(_lib.Stats) == 4 * 4 + 15 * 8
    assert C.sizeof(_lib.IterRecord) == 2 * 4 + 10 * 8


def test_no_cpu_fallback_without_device(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises((ipm.IpmLibraryError, ipm.IpmError)):
        ipm.solve([[1.0, 2.0]], [1.0], [1.0, 1.0])
    lib = ipm.load_library()
    h = C.c_void_p()
    rc = lib.ipm_create(0, 4, 8, None, None, 0, None, C.byref(h))
    assert rc == -3 and not h.value                      # IPM_ERR_NO_DEVICE, never a CPU path


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "interiorpointmethod_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
