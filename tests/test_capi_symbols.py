"""The C-ABI library loads and exports every symbol include/apexgpu.h declares (no GPU needed;
no compute call is made)."""
import ctypes as C
import os
import re

import pytest

import apex_solver_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "apexgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(apexgpu_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(pkg.capi.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    L = pkg.capi.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"libapexgpu.so does not export {n}"
    assert sorted(pkg.capi.SYMBOLS) == names  # the Python binding covers the whole header
    assert b"gfx950" in L.apexgpu_version()


def test_struct_layouts_match_header():
    # apexgpu_lm_config: int + 11 doubles + int (with padding) ; apexgpu_lm_iter: 8 doubles
    assert C.sizeof(pkg.capi.LmIterC) == 64
    assert C.sizeof(pkg.capi.LmConfigC) == 8 + 11 * 8 + 8
    assert C.sizeof(pkg.capi.LmResultC) == 8 + 5 * 8 + 16


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: without a visible MI355X the handle cannot even be created."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.capi.LinAlgError) as e:
        pkg.capi.Handle(4, 10, 30, 1, 0)
    assert e.value.kind == "DeviceError"


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under apex-solver_amd/ may reference it."""
    pk = os.path.join(ROOT, "apex-solver_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "ba_device.hpp", os.path.join(dp, f)


def test_host_block_cache_trims():
    """apexgpu_trim_host_cache answers without a GPU (nothing kept in a fresh process: 0 bytes) -- the set-up's host blocks
    are kept by the process for the next handle and this call hands them back."""
    import ctypes as C

    from apex_solver_amd import capi

    L = capi.load()
    n = C.c_int64(-1)
    assert L.apexgpu_trim_host_cache(C.byref(n)) == 0 and n.value >= 0
    assert L.apexgpu_trim_host_cache(None) == 0
