"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/compv_hip.h
declares; without a GPU the product fails loudly instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "compv_hip.h")).read()
    return sorted(set(re.findall(r"COMPVHIP_API\s+[\w\s\*]+?\b(compvhip_\w+)\s*\(", txt)))


def test_header_declares_the_binding_list():
    from compv_amd import capi
    assert _declared() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    from compv_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in _declared():
        assert hasattr(lib, s), s


def test_dims_helper_matches_reference_geometry():
    from compv_amd import capi
    lib = capi.load()
    R, T, st = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_float()
    assert lib.compvhip_houghsht_dims(1920, 1080, 1.0, ctypes.byref(R), ctypes.byref(T), ctypes.byref(st)) == 0
    assert (R.value, T.value) == (6001, 180)          # R = 2(W+H)+1, T = round(pi/theta)
    assert lib.compvhip_houghsht_dims(640, 480, 0.0, ctypes.byref(R), ctypes.byref(T), ctypes.byref(st)) == capi.E_INVALID_PARAMETER


def test_no_gpu_means_loud_failure():
    from compv_amd import capi
    lib = capi.load()
    if lib.compvhip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.CompvHipError):
        capi.Context(0)


def test_product_never_touches_the_oracle():
    """Nothing under compv_amd/ may import, link or open anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "compv_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)
