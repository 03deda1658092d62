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
    # KHT vote map (initCoords, houghkht.cxx:501-541): rhoN = (sqrt(W^2 + H^2) + 1) / rho, T = 180 / theta
    assert capi.houghkht_dims(3840, 2160, 1.0, 1.0) == (180, 4406)
    with pytest.raises(capi.CompvHipError):
        capi.houghkht_dims(3840, 2160, 2.0, 1.0)      # rho must be in (0, 1]


def test_vote_grid_helper():
    """compvhip_houghsht_vote_grid: the tile grid a plan votes with (host arithmetic).  Batches take the smallest grid whose rho windows fit the LDS
    (4 x 3 at 4K, 2 x 2 at 1080p); a single frame -- a launch that would leave most CUs idle -- takes a finer one; every window fits 1264 rows."""
    from compv_amd import capi
    lib = capi.load()

    def grid(W, H, frames, deg=1.0):
        nx, ny, rw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = lib.compvhip_houghsht_vote_grid(W, H, ctypes.c_float(deg), frames, ctypes.byref(nx), ctypes.byref(ny), ctypes.byref(rw))
        return rc, nx.value, ny.value, rw.value
    assert grid(3840, 2160, 32)[:3] == (0, 4, 3)
    assert grid(1920, 1080, 32)[:3] == (0, 2, 2)
    assert grid(1280, 720, 32)[:3] == (0, 2, 1)
    rc, nx, ny, rw = grid(3840, 2160, 1)
    assert rc == 0 and nx * ny > 12
    for W, H, F in ((3840, 2160, 32), (3840, 2160, 1), (32767, 64, 1), (64, 32767, 1), (8192, 8192, 1), (17, 9, 4), (641, 480, 1)):
        rc, nx, ny, rw = grid(W, H, F)
        assert rc == 0 and nx >= 1 and ny >= 1 and 16 <= rw <= 1264 and rw % 16 == 0, (W, H, F, rc, nx, ny, rw)
    assert grid(640, 480, 0)[0] == capi.E_INVALID_PARAMETER
    assert grid(640, 480, 1, 0.0)[0] == capi.E_INVALID_PARAMETER


def test_host_cpu_budget_is_what_the_host_grants():
    """compvhip_host_cpu_budget = min(hardware threads, affinity mask, cgroup CPU quota): the default worker count of compvhip_plan_houghkht."""
    from compv_amd import capi
    n = capi.host_cpu_budget()
    want = os.cpu_count() or 1
    try:
        want = min(want, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q = open(path).read().split()
            if q[0] != "max":
                want = min(want, max(1, int(q[0]) // int(q[1])))
        except (OSError, ValueError, IndexError):
            pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0 and not os.path.exists("/sys/fs/cgroup/cpu.max"):
            want = min(want, max(1, quota // period))
    except (OSError, ValueError):
        pass
    assert n == want and n >= 1


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


def test_to_cartesian_host_helpers_match_oracle_and_reference(oracle):
    """compvhip_houghsht_to_cartesian / compvhip_houghkht_to_cartesian (host float32 arithmetic behind the C ABI, what the plugin
    classes call) against the oracle and, where the compiled reference is present, against CompVHoughSht / CompVHoughKht::toCartesian
    themselves: bit patterns of all four endpoints, including the theta == 0 branch."""
    import numpy as np
    from compv_amd import capi
    from oracle_bindings import RefShim, have_refshim
    rng = np.random.default_rng(7)
    W, H = 1282, 720
    lines = [(float(np.float32(r)), float(np.float32(t))) for r, t in zip(rng.uniform(-1500, 1500, 400), rng.uniform(0.0, 3.14159, 400))]
    lines += [(123.0, 0.0), (-77.5, 0.0), (0.0, float(np.float32(1.5707964))), (640.0, float(np.float32(3.1241393)))]
    ref = RefShim(1) if have_refshim() else None
    for kht in (False, True):
        got = capi.to_cartesian(W, H, lines, kht=kht)
        exp = oracle.sht_to_cartesian(W, H, lines, kht=kht)
        assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), kht
        if ref is not None:
            assert got.view(np.uint32).tolist() == ref.sht_to_cartesian(W, H, lines, kht=kht).view(np.uint32).tolist(), kht
    assert capi.to_cartesian(W, H, []).shape == (0, 4)
