import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(d, "golden.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(d, "golden_arrays.npz"))
    return meta, arrays


@pytest.fixture(scope="session")
def refshim():
    from oracle_bindings import RefShim, have_refshim
    if not have_refshim():
        pytest.skip("oracle/_ref not built (needs /root/reference; see oracle/build_ref.sh)")
    return RefShim(1)


@pytest.fixture(scope="session")
def hip_ctx():
    """GPU context through the C ABI; the HIP extension is mandatory (no fallback)."""
    from compv_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
