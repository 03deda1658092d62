"""GPU test of the drop-in boundary against the REAL CompV library: integration/_build/headless_samples runs the
samples' call sequences (samples/edges_canny, samples/hough_lines) through CompVEdgeDete::newObj / CompVHough::newObj
twice -- stock CPU factories, then the HIP factories registered by id with CompVFeature::addFactory -- and compares.
The binary is built in the build container (integration/build.sh needs the CompV checkout) and travels as a prebuilt."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "integration", "_build", "headless_samples")


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,frames", [(641, 480, 1), (1280, 720, 2), (1920, 1080, 1)])
def test_real_compv_with_hip_factories(W, H, frames):
    if not os.path.exists(BIN):
        pytest.skip("integration/_build/headless_samples not built (needs a CompV checkout)")
    r = subprocess.run([BIN, str(W), str(H), str(frames)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert "DROP-IN PARITY OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.returncode == 0
