"""CPU: the oracle with the reference's own unit-test parameters (unittests/houghsht.cxx:17-21 -- Canny(0.8,1.6), theta step
kfMathTrigPiOver180 'degrees' = 10313 theta bins, threshold 100) against the fixture the compiled reference produced
(tests/golden/make_golden_unittest.py)."""
import json
import os

import numpy as np
import pytest

from oracle_bindings import md5_rows, synth_frame

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name", ["unittest_200x258", "unittest_320x240"])
def test_oracle_reference_unittest_parameters(oracle, name):
    with open(os.path.join(HERE, "golden", "golden_unittest.json")) as f:
        m = json.load(f)[name]
    W, H = m["W"], m["H"]
    rc, can = oracle.canny(synth_frame(W, H, m["seed"]), m["tLow"], m["tHigh"])
    assert rc == 0 and md5_rows(can) == m["canny_md5"] and int((can != 0).sum()) == m["canny_edges"]
    R, T, _ = oracle.sht_dims(W, H, m["theta_deg"])
    assert T == 10313
    lines = oracle.sht(can, m["theta_deg"], m["threshold"])
    assert len(lines) == m["lines"]
    assert float(np.sum(np.array([np.float32(l[0]) for l in lines], np.float64))) == m["sum_rho"]
    assert abs(float(np.sum(np.array([np.float32(l[1]) for l in lines], np.float64))) - m["sum_theta"]) <= 0.0009765625   # houghsht.cxx:72
    assert int(sum(l[2] for l in lines)) == m["sum_strength"]
    assert max(l[2] for l in lines) == m["max_strength"]
