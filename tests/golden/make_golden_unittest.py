"""Golden vector for the reference's OWN unit-test parameters (unittests/houghsht.cxx:17-21,54-76): Canny(0.8, 1.6) and
CompVHough::newObj(COMPV_HOUGHSHT_ID, rho = 1, theta = kfMathTrigPiOver180, threshold = 100).  The unit test passes
kfMathTrigPiOver180 (0.01745...) where the factory expects DEGREES, so the reference really runs with a 0.01745-degree
theta step: T = 10313 theta bins.  The unit test's images are not in the tree; this script runs the same calls with the
REAL CompV library (oracle/_ref) on the synthetic 1282x720 frame (the size of the unit test's first image) and records
what the unit test checks: the number of lines, sum(rho), sum(theta), sum(strength).  Build container only:

    python tests/golden/make_golden_unittest.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_bindings import RefShim, md5_rows, synth_frame  # noqa: E402

PI_OVER_180 = float(np.float32(3.1415926535897932384626433) / np.float32(180.0))   # kfMathTrigPiOver180 (base/math/compv_math.cxx:30)

CASES = [("unittest_1282x720", 1282, 720, 12345), ("unittest_200x258", 200, 258, 12345), ("unittest_320x240", 320, 240, 777)]


def main():
    ref = RefShim(1)
    assert ref.avx2
    meta = {}
    for name, W, H, seed in CASES:
        img = synth_frame(W, H, seed)
        rc, can = ref.canny(img, 0.8, 1.6)
        assert rc == 0
        t0 = time.time()
        lines = ref.sht(can, PI_OVER_180, 100, cap=1 << 22)
        dt = time.time() - t0
        rho = np.array([l[0] for l in lines], np.float64)
        theta = np.array([l[1] for l in lines], np.float64)
        strength = np.array([l[2] for l in lines], np.int64)
        meta[name] = {"W": W, "H": H, "seed": seed, "tLow": 0.8, "tHigh": 1.6, "theta_deg": PI_OVER_180, "threshold": 100,
                      "canny_md5": md5_rows(can), "canny_edges": int((can != 0).sum()), "lines": len(lines),
                      "sum_rho": float(rho.sum()), "sum_theta": float(theta.sum()), "sum_strength": int(strength.sum()),
                      "max_strength": int(strength.max()) if len(lines) else 0}
        print(name, meta[name], "%.1f s" % dt)
    with open(os.path.join(HERE, "golden_unittest.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
