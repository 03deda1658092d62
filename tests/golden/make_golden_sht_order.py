"""Golden vectors for the ORDER of the reference's SHT line list: CompVHoughSht::process sorts with an unstable std::sort on
the strength alone (core/features/hough/compv_core_feature_houghsht.cxx:241-249), so the order inside equal-strength groups
-- and which of them survive maxLines -- is a property of the reference built with this toolchain.  This script runs the REAL
CompV library (oracle/_ref) and stores, per case, the exact list it returns (rho and theta as float32 bit patterns, strength)
as an md5 plus the first and last 64 entries.  Build container only:

    python tests/golden/make_golden_sht_order.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_bindings import RefShim, libstdcxx_version, md5_rows, synth_frame  # noqa: E402

# name, W, H, seed, tLow, tHigh, theta_deg, threshold, maxLines
CASES = [("vga_all", 640, 480, 77, 59.0, 119.0, 1.0, 30, 0), ("vga_top100", 640, 480, 77, 59.0, 119.0, 1.0, 30, 100),
         ("hd_halfdeg", 1280, 720, 77, 59.0, 119.0, 0.5, 60, 0), ("ragged_top40", 333, 77, 77, 0.8, 1.6, 0.5, 5, 40),
         ("calib_like", 1280, 720, 5, 59.0, 119.0, 0.5, 5, 1080)]


def pack(lines):
    a = np.zeros((len(lines), 3), np.uint32)
    a[:, 0] = np.array([l[0] for l in lines], np.float32).view(np.uint32)
    a[:, 1] = np.array([l[1] for l in lines], np.float32).view(np.uint32)
    a[:, 2] = np.array([l[2] for l in lines], np.uint32)
    return a


def main():
    ref = RefShim(1)
    meta = {}
    for name, W, H, seed, tl, th, deg, thr, maxl in CASES:
        img = synth_frame(W, H, seed)
        rc, can = ref.canny(img, tl, th)
        assert rc == 0
        a = pack(ref.sht(can, deg, thr, maxl))
        s = a[:, 2]
        meta[name] = {"W": W, "H": H, "seed": seed, "tLow": tl, "tHigh": th, "theta_deg": deg, "threshold": thr, "max_lines": maxl,
                      "canny_md5": md5_rows(can), "lines": int(len(a)), "equal_strength_pairs": int((np.diff(s.astype(np.int64)) == 0).sum()),
                      "md5": hashlib.md5(a.tobytes()).hexdigest(), "head": a[:64].tolist(), "tail": a[-64:].tolist()}
        print(name, len(a), meta[name]["equal_strength_pairs"], meta[name]["md5"])
    meta["_runtime"] = {"libstdcxx": libstdcxx_version(), "note": "tie order = this runtime's std::sort; a different libstdc++ may order equal strengths differently"}
    with open(os.path.join(HERE, "golden_sht_order.json"), "w") as f:
        json.dump(meta, f, sort_keys=True)


if __name__ == "__main__":
    main()
