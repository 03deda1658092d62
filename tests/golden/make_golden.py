"""Generate the golden vectors in this directory with the REAL CompV library (oracle/_ref, built from
/root/reference by oracle/build_ref.sh).  Run in the build container only:

    python tests/golden/make_golden.py

Inputs are the deterministic synthetic frames of SURVEY.md 8(d) (regenerated from (W, H, seed) by
tests/oracle_bindings.synth_frame), so only expected OUTPUTS are stored: MD5 of the uint8 Sobel / Canny maps
(over the valid bytes of each row, the reference's compv_tests_md5 convention), edge-pixel counts, and the SHT line
sets (rho, theta, strength) in canonical order.  The small cases also store the maps themselves (bit-packed).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_bindings import RefShim, md5_rows, synth_frame  # noqa: E402

CASES = [
    # name, W, H, seed, (tLow, tHigh), sht (thetaDeg, threshold) or None, store_maps
    ("tiny_20x20", 20, 20, 12345, (59.0, 119.0), None, True),
    ("q1_200x258", 200, 258, 12345, (59.0, 119.0), (1.0, 30), True),          # quirk Q1 (gmax lane subset)
    ("small_320x240", 320, 240, 12345, (59.0, 119.0), (1.0, 40), True),
    ("q3_641x480", 641, 480, 12345, (59.0, 119.0), (1.0, 100), True),        # quirk Q3 (column coverage)
    ("ragged_333x77", 333, 77, 777, (0.8, 1.6), (1.0, 20), True),             # dense edges, ragged size
    ("dense_1282x720", 1282, 720, 12345, (0.8, 1.6), (1.0, 100), False),      # quirk Q2 (SHT NMS columns)
    ("hd_1280x720", 1280, 720, 12345, (59.0, 119.0), (1.0, 100), False),
    ("fhd_1920x1080", 1920, 1080, 12345, (59.0, 119.0), (1.0, 100), False),
    ("uhd_3840x2160", 3840, 2160, 12345, (59.0, 119.0), (1.0, 100), False),
    ("fhd_seed7", 1920, 1080, 12352, (59.0, 119.0), (1.0, 100), False),       # batch frame f uses seed 12345+f
    ("theta_half_640x480", 640, 480, 12345, (59.0, 119.0), (0.5, 50), False),
    ("mean_640x480", 640, 480, 12345, (0.68, 1.36), None, True),              # PERCENT_OF_MEAN thresholds
]


FULL_SIZE_EXTRAS = ("hd_1280x720", "fhd_1920x1080", "uhd_3840x2160")

KHT_CASES = ("small_320x240", "q3_641x480", "ragged_333x77", "hd_1280x720", "fhd_1920x1080", "uhd_3840x2160", "dense_1282x720")


def main():
    ref = RefShim(1)
    assert ref.avx2, "goldens must come from the AVX2 intrinsics path"
    meta = {}
    arrays = {}
    for name, W, H, seed, (tl, th), sht, store in CASES:
        img = synth_frame(W, H, seed)
        sob = ref.sobel(img)
        typ = 1 if name.startswith("mean_") else 0
        rc, can = ref.canny(img, tl, th, 3, typ)
        assert rc == 0
        m = {"W": W, "H": H, "seed": seed, "tLow": tl, "tHigh": th, "threshold_type": typ,
             "input_md5": md5_rows(img), "sobel_md5": md5_rows(sob), "canny_md5": md5_rows(can),
             "canny_edges": int((can != 0).sum())}
        if name in FULL_SIZE_EXTRAS:
            # the packed 5x5 Canny kernel and the Scharr / Prewitt detectors at the sizes the bench times them
            # (5x5 thresholds: the 3x3 pair, and x 12 = the 5x5 kernel's gain on a step edge, the bench's "same edge density" point)
            for tag, (l5, h5) in (("canny5", (tl, th)), ("canny5_x12", (tl * 12.0, th * 12.0))):
                rc5, can5 = ref.canny(img, l5, h5, 5, 0)
                assert rc5 == 0
                m[tag] = {"tLow": l5, "tHigh": h5, "md5": md5_rows(can5), "edges": int((can5 != 0).sum())}
            m["scharr_md5"] = md5_rows(ref.edge_dete(img, 2))
            m["prewitt_md5"] = md5_rows(ref.edge_dete(img, 3))
            assert md5_rows(ref.edge_dete(img, 0)) == m["sobel_md5"]
        if store:
            arrays[name + "/sobel"] = sob
            arrays[name + "/canny_bits"] = np.packbits(can != 0, axis=1)
        if sht:
            deg, thr = sht
            lines = ref.sht(can, deg, thr)
            # canonical order: strength desc, then rho desc (= accumulator row asc), then theta asc
            lines = sorted(lines, key=lambda l: (-l[2], -l[0], l[1]))
            m["sht"] = {"theta_deg": deg, "threshold": thr, "lines": len(lines),
                        "sum_strength": int(sum(l[2] for l in lines)),
                        "sum_rho": float(np.sum(np.array([l[0] for l in lines], np.float64))),
                        "sum_theta": float(np.sum(np.array([l[1] for l in lines], np.float64)))}
            keep = lines if len(lines) <= 4096 else lines[:4096]
            arrays[name + "/sht_lines"] = np.array([(l[0], l[1], l[2]) for l in keep], np.float64).reshape(-1, 3)
        if name in KHT_CASES:
            kl, gs = ref.kht(can, 1.0, 1.0, 1)
            m["kht"] = {"rho": 1.0, "theta_deg": 1.0, "threshold": 1, "lines": len(kl), "gs": repr(gs),
                        "sum_strength": int(sum(l[2] for l in kl))}
            arrays[name + "/kht_lines"] = np.array([(l[0], l[1], l[2]) for l in kl], np.float64).reshape(-1, 3)  # reference order
        meta[name] = m
        print(name, m["canny_edges"], m.get("sht", {}).get("lines"))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_arrays.npz"), **arrays)


if __name__ == "__main__":
    main()
