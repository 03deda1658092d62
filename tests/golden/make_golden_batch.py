"""Expected outputs of the BENCHMARK workload itself (BASELINE.json config 4: 256 frames of 3840x2160, seeds 12345 .. 12345+255,
Canny(59,119) -> SHT(rho 1, theta 1 deg, threshold 100)), produced by the REAL CompV library (oracle/_ref, AVX2 intrinsics path,
one thread: the multi-threaded gradient of the reference races).  Run in the build container only:

    python tests/golden/make_golden_batch.py            (about a minute)  -> golden_batch.json
    python tests/golden/make_golden_batch.py fhd        (1920 x 1080, 64 frames: bench.py's configs_extra)  -> golden_batch_fhd.json
    python tests/golden/make_golden_batch.py kht        (BASELINE config 5 on the first resident batch: 32 frames of 3840 x 2160, seeds 12345 .. 12376,
                                                         CompVHoughKht rho 1, theta 1 deg, threshold 1 on the reference's Canny maps)  -> golden_batch_kht.json

Per frame: MD5 of the edge map (rows of W bytes), number of edge pixels, number of lines, sum of their strengths and an
order-independent 64-bit hash of the line set (the reference leaves the order of equal-strength lines to an unstable sort):
    line_hash = sum over lines of ((rho + 32768) * 1000003 + col * 7919 + strength * 31337)   mod 2^64
with rho the integer accumulator rho (= barrier - row) and col = the theta bin.  What the reference's own tests assert for these
stages: unittests/canny.cxx:43-54 (MD5 of the edge map), unittests/houghsht.cxx:54-76 (line count + sums).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_bindings import RefShim, md5_rows, synth_frame  # noqa: E402

FHD = len(sys.argv) > 1 and sys.argv[1] == "fhd"
W, H = (1920, 1080) if FHD else (3840, 2160)
FIRST_SEED, FRAMES = 12345, (64 if FHD else 256)
T_LOW, T_HIGH, THETA_DEG, THRESHOLD = 59.0, 119.0, 1.0, 100
M64 = (1 << 64) - 1


def line_hash(rho, col, strength):
    """rho, col, strength: integer numpy arrays of one frame's lines (any order)."""
    rho = np.asarray(rho, np.int64).astype(np.uint64)
    col = np.asarray(col, np.int64).astype(np.uint64)
    st = np.asarray(strength, np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        v = (rho + np.uint64(32768)) * np.uint64(1000003) + col * np.uint64(7919) + st * np.uint64(31337)
        return int(v.sum(dtype=np.uint64)) & M64


def kht_main():
    """Per frame: number of KHT lines, sum of their strengths, gs (repr: every digit) and an ORDER-DEPENDENT hash of the list (the KHT list's order
    is part of the parity contract): h = (h * 1000003 + bits(rho f32) * 7919 + bits(theta f32) * 31337 + strength) mod 2^64 over the lines in order."""
    ref = RefShim(1)
    assert ref.avx2, "goldens must come from the AVX2 intrinsics path"
    frames = []
    for f in range(32):
        img = synth_frame(3840, 2160, FIRST_SEED + f)
        rc, can = ref.canny(img, T_LOW, T_HIGH, 3, 0)
        assert rc == 0
        kl, gs = ref.kht(can, 1.0, THETA_DEG, 1)
        frames.append({"seed": FIRST_SEED + f, "canny_md5": md5_rows(can), "lines": len(kl), "sum_strength": int(sum(l[2] for l in kl)), "gs": repr(gs),
                       "list_hash": "%016x" % kht_list_hash([l[0] for l in kl], [l[1] for l in kl], [l[2] for l in kl])})
        print(f, frames[-1], flush=True)
    out = {"W": 3840, "H": 2160, "tLow": T_LOW, "tHigh": T_HIGH, "rho": 1.0, "theta_deg": THETA_DEG, "threshold": 1, "first_seed": FIRST_SEED,
           "source": "CompV (oracle/_ref, AVX2 intrinsics, 1 thread)", "frames": frames}
    with open(os.path.join(HERE, "golden_batch_kht.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)


def kht_list_hash(rho, theta, strength):
    rb = np.asarray(rho, np.float32).view(np.uint32).astype(np.uint64)
    tb = np.asarray(theta, np.float32).view(np.uint32).astype(np.uint64)
    st = np.asarray(strength, np.int64).astype(np.uint64)
    h = 0
    for r, t, s in zip(rb.tolist(), tb.tolist(), st.tolist()):
        h = (h * 1000003 + r * 7919 + t * 31337 + s) & M64
    return h


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "kht":
        return kht_main()
    ref = RefShim(1)
    assert ref.avx2, "goldens must come from the AVX2 intrinsics path"
    step = np.float32(THETA_DEG) * (np.float32(3.1415926535897932384626433) / np.float32(180.0))
    frames = []
    for f in range(FRAMES):
        img = synth_frame(W, H, FIRST_SEED + f)
        rc, can = ref.canny(img, T_LOW, T_HIGH, 3, 0)
        assert rc == 0
        lines = ref.sht(can, THETA_DEG, THRESHOLD)
        rho = np.array([l[0] for l in lines], np.float64)
        theta = np.array([l[1] for l in lines], np.float32)
        col = np.rint(theta / step).astype(np.int64)
        assert np.array_equal((col.astype(np.float32) * step), theta), "theta is col * step in f32 (houghsht.cxx:662)"
        st = np.array([l[2] for l in lines], np.int64)
        frames.append({"seed": FIRST_SEED + f, "canny_md5": md5_rows(can), "edges": int((can != 0).sum()), "lines": len(lines),
                       "sum_strength": int(st.sum()), "line_hash": "%016x" % line_hash(rho.astype(np.int64), col, st)})
        print(f, frames[-1], flush=True)
    out = {"W": W, "H": H, "tLow": T_LOW, "tHigh": T_HIGH, "theta_deg": THETA_DEG, "threshold": THRESHOLD, "first_seed": FIRST_SEED,
           "source": "CompV (oracle/_ref, AVX2 intrinsics, 1 thread)", "frames": frames}
    with open(os.path.join(HERE, "golden_batch_fhd.json" if FHD else "golden_batch.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
