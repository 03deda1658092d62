"""Oracle (oracle/compv_oracle.c) vs the compiled reference for the samples' caller-side pre-processing
(SURVEY 8f row 1): CompVImage::convertGrayscale and CompVImage::thresholdOtsu (samples/hough_lines/main.cxx:102-105)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from oracle_bindings import synth_frame

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_preproc import otsu_input, packed_input  # noqa: E402  (input generators only; no reference needed)

GOLDEN_PREPROC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_preproc.json")))

FMT_NAMES = ["RGBA32", "ARGB32", "BGRA32", "RGB24", "BGR24", "RGB565LE", "RGB565BE", "BGR565LE", "BGR565BE", "YUYV422", "UYVY422", "Y"]


def packed_frame(oracle, fmt, W, H, S, seed):
    bpp = oracle.fmt_bytes(fmt)
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(H, S * bpp), dtype=np.uint8), bpp


@pytest.mark.parametrize("fmt", range(len(FMT_NAMES)), ids=FMT_NAMES)
@pytest.mark.parametrize("W,H,S", [(64, 8, 64), (33, 5, 48), (130, 17, 160), (641, 31, 704)])
def test_grayscale_oracle_matches_reference(oracle, refshim, fmt, W, H, S):
    if FMT_NAMES[fmt] in ("YUYV422", "UYVY422") and (W & 1):
        W += 1  # packed 4:2:2 needs an even width
    data, bpp = packed_frame(oracle, fmt, W, H, S, 1000 * fmt + W)
    exp = refshim.grayscale(data, fmt, W, bpp)
    got = oracle.grayscale(data, fmt, W)
    assert (got == exp).all(), int((got != exp).sum())


def test_grayscale_known_values(oracle):
    # Y = ((33 R + 65 G + 13 B) >> 7) + 16 (compv_image_conv_rgbfamily.cxx:108): black -> 16, white -> 237, pure red -> 81
    px = np.array([[0, 0, 0, 255, 255, 255, 255, 0, 0]], np.uint8)
    assert oracle.grayscale(px, 3, 3).tolist() == [[16, 237, 81]]
    assert oracle.grayscale(px, 4, 3).tolist() == [[16, 237, 41]]     # the same bytes read as BGR: pure blue -> (13*255 >> 7) + 16


@pytest.mark.parametrize("W,H,seed", [(20, 20, 1), (333, 77, 2), (640, 480, 3), (1282, 720, 4), (1920, 1080, 5)])
def test_otsu_oracle_matches_reference(oracle, refshim, W, H, seed):
    img = synth_frame(W, H, 12345 + seed)
    assert oracle.otsu(img) == int(refshim.otsu(img))
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    assert oracle.otsu(noise) == int(refshim.otsu(noise))
    bimodal = np.where(rng.random((H, W)) < 0.3, rng.integers(150, 220, (H, W)), rng.integers(10, 90, (H, W))).astype(np.uint8)
    assert oracle.otsu(bimodal) == int(refshim.otsu(bimodal))
    flat = np.full((H, W), 77, np.uint8)                      # single grey level: q2 == 0 at once -> threshold 0
    assert oracle.otsu(flat) == int(refshim.otsu(flat)) == 0


def test_otsu_hist_and_canny_thresholds(oracle):
    img = synth_frame(320, 240)
    h = oracle.hist256(img)
    assert int(h.sum()) == 320 * 240 and (h == np.bincount(img.ravel(), minlength=256)).all()
    # samples/hough_lines/main.cxx:104-105: LOW = (float)(t*0.5), HIGH = (float)t, then the COMPARE_TO_GRADIENT clamp
    assert oracle.otsu_canny_thresholds(119) == (59, 119)
    assert oracle.otsu_canny_thresholds(0) == (1, 3)
    assert oracle.otsu_canny_thresholds(1) == (1, 3)
    assert oracle.otsu_canny_thresholds(255) == (127, 255)


# ---- committed fixtures generated from the compiled reference (tests/golden/make_golden_preproc.py): these run on any box ----
@pytest.mark.parametrize("case", GOLDEN_PREPROC["grayscale"], ids=lambda c: "%s_%dx%d" % (c["name"], c["W"], c["H"]))
def test_grayscale_oracle_matches_golden(oracle, case):
    data = packed_input(case["fmt"], case["W"], case["H"], case["S"], case["seed"])
    g = oracle.grayscale(data, case["fmt"], case["W"])
    assert hashlib.md5(np.ascontiguousarray(g).tobytes()).hexdigest() == case["md5"]


@pytest.mark.parametrize("case", [c for c in GOLDEN_PREPROC["otsu"] if c["W"] <= 1920], ids=lambda c: "%s_%dx%d" % (c["kind"], c["W"], c["H"]))
def test_otsu_oracle_matches_golden(oracle, case):
    assert oracle.otsu(otsu_input(case["kind"], case["W"], case["H"], case["seed"])) == case["threshold"]
