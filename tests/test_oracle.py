"""CPU tests: the oracle (oracle/compv_oracle.c) against the reference's own known-answer vectors and against the
golden fixtures generated from the compiled reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle_bindings import md5_rows, synth_frame

SMALL = ["tiny_20x20", "q1_200x258", "small_320x240", "q3_641x480", "ragged_333x77", "mean_640x480", "theta_half_640x480"]
MEDIUM = ["hd_1280x720", "dense_1282x720", "fhd_1920x1080", "fhd_seed7"]
LARGE = ["uhd_3840x2160"]


def _convlt_inputs():
    # unittests/math_convlt.cxx:100-165: 1285x720, stride 1344, k = 7
    W, H, S = 1285, 720, 1344
    i = np.arange(W)[None, :]
    j = np.arange(H)[:, None]
    d8 = np.zeros((H, S), np.uint8)
    d8[:, :W] = ((i * j) + 53).astype(np.uint8)                               # :107
    d16 = np.zeros((H, S), np.int16)
    d16[:, :W] = (((i * j) + 53) * np.where(i & 1, -1, 1)).astype(np.int16)    # :129
    k = np.array([(t + 53) * (-1 if t & 1 else 1) for t in range(7)], np.int16)  # :158
    return d8[:, :W], d16[:, :W], k


def test_convlt_reference_known_answers(oracle):
    """The two hot-path convolution instantiations reproduce the reference's own goldens
    (unittests/math_convlt.cxx:24-25, cases 5 and 6)."""
    d8, d16, k = _convlt_inputs()
    rc, out = oracle.convlt_8u(d8, k, k)
    assert rc == 0 and md5_rows(out) == "7f1116ade2a1cdb37842084c781ee05e"
    rc, out = oracle.convlt_16s(d16, k, k)
    assert rc == 0 and md5_rows(out) == "cad2f4d2fd66e171997f39804e667699"


def test_convlt_rejects_bad_geometry(oracle):
    img = np.zeros((2, 8), np.uint8)
    rc, _ = oracle.convlt_8u(img, [1, 2, 1], [-1, 0, 1])   # H < k  (compv_math_convlt.h:100)
    assert rc != 0


def test_synth_generator_matches_c(oracle):
    for (W, H, seed) in [(64, 48, 12345), (333, 77, 777), (641, 17, 1)]:
        assert (synth_frame(W, H, seed) == oracle.synth(W, H, seed)).all()


@pytest.mark.parametrize("name", SMALL + MEDIUM)
def test_golden_sobel_canny(oracle, golden, name):
    meta, arrays = golden
    m = meta[name]
    img = synth_frame(m["W"], m["H"], m["seed"])
    assert md5_rows(img) == m["input_md5"]
    sob, _ = oracle.edge_dete(img)
    assert md5_rows(sob) == m["sobel_md5"]
    rc, can = oracle.canny(img, m["tLow"], m["tHigh"], 3, m["threshold_type"])
    assert rc == 0
    assert md5_rows(can) == m["canny_md5"]
    assert int((can != 0).sum()) == m["canny_edges"]
    if name + "/sobel" in arrays:
        assert (arrays[name + "/sobel"] == sob).all()
        bits = np.unpackbits(arrays[name + "/canny_bits"], axis=1)[:, :m["W"]].astype(bool)
        assert ((can != 0) == bits).all()


@pytest.mark.parametrize("name", [n for n in SMALL + MEDIUM if n not in ("tiny_20x20", "mean_640x480")])
def test_golden_sht(oracle, golden, name):
    meta, arrays = golden
    m = meta[name]
    s = m["sht"]
    img = synth_frame(m["W"], m["H"], m["seed"])
    rc, can = oracle.canny(img, m["tLow"], m["tHigh"], 3, m["threshold_type"])
    lines = oracle.sht(can, s["theta_deg"], s["threshold"])
    assert len(lines) == s["lines"]
    assert sum(l[2] for l in lines) == s["sum_strength"]
    exp = arrays[name + "/sht_lines"]
    got = np.array([(l[0], l[1], l[2]) for l in lines[:len(exp)]], np.float64).reshape(-1, 3)
    assert (got == exp).all()   # rho, theta (f32 values) and strength, canonical order


def test_golden_uhd(oracle, golden):
    meta, arrays = golden
    m = meta["uhd_3840x2160"]
    img = synth_frame(m["W"], m["H"], m["seed"])
    rc, can = oracle.canny(img, m["tLow"], m["tHigh"])
    assert md5_rows(can) == m["canny_md5"] and int((can != 0).sum()) == m["canny_edges"] == 369465
    lines = oracle.sht(can, 1.0, 100)
    assert len(lines) == m["sht"]["lines"] == 57350
    assert sum(l[2] for l in lines) == m["sht"]["sum_strength"]


@pytest.mark.parametrize("name", ["hd_1280x720", "fhd_1920x1080", "uhd_3840x2160"])
def test_golden_5x5_canny_and_detectors_full_size(oracle, golden, name):
    """The restatement against the compiled reference's 5x5 Canny MD5s / edge counts and its Scharr / Prewitt detector MD5s at the bench's sizes
    (tests/golden/make_golden.py: FULL_SIZE_EXTRAS)."""
    meta, _ = golden
    m = meta[name]
    img = synth_frame(m["W"], m["H"], m["seed"])
    for key in ("canny5", "canny5_x12"):
        g = m[key]
        rc, can = oracle.canny(img, g["tLow"], g["tHigh"], 5)
        assert rc == 0 and md5_rows(can) == g["md5"] and int((can != 0).sum()) == g["edges"], (name, key)
    assert md5_rows(oracle.edge_dete(img, 2)[0]) == m["scharr_md5"]
    assert md5_rows(oracle.edge_dete(img, 3)[0]) == m["prewitt_md5"]


def test_thresholds(oracle):
    # compv_core_feature_canny_dete.cxx:251-266
    assert oracle.canny_thresholds(59.0, 119.0) == (0, 59, 119)
    assert oracle.canny_thresholds(0.8, 1.6) == (0, 1, 3)            # clip to 1, then tHigh = max(tLow+2, .)
    assert oracle.canny_thresholds(0.68, 1.36, 1, 124 * 640 * 480, 640, 480) == (0, 84, 168)
    assert oracle.canny_thresholds(5.0, 5.0)[0] != 0                 # tLow >= tHigh -> E_INVALID_STATE (:126)


def test_coverage_quirk_q3(oracle):
    # W = 641: columns 625..639 are neither NMS'ed nor seed-scanned; W in {1280,1920,3840}: full coverage
    assert oracle.canny_coverage(641) == (625, 640)
    for W in (1280, 1920, 3840):
        se, cs = oracle.canny_coverage(W)
        assert cs <= se and cs < W - 1


def test_sht_vote_total_and_dims(oracle):
    img = synth_frame(320, 240)
    rc, can = oracle.canny(img, 59.0, 119.0)
    R, T, th = oracle.sht_dims(320, 240, 1.0)
    assert (R, T) == (2 * (320 + 240) + 1, 180)
    acc = oracle.sht_acc(can, 1.0)
    assert acc.sum() == int((can != 0).sum()) * T      # every edge votes once per theta
    assert oracle.sht_dims(1920, 1080, 1.0)[:2] == (6001, 180)


def _sht_order_golden():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_sht_order.json")))


def _pack_lines(rho, theta, strength):
    import numpy as np
    a = np.zeros((len(rho), 3), np.uint32)
    a[:, 0] = np.asarray(rho, np.float32).view(np.uint32)
    a[:, 1] = np.asarray(theta, np.float32).view(np.uint32)
    a[:, 2] = np.asarray(strength, np.uint32)
    return a


def test_sht_tie_order_runtime_is_the_fixtures_runtime():
    """The order of equal-strength lines is "what this C++ runtime's std::sort does", in the reference and in compvhip_houghsht_u8 /
    compvhip_houghkht_u8 that reproduce it (api.cpp referenceLineOrder, kht_host.cpp khtPeaks).  The fixtures record the libstdc++ they
    were generated with; on another runtime they would have to be regenerated (with the reference rebuilt against it): fail loudly."""
    from oracle_bindings import libstdcxx_version
    want = _sht_order_golden()["_runtime"]["libstdcxx"]
    assert libstdcxx_version() == want, ("tie-order fixtures were generated with %s, this process runs %s: regenerate "
                                         "tests/golden/golden_sht_order.json with the reference built against this runtime" % (want, libstdcxx_version()))


@pytest.mark.parametrize("name", ["vga_all", "vga_top100", "hd_halfdeg", "ragged_top40", "calib_like"])
def test_sht_reference_line_order_fixture(oracle, name):
    """The compiled reference's line list, element by element (tests/golden/make_golden_sht_order.py): thousands of equal-strength
    pairs, with and without a maxLines cut."""
    import hashlib
    m = _sht_order_golden()[name]
    rc, can = oracle.canny(synth_frame(m["W"], m["H"], m["seed"]), m["tLow"], m["tHigh"])
    assert md5_rows(can) == m["canny_md5"]
    lines = oracle.sht(can, m["theta_deg"], m["threshold"], m["max_lines"], reference_order=True)
    a = _pack_lines([l[0] for l in lines], [l[1] for l in lines], [l[2] for l in lines])
    assert len(a) == m["lines"] and m["equal_strength_pairs"] > 20
    assert a[:64].tolist() == m["head"] and a[-64:].tolist() == m["tail"]
    assert hashlib.md5(a.tobytes()).hexdigest() == m["md5"]
