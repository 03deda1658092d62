"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI (compv_amd.capi ->
libcompv_hip.so), against the oracle on the same seeded inputs, against the committed golden fixtures generated from
the compiled reference, and -- at BASELINE.json's full sizes -- through size-independent properties.

Bars: bit-exact for the uint8 Sobel / Canny maps, the int32 Hough accumulator and the line set (rho, theta as f32
bit patterns, strength); there is no floating-point tolerance on this path (the only f32 operations are one
division + one multiply in the Sobel normalisation and col*thetaStep, all correctly rounded single operations).
"""
import os

import numpy as np
import pytest

from oracle_bindings import md5_rows, synth_frame

pytestmark = pytest.mark.gpu


def _lines_tuple(lines):
    return [(float(l["rho"]), float(l["theta"]), int(l["strength"]), int(l["row"]), int(l["col"])) for l in lines]


def _orc_tuple(lines):
    return [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2]), int(l[3]), int(l[4])) for l in lines]


# ---------------------------------------------------------------------------------------------------------------
# host entry points vs oracle
# ---------------------------------------------------------------------------------------------------------------
SIZES = [(3, 3), (9, 9), (16, 16), (17, 9), (20, 20), (33, 200), (64, 64), (97, 33), (200, 258), (257, 65), (320, 240),
         (512, 64), (513, 65), (641, 480), (1023, 129), (1282, 720)]


@pytest.mark.parametrize("W,H", SIZES)
def test_edge_dete_matches_oracle(hip_ctx, oracle, W, H):
    from compv_amd import capi
    img = synth_frame(W, H, 100 + W)
    for op, oop in [(capi.OP_SOBEL, 0), (capi.OP_SCHARR, 2), (capi.OP_PREWITT, 3)]:
        exp, _ = oracle.edge_dete(img, oop)
        got = hip_ctx.edge_dete(img, op)
        assert (got == exp).all(), (W, H, op, int((got != exp).sum()))


def _extreme_images(W, H):
    yy, xx = np.mgrid[0:H, 0:W]
    rng = np.random.RandomState(W * 131 + H)
    return {"vstripes": ((xx & 1) * 255).astype(np.uint8), "hstripes": ((yy & 1) * 255).astype(np.uint8),
            "checker1": (((xx + yy) & 1) * 255).astype(np.uint8), "checker2": ((((xx >> 1) + (yy >> 1)) & 1) * 255).astype(np.uint8),
            "binary_noise": (rng.rand(H, W) < 0.5).astype(np.uint8) * 255, "noise": rng.randint(0, 256, (H, W)).astype(np.uint8),
            "steps": ((xx >= W // 2) * 255).astype(np.uint8), "ramp": ((xx * 255) // max(W - 1, 1)).astype(np.uint8)}


@pytest.mark.parametrize("W,H", [(130, 70), (515, 67), (256, 64), (300, 129)])
def test_packed_gradient_at_extreme_contrast(hip_ctx, oracle, W, H):
    """The packed (two pixels per register, biased 16-bit halves) gradient of the Canny tile kernel and of the detector kernels at the ends of its
    value range: 0 / 255 stripes, checkerboards and noise drive |gx|, |gy| to their maxima (1020 for Sobel, 4080 for Scharr) with both signs in adjacent
    pixels -- a carry or borrow between the halves of a register would show here first."""
    from compv_amd import capi
    for name, img in _extreme_images(W, H).items():
        for op, oop in [(capi.OP_SOBEL, 0), (capi.OP_SCHARR, 2), (capi.OP_PREWITT, 3)]:
            exp, _ = oracle.edge_dete(img, oop)
            got = hip_ctx.edge_dete(img, op)
            assert (got == exp).all(), (name, op, int((got != exp).sum()))
        for tl, th in ((59.0, 119.0), (1.0, 3.0), (900.0, 2000.0)):
            rc, exp = oracle.canny(img, tl, th)
            assert rc == 0
            got = hip_ctx.canny(img, tl, th)
            assert (got == exp).all(), (name, tl, th, int((got != exp).sum()))


def test_edge_dete_constant_image_is_zero(hip_ctx, oracle):
    img = np.full((48, 80), 77, np.uint8)     # gmax == 0 -> scale = inf -> all zeros (edge_dete.cxx:199)
    exp, gmax = oracle.edge_dete(img)
    assert gmax == 0 and not exp.any()
    assert not hip_ctx.edge_dete(img).any()


@pytest.mark.parametrize("W,H", SIZES)
def test_canny_matches_oracle(hip_ctx, oracle, W, H):
    rng = np.random.default_rng(W * 31 + H)
    y, x = np.mgrid[0:H, 0:W]
    imgs = [synth_frame(W, H, 7 + H),
            rng.integers(0, 256, (H, W), dtype=np.uint8),
            ((np.sin(x / 7.0) + np.cos(y / 5.0)) * 40 + 128 + rng.integers(0, 6, (H, W))).astype(np.uint8)]
    for k, img in enumerate(imgs):
        for (tl, th) in [(59.0, 119.0), (0.8, 1.6), (20.0, 300.0)]:
            rc, exp = oracle.canny(img, tl, th)
            assert rc == 0
            got = hip_ctx.canny(img, tl, th)
            assert (got == exp).all(), (W, H, k, tl, th, int((got != exp).sum()))


@pytest.mark.parametrize("W,H", [(9, 9), (20, 20), (64, 64), (129, 130), (513, 65), (641, 333), (1282, 720)])
def test_canny_5x5_sobel_matches_oracle(hip_ctx, oracle, W, H):
    """Kernel size 5 (COMPV_CANNY_SET_INT_KERNEL_SIZE): 2-px zero border, |g| up to 24480."""
    rng = np.random.default_rng(W + H)
    for img in (synth_frame(W, H, 5), rng.integers(0, 256, (H, W), dtype=np.uint8)):
        for (tl, th) in [(400.0, 900.0), (0.8, 1.6), (2000.0, 6000.0)]:
            rc, exp = oracle.canny(img, tl, th, 5)
            assert rc == 0
            got = hip_ctx.canny(img, tl, th, ksize=5)
            assert (got == exp).all(), (W, H, tl, th, int((got != exp).sum()))


def test_canny_long_weak_chains(hip_ctx, oracle):
    """Hysteresis stress: weak spirals / long chains with one strong seed cross many tiles and bands."""
    W, H = 1100, 700
    img = np.full((H, W), 100, np.uint8)
    # a serpentine of low-contrast lines (weak everywhere) ...
    for k, yy in enumerate(range(20, H - 20, 12)):
        img[yy:yy + 3, 15:W - 15] = 112
        xs = W - 30 if (k % 2 == 0) else 15
        img[yy:yy + 15, xs:xs + 3] = 112
    # ... and one strong blob touching its start
    img[18:26, 10:20] = 255
    rc, exp = oracle.canny(img, 10.0, 200.0)
    assert rc == 0 and exp.any()
    got = hip_ctx.canny(img, 10.0, 200.0)
    assert (got == exp).all(), int((got != exp).sum())


def test_canny_more_hysteresis_rounds_than_flag_slots(oracle, monkeypatch):
    """A weak chain that zigzags across band borders needs one resolve round per crossing; when the rounds outnumber the flag slots
    (4096; lowered to 8 here) the slots are reused instead of giving up -- the reference has no such limit."""
    from compv_amd import capi
    W, H = 1100, 700
    img = np.full((H, W), 100, np.uint8)
    for k, yy in enumerate(range(20, H - 20, 12)):
        img[yy:yy + 3, 15:W - 15] = 112
        xs = W - 30 if (k % 2 == 0) else 15
        img[yy:yy + 15, xs:xs + 3] = 112
    img[18:26, 10:20] = 255
    rc, exp = oracle.canny(img, 10.0, 200.0)
    assert rc == 0 and exp.any()
    monkeypatch.setenv("COMPVHIP_RESOLVE_WRAP", "8")
    ctx = capi.Context(0)
    try:
        got = ctx.canny(img, 10.0, 200.0)
        assert (got == exp).all(), int((got != exp).sum())
        again = ctx.canny(img, 10.0, 200.0)
        assert (again == exp).all()
    finally:
        ctx.close()


def test_canny_in_place_and_strided(hip_ctx, oracle):
    W, H, S = 300, 200, 384
    buf = np.zeros((H, S), np.uint8)
    buf[:, :W] = synth_frame(W, H, 5)
    view = buf[:, :W]
    rc, exp = oracle.canny(np.ascontiguousarray(view), 59.0, 119.0)
    got = hip_ctx.canny(view, 59.0, 119.0, out=view)        # samples/edges_canny/main.cxx:72 does process(mat,&mat)
    assert (got == exp).all()
    assert not buf[:, W:].any()                              # stride padding untouched


def test_canny_mean_threshold_mode(hip_ctx, oracle):
    from compv_amd import capi
    for (W, H) in [(640, 480), (333, 77)]:
        img = synth_frame(W, H)
        rc, exp = oracle.canny(img, 0.68, 1.36, 3, 1)
        got = hip_ctx.canny(img, 0.68, 1.36, 3, capi.THRESHOLD_PERCENT_OF_MEAN)
        assert (got == exp).all()


def test_canny_error_behaviour(hip_ctx):
    from compv_amd import capi
    img = synth_frame(64, 64)
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.canny(img, 100.0, 50.0)                      # tLow >= tHigh -> E_INVALID_STATE (canny_dete.cxx:126)
    assert e.value.code == capi.E_INVALID_STATE
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.canny(img, 10.0, 50.0, ksize=7)              # canny_dete.cxx:101
    assert e.value.code == capi.E_INVALID_PARAMETER
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.canny(np.zeros((2, 2), np.uint8), 10.0, 50.0)  # smaller than the kernel (compv_math_convlt.h:100)
    assert e.value.code == capi.E_INVALID_PARAMETER


@pytest.mark.parametrize("W,H,tl,th,deg,thr", [(320, 240, 59., 119., 1.0, 40), (333, 77, 0.8, 1.6, 1.0, 20), (641, 480, 59., 119., 1.0, 100),
                                               (640, 480, 59., 119., 0.5, 50), (640, 480, 59., 119., 2.0, 30), (320, 240, 59., 119., 1.3, 10),
                                               (1282, 720, 0.8, 1.6, 1.0, 100)])
def test_houghsht_matches_oracle(hip_ctx, oracle, W, H, tl, th, deg, thr):
    img = synth_frame(W, H)
    rc, edges = oracle.canny(img, tl, th)
    acc_exp = oracle.sht_acc(edges, deg)
    exp = oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, deg, thr)
    lines, acc = hip_ctx.houghsht(edges, deg, thr, want_acc=True)
    assert acc.shape == acc_exp.shape and (acc == acc_exp).all(), int((acc != acc_exp).sum())   # vote histogram bit-exact
    assert _lines_tuple(lines) == _orc_tuple(exp)
    # maxLines keeps the strongest
    if len(exp) > 5:
        top = hip_ctx.houghsht(edges, deg, thr, max_lines=5)
        assert _lines_tuple(top) == _orc_tuple(exp[:5])


@pytest.mark.parametrize("W,H", [(8000, 191), (8000, 192), (40, 8300), (12000, 48)])
def test_houghsht_wide_rho_range(hip_ctx, oracle, W, H):
    """Wide rho ranges (tens of tiles per frame in the tiled vote kernel; W+H = 8191 / 8192 straddle the two variants of the
    first-generation kernel)."""
    rng = np.random.default_rng(W + H)
    edges = np.where(rng.random((H, W)) < 0.02, 0xff, 0).astype(np.uint8)
    edges[H // 2, :] = 0xff
    edges[:, W // 3] = 0xff
    acc_exp = oracle.sht_acc(edges, 1.0)
    exp = oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, 1.0, 30)
    lines, acc = hip_ctx.houghsht(edges, 1.0, 30, want_acc=True)
    assert (acc == acc_exp).all(), int((acc != acc_exp).sum())
    assert _lines_tuple(lines) == _orc_tuple(exp)


def test_canny_documented_deviations(hip_ctx, oracle):
    """Two DEFINED deviations from the reference, pinned as such:
    * thresholds above 32767: the reference's SIMD leaves compare them as signed int16 and its scalar remainder as unsigned -- an artefact
      regime (g <= 24480 can never exceed them) that this implementation rejects with E_INVALID_PARAMETER instead of reproducing;
    * an Otsu level of 0 (device OTSU threshold mode): the reference's set() rejects the threshold pair and the sample skips the frame
      (canny_dete.cxx:86-99, samples/hough_lines/main.cxx:103-105); a batch cannot skip a frame: thresholds (1, 3) are used."""
    import torch
    from compv_amd import capi
    img = synth_frame(64, 64)
    for (tl, th) in [(10.0, 40000.0), (33000.0, 40000.0)]:
        with pytest.raises(capi.CompvHipError) as e:
            hip_ctx.canny(img, tl, th)
        assert e.value.code == capi.E_INVALID_PARAMETER
    hip_ctx.canny(img, 10.0, 32767.0)      # the largest accepted pair still works (and selects nothing strong: g <= 2040)
    W, H, F = 320, 96, 2
    frames = np.zeros((F, H, W), np.uint8)
    frames[0, 20:60, 50:200] = 200          # two grey levels {0, 200}: every Otsu level 0..199 has the same variance -> level 0
    frames[1] = synth_frame(W, H, 3)
    assert oracle.otsu(frames[0]) == 0
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(frames).to(dev)
    d_e = torch.empty_like(d_in)
    d_t = torch.zeros(F, dtype=torch.int32, device=dev)
    plan = capi.Plan(hip_ctx, W, H, W, F, 1.0)
    try:
        plan.otsu(d_in.data_ptr(), d_t.data_ptr())
        plan.canny(d_in.data_ptr(), 0.5, 1.0, d_e.data_ptr(), threshold_type=capi.THRESHOLD_OTSU)
        torch.cuda.synchronize()
        assert d_t.cpu().tolist() == [0, int(oracle.otsu(frames[1]))]      # the level is reported, the frame is not skipped
        rc, exp0 = oracle.canny(frames[0], 1.0, 3.0)
        assert rc == 0 and exp0.any()
        assert (d_e[0].cpu().numpy() == exp0).all()
        lo, hi = oracle.otsu_canny_thresholds(int(oracle.otsu(frames[1])))
        rc, exp1 = oracle.canny(frames[1], float(lo), float(hi))
        assert (d_e[1].cpu().numpy() == exp1).all()
    finally:
        plan.close()


@pytest.mark.parametrize("W,H", [(20480, 16), (32767, 64), (64, 32767), (8192, 8192)])
def test_houghsht_beyond_the_first_generation_limit(hip_ctx, oracle, W, H):
    """The reference accepts every size up to 32767 x 32767 (core/features/hough/compv_core_feature_houghsht.cxx:318-348); the
    first-generation vote kernel stopped at W + H = 20479 (one whole rho column per workgroup in LDS).  The tiled kernel has no such
    limit: sparse maps at the extreme sizes, accumulator and line set against the oracle."""
    rng = np.random.default_rng(W * 3 + H)
    edges = np.zeros((H, W), np.uint8)
    n = 60000
    edges[rng.integers(0, H, n), rng.integers(0, W, n)] = 0xff
    edges[H // 2, :] = 0xff                                   # one full row and one full column: long collinear runs
    edges[:, W // 3] = 0xff
    edges[np.arange(min(W, H)), np.arange(min(W, H))] = 0xff  # and a diagonal
    acc_exp = oracle.sht_acc(edges, 1.0)
    thr = 40
    exp = oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, 1.0, thr)
    lines, acc = hip_ctx.houghsht(edges, 1.0, thr, want_acc=True, cap=max(1 << 16, len(exp)))
    assert acc.shape == acc_exp.shape and (acc == acc_exp).all(), int((acc != acc_exp).sum())
    assert _lines_tuple(lines) == _orc_tuple(exp)


def test_houghsht_empty_and_full_maps(hip_ctx, oracle):
    W, H = 160, 120
    none = np.zeros((H, W), np.uint8)
    assert len(hip_ctx.houghsht(none, 1.0, 1)) == 0
    full = np.full((H, W), 0xff, np.uint8)                    # every pixel votes; any non-zero byte is an edge
    full[::3, ::5] = 1
    acc_exp = oracle.sht_acc(full, 1.0)
    lines, acc = hip_ctx.houghsht(full, 1.0, 50, want_acc=True)
    assert (acc == acc_exp).all()
    assert _lines_tuple(lines) == _orc_tuple(oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, 1.0, 50))


def test_houghsht_counts_above_255_in_several_tiles(hip_ctx, oracle):
    """The partial windows of the voting tiles travel as byte planes: low bytes always, high bytes only for the (tile, theta) columns
    that hold a count of 256 or more.  A nearly full 1600x1300 map spans several tiles and puts counts of up to ~1600 (high bytes up to
    6) into many columns of every tile, next to columns that stay below 256; a sparse call afterwards must not see stale high bytes."""
    W, H = 1600, 1300
    rng = np.random.default_rng(11)
    full = np.full((H, W), 0xff, np.uint8)
    full[rng.random((H, W)) < 0.02] = 0
    full[:, 700:760] = 0                                      # a gap: some columns of some tiles stay small
    acc_exp = oracle.sht_acc(full, 1.0)
    assert acc_exp.max() > 1500
    lines, acc = hip_ctx.houghsht(full, 1.0, 1200, want_acc=True)
    assert (acc == acc_exp).all(), int((acc != acc_exp).sum())
    assert _lines_tuple(lines) == _orc_tuple(oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, 1.0, 1200))
    sparse = (rng.random((H, W)) < 0.01).astype(np.uint8)
    acc_exp = oracle.sht_acc(sparse, 1.0)
    lines, acc = hip_ctx.houghsht(sparse, 1.0, 40, want_acc=True)
    assert (acc == acc_exp).all(), int((acc != acc_exp).sum())
    assert _lines_tuple(lines) == _orc_tuple(oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, 1.0, 40))


def test_houghsht_error_behaviour(hip_ctx):
    from compv_amd import capi
    e = np.zeros((32, 32), np.uint8)
    with pytest.raises(capi.CompvHipError) as ex:
        hip_ctx.houghsht(e, 1.0, 10, rho=0.5)                  # SHT requires rho == 1 (houghsht.cxx:306-316)
    assert ex.value.code == capi.E_INVALID_PARAMETER
    with pytest.raises(capi.CompvHipError):
        hip_ctx.houghsht(e, 1.0, 0)                            # threshold must be > 0 (:82)


# ---------------------------------------------------------------------------------------------------------------
# golden fixtures from the compiled reference
# ---------------------------------------------------------------------------------------------------------------
GOLD = ["tiny_20x20", "q1_200x258", "small_320x240", "q3_641x480", "ragged_333x77", "dense_1282x720", "hd_1280x720",
        "fhd_1920x1080", "fhd_seed7", "theta_half_640x480", "mean_640x480", "uhd_3840x2160"]


@pytest.mark.parametrize("name", GOLD)
def test_golden(hip_ctx, golden, name):
    meta, arrays = golden
    m = meta[name]
    W, H = m["W"], m["H"]
    img = synth_frame(W, H, m["seed"])
    assert md5_rows(img) == m["input_md5"]
    sob = hip_ctx.edge_dete(img)
    assert md5_rows(sob) == m["sobel_md5"]
    can = hip_ctx.canny(img, m["tLow"], m["tHigh"], 3, m["threshold_type"])
    assert md5_rows(can) == m["canny_md5"]
    assert int((can != 0).sum()) == m["canny_edges"]
    assert set(np.unique(can)) <= {0, 255}
    if "sht" in m:
        s = m["sht"]
        lines = hip_ctx.houghsht(can, s["theta_deg"], s["threshold"], cap=max(1, s["lines"]))
        assert len(lines) == s["lines"]
        assert int(lines["strength"].astype(np.int64).sum()) == s["sum_strength"]
        exp = arrays[name + "/sht_lines"]
        # the fixture holds the canonical order (strength desc, rho desc, theta asc); the host entry point returns the reference's own
        # order (test_houghsht_host_path_returns_the_references_line_order checks that one): canonicalise before comparing
        o = np.lexsort((lines["theta"], -lines["rho"], -lines["strength"].astype(np.int64)))[:len(exp)]
        got = np.stack([lines["rho"][o].astype(np.float64), lines["theta"][o].astype(np.float64), lines["strength"][o].astype(np.float64)], axis=1)
        assert (got == exp).all()


@pytest.mark.parametrize("name", ["hd_1280x720", "fhd_1920x1080", "uhd_3840x2160"])
def test_golden_5x5_canny_and_detectors_full_size(hip_ctx, golden, name):
    """the packed 5x5 Canny kernel (canny_swar_tile_kernel<.., KS = 5>) and the Scharr / Prewitt detectors at 720p / 1080p / 4K against
    MD5s + edge counts recorded from the compiled reference (tests/golden/make_golden.py: FULL_SIZE_EXTRAS), at the benchmark's thresholds (36 % of the
    pixels are edges under a 5x5 gradient: the sparse stage is dense) and at thresholds x 12 (the benchmark's edge density)."""
    from compv_amd import capi
    meta, _ = golden
    m = meta[name]
    img = synth_frame(m["W"], m["H"], m["seed"])
    for key in ("canny5", "canny5_x12"):
        g = m[key]
        can = hip_ctx.canny(img, g["tLow"], g["tHigh"], ksize=5)
        assert md5_rows(can) == g["md5"], (name, key)
        assert int((can != 0).sum()) == g["edges"]
    assert md5_rows(hip_ctx.edge_dete(img, capi.OP_SCHARR)) == m["scharr_md5"]
    assert md5_rows(hip_ctx.edge_dete(img, capi.OP_PREWITT)) == m["prewitt_md5"]


def test_plan_canny_5x5_batch_full_size(hip_ctx, golden):
    """The plan entry the bench's kernels_extra times (compvhip_plan_canny, ksize 5) on a batch of 4K frames: frame 0 = the reference fixture, every frame's
    edge count differs from frame 0's (distinct seeds) and repeats exactly on a second run."""
    import torch
    from compv_amd import capi
    meta, _ = golden
    m = meta["uhd_3840x2160"]
    W, H, F = m["W"], m["H"], 3
    dev = torch.device("cuda:0")
    frames = np.stack([synth_frame(W, H, m["seed"] + f) for f in range(F)])
    d_in = torch.from_numpy(frames).to(dev)
    d_e = torch.empty_like(d_in)
    plan = capi.Plan(hip_ctx, W, H, W, F, 1.0)
    try:
        for key in ("canny5_x12", "canny5"):
            g = m[key]
            plan.canny(d_in.data_ptr(), g["tLow"], g["tHigh"], d_e.data_ptr(), ksize=5)
            torch.cuda.synchronize()
            e = d_e.cpu().numpy()
            assert md5_rows(e[0]) == g["md5"] and int((e[0] != 0).sum()) == g["edges"], key
            plan.canny(d_in.data_ptr(), g["tLow"], g["tHigh"], d_e.data_ptr(), ksize=5)
            torch.cuda.synchronize()
            assert (d_e.cpu().numpy() == e).all()
    finally:
        plan.close()


def test_plan_houghkht_batch_uhd_against_the_reference_fixture(hip_ctx):
    """What bench.py's kht figure times: compvhip_plan_houghkht on 4K edge maps of the benchmark's first frames, every frame against
    tests/golden/golden_batch_kht.json (real CompV: line count, strength sum, GS to the last digit, order-dependent hash of the list)."""
    import json
    import torch
    from compv_amd import capi
    gk = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_batch_kht.json")))
    W, H, F = gk["W"], gk["H"], 9                       # two groups of 8 -> the group pipeline is exercised
    dev = torch.device("cuda:0")
    frames = np.stack([synth_frame(W, H, gk["first_seed"] + f) for f in range(F)])
    d_in = torch.from_numpy(frames).to(dev)
    d_e = torch.empty_like(d_in)
    plan = capi.Plan(hip_ctx, W, H, W, F, 1.0)
    try:
        plan.canny(d_in.data_ptr(), gk["tLow"], gk["tHigh"], d_e.data_ptr())
        torch.cuda.synchronize()
        e = d_e.cpu().numpy()
        lines, gs = plan.houghkht(d_e.data_ptr(), gk["rho"], gk["theta_deg"], gk["threshold"])
        for f in range(F):
            g = gk["frames"][f]
            assert md5_rows(e[f]) == g["canny_md5"]
            l = lines[f]
            h = 0
            for r, t, sv in zip(l["rho"].astype(np.float32).view(np.uint32).tolist(), l["theta"].astype(np.float32).view(np.uint32).tolist(), l["strength"].astype(np.int64).tolist()):
                h = (h * 1000003 + r * 7919 + t * 31337 + sv) & ((1 << 64) - 1)
            assert (len(l), int(l["strength"].astype(np.int64).sum()), repr(gs[f]), "%016x" % h) == (g["lines"], g["sum_strength"], g["gs"], g["list_hash"]), f
    finally:
        plan.close()


# ---------------------------------------------------------------------------------------------------------------
# device-resident batched plan (what bench.py times) at BASELINE sizes, via torch device memory
# ---------------------------------------------------------------------------------------------------------------
def _plan_run(hip_ctx, frames_np, tl, th, thr, line_cap):
    import torch
    from compv_amd import capi
    n, H, W = frames_np.shape
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(frames_np).to(dev)
    d_edges = torch.empty_like(d_in)
    d_lines = torch.zeros((n, line_cap, 5), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
    plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
    st = torch.cuda.current_stream().cuda_stream
    plan.pipeline(d_in.data_ptr(), tl, th, thr, 0, d_edges.data_ptr(), d_lines.data_ptr(), line_cap, d_counts.data_ptr(), st)
    torch.cuda.synchronize()
    accs = []
    _, R, T, _ = plan.acc(0)
    for f in range(n):
        a = torch.zeros((R, T), dtype=torch.int32, device=dev)
        plan.acc_export(f, a.data_ptr(), T, st)     # reference layout [R][T]
        torch.cuda.synchronize()
        accs.append(a.cpu().numpy())
    out = (d_edges.cpu().numpy(), d_lines.cpu().numpy().view(np.uint8).reshape(n, line_cap, 20), d_counts.cpu().numpy(), accs)
    plan.close()
    return out


def test_plan_pipeline_batch_matches_oracle(hip_ctx, oracle):
    from compv_amd import capi
    W, H, n = 640, 480, 5
    frames = np.stack([synth_frame(W, H, 12345 + f) for f in range(n)])
    edges, lines_raw, counts, accs = _plan_run(hip_ctx, frames, 59.0, 119.0, 60, 4096)
    for f in range(n):
        rc, e = oracle.canny(frames[f], 59.0, 119.0)
        assert (edges[f] == e).all(), f
        acc = oracle.sht_acc(e, 1.0)
        assert (accs[f] == acc).all(), f
        exp = oracle.sht_lines_from_acc(acc, W, H, 1.0, 60)
        assert counts[f] == len(exp)
        got = np.frombuffer(lines_raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:len(exp)]
        assert _lines_tuple(got) == _orc_tuple(exp)


def test_plan_pipeline_nine_frames_and_fine_theta(hip_ctx, oracle):
    """The kernels placed XCD-aware (voting, count, lines) deal frames to the 8 XCDs in groups of eight: 9 frames = one full group and a
    group with a single frame.  And a theta step of 0.25 degrees (T = 720: 90 groups of 8 columns) makes the line kernels walk the flag
    planes in three chunks of 32 column groups."""
    from compv_amd import capi
    W, H, n = 200, 150, 9
    frames = np.stack([synth_frame(W, H, 777 + f) for f in range(n)])
    edges, lines_raw, counts, accs = _plan_run(hip_ctx, frames, 59.0, 119.0, 25, 4096)
    for f in range(n):
        rc, e = oracle.canny(frames[f], 59.0, 119.0)
        assert (edges[f] == e).all(), f
        acc = oracle.sht_acc(e, 1.0)
        assert (accs[f] == acc).all(), f
        exp = oracle.sht_lines_from_acc(acc, W, H, 1.0, 25)
        assert counts[f] == len(exp)
        got = np.frombuffer(lines_raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:len(exp)]
        assert _lines_tuple(got) == _orc_tuple(exp)
    rc, e = oracle.canny(frames[0], 59.0, 119.0)
    acc_exp = oracle.sht_acc(e, 0.25)
    assert acc_exp.shape[1] == 720
    lines, acc = hip_ctx.houghsht(e, 0.25, 12, want_acc=True)
    assert (acc == acc_exp).all()
    assert _lines_tuple(lines) == _orc_tuple(oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, 0.25, 12))
    assert len(lines) > 100


def test_plan_houghsht_foreign_edge_maps_with_empty_and_full_frames(hip_ctx, oracle):
    """compvhip_plan_houghsht on caller-provided edge maps (any non-zero byte is an edge, houghsht.cxx:159-165): a batch that mixes
    an EMPTY frame (no tile has an edge), a frame with every pixel set (the densest possible tile lists) and sparse frames -- the
    frames of a batch must not influence each other."""
    import torch
    from compv_amd import capi
    W, H = 704, 200
    rng = np.random.default_rng(7)
    frames = np.zeros((4, H, W), np.uint8)
    frames[1][:] = 0xff
    frames[2] = (rng.random((H, W)) < 0.03).astype(np.uint8) * 7          # non-zero, not 0xff
    frames[3][H // 2, :] = 1
    dev = torch.device("cuda:0")
    d_edges = torch.from_numpy(frames).to(dev)
    cap = 1 << 16
    d_lines = torch.zeros((4, cap, 5), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
    plan = capi.Plan(hip_ctx, W, H, W, 4, 1.0)
    st = torch.cuda.current_stream().cuda_stream
    thr = 25
    for _ in range(2):                                                     # twice: no state may leak between calls either
        plan.houghsht(d_edges.data_ptr(), thr, 0, d_lines.data_ptr(), cap, d_counts.data_ptr(), st)
        torch.cuda.synchronize()
        counts = d_counts.cpu().numpy()
        raw = d_lines.cpu().numpy().view(np.uint8).reshape(4, cap, 20)
        _, R, T, _ = plan.acc(0)
        for f in range(4):
            a = torch.zeros((R, T), dtype=torch.int32, device=dev)
            plan.acc_export(f, a.data_ptr(), T, st)
            torch.cuda.synchronize()
            acc = oracle.sht_acc(frames[f], 1.0)
            assert (a.cpu().numpy() == acc).all(), f
            exp = oracle.sht_lines_from_acc(acc, W, H, 1.0, thr)
            assert counts[f] == len(exp), (f, counts[f], len(exp))
            got = np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:len(exp)]
            assert _lines_tuple(got) == _orc_tuple(exp), f
        assert counts[0] == 0
    plan.close()


@pytest.mark.parametrize("name", ["fhd_1920x1080", "uhd_3840x2160"])
def test_plan_full_size_golden_and_properties(hip_ctx, golden, name):
    """BASELINE configs 2-4: full Canny and SHT at 1080p / 4K through the batched plan: MD5/sums from the compiled
    reference, plus size-independent properties (sum(acc) == E*T, idempotent re-run, frame independence)."""
    meta, _ = golden
    m = meta[name]
    W, H = m["W"], m["H"]
    f0 = synth_frame(W, H, m["seed"])
    f1 = synth_frame(W, H, m["seed"] + 1)
    frames = np.stack([f0, f1, f0])
    edges, lines_raw, counts, accs = _plan_run(hip_ctx, frames, m["tLow"], m["tHigh"], m["sht"]["threshold"], 1 << 16)
    assert md5_rows(edges[0]) == m["canny_md5"]
    assert (edges[0] == edges[2]).all() and not (edges[0] == edges[1]).all()      # frames are independent units
    E = int((edges[0] != 0).sum())
    assert E == m["canny_edges"]
    assert int(accs[0].sum()) == E * 180                                           # every edge votes once per theta
    assert (accs[0] == accs[2]).all()
    assert counts[0] == m["sht"]["lines"] == counts[2]
    from compv_amd import capi
    got = np.frombuffer(lines_raw[0].tobytes(), dtype=capi.LINE_DTYPE)[:counts[0]]
    assert int(got["strength"].astype(np.int64).sum()) == m["sht"]["sum_strength"]
    assert (np.diff(got["strength"].astype(np.int64)) <= 0).all()                  # sorted by strength descending
    assert accs[0].max() == got["strength"][0]


def test_plan_edge_dete_and_canny_modes_batch(hip_ctx, oracle):
    """Device-resident batched Sobel detector and Canny (mean-threshold mode, 5x5 kernel, aliased in/out) vs oracle."""
    import torch
    from compv_amd import capi
    W, H, n = 648, 200, 3
    frames = np.stack([synth_frame(W, H, 40 + f) for f in range(n)])
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(frames).to(dev)
    d_out = torch.empty_like(d_in)
    plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
    st = torch.cuda.current_stream().cuda_stream
    for op, oop in [(capi.OP_SOBEL, 0), (capi.OP_SCHARR, 2), (capi.OP_PREWITT, 3)]:
        plan.edge_dete(d_in.data_ptr(), op, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        for f in range(n):
            assert (got[f] == oracle.edge_dete(frames[f], oop)[0]).all(), (op, f)
    plan.canny(d_in.data_ptr(), 0.68, 1.36, d_out.data_ptr(), 3, capi.THRESHOLD_PERCENT_OF_MEAN, st)   # per-frame mean thresholds on device
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for f in range(n):
        assert (got[f] == oracle.canny(frames[f], 0.68, 1.36, 3, 1)[1]).all(), f
    plan.canny(d_in.data_ptr(), 400.0, 900.0, d_out.data_ptr(), 5, capi.THRESHOLD_COMPARE_TO_GRADIENT, st)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for f in range(n):
        assert (got[f] == oracle.canny(frames[f], 400.0, 900.0, 5)[1]).all(), f
    d_alias = d_in.clone()
    plan.canny(d_alias.data_ptr(), 59.0, 119.0, d_alias.data_ptr(), 3, capi.THRESHOLD_COMPARE_TO_GRADIENT, st)  # in place on the device
    torch.cuda.synchronize()
    got = d_alias.cpu().numpy()
    for f in range(n):
        assert (got[f] == oracle.canny(frames[f], 59.0, 119.0)[1]).all(), f
    plan.close()


def test_no_leaks(oracle):
    """hipMalloc/hipFree balance (the analogue of COMPV_DEBUG_CHECK_FOR_MEMORY_LEAKS, compv_api.h:148-155)."""
    from compv_amd import capi
    ctx = capi.Context(0)
    img = synth_frame(200, 100)
    e = ctx.canny(img, 59.0, 119.0)
    ctx.houghsht(e, 1.0, 20)
    ctx.edge_dete(img)
    ctx.houghkht(e)
    assert ctx.live_allocations() > 0
    plan = capi.Plan(ctx, 200, 104, 200, 2, 1.0)
    before = ctx.live_allocations()
    plan.close()
    assert ctx.live_allocations() < before          # every plan buffer was released
    ctx.close()
    assert ctx.h is None


# ---------------------------------------------------------------------------------------------------------------
# KHT (BASELINE config 5): hybrid host/GPU path vs oracle and vs the fixtures from the compiled reference
# ---------------------------------------------------------------------------------------------------------------
def _kht_tuple(lines):
    return [(float(l["rho"]), float(l["theta"]), int(l["strength"])) for l in lines]


@pytest.mark.parametrize("W,H,tl,th,rho,deg,thr", [(320, 240, 59., 119., 1.0, 1.0, 1), (641, 480, 59., 119., 1.0, 1.0, 1), (640, 480, 59., 119., 0.5, 1.0, 1),
                                                     (480, 360, 59., 119., 1.0, 0.5, 150), (257, 129, 0.8, 1.6, 1.0, 2.0, 1), (333, 77, 0.8, 1.6, 1.0, 1.0, 1)])
def test_houghkht_matches_oracle(hip_ctx, oracle, W, H, tl, th, rho, deg, thr):
    img = synth_frame(W, H, 99)
    rc, edges = oracle.canny(img, tl, th)
    exp, gs_exp = oracle.kht(edges, rho, deg, thr)
    got, gs = hip_ctx.houghkht(edges, rho, deg, thr)
    assert gs == gs_exp                                      # COMPV_HOUGHKHT_GET_FLT64_GS, bit-exact
    assert _kht_tuple(got) == [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2])) for l in exp]   # values AND order
    if len(exp) > 3:
        top, _ = hip_ctx.houghkht(edges, rho, deg, thr, max_lines=3)
        assert _kht_tuple(top) == _kht_tuple(got[:3])


@pytest.mark.parametrize("W,H", [(32767, 48), (48, 32767), (8191, 200)])
def test_houghkht_at_the_coordinate_limits(hip_ctx, oracle, W, H):
    """Images as wide / tall as the API takes (W, H <= 32 767: the reference's int16 coordinates, and the 4-byte points of the product's strings): pixel
    coordinates up to 32 765, 32 768 rho bins.  Lines (values and order) and GS against the oracle."""
    img = synth_frame(W, H, 99)
    rc, edges = oracle.canny(img, 59.0, 119.0)
    assert rc == 0
    edges = edges.copy()
    edges[2:H - 2, W - 2] = 255; edges[H - 2, 2:W - 2] = 255      # strings along the last interior column and row: x = W - 2, y = H - 2
    edges[H // 2, W - 40:W - 1] = 255
    exp, gs_exp = oracle.kht(edges, 1.0, 1.0, 1)
    got, gs = hip_ctx.houghkht(edges, 1.0, 1.0, 1)
    assert gs == gs_exp and len(exp) > 100
    assert _kht_tuple(got) == [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2])) for l in exp]


@pytest.mark.parametrize("W,H,p,amp", [(697, 34, 4, 200), (700, 40, 8, 150), (946, 27, 3, 180), (1093, 224, 5, 230)])
def test_houghkht_exactly_collinear_clusters(hip_ctx, oracle, W, H, p, amp):
    """Checkerboards give EXACTLY collinear clusters: the kernels' sigmas hit their floor and the Gaussian's peak vote exceeds 2^31.  The reference
    converts it with static_cast<int32_t> -- cvttsd2si on its x86 build: INT_MIN, i.e. "no vote" -- and the device conversion saturates to INT_MAX
    instead unless it is told not to (kht_kernels.hip, cvttsd2si; found by tools/fuzz_parity.py in round 4: lines of strength 2147483627)."""
    yy, xx = np.mgrid[0:H, 0:W]
    img = ((((xx // p) + (yy // p)) & 1) * amp).astype(np.uint8)
    rc, edges = oracle.canny(img, 59.0, 119.0)
    assert rc == 0
    exp, gs_exp = oracle.kht(edges, 1.0, 1.0, 1)
    got, gs = hip_ctx.houghkht(edges, 1.0, 1.0, 1)
    assert gs == gs_exp and len(exp) > 0
    assert _kht_tuple(got) == [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2])) for l in exp]


@pytest.mark.parametrize("W,H,tl,th,min_dev,min_size", [(320, 240, 59., 119., 2.0, 10), (1282, 720, 0.8, 1.6, 2.0, 10), (641, 480, 59., 119., 0.5, 5),
                                                         (1920, 1080, 59., 119., 2.0, 10), (333, 77, 0.8, 1.6, 4.0, 3),
                                                         (320, 240, 59., 119., 2.0, 2),      # smallest clusters the recursion can produce
                                                         (641, 480, 59., 119., 0.0, 4),      # no deviation floor: ratios may be +inf
                                                         (3840, 2160, 59., 119., 2.0, 10)])
def test_houghkht_cluster_statistics_kernel_bit_exact(hip_ctx, oracle, W, H, tl, th, min_dev, min_size):
    """kht_subdivide_kernel (one wave per string) + kht_stats_kernel (one thread per cluster, float64, __ddiv_rn / __dsqrt_rn) against the
    oracle's clusters_find + voting_Algorithm2_Kernels: the same clusters in the same order, all seven fields of every kernel and hmax,
    bit for bit (theta goes through the host libm acos on both sides)."""
    img = synth_frame(W, H, 4242)
    rc, edges = oracle.canny(img, tl, th)
    exp, hmax_exp = oracle.kht_kernels(edges, min_dev, min_size)
    got, hmax = hip_ctx.houghkht_kernels(edges, min_dev, min_size)
    assert got.shape == exp.shape and len(exp) > 0
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
    assert hmax == hmax_exp


@pytest.mark.parametrize("name", ["small_320x240", "hd_1280x720", "fhd_1920x1080", "dense_1282x720", "uhd_3840x2160"])
def test_houghkht_golden(hip_ctx, golden, name):
    """Line set (values and order) and GS recorded from the compiled reference; 4K = BASELINE config 5."""
    meta, arrays = golden
    m = meta[name]
    img = synth_frame(m["W"], m["H"], m["seed"])
    can = hip_ctx.canny(img, m["tLow"], m["tHigh"], 3, m["threshold_type"])
    lines, gs = hip_ctx.houghkht(can, 1.0, 1.0, 1)
    k = m["kht"]
    assert repr(gs) == k["gs"]
    assert len(lines) == k["lines"]
    exp = arrays[name + "/kht_lines"]
    got = np.stack([lines["rho"].astype(np.float64), lines["theta"].astype(np.float64), lines["strength"].astype(np.float64)], axis=1)
    assert (got == exp).all()


def test_houghkht_empty_and_errors(hip_ctx):
    from compv_amd import capi
    e = np.zeros((64, 64), np.uint8)
    lines, gs = hip_ctx.houghkht(e)
    assert len(lines) == 0 and gs == 1.0
    with pytest.raises(capi.CompvHipError) as ex:
        hip_ctx.houghkht(e, rho=1.5)                           # rho must be in (0,1] (houghkht.cxx:491)
    assert ex.value.code == capi.E_INVALID_PARAMETER
    # defined deviation: clusterMinSize = 1 sends the reference's clusters_subdivision into an unbounded recursion (houghkht.cxx:795-821: a stack
    # overflow, which the C restatement reproduces); the HIP path refuses the value instead
    for fn in (lambda: hip_ctx.houghkht(e, min_size=1), lambda: hip_ctx.houghkht_kernels(e, 2.0, 1)):
        with pytest.raises(capi.CompvHipError) as ex:
            fn()
        assert ex.value.code == capi.E_INVALID_PARAMETER
    # a parameter space whose cell key theta * 2 (rhoN + 2) + rho no longer fits 32 bits is refused instead of decoded wrongly:
    # 4K with rho = 0.01, theta = 0.02 deg has ~7.9e9 keys
    big = np.zeros((2160, 3840), np.uint8)
    with pytest.raises(capi.CompvHipError) as ex:
        hip_ctx.houghkht(big, rho=0.01, theta_deg=0.02)
    assert ex.value.code == capi.E_INVALID_PARAMETER


def test_plan_to_cartesian_on_device(hip_ctx, oracle):
    """CompVHoughSht::toCartesian on the device line arrays (samples/hough_lines/main.cxx:108) vs the oracle, 2 frames; one frame
    holds a perfectly vertical line (theta == 0 branch)."""
    import torch
    from compv_amd import capi
    W, H, F, cap = 640, 480, 2, 4096
    frames = np.stack([synth_frame(W, H, 99), np.full((H, W), 30, np.uint8)])
    frames[1][:, 300:304] = 250                      # vertical bar -> theta = 0 lines
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(frames).to(dev)
    d_edges = torch.empty_like(d_in)
    d_lines = torch.zeros((F, cap, 5), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
    d_cart = torch.full((F, cap, 4), float("nan"), dtype=torch.float32, device=dev)
    plan = capi.Plan(hip_ctx, W, H, W, F, 1.0)
    try:
        plan.pipeline(d_in.data_ptr(), 59.0, 119.0, 60, 0, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr())
        plan.to_cartesian(d_lines.data_ptr(), d_counts.data_ptr(), cap, d_cart.data_ptr())
        torch.cuda.synchronize()
        counts = d_counts.cpu().numpy()
        lines = d_lines.cpu().numpy().view(np.uint8).reshape(F, cap, 20)
        cart = d_cart.cpu().numpy()
        saw_vertical = False
        for f in range(F):
            n = int(counts[f])
            assert 0 < n <= cap
            rec = np.frombuffer(lines[f][:n].tobytes(), dtype=capi.LINE_DTYPE)
            polar = [(float(r["rho"]), float(r["theta"])) for r in rec]
            saw_vertical |= any(t == 0.0 for _, t in polar)
            exp = oracle.sht_to_cartesian(W, H, polar)
            assert cart[f][:n].view(np.uint32).tolist() == exp.view(np.uint32).tolist()
            assert np.isnan(cart[f][n:]).all()      # nothing written past the line count
        assert saw_vertical
    finally:
        plan.close()


def test_random_shape_sweep(hip_ctx, oracle):
    """Seeded sweep over odd sizes / thresholds / operators: Sobel-family detector, Canny (3x3 and 5x5, both threshold modes) and
    SHT (accumulator + line set) against the oracle.  Catches tile-edge and ragged-tail cases no hand-picked size covers."""
    from compv_amd import capi
    rng = np.random.default_rng(20240926)
    for it in range(40):
        W = int(rng.integers(5, 1100)); H = int(rng.integers(5, 300))
        kind = it % 4
        if kind == 0:
            img = synth_frame(W, H, 1000 + it)
        elif kind == 1:
            img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
        elif kind == 2:
            img = (np.add.outer(np.arange(H) * 3, np.arange(W) * 2) % 256).astype(np.uint8)
            img[rng.random((H, W)) < 0.02] = 255
        else:
            img = np.zeros((H, W), np.uint8)
            img[H // 3: 2 * H // 3 + 1, W // 4: 3 * W // 4 + 1] = int(rng.integers(60, 256))
        op = [0, 2, 3][it % 3]
        assert (hip_ctx.edge_dete(img, op) == oracle.edge_dete(img, op)[0]).all(), (it, W, H, op)
        tl = float(rng.uniform(5, 120)); th = tl * float(rng.uniform(1.2, 3.0))
        ks = 5 if (it % 5 == 0 and W >= 5 and H >= 5) else 3
        rc, exp = oracle.canny(img, tl, th, ks)
        got = hip_ctx.canny(img, tl, th, ksize=ks)
        assert rc == 0 and (got == exp).all(), (it, W, H, tl, th, ks, int((got != exp).sum()))
        rc, exp_m = oracle.canny(img, 0.6, 1.3, 3, 1)
        got_m = hip_ctx.canny(img, 0.6, 1.3, threshold_type=capi.THRESHOLD_PERCENT_OF_MEAN)
        assert rc == 0 and (got_m == exp_m).all(), (it, W, H, "mean")
        deg = [1.0, 0.5, 2.0, 1.5][it % 4]
        thr = int(rng.integers(5, 60))
        acc_exp = oracle.sht_acc(exp, deg)
        lines, acc = hip_ctx.houghsht(exp, deg, thr, want_acc=True)
        assert (acc == acc_exp).all(), (it, W, H, deg)
        assert _lines_tuple(lines) == _orc_tuple(oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, deg, thr)), (it, W, H, deg, thr)


def test_full_size_properties_linearity_and_monotonicity(hip_ctx):
    """Size-independent properties at BASELINE's full 4K size (no oracle needed): the Hough accumulator is linear in the edge
    set (acc(A u B) = acc(A) + acc(B) for disjoint A, B; sum(acc) = |E| * T), Canny's edge set grows monotonically when tLow
    drops, is a subset of the weak set and a superset of the seeds, and is idempotent under re-thresholding of its own output."""
    W, H = 3840, 2160
    img = synth_frame(W, H, 424242)
    e_hi = hip_ctx.canny(img, 59.0, 119.0)
    e_lo = hip_ctx.canny(img, 20.0, 119.0)
    assert set(np.unique(e_hi)) <= {0, 255}
    assert ((e_hi != 0) <= (e_lo != 0)).all()                  # lower tLow only adds weak pixels reachable from the same seeds
    assert (e_lo != 0).sum() > (e_hi != 0).sum()
    e_seed = hip_ctx.canny(img, 118.0, 119.0)                 # tLow ~ tHigh: (almost) only the seeds survive
    assert ((e_seed != 0) <= (e_hi != 0)).all()
    A = e_hi.copy(); B = e_hi.copy()
    A[:, W // 2:] = 0; B[:, :W // 2] = 0                       # disjoint halves of the edge set
    _, acc_a = hip_ctx.houghsht(A, 1.0, 100, want_acc=True)
    _, acc_b = hip_ctx.houghsht(B, 1.0, 100, want_acc=True)
    _, acc = hip_ctx.houghsht(e_hi, 1.0, 100, want_acc=True)
    assert (acc_a.astype(np.int64) + acc_b == acc).all()
    T = acc.shape[1]
    assert int(acc.sum()) == int((e_hi != 0).sum()) * T


@pytest.mark.parametrize("W,H", [(32767, 5), (5, 32767), (16384, 3), (3, 4097)])
def test_extreme_aspect_ratios(hip_ctx, oracle, W, H):
    """The reference's coordinate limit is 32767 (int16 in the hysteresis stack, canny_dete.cxx:617-621): maximum widths/heights."""
    rng = np.random.default_rng(W + 7 * H)
    img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    assert (hip_ctx.edge_dete(img) == oracle.edge_dete(img)[0]).all()
    rc, exp = oracle.canny(img, 40.0, 90.0)
    assert rc == 0 and (hip_ctx.canny(img, 40.0, 90.0) == exp).all()
    from compv_amd import capi
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.canny(np.zeros((4, 32768), np.uint8), 40.0, 90.0)     # one past the limit
    assert e.value.code == capi.E_INVALID_PARAMETER


def test_houghsht_line_buffer_contract(hip_ctx, oracle):
    """C-ABI contract of the caller's line buffer (include/compv_hip.h): too small -> COMPVHIP_E_OUT_OF_BOUND, *n = lines found, the
    first `cap` (strongest) lines written; and more lines than the library's initial device key buffer (65 536) are handled."""
    import ctypes as C
    from compv_amd import capi
    W, H, deg = 1282, 720, 0.25
    rng = np.random.default_rng(5)
    edges = np.where(rng.random((H, W)) < 0.3, 0xff, 0).astype(np.uint8)
    acc = oracle.sht_acc(edges, deg)
    exp = oracle.sht_lines_from_acc_reference_order(acc, W, H, deg, 1)
    assert len(exp) > (1 << 16)                                # more than the initial device key capacity
    got = hip_ctx.houghsht(edges, deg, 1)                      # the wrapper retries with the reported size
    assert _lines_tuple(got) == _orc_tuple(exp)
    cap = 7
    lines = np.zeros(cap, capi.LINE_DTYPE)
    n = C.c_size_t(0)
    rc = hip_ctx.lib.compvhip_houghsht_u8(hip_ctx.h, edges.ctypes.data, W, H, W, 1.0, deg, 1, 0, lines.ctypes.data, cap, C.byref(n), None, 0)
    assert rc == capi.E_OUT_OF_BOUND and n.value == len(exp)
    assert _lines_tuple(lines) == _orc_tuple(exp[:cap])


def test_plan_small_line_capacity_keeps_the_strongest(hip_ctx, oracle):
    """Plan API: lineCap smaller than the number of lines -> d_counts reports the lines found, d_lines holds the strongest lineCap."""
    import torch
    from compv_amd import capi
    W, H, cap = 640, 480, 16
    img = synth_frame(W, H, 31337)
    rc, e = oracle.canny(img, 59.0, 119.0)
    exp = oracle.sht(e, 1.0, 40)
    assert len(exp) > cap
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(img[None]).to(dev)
    d_edges = torch.empty_like(d_in)
    d_lines = torch.zeros((1, cap, 5), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = capi.Plan(hip_ctx, W, H, W, 1, 1.0)
    try:
        plan.pipeline(d_in.data_ptr(), 59.0, 119.0, 40, 0, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr())
        torch.cuda.synchronize()
        assert int(d_counts.cpu()[0]) == len(exp)
        rec = np.frombuffer(d_lines.cpu().numpy().tobytes(), dtype=capi.LINE_DTYPE)
        assert _lines_tuple(rec) == _orc_tuple(exp[:cap])
    finally:
        plan.close()


# ---------------------------------------------------------------------------------------------------------------
# round-2 evidence: cell-by-cell vote histograms at BASELINE's full sizes, the reference's own unit-test parameters,
# the key-buffer contract with a small lineCap, the asynchronous step
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["fhd_1920x1080", "uhd_3840x2160"])
def test_plan_full_size_accumulator_cell_by_cell(hip_ctx, oracle, golden, name):
    """BASELINE config 3/4: the Hough vote histogram of the batched plan equals the oracle's CELL BY CELL at 1920x1080 and
    3840x2160 (int32 [R][T], reference layout), and so does the complete line set."""
    from compv_amd import capi
    meta, _ = golden
    m = meta[name]
    W, H = m["W"], m["H"]
    frames = np.stack([synth_frame(W, H, m["seed"]), synth_frame(W, H, m["seed"] + 3)])
    thr = m["sht"]["threshold"]
    edges, lines_raw, counts, accs = _plan_run(hip_ctx, frames, m["tLow"], m["tHigh"], thr, 1 << 16)
    assert md5_rows(edges[0]) == m["canny_md5"]
    for f in range(2):
        rc, e = oracle.canny(frames[f], m["tLow"], m["tHigh"])
        assert rc == 0 and (edges[f] == e).all()
        acc = oracle.sht_acc(e, 1.0)
        assert accs[f].shape == acc.shape
        assert (accs[f] == acc).all(), (f, int((accs[f] != acc).sum()))
        exp = oracle.sht_lines_from_acc(acc, W, H, 1.0, thr)
        assert counts[f] == len(exp)
        got = np.frombuffer(lines_raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:len(exp)]
        assert _lines_tuple(got) == _orc_tuple(exp)


def _unittest_golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_unittest.json")) as f:
        return json.load(f)


def test_houghsht_reference_unittest_parameters_small(hip_ctx, oracle):
    """unittests/houghsht.cxx:17-21: Canny(0.8, 1.6), theta = kfMathTrigPiOver180 'degrees' (T = 10313 theta bins, 5157 vote
    workgroups, 26-bit cell indices), threshold 100 -- accumulator and line set against the oracle, sums against the fixture the
    compiled reference produced (tests/golden/make_golden_unittest.py)."""
    g = _unittest_golden()
    for name in ("unittest_200x258", "unittest_320x240"):
        m = g[name]
        W, H = m["W"], m["H"]
        can = hip_ctx.canny(synth_frame(W, H, m["seed"]), m["tLow"], m["tHigh"])
        assert md5_rows(can) == m["canny_md5"]
        R, T, _ = hip_ctx.houghsht_dims(W, H, m["theta_deg"])
        assert T == 10313 and R == 2 * (W + H) + 1
        acc_exp = oracle.sht_acc(can, m["theta_deg"])
        exp = oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, m["theta_deg"], m["threshold"])
        lines, acc = hip_ctx.houghsht(can, m["theta_deg"], m["threshold"], cap=max(1, m["lines"]), want_acc=True)
        assert (acc == acc_exp).all(), int((acc != acc_exp).sum())
        assert len(lines) == m["lines"] == len(exp)
        assert _lines_tuple(lines) == _orc_tuple(exp)
        assert float(lines["rho"].astype(np.float64).sum()) == m["sum_rho"]
        assert abs(float(lines["theta"].astype(np.float64).sum()) - m["sum_theta"]) <= 0.0009765625      # the unit test's own tolerance
        assert int(lines["strength"].astype(np.int64).sum()) == m["sum_strength"]


@pytest.mark.parametrize("name", ["vga_all", "vga_top100", "hd_halfdeg", "ragged_top40", "calib_like"])
def test_houghsht_host_path_returns_the_references_line_order(hip_ctx, name):
    """compvhip_houghsht_u8 against the compiled reference's list, element by element -- the order inside equal-strength groups and the
    survivors of the maxLines cut included (fixture: tests/golden/make_golden_sht_order.py)."""
    import hashlib, json
    m = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_sht_order.json")))[name]
    can = hip_ctx.canny(synth_frame(m["W"], m["H"], m["seed"]), m["tLow"], m["tHigh"])
    assert md5_rows(can) == m["canny_md5"]
    lines = hip_ctx.houghsht(can, m["theta_deg"], m["threshold"], max_lines=m["max_lines"])
    a = np.zeros((len(lines), 3), np.uint32)
    a[:, 0] = lines["rho"].astype(np.float32).view(np.uint32)
    a[:, 1] = lines["theta"].astype(np.float32).view(np.uint32)
    a[:, 2] = lines["strength"].astype(np.uint32)
    assert len(a) == m["lines"]
    assert a[:64].tolist() == m["head"] and a[-64:].tolist() == m["tail"]
    assert hashlib.md5(a.tobytes()).hexdigest() == m["md5"]


def test_houghsht_reference_unittest_parameters_1282x720(hip_ctx):
    """The same at the size of the unit test's first image (1282x720): 347 625 edge pixels x 10 313 theta bins, 1.3 million lines
    (20x the initial device key buffer) -- the three sums the unit test checks (houghsht.cxx:64-73) against the compiled reference."""
    m = _unittest_golden()["unittest_1282x720"]
    W, H = m["W"], m["H"]
    can = hip_ctx.canny(synth_frame(W, H, m["seed"]), m["tLow"], m["tHigh"])
    assert md5_rows(can) == m["canny_md5"] and int((can != 0).sum()) == m["canny_edges"]
    lines = hip_ctx.houghsht(can, m["theta_deg"], m["threshold"], cap=m["lines"])
    assert len(lines) == m["lines"]
    assert float(lines["rho"].astype(np.float64).sum()) == m["sum_rho"]
    assert abs(float(lines["theta"].astype(np.float64).sum()) - m["sum_theta"]) <= 0.0009765625
    assert int(lines["strength"].astype(np.int64).sum()) == m["sum_strength"]
    assert int(lines["strength"][0]) == m["max_strength"]
    assert (np.diff(lines["strength"].astype(np.int64)) <= 0).all()


def test_plan_small_line_capacity_with_many_candidates(hip_ctx, oracle):
    """Key-buffer contract (include/compv_hip.h, compvhip_plan_houghsht): with MORE than 1024 candidate lines per frame and a
    tiny lineCap / maxLines the plan still returns the strongest lines (the device key buffer holds max(lineCap, 65536))."""
    import torch
    from compv_amd import capi
    W, H, n = 800, 600, 2
    frames = np.stack([synth_frame(W, H, 4242 + f) for f in range(n)])
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(frames).to(dev)
    d_edges = torch.empty_like(d_in)
    for cap, max_lines in ((8, 0), (64, 5), (2000, 0)):
        d_lines = torch.zeros((n, cap, 5), dtype=torch.int32, device=dev)
        d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
        plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
        try:
            plan.pipeline(d_in.data_ptr(), 0.8, 1.6, 30, max_lines, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr())
            torch.cuda.synchronize()
            raw = d_lines.cpu().numpy().view(np.uint8).reshape(n, cap, 20)
            for f in range(n):
                rc, e = oracle.canny(frames[f], 0.8, 1.6)
                exp = oracle.sht(e, 1.0, 30)
                assert len(exp) > 1024                          # more candidates than the old minimum key capacity
                assert int(d_counts.cpu()[f]) == len(exp)
                keep = min(cap, max_lines) if max_lines > 0 else cap
                rec = np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:keep]
                assert _lines_tuple(rec) == _orc_tuple(exp[:keep]), (cap, max_lines, f)
        finally:
            plan.close()


def test_plan_line_sort_over_several_chunks_with_an_empty_frame(hip_ctx, oracle):
    """The device-sized line sort (sht_sort_kernels.hip): a frame whose lines fill several 4096-line chunks (the last one partly), a frame without a
    single line between two that have some, a line capacity that is no multiple of the chunk and a maxLines cut inside a chunk -- every list against
    the oracle's canonical order, element by element."""
    import torch
    from compv_amd import capi
    W, H, n = 1024, 768, 4
    bars = np.full((H, W), 90, np.uint8)                              # nine bars: strengths up to the bars' height, where the noise frames 0 and 2 stay near the threshold
    for b in range(9):
        bars[40:H - 40, 60 + 100 * b:63 + 100 * b] = 200
    frames = np.stack([synth_frame(W, H, 777), np.full((H, W), 90, np.uint8), synth_frame(W, H, 778), bars])
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(frames).to(dev)
    d_edges = torch.empty_like(d_in)
    exp = []
    for f in range(n):
        rc, e = oracle.canny(frames[f], 0.8, 1.6)
        exp.append(oracle.sht(e, 1.0, 12))
    assert len(exp[0]) > 2 * 4096 and len(exp[0]) % 4096 != 0 and len(exp[1]) == 0 and len(exp[3]) > 4096
    for cap, max_lines in ((70000, 0), (10000, 0), (10000, 5000), (4096, 4097)):
        d_lines = torch.zeros((n, cap, 5), dtype=torch.int32, device=dev)
        d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
        plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
        try:
            plan.pipeline(d_in.data_ptr(), 0.8, 1.6, 12, max_lines, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr())
            torch.cuda.synchronize()
            raw = d_lines.cpu().numpy().view(np.uint8).reshape(n, cap, 20)
            counts = d_counts.cpu().numpy()
            for f in range(n):
                assert int(counts[f]) == len(exp[f]), (cap, max_lines, f)
                keep = min(cap, max_lines, len(exp[f])) if max_lines > 0 else min(cap, len(exp[f]))
                if len(exp[f]) <= max(cap, 65536):                # (beyond the key capacity the kept subset is arbitrary: include/compv_hip.h)
                    rec = np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:keep]
                    assert _lines_tuple(rec) == _orc_tuple(exp[f][:keep]), (cap, max_lines, f)
        finally:
            plan.close()


def test_houghsht_dense_local_maxima_past_the_lds_stage(hip_ctx, oracle):
    """sht_lines_kernel stages up to 512 (key, cell) pairs of a 64-row block in the LDS and stores denser blocks directly: a 2 % random
    edge map with threshold 1 has up to ~1600 local maxima per block (and blocks on both sides of the limit)."""
    W, H = 512, 384
    rng = np.random.default_rng(5)
    e = (rng.random((H, W)) < 0.02).astype(np.uint8) * 255
    acc_exp = oracle.sht_acc(e, 1.0)
    exp = oracle.sht_lines_from_acc_reference_order(acc_exp, W, H, 1.0, 1)
    per_block = np.bincount(np.array([l[3] for l in exp]) // 64)
    assert per_block.max() > 1024 and (per_block[per_block > 0] < 512).any()
    lines, acc = hip_ctx.houghsht(e, 1.0, 1, want_acc=True)
    assert (acc == acc_exp).all()
    assert _lines_tuple(lines) == _orc_tuple(exp)


def test_plan_pipeline_async_matches_sync(hip_ctx, oracle):
    """compvhip_plan_pipeline_async / compvhip_plan_wait: two steps in flight on two buffer sets, same results as the
    synchronous call; includes a frame whose hysteresis needs more than the speculative resolve rounds (replay path)."""
    import torch
    from compv_amd import capi
    W, H, n, cap = 1104, 700, 2, 4096                           # the plan API needs a stride that is a multiple of 8
    serp = np.full((H, W), 100, np.uint8)                       # long weak chains crossing many 64-row bands (test_canny_long_weak_chains)
    for k, yy in enumerate(range(20, H - 20, 12)):
        serp[yy:yy + 3, 15:W - 15] = 112
        xs = W - 30 if (k % 2 == 0) else 15
        serp[yy:yy + 15, xs:xs + 3] = 112
    serp[18:26, 10:20] = 255
    batches = [np.stack([synth_frame(W, H, 5), synth_frame(W, H, 6)]), np.stack([serp, synth_frame(W, H, 7)])]
    params = [(59.0, 119.0), (10.0, 200.0)]
    dev = torch.device("cuda:0")
    plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
    st = torch.cuda.Stream(device=dev)
    try:
        bufs = []
        for b in batches:
            bufs.append((torch.from_numpy(b).to(dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev),
                         torch.zeros((n, cap, 5), dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        tickets = []
        for (d_in, d_e, d_l, d_c), (tl, th) in zip(bufs, params):
            tickets.append(plan.pipeline_async(d_in.data_ptr(), tl, th, 40, 0, d_e.data_ptr(), d_l.data_ptr(), cap, d_c.data_ptr(), st.cuda_stream))
        # results are checked in issue order: each wait() may replay its step synchronously
        for k, t in enumerate(tickets):
            plan.wait(t)
            st.synchronize()
            d_in, d_e, d_l, d_c = bufs[k]
            tl, th = params[k]
            if k == 0:
                edges = d_e.cpu().numpy(); counts = d_c.cpu().numpy(); raw = d_l.cpu().numpy().view(np.uint8).reshape(n, cap, 20)
                for f in range(n):
                    rc, e = oracle.canny(batches[k][f], tl, th)
                    assert (edges[f] == e).all(), (k, f)
                    exp = oracle.sht(e, 1.0, 40)
                    assert counts[f] == len(exp)
                    assert _lines_tuple(np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:min(len(exp), cap)]) == _orc_tuple(exp[:cap])
        # step 0's buffers may have been produced while step 1 was already running; step 1 (the hard one) is verified last
        d_in, d_e, d_l, d_c = bufs[1]
        edges = d_e.cpu().numpy(); counts = d_c.cpu().numpy(); raw = d_l.cpu().numpy().view(np.uint8).reshape(n, cap, 20)
        for f in range(n):
            rc, e = oracle.canny(batches[1][f], *params[1])
            assert (edges[f] == e).all(), ("step1", f, int((edges[f] != e).sum()))
            exp = oracle.sht(e, 1.0, 40)
            assert counts[f] == len(exp)
            assert _lines_tuple(np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:min(len(exp), cap)]) == _orc_tuple(exp[:cap])
        with pytest.raises(capi.CompvHipError):
            plan.wait(tickets[0])                                 # a ticket can be waited for once
    finally:
        plan.close()


def test_plan_wide_frames_use_the_library_sort_sized_by_the_lines(hip_ctx, oracle):
    """max(W, H) > 4095: a strength needs 14 bits, so the line sort is the library's radix sort over the key slots in use -- read back by the
    synchronous step, predicted from the plan's earlier steps by the asynchronous one.  Three asynchronous steps (the second has far more lines
    than the first lets it predict: the replay of compvhip_plan_wait; the third has fewer), then the synchronous and the stream-ordered
    entry points on the same plan; every list against the oracle, element by element, in the canonical order."""
    import torch
    from compv_amd import capi
    W, H, n, cap = 4104, 72, 2, 8192
    dev = torch.device("cuda:0")

    def frames_with(nbars):
        out = []
        for f in range(n):
            img = np.full((H, W), 40, np.uint8)
            for b in range(nbars + f):
                x = 20 + 37 * b
                img[6:H - 6, x:x + 3] = 200
            out.append(img)
        return np.stack(out)

    batches = [frames_with(2), frames_with(100), frames_with(40)]
    plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
    st = torch.cuda.Stream(device=dev)

    def check(batch, d_e, d_l, d_c, thr, what):
        edges = d_e.cpu().numpy(); counts = d_c.cpu().numpy(); raw = d_l.cpu().numpy().view(np.uint8).reshape(n, cap, 20)
        for f in range(n):
            rc, e = oracle.canny(batch[f], 59.0, 119.0)
            assert (edges[f] == e).all(), (what, f)
            exp = oracle.sht(e, 1.0, thr)
            assert counts[f] == len(exp), (what, f, counts[f], len(exp))
            assert _lines_tuple(np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:min(len(exp), cap)]) == _orc_tuple(exp[:cap]), (what, f)
        return int(counts.sum())

    try:
        totals = []
        for k, b in enumerate(batches):
            d_in = torch.from_numpy(b).to(dev); d_e = torch.empty_like(d_in)
            d_l = torch.zeros((n, cap, 5), dtype=torch.int32, device=dev); d_c = torch.zeros(n, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            t = plan.pipeline_async(d_in.data_ptr(), 59.0, 119.0, 30, 0, d_e.data_ptr(), d_l.data_ptr(), cap, d_c.data_ptr(), st.cuda_stream)
            plan.wait(t); st.synchronize()
            totals.append(check(b, d_e, d_l, d_c, 30, "async %d" % k))
        assert totals[0] > 0 and totals[1] > totals[0] + (totals[0] >> 4) + 4096, totals   # step 1 has more lines than step 0 lets it predict: replayed
        b = batches[2]
        d_in = torch.from_numpy(b).to(dev); d_e = torch.empty_like(d_in)
        d_l = torch.zeros((n, cap, 5), dtype=torch.int32, device=dev); d_c = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        plan.pipeline(d_in.data_ptr(), 59.0, 119.0, 30, 0, d_e.data_ptr(), d_l.data_ptr(), cap, d_c.data_ptr(), st.cuda_stream)
        st.synchronize()
        check(b, d_e, d_l, d_c, 30, "sync")
        d_l.zero_(); d_c.zero_(); torch.cuda.synchronize()
        plan.houghsht(d_e.data_ptr(), 30, 0, d_l.data_ptr(), cap, d_c.data_ptr(), st.cuda_stream)   # stream-ordered: the whole capacity is sorted
        st.synchronize()
        check(b, d_e, d_l, d_c, 30, "stream-ordered")
    finally:
        plan.close()


def test_async_speculative_rounds_adapt_and_recover(hip_ctx, oracle):
    """Round 6: a plan enqueues as many speculative hysteresis rounds as its recent asynchronous steps needed (2 once four easy steps have been seen, never
    more than 3).  Six easy steps teach it 2; then a frame whose weak chains cross a dozen bands needs far more -- compvhip_plan_wait must replay it and return
    the reference's edge map --, and an easy step after it is right again."""
    import torch
    from compv_amd import capi
    W, H, n, cap = 1104, 700, 1, 4096
    serp = np.full((H, W), 100, np.uint8)
    for k, yy in enumerate(range(20, H - 20, 12)):
        serp[yy:yy + 3, 15:W - 15] = 112
        xs = W - 30 if (k % 2 == 0) else 15
        serp[yy:yy + 15, xs:xs + 3] = 112
    serp[18:26, 10:20] = 255
    easy = [synth_frame(W, H, 40 + i) for i in range(7)]
    seq = [(e, 59.0, 119.0) for e in easy[:6]] + [(serp, 10.0, 200.0), (easy[6], 59.0, 119.0)]
    dev = torch.device("cuda:0")
    plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
    st = torch.cuda.Stream(device=dev)
    try:
        for i, (img, tl, th) in enumerate(seq):
            d_in = torch.from_numpy(img[None]).to(dev)
            d_e = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
            d_l = torch.zeros((n, cap, 5), dtype=torch.int32, device=dev); d_c = torch.zeros(n, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            t = plan.pipeline_async(d_in.data_ptr(), tl, th, 40, 0, d_e.data_ptr(), d_l.data_ptr(), cap, d_c.data_ptr(), st.cuda_stream)
            plan.wait(t)
            st.synchronize()
            rc, e = oracle.canny(img, tl, th)
            got = d_e.cpu().numpy()[0]
            assert (got == e).all(), (i, int((got != e).sum()))
            assert int(d_c.cpu().numpy()[0]) == len(oracle.sht(e, 1.0, 40)), i
    finally:
        plan.close()


def test_async_replay_with_shared_output_buffers(hip_ctx, oracle):
    """compvhip_plan_wait's replay rule (include/compv_hip.h): steps in flight may SHARE their output buffers; when an earlier step is replayed
    (its hysteresis needed more rounds than were enqueued), the later steps are replayed too when they are waited for, so after wait(t) the
    buffers hold step t's results.  Step 0 = the serpentine frame that forces the replay, step 1 = an ordinary frame; both write ONE buffer set."""
    import torch
    from compv_amd import capi
    W, H, n, cap = 1104, 700, 1, 4096
    serp = np.full((H, W), 100, np.uint8)
    for k, yy in enumerate(range(20, H - 20, 12)):
        serp[yy:yy + 3, 15:W - 15] = 112
        xs = W - 30 if (k % 2 == 0) else 15
        serp[yy:yy + 15, xs:xs + 3] = 112
    serp[18:26, 10:20] = 255
    frames = [serp[None].copy(), synth_frame(W, H, 9)[None].copy()]
    params = [(10.0, 200.0), (59.0, 119.0)]
    dev = torch.device("cuda:0")
    plan = capi.Plan(hip_ctx, W, H, W, n, 1.0)
    st = torch.cuda.Stream(device=dev)
    try:
        d_in = [torch.from_numpy(f).to(dev) for f in frames]
        d_e = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
        d_l = torch.zeros((n, cap, 5), dtype=torch.int32, device=dev)
        d_c = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        tickets = [plan.pipeline_async(d_in[k].data_ptr(), params[k][0], params[k][1], 40, 0, d_e.data_ptr(), d_l.data_ptr(), cap, d_c.data_ptr(), st.cuda_stream)
                   for k in range(2)]
        for k, t in enumerate(tickets):
            plan.wait(t)                                          # wait(0) replays step 0 AFTER step 1 ran; wait(1) must bring step 1's results back
            st.synchronize()
            rc, e = oracle.canny(frames[k][0], *params[k])
            assert rc == 0
            assert (d_e.cpu().numpy()[0] == e).all(), ("step", k)
            exp = oracle.sht(e, 1.0, 40)
            assert int(d_c.cpu().numpy()[0]) == len(exp), ("step", k)
            raw = d_l.cpu().numpy().view(np.uint8).reshape(n, cap, 20)
            assert _lines_tuple(np.frombuffer(raw[0].tobytes(), dtype=capi.LINE_DTYPE)[:min(len(exp), cap)]) == _orc_tuple(exp[:cap]), ("step", k)
    finally:
        plan.close()


def test_two_plans_in_flight_on_two_streams(hip_ctx, oracle):
    """bench.py's default step mode: two plans, each with its own buffers and HIP stream, take asynchronous steps in turn so that kernels
    of different batches overlap on the GPU.  Different inputs per plan, several steps each; every plan's LAST result is compared with
    the oracle (nothing may leak between plans that share a context)."""
    import torch
    from compv_amd import capi
    W, H, n, cap, steps = 1104, 620, 3, 8192, 4
    dev = torch.device("cuda:0")
    lanes = []
    for k in range(2):
        frames = np.stack([synth_frame(W, H, 100 * k + f) for f in range(n)])
        lanes.append({"frames": frames, "plan": capi.Plan(hip_ctx, W, H, W, n, 1.0), "st": torch.cuda.Stream(device=dev),
                      "in": torch.from_numpy(frames).to(dev), "e": torch.empty((n, H, W), dtype=torch.uint8, device=dev),
                      "l": torch.zeros((n, cap, 5), dtype=torch.int32, device=dev), "c": torch.zeros(n, dtype=torch.int32, device=dev)})
    torch.cuda.synchronize()
    try:
        pend = []
        for s in range(steps):
            for q in lanes:
                t = q["plan"].pipeline_async(q["in"].data_ptr(), 59.0, 119.0, 45, 0, q["e"].data_ptr(), q["l"].data_ptr(), cap, q["c"].data_ptr(),
                                             q["st"].cuda_stream)
                pend.append((q, t))
                if len(pend) > 2:
                    q0, t0 = pend.pop(0)
                    q0["plan"].wait(t0)
        for q0, t0 in pend:
            q0["plan"].wait(t0)
        torch.cuda.synchronize()
        for k, q in enumerate(lanes):
            edges = q["e"].cpu().numpy(); counts = q["c"].cpu().numpy(); raw = q["l"].cpu().numpy().view(np.uint8).reshape(n, cap, 20)
            for f in range(n):
                rc, e = oracle.canny(q["frames"][f], 59.0, 119.0)
                assert (edges[f] == e).all(), (k, f)
                exp = oracle.sht(e, 1.0, 45)
                assert counts[f] == len(exp) and len(exp) <= cap
                assert _lines_tuple(np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:len(exp)]) == _orc_tuple(exp), (k, f)
    finally:
        for q in lanes:
            q["plan"].close()


def test_bench_batches_match_the_reference_fixture(hip_ctx):
    """The benchmark's own workload and step mode: two resident batches of 32 4K frames (seeds 12345 .. 12345+63 of BASELINE config 4),
    two plans on two streams, asynchronous steps in flight at the same time -- every frame's edge map (MD5), edge count, line count,
    strength sum and line-set hash against tests/golden/golden_batch.json, which the real CompV library produced
    (tests/golden/make_golden_batch.py).  bench.py runs the same check over all 256 frames after its timed region."""
    import hashlib
    import json
    import torch
    import bench
    from compv_amd import capi
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_batch.json")))
    W, H, F, cap = gold["W"], gold["H"], 32, 1 << 16
    by_seed = {g["seed"]: g for g in gold["frames"]}
    dev = torch.device("cuda:0")
    synth = bench.FrameSynth(torch, dev, W, H)
    assert np.array_equal(synth.frame(12345 + 40).cpu().numpy(), synth_frame(W, H, 12345 + 40))
    lanes = []
    for k in range(2):
        seeds = [gold["first_seed"] + k * F + f for f in range(F)]
        lanes.append({"seeds": seeds, "in": synth.batch(seeds), "plan": capi.Plan(hip_ctx, W, H, W, F, gold["theta_deg"]), "st": torch.cuda.Stream(device=dev),
                      "e": torch.empty((F, H, W), dtype=torch.uint8, device=dev), "l": torch.zeros((F, cap, 5), dtype=torch.int32, device=dev),
                      "c": torch.zeros(F, dtype=torch.int32, device=dev)})
    del synth
    torch.cuda.synchronize()
    try:
        for rep in range(2):      # the second round re-uses every buffer of both plans
            tickets = [(q, q["plan"].pipeline_async(q["in"].data_ptr(), gold["tLow"], gold["tHigh"], gold["threshold"], 0, q["e"].data_ptr(), q["l"].data_ptr(),
                                                    cap, q["c"].data_ptr(), q["st"].cuda_stream)) for q in lanes]
            for q, t in tickets:
                q["plan"].wait(t)
        torch.cuda.synchronize()
        for q in lanes:
            counts = q["c"].cpu().numpy()
            assert int(counts.max()) <= cap
            ln = q["l"].to(torch.int64)
            valid = torch.arange(cap, device=dev)[None, :] < q["c"].to(torch.int64)[:, None]
            hv = (W + H - ln[:, :, 3] + 32768) * 1000003 + ln[:, :, 4] * 7919 + ln[:, :, 2] * 31337
            hv = torch.where(valid, hv, torch.zeros_like(hv)).sum(dim=1).cpu().numpy()
            sums = torch.where(valid, ln[:, :, 2], torch.zeros_like(ln[:, :, 2])).sum(dim=1).cpu().numpy()
            edges = q["e"].cpu().numpy()
            for f in range(F):
                g = by_seed[q["seeds"][f]]
                got = {"canny_md5": hashlib.md5(edges[f].tobytes()).hexdigest(), "edges": int((edges[f] != 0).sum()), "lines": int(counts[f]),
                       "sum_strength": int(sums[f]), "line_hash": "%016x" % (int(hv[f]) & bench.M64)}
                assert got == {k: g[k] for k in got}, (q["seeds"][f], got, g)
    finally:
        for q in lanes:
            q["plan"].close()


def test_plan_houghkht_several_groups_in_flight(hip_ctx, oracle):
    """compvhip_plan_houghkht puts its frames through the stages in groups of 8, up to four groups at a time (own controller thread, stream and buffers
    each): 19 frames = groups of 8, 8 and 3 -- with 12 host threads three controllers, with 5 one, with 32 four -- every frame's line list (values and
    order) and GS against the oracle; an empty frame sits in the middle of a group."""
    import torch
    from compv_amd import capi
    W, H, F = 640, 480, 19
    dev = torch.device("cuda:0")
    frames = np.stack([synth_frame(W, H, 4000 + f) for f in range(F)])
    frames[10] = 31
    d_in = torch.from_numpy(frames).to(dev)
    d_e = torch.empty_like(d_in)
    plan = capi.Plan(hip_ctx, W, H, W, F, 1.0)
    try:
        plan.canny(d_in.data_ptr(), 59.0, 119.0, d_e.data_ptr())
        torch.cuda.synchronize()
        exp = []
        for f in range(F):
            rc, e = oracle.canny(frames[f], 59.0, 119.0)
            exp.append(oracle.kht(e, 1.0, 1.0, 20))
        assert sum(len(el) for el, _ in exp) > 0
        for threads in (12, 5, 32):
            lines, gs = plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 20, threads=threads)
            for f in range(F):
                el, egs = exp[f]
                assert _kht_tuple(lines[f]) == [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2])) for l in el], (threads, f)
                assert (gs[f] == egs) if len(el) else (gs[f] is None), (threads, f)
        assert len(lines[10]) == 0
    finally:
        plan.close()


def test_plan_houghkht_batch(hip_ctx, oracle, golden):
    """compvhip_plan_houghkht (BASELINE config 5 as a throughput path): a batch of device edge maps, host linking on a pool of worker
    threads pipelined with the GPU stages.  Every frame's line list (values AND order) and GS against the oracle -- frame 0 also against the
    fixture recorded from the compiled reference -- for several pool sizes, with an empty frame in the batch, then the error contract."""
    import torch
    from compv_amd import capi
    meta, arrays = golden
    W, H, F = 1920, 1080, 7
    dev = torch.device("cuda:0")
    frames = np.stack([synth_frame(W, H, 12345 + f) for f in range(F)])
    frames[5] = 77                                   # constant frame: no edges, no lines
    d_in = torch.from_numpy(frames).to(dev)
    d_e = torch.empty_like(d_in)
    plan = capi.Plan(hip_ctx, W, H, W, F, 1.0)
    try:
        plan.canny(d_in.data_ptr(), 59.0, 119.0, d_e.data_ptr())
        torch.cuda.synchronize()
        exp = []
        for f in range(F):
            rc, e = oracle.canny(frames[f], 59.0, 119.0)
            exp.append(oracle.kht(e, 1.0, 1.0, 1))
        for threads in (1, 3, 0):
            lines, gs = plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 1, threads=threads)
            for f in range(F):
                el, egs = exp[f]
                assert _kht_tuple(lines[f]) == [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2])) for l in el], (threads, f)
                assert (gs[f] == egs) if len(el) else (gs[f] is None), (threads, f)
            st = plan.houghkht_stage_ms()
            assert st["threads"] == (threads if threads else st["threads"]) and st["wall_ms"] > 0 and st["stages"]["link"] > 0
        assert len(lines[5]) == 0
        m = meta["fhd_1920x1080"]["kht"]
        assert repr(gs[0]) == m["gs"] and len(lines[0]) == m["lines"]
        got = np.stack([lines[0]["rho"].astype(np.float64), lines[0]["theta"].astype(np.float64), lines[0]["strength"].astype(np.float64)], axis=1)
        assert (got == arrays["fhd_1920x1080/kht_lines"]).all()
        top, _ = plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 1, max_lines=3)
        assert _kht_tuple(top[0]) == _kht_tuple(lines[0][:3])
        with pytest.raises(capi.CompvHipError) as err:
            plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 1, cap=2)            # fewer slots than lines
        assert err.value.code == capi.E_OUT_OF_BOUND
        with pytest.raises(capi.CompvHipError) as err:
            plan.houghkht(d_e.data_ptr(), 2.0, 1.0, 1)                   # rho must be in (0, 1] (houghkht.cxx:146-163)
        assert err.value.code == capi.E_INVALID_PARAMETER
    finally:
        plan.close()
    assert hip_ctx.live_allocations() >= 0
