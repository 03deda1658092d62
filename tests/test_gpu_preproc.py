"""GPU parity of the samples' caller-side pre-processing through the C ABI:
compvhip_grayscale_u8 / compvhip_otsu_u8 / plan variants / Canny with per-frame Otsu thresholds, against the oracle
(pinned to the compiled reference by tests/test_preproc_oracle.py)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from oracle_bindings import synth_frame

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_preproc import otsu_input, packed_input  # noqa: E402  (input generators only)

GOLDEN_PREPROC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_preproc.json")))

pytestmark = pytest.mark.gpu

FMT_NAMES = ["RGBA32", "ARGB32", "BGRA32", "RGB24", "BGR24", "RGB565LE", "RGB565BE", "BGR565LE", "BGR565BE", "YUYV422", "UYVY422", "Y"]


@pytest.mark.parametrize("fmt", range(len(FMT_NAMES)), ids=FMT_NAMES)
@pytest.mark.parametrize("W,H,S", [(64, 8, 64), (34, 5, 48), (642, 31, 704), (1920, 16, 1920)])
def test_grayscale_matches_oracle(hip_ctx, oracle, fmt, W, H, S):
    bpp = oracle.fmt_bytes(fmt)
    rng = np.random.default_rng(100 * fmt + W)
    packed = rng.integers(0, 256, size=(H, S * bpp), dtype=np.uint8)
    exp = oracle.grayscale(packed, fmt, W)
    got = hip_ctx.grayscale(packed, fmt, W)
    assert (got == exp).all(), int((got != exp).sum())


def test_grayscale_errors(hip_ctx):
    from compv_amd import capi
    px = np.zeros((4, 64 * 2), np.uint8)
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.grayscale(px, capi.FMT_YUYV422, 33)           # odd width for packed 4:2:2
    assert e.value.code == capi.E_INVALID_PARAMETER
    with pytest.raises(capi.CompvHipError) as e:
        hip_ctx.lib.compvhip_grayscale_u8.restype = int
        hip_ctx._chk(hip_ctx.lib.compvhip_grayscale_u8(hip_ctx.h, px.ctypes.data, 99, 32, 4, 64, px.ctypes.data, 64))
    assert e.value.code == capi.E_NOT_IMPLEMENTED              # conv_to_grayscale.cxx:86-88


@pytest.mark.parametrize("W,H,seed", [(20, 20, 1), (333, 77, 2), (641, 480, 3), (1282, 720, 4), (1920, 1080, 5), (3840, 2160, 6)])
def test_otsu_matches_oracle(hip_ctx, oracle, W, H, seed):
    rng = np.random.default_rng(seed)
    imgs = [synth_frame(W, H, 12345 + seed),
            rng.integers(0, 256, size=(H, W), dtype=np.uint8),
            np.where(rng.random((H, W)) < 0.3, rng.integers(150, 220, (H, W)), rng.integers(10, 90, (H, W))).astype(np.uint8),
            np.full((H, W), 77, np.uint8),                                  # one grey level -> 0
            (np.arange(W * H, dtype=np.uint32).reshape(H, W) % 251).astype(np.uint8)]
    for img in imgs:
        assert hip_ctx.otsu(img) == float(oracle.otsu(img))


def test_plan_sample_sequence_on_device(hip_ctx, oracle):
    """samples/hough_lines/main.cxx:102-108 with every step on the device: convertGrayscale -> thresholdOtsu ->
    Canny(LOW = t*0.5, HIGH = t) -> HoughSHT, 3 RGB24 frames in one batch."""
    import torch
    from compv_amd import capi
    W, H, S, F = 640, 480, 640, 3
    dev = torch.device("cuda", 0)
    rgb = np.zeros((F, H, S * 3), np.uint8)
    grays, otsus, edges_exp = [], [], []
    for f in range(F):
        base = synth_frame(W, H, 777 + f).astype(np.int32)
        frame = np.stack([np.clip(base + 20 * f, 0, 255), np.clip(base * 3 // 4, 0, 255), np.clip(255 - base, 0, 255)], axis=-1).astype(np.uint8)
        rgb[f] = frame.reshape(H, W * 3)
        g = oracle.grayscale(rgb[f], capi.FMT_RGB24, W)
        t = oracle.otsu(g)
        lo, hi = oracle.otsu_canny_thresholds(t)
        rc, e = oracle.canny(g, float(lo), float(hi))
        assert rc == 0
        grays.append(g); otsus.append(t); edges_exp.append(e)
    d_rgb = torch.from_numpy(rgb).to(dev)
    d_gray = torch.empty((F, H, S), dtype=torch.uint8, device=dev)
    d_edges = torch.empty_like(d_gray)
    d_otsu = torch.zeros(F, dtype=torch.int32, device=dev)
    plan = capi.Plan(hip_ctx, W, H, S, F, 1.0)
    try:
        plan.grayscale(d_rgb.data_ptr(), capi.FMT_RGB24, d_gray.data_ptr())
        plan.otsu(d_gray.data_ptr(), d_otsu.data_ptr())
        plan.canny(d_gray.data_ptr(), 0.5, 1.0, d_edges.data_ptr(), threshold_type=capi.THRESHOLD_OTSU)
        line_cap = 4096
        d_lines = torch.zeros((F, line_cap, 5), dtype=torch.int32, device=dev)
        d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
        plan.houghsht(0, 60, 0, d_lines.data_ptr(), line_cap, d_counts.data_ptr())
        torch.cuda.synchronize()
        assert d_otsu.cpu().tolist() == otsus
        g = d_gray.cpu().numpy()
        e = d_edges.cpu().numpy()
        counts = d_counts.cpu().numpy()
        for f in range(F):
            assert (g[f][:, :W] == grays[f]).all()
            assert (e[f][:, :W] == edges_exp[f]).all(), f
            exp_lines = oracle.sht(edges_exp[f], 1.0, 60)
            assert counts[f] == len(exp_lines)
    finally:
        plan.close()


@pytest.mark.parametrize("case", GOLDEN_PREPROC["grayscale"], ids=lambda c: "%s_%dx%d" % (c["name"], c["W"], c["H"]))
def test_grayscale_matches_reference_golden(hip_ctx, case):
    data = packed_input(case["fmt"], case["W"], case["H"], case["S"], case["seed"])
    g = hip_ctx.grayscale(data, case["fmt"], case["W"])
    assert hashlib.md5(np.ascontiguousarray(g).tobytes()).hexdigest() == case["md5"]


@pytest.mark.parametrize("case", GOLDEN_PREPROC["otsu"], ids=lambda c: "%s_%dx%d" % (c["kind"], c["W"], c["H"]))
def test_otsu_matches_reference_golden(hip_ctx, case):
    assert hip_ctx.otsu(otsu_input(case["kind"], case["W"], case["H"], case["seed"])) == float(case["threshold"])


def test_sample_loop_as_one_enqueue(hip_ctx, oracle):
    """compvhip_plan_pipeline_ex: the per-frame sequence of samples/hough_lines/main.cxx:102-109 -- convertGrayscale -> thresholdOtsu ->
    Canny(Otsu * 0.5, Otsu) -> HoughSHT -> toCartesian -- as ONE call on a batch of packed RGB24 frames, synchronous and as an asynchronous
    ticket; then the other configurations of the step (5x5 Sobel with PERCENT_OF_MEAN thresholds on a luma plane)."""
    import torch
    from compv_amd import capi
    W, H, S, F, cap = 640, 480, 640, 3, 4096
    dev = torch.device("cuda", 0)
    rgb = np.zeros((F, H, S * 3), np.uint8)
    grays, otsus, edges_exp, lines_exp = [], [], [], []
    for f in range(F):
        base = synth_frame(W, H, 901 + f).astype(np.int32)
        frame = np.stack([np.clip(base + 15 * f, 0, 255), np.clip(base * 3 // 4, 0, 255), np.clip(255 - base, 0, 255)], axis=-1).astype(np.uint8)
        rgb[f] = frame.reshape(H, W * 3)
        g = oracle.grayscale(rgb[f], capi.FMT_RGB24, W)
        t = oracle.otsu(g)
        lo, hi = oracle.otsu_canny_thresholds(t)
        rc, e = oracle.canny(g, float(lo), float(hi))
        assert rc == 0
        grays.append(g); otsus.append(t); edges_exp.append(e); lines_exp.append(oracle.sht(e, 1.0, 60))
    d_rgb = torch.from_numpy(rgb).to(dev)
    plan = capi.Plan(hip_ctx, W, H, S, F, 1.0)
    try:
        for asynchronous in (False, True):
            d_gray = torch.zeros((F, H, S), dtype=torch.uint8, device=dev)
            d_edges = torch.zeros_like(d_gray)
            d_otsu = torch.zeros(F, dtype=torch.int32, device=dev)
            d_lines = torch.zeros((F, cap, 5), dtype=torch.int32, device=dev)
            d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
            d_cart = torch.zeros((F, cap, 4), dtype=torch.float32, device=dev)
            t = plan.pipeline_ex(d_rgb.data_ptr(), 0.5, 1.0, 60, 0, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr(),
                                 threshold_type=capi.THRESHOLD_OTSU, pixfmt=capi.FMT_RGB24, d_gray=d_gray.data_ptr(), d_otsu=d_otsu.data_ptr(),
                                 d_cart=d_cart.data_ptr(), asynchronous=asynchronous)
            if asynchronous:
                plan.wait(t)
            torch.cuda.synchronize()
            assert d_otsu.cpu().tolist() == otsus
            g = d_gray.cpu().numpy(); e = d_edges.cpu().numpy(); counts = d_counts.cpu().numpy()
            raw = d_lines.cpu().numpy().view(np.uint8).reshape(F, cap, 20); cart = d_cart.cpu().numpy()
            for f in range(F):
                assert (g[f][:, :W] == grays[f]).all() and (e[f][:, :W] == edges_exp[f]).all(), (asynchronous, f)
                exp = lines_exp[f]
                assert counts[f] == len(exp) and 0 < len(exp) <= cap
                got = np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:len(exp)]
                assert [(int(l["row"]), int(l["col"]), int(l["strength"])) for l in got] == [(l[3], l[4], l[2]) for l in exp]
                ce = oracle.sht_to_cartesian(W, H, [(float(np.float32(l[0])), float(np.float32(l[1]))) for l in exp])
                assert (cart[f][:len(exp)].view(np.uint32) == ce.view(np.uint32)).all(), (asynchronous, f)
        # the sample's line cut (HOUGH_MAXLINES = 20, samples/hough_lines/main.cxx) with the Cartesian output: d_counts keeps the UNCUT count, only
        # the first 20 slots of d_lines are decoded -- toCartesian must stop there too (slots past the cut hold whatever the buffer held)
        d_lines = torch.full((F, cap, 5), 0x7f7f7f7f, dtype=torch.int32, device=dev)
        d_cart = torch.full((F, cap, 4), float("nan"), dtype=torch.float32, device=dev)
        d_counts.zero_()
        plan.pipeline_ex(d_rgb.data_ptr(), 0.5, 1.0, 60, 20, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr(),
                         threshold_type=capi.THRESHOLD_OTSU, pixfmt=capi.FMT_RGB24, d_cart=d_cart.data_ptr())
        torch.cuda.synchronize()
        counts = d_counts.cpu().numpy(); cart = d_cart.cpu().numpy()
        raw = d_lines.cpu().numpy().view(np.uint8).reshape(F, cap, 20)
        for f in range(F):
            exp = lines_exp[f]
            assert counts[f] == len(exp) > 20
            got = np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:20]
            # the cut keeps the 20 strongest; lines of equal strength at the cut are an arbitrary choice of the reference's unstable sort: compare strengths
            assert [int(l["strength"]) for l in got] == [l[2] for l in exp[:20]]
            ce = oracle.sht_to_cartesian(W, H, [(float(l["rho"]), float(l["theta"])) for l in got])
            assert (cart[f][:20].view(np.uint32) == ce.view(np.uint32)).all(), f
            assert np.isnan(cart[f][20:]).all(), f                  # nothing converted past the cut
        # kernel size 5 + PERCENT_OF_MEAN thresholds on the luma planes, no optional outputs (the plan keeps its own scratch)
        d_y = torch.from_numpy(np.stack(grays)).to(dev)
        d_edges = torch.zeros_like(d_y); d_lines.zero_(); d_counts.zero_()
        plan.pipeline_ex(d_y.data_ptr(), 0.68, 1.36, 40, 0, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr(), ksize=5,
                         threshold_type=capi.THRESHOLD_PERCENT_OF_MEAN)
        torch.cuda.synchronize()
        e = d_edges.cpu().numpy(); counts = d_counts.cpu().numpy()
        for f in range(F):
            rc, exp_e = oracle.canny(grays[f], 0.68, 1.36, 5, 1)
            assert rc == 0 and (e[f] == exp_e).all(), f
            assert counts[f] == len(oracle.sht(exp_e, 1.0, 40))
        with pytest.raises(capi.CompvHipError) as err:
            plan.pipeline_ex(d_y.data_ptr(), 10.0, 20.0, 40, 0, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr(), pixfmt=99)
        assert err.value.code == capi.E_NOT_IMPLEMENTED
    finally:
        plan.close()
