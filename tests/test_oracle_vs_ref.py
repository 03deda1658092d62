"""CPU tests (only where oracle/_ref exists): the C restatement against the REAL CompV library, beyond the committed
fixtures -- random images, ragged sizes, the accumulator through the reference's own SIMD voting leaves."""
import numpy as np
import pytest

from oracle_bindings import synth_frame


def _canon_ref(lines):
    return sorted([(-s, -rho, th) for (rho, th, s) in lines])


def _canon_orc(lines):
    return sorted([(-s, -rho, th) for (rho, th, s, row, col) in lines])


@pytest.mark.parametrize("W,H", [(64, 64), (97, 33), (129, 130), (257, 65), (640, 480), (17, 9), (16, 16), (9, 9), (33, 200)])
def test_sobel_canny_random_images(oracle, refshim, W, H):
    rng = np.random.default_rng(W * 1000 + H)
    for kind in range(3):
        if kind == 0:
            img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        elif kind == 1:
            img = synth_frame(W, H, 4242 + kind)
        else:  # smooth gradients + a few blobs: long weak chains
            y, x = np.mgrid[0:H, 0:W]
            img = ((np.sin(x / 7.0) + np.cos(y / 5.0)) * 40 + 128 + rng.integers(0, 6, (H, W))).astype(np.uint8)
        so, _ = oracle.edge_dete(img)
        assert (so == refshim.sobel(img)).all()
        for (tl, th) in [(59.0, 119.0), (0.8, 1.6), (20.0, 300.0)]:
            rc, a = oracle.canny(img, tl, th)
            rc2, b = refshim.canny(img, tl, th)
            assert rc == 0 and rc2 == 0
            assert (a == b).all(), (W, H, kind, tl, th, int((a != b).sum()))


def test_canny_mean_mode(oracle, refshim):
    img = synth_frame(640, 480)
    rc, a = oracle.canny(img, 0.68, 1.36, 3, 1)
    rc2, b = refshim.canny(img, 0.68, 1.36, 3, 1)
    assert rc == 0 and rc2 == 0 and (a == b).all()


@pytest.mark.parametrize("W,H,tl,th,thr,deg", [(320, 240, 59., 119., 10, 1.3), (333, 77, 0.8, 1.6, 20, 1.0), (640, 480, 59., 119., 30, 2.0)])
def test_sht_acc_and_lines(oracle, refshim, W, H, tl, th, thr, deg):
    img = synth_frame(W, H)
    rc, e = refshim.canny(img, tl, th)
    R, T, _ = oracle.sht_dims(W, H, deg)
    s, c = oracle.sht_tables(deg, T)
    acc = oracle.sht_acc(e, deg)
    assert (acc == refshim.sht_acc(e, s, c, R)).all()     # the reference's own SSE4.1/AVX2 voting leaves
    assert _canon_orc(oracle.sht_lines_from_acc(acc, W, H, deg, thr)) == _canon_ref(refshim.sht(e, deg, thr))


@pytest.mark.parametrize("W,H,tl,th,thr,deg,maxl", [(640, 480, 59., 119., 30, 1.0, 0), (640, 480, 59., 119., 30, 1.0, 100), (1282, 720, 0.8, 1.6, 100, 1.0, 0),
                                                     (333, 77, 0.8, 1.6, 5, 0.5, 40), (1280, 720, 59., 119., 60, 0.5, 0)])
def test_sht_line_order_is_the_references_order(oracle, refshim, W, H, tl, th, thr, deg, maxl):
    """The reference orders its lines with an unstable std::sort on the strength alone (houghsht.cxx:241-249): the order inside
    equal-strength groups -- and which of them survive maxLines -- is what libstdc++'s introsort makes of the (row, col) emission
    order.  orc_sht_reference_order (oracle/kht_sort.cpp) must reproduce the compiled reference's list element by element."""
    img = synth_frame(W, H, 77)
    rc, edges = oracle.canny(img, tl, th)
    got = oracle.sht(edges, deg, thr, maxl, reference_order=True)
    exp = refshim.sht(edges, deg, thr, maxl)
    assert len(exp) > 20 and len({l[2] for l in exp}) < len(exp)                 # there ARE equal-strength groups
    assert [(np.float32(l[0]), np.float32(l[1]), int(l[2])) for l in got] == [(np.float32(l[0]), np.float32(l[1]), int(l[2])) for l in exp]


@pytest.mark.parametrize("W,H", [(64, 64), (129, 130), (640, 480), (641, 333), (20, 20), (9, 9)])
def test_canny_5x5_sobel(oracle, refshim, W, H):
    """COMPV_CANNY_SET_INT_KERNEL_SIZE = 5 (kernels compv_features.h:129-130)."""
    rng = np.random.default_rng(W + H)
    for img in (synth_frame(W, H, 5), rng.integers(0, 256, (H, W), dtype=np.uint8)):
        for (tl, th) in [(400.0, 900.0), (0.8, 1.6), (2000.0, 6000.0)]:
            rc, a = oracle.canny(img, tl, th, 5)
            rc2, b = refshim.canny(img, tl, th, 5)
            assert rc == 0 and rc2 == 0
            assert (a == b).all(), (W, H, tl, th, int((a != b).sum()))


def test_sht_to_cartesian_matches_reference(oracle, refshim):
    """CompVHoughSht::toCartesian (houghsht.cxx:566-589) incl. the theta == 0 special case."""
    import math
    step = np.float32(math.pi / 180.0)
    lines = [(float(rho), float(np.float32(col) * step)) for col in range(0, 180, 7) for rho in (-1500.0, -3.0, 0.0, 17.0, 2400.0)]
    for (W, H) in ((640, 480), (3840, 2160)):
        exp = refshim.sht_to_cartesian(W, H, lines)
        got = oracle.sht_to_cartesian(W, H, lines)
        assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist()
