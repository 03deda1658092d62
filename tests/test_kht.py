"""CPU tests of the KHT oracle (oracle/kht_oracle.c + kht_sort.cpp): GS to the last bit and the complete line set in the
reference's order, against fixtures generated from the compiled reference and (where oracle/_ref exists) against the
reference itself on further inputs."""
import numpy as np
import pytest

from oracle_bindings import synth_frame

KHT = ["small_320x240", "q3_641x480", "ragged_333x77", "hd_1280x720", "fhd_1920x1080", "dense_1282x720", "uhd_3840x2160"]


@pytest.mark.parametrize("name", KHT)
def test_kht_golden(oracle, golden, name):
    meta, arrays = golden
    m = meta[name]
    img = synth_frame(m["W"], m["H"], m["seed"])
    rc, can = oracle.canny(img, m["tLow"], m["tHigh"], 3, m["threshold_type"])
    lines, gs = oracle.kht(can, 1.0, 1.0, 1)
    k = m["kht"]
    assert repr(gs) == k["gs"]                                   # COMPV_HOUGHKHT_GET_FLT64_GS, all 17 digits
    assert len(lines) == k["lines"]
    exp = arrays[name + "/kht_lines"]
    got = np.array([(l[0], l[1], l[2]) for l in lines], np.float64).reshape(-1, 3)
    assert (got == exp).all()                                    # rho, theta (f32), strength AND order


@pytest.mark.parametrize("W,H,tl,th,rho,deg,thr", [(640, 480, 59., 119., 1.0, 1.0, 1), (640, 480, 59., 119., 0.5, 1.0, 1), (480, 360, 59., 119., 1.0, 0.5, 150),
                                                     (257, 129, 0.8, 1.6, 1.0, 2.0, 1), (97, 64, 20., 60., 1.0, 1.0, 1)])
def test_kht_vs_reference(oracle, refshim, W, H, tl, th, rho, deg, thr):
    img = synth_frame(W, H, 99)
    rc, e = refshim.canny(img, tl, th)
    lo, gso = oracle.kht(e, rho, deg, thr)
    lr, gsr = refshim.kht(e, rho, deg, thr)
    assert gso == gsr
    assert [(l[0], l[1], l[2]) for l in lo] == [(l[0], l[1], l[2]) for l in lr]


def test_kht_empty_and_maxlines(oracle):
    e = np.zeros((64, 64), np.uint8)
    lines, gs = oracle.kht(e)
    assert lines == [] and gs == 1.0
    img = synth_frame(320, 240)
    rc, can = oracle.canny(img, 59., 119.)
    full, _ = oracle.kht(can)
    top, _ = oracle.kht(can, max_lines=5)
    assert top == full[:5]


@pytest.mark.parametrize("W,H,tl,th,min_size", [(320, 240, 59., 119., 10), (641, 480, 59., 119., 5), (257, 129, 0.8, 1.6, 3), (64, 64, 59., 119., 10),
                                                 (130, 70, 59., 119., 2), (1282, 720, 0.8, 1.6, 10), (1920, 1080, 59., 119., 10)])
def test_bit_plane_linker_matches_the_restated_byte_walk(oracle, W, H, tl, th, min_size):
    """compvhip_houghkht_link_u8 -- the product's host stage of KHT: edge map as a bit plane, 8-bit neighbour code in the reference's priority order +
    count-trailing-zeros, run following, word-wise seed scan (compv_amd/csrc/kht_host.cpp) -- against the oracle's restatement of the reference's byte
    walk (linking_AppendixA, houghkht.cxx:544-760): the same strings with the same points in the same order.  No device involved."""
    from compv_amd import capi
    img = synth_frame(W, H, 777)
    rc, edges = oracle.canny(img, tl, th)
    assert rc == 0
    for e in (edges, np.ascontiguousarray(edges[:, ::-1]), np.ascontiguousarray(edges.T)):   # + mirrored and transposed maps (other walk directions)
        exp_xy, exp_ends = oracle.kht_link(e, min_size)
        got_xy, got_ends = capi.houghkht_link(e, min_size)
        assert np.array_equal(got_ends, exp_ends)
        assert np.array_equal(got_xy, exp_xy)


def test_bit_plane_linker_borders_and_word_boundaries(oracle):
    """Pixels on the image border (never seeds, but reachable by a walk), runs across 64-bit word boundaries, isolated pixels, full rows."""
    from compv_amd import capi
    rng = np.random.RandomState(5)
    for W, H in ((63, 9), (64, 9), (65, 9), (128, 5), (129, 33), (200, 3), (3, 200), (32767, 6), (5, 32767)):   # ... and the largest coordinates the API takes
        e = np.zeros((H, W), np.uint8)
        e[H // 2, :] = 255                       # a full row: leftward and rightward runs over every word boundary
        e[:, W // 2] = 255                       # a full column
        e[0, :] = 255; e[:, 0] = 255; e[H - 1, ::2] = 255; e[::2, W - 1] = 255     # the border
        e |= (rng.rand(H, W) < 0.08).astype(np.uint8) * 255
        for ms in (1, 2, 10):
            exp_xy, exp_ends = oracle.kht_link(e, ms)
            got_xy, got_ends = capi.houghkht_link(e, ms)
            assert np.array_equal(got_ends, exp_ends), (W, H, ms)
            assert np.array_equal(got_xy, exp_xy), (W, H, ms)


def test_bit_plane_linker_long_runs_switch_the_shortcut_on(oracle):
    """Line art: the linker's horizontal-run shortcut is off until 32 horizontal single steps in a row were walked and stays on while its runs are long.  Long
    stripes with gaps, pixels in the row above that end a run (a pixel a run would leave has a neighbour above: the reference turns upwards there), stripes two
    rows apart, bars crossing them, noise that switches the shortcut off again -- walked rightwards (forward walks) and leftwards (backward walks, mirrored map)."""
    from compv_amd import capi
    rng = np.random.RandomState(11)
    for W, H, noise in ((700, 64, 0.0), (1000, 90, 0.002), (513, 70, 0.01), (2000, 40, 0.0005)):
        e = np.zeros((H, W), np.uint8)
        for y in range(3, H - 3, 5):
            e[y, 2:W - 2] = 255
            for g in rng.randint(3, W - 3, size=W // 150):           # gaps
                e[y, g:g + rng.randint(1, 4)] = 0
            for g in rng.randint(3, W - 3, size=W // 100):           # blockers in the row above
                e[y - 1, g] = 255
        e[6:H - 6:10, 5:W - 5:3] = 255                              # dotted rows two above a stripe
        e[2:H - 2, W // 3] = 255; e[2:H - 2, (2 * W) // 3 + 1] = 255   # bars
        e |= (rng.rand(H, W) < noise).astype(np.uint8) * 255
        for m in (e, np.ascontiguousarray(e[:, ::-1]), np.ascontiguousarray(e[::-1, :])):
            for ms in (2, 10):
                exp_xy, exp_ends = oracle.kht_link(m, ms)
                got_xy, got_ends = capi.houghkht_link(m, ms)
                assert np.array_equal(got_ends, exp_ends), (W, H, noise, ms)
                assert np.array_equal(got_xy, exp_xy), (W, H, noise, ms)

