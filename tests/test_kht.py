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
