"""Latency of the host (drop-in) entry points: mean ms per call, including H2D/D2H and synchronisation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from compv_amd import capi
from oracle_bindings import synth_frame
ctx = capi.Context(0)
def t(fn, n=20):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
for (W, H) in [(1280, 720), (1920, 1080), (3840, 2160)]:
    img = synth_frame(W, H)
    e = ctx.canny(img, 59., 119.)
    print("%dx%d  sobel %.3f ms  canny %.3f ms  sht %.3f ms  kht %.3f ms" % (
        W, H, t(lambda: ctx.edge_dete(img)), t(lambda: ctx.canny(img, 59., 119.)), t(lambda: ctx.houghsht(e, 1.0, 100)), t(lambda: ctx.houghkht(e), 5)), flush=True)
