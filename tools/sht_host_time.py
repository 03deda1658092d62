import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from compv_amd import capi
from oracle_bindings import synth_frame
W, H = 3840, 2160
img = synth_frame(W, H, 12345)
ctx = capi.Context(0)
edges = ctx.canny(img, 59.0, 119.0)
def best(fn, n=5):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r
for thr in (100, 300, 1000, 3000):
    ms, lines = best(lambda: ctx.houghsht(edges, 1.0, thr))
    print("threshold", thr, "lines", len(lines), "ms %.3f" % ms)
ms, e = best(lambda: ctx.canny(img, 59.0, 119.0)); print("canny ms %.3f" % ms)
z = np.zeros_like(edges)
ms, lines = best(lambda: ctx.houghsht(z, 1.0, 100)); print("empty map: lines", len(lines), "ms %.3f" % ms)
