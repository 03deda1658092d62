"""GPU debugging aid: Hough accumulator of a few isolated edge pixels vs the oracle, per theta."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_bindings import *
from compv_amd import capi
o = Oracle(); ctx = capi.Context(0)
for (W, H, pts) in [(320, 240, [(100, 50)]), (320, 240, [(0, 0)]), (320, 240, [(319, 239)]), (320, 240, [(33, 7), (200, 100), (201, 100), (5, 239)]), (64, 64, [(10, 20)])]:
    e = np.zeros((H, W), np.uint8)
    for (x, y) in pts: e[y, x] = 255
    lines, acc = ctx.houghsht(e, 1.0, 1000, want_acc=True)
    ea = o.sht_acc(e, 1.0)
    d = acc != ea
    print(W, H, pts, 'diff cells', int(d.sum()), 'sum', int(acc.sum()), int(ea.sum()))
    if d.any():
        for t in [0, 1, 31, 32, 33, 63, 64, 65, 90, 127, 128, 179]:
            got = np.nonzero(acc[:, t])[0]; exp = np.nonzero(ea[:, t])[0]
            print('   theta', t, 'got rows', got[:6], 'exp rows', exp[:6])
