"""First-contact GPU debugging aid: prints HIP-vs-oracle differences instead of asserting."""
import sys, os, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_bindings import *
from compv_amd import capi
o = Oracle(); ctx = capi.Context(0)
for (W, H) in [(64, 64), (320, 240), (641, 480), (1282, 720)]:
    img = synth_frame(W, H)
    try:
        s = ctx.edge_dete(img); es, _ = o.edge_dete(img); print(W, H, 'sobel diff', int((s != es).sum()), flush=True)
    except Exception as e:
        print('sobel err', e)
    try:
        c = ctx.canny(img, 59., 119.); rc, ec = o.canny(img, 59., 119.); d = (c != ec)
        print(W, H, 'canny diff', int(d.sum()), 'edges', int((ec != 0).sum()), 'got', int((c != 0).sum()), flush=True)
        if d.any():
            ys, xs = np.nonzero(d); print(' first diffs', list(zip(ys[:10], xs[:10])), 'vals', c[ys[:10], xs[:10]], ec[ys[:10], xs[:10]])
    except Exception as e:
        print('canny err', e)
    try:
        rc, ec = o.canny(img, 59., 119.)
        lines, acc = ctx.houghsht(ec, 1.0, 50, want_acc=True); ea = o.sht_acc(ec, 1.0)
        print(W, H, 'acc diff', int((acc != ea).sum()), 'sum', acc.sum(), ea.sum(), 'lines', len(lines), len(o.sht_lines_from_acc(ea, W, H, 1.0, 50)), flush=True)
    except Exception as e:
        print('sht err', e)
