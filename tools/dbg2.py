"""GPU debugging aid: plan.canny on a strided batch vs the oracle, prints where the maps differ."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle_bindings import *
from compv_amd import capi
o = Oracle(); ctx = capi.Context(0)
dev = torch.device("cuda", 0)
for (W, H, S, F, tl, th) in [(1282, 720, 1344, 2, 30.0, 70.0), (1282, 720, 1288, 1, 30.0, 70.0), (1280, 720, 1280, 1, 30.0, 70.0), (1282, 720, 1344, 1, 59.0, 119.0)]:
    frames = np.zeros((F, H, S), np.uint8)
    for f in range(F):
        frames[f, :, :W] = synth_frame(W, H, 4321 + f)
    d = torch.from_numpy(frames).to(dev); de = torch.empty_like(d)
    plan = capi.Plan(ctx, W, H, S, F, 1.0)
    plan.set_timing(1)
    plan.canny(d.data_ptr(), tl, th, de.data_ptr())
    torch.cuda.synchronize()
    print(W, H, S, F, tl, th, [n for n, _ in plan.get_timing()])
    e = de.cpu().numpy()
    for f in range(F):
        rc, ex = o.canny(np.ascontiguousarray(frames[f][:, :W]), tl, th)
        dd = e[f][:, :W] != ex
        print('  frame', f, 'diff', int(dd.sum()), 'edges', int((ex != 0).sum()))
        if dd.any():
            ys, xs = np.nonzero(dd)
            print('   rows', ys.min(), ys.max(), 'cols', xs.min(), xs.max(), 'first', list(zip(ys[:8], xs[:8])), 'got', e[f][ys[:8], xs[:8]], 'exp', ex[ys[:8], xs[:8]])
            print('   col hist /240', np.bincount(xs // 240), 'row hist /64', np.bincount(ys // 64))
    plan.close()
