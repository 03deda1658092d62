"""GPU debugging aid: repeats blur -> Canny on a strided batch (tests/test_gpu_gauss.py::test_plan_blur_then_canny_batch) and prints where the maps differ."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle_bindings import *
from compv_amd import capi
o = Oracle(); ctx = capi.Context(0)
dev = torch.device("cuda", 0)
W, H, S, F = 1282, 720, 1344, 4
frames = np.zeros((F, H, S), np.uint8)
kern = capi.gauss_kernel_fixedpoint(5, 1.0)
exp = []
for f in range(F):
    frames[f, :, :W] = synth_frame(W, H, 4321 + f)
    rc, b = o.convlt_fxp(frames[f][:, :W], kern, kern)
    rc2, e = o.canny(np.ascontiguousarray(b), 30.0, 70.0)
    exp.append((b, e))
for rep in range(12):
    d = torch.from_numpy(frames).to(dev); de = torch.empty_like(d)
    plan = capi.Plan(ctx, W, H, S, F, 1.0)
    plan.convlt_fixedpoint(d.data_ptr(), kern, kern, d.data_ptr())
    plan.canny(d.data_ptr(), 30.0, 70.0, de.data_ptr())
    torch.cuda.synchronize()
    b = d.cpu().numpy(); e = de.cpu().numpy()
    for f in range(F):
        db = int((b[f][:, :W] != exp[f][0]).sum())
        dd = e[f][:, :W] != exp[f][1]
        if db or dd.any():
            ys, xs = np.nonzero(dd)
            print('rep', rep, 'frame', f, 'blur diff', db, 'canny diff', int(dd.sum()), 'rows', ys.min() if len(ys) else None, ys.max() if len(ys) else None,
                  'cols', xs.min() if len(xs) else None, xs.max() if len(xs) else None, list(zip(ys[:6], xs[:6])), 'got', e[f][ys[:6], xs[:6]], flush=True)
    plan.close()
print('done')
