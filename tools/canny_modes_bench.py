"""Canny with the 5x5 Sobel (compvhip_plan_canny, ksize = 5) and the Sobel / Scharr / Prewitt detector on 32 resident 4K frames: per-kernel HIP-event times
of the plan's timing API (what bench.py's kernels_extra object reports)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from compv_amd import capi
W, H, F = 3840, 2160, 32
dev = torch.device("cuda:0")
synth = bench.FrameSynth(torch, dev, W, H)
d_in = synth.batch([12345 + f for f in range(F)])
d_e = torch.empty_like(d_in)
ctx = capi.Context(0); plan = capi.Plan(ctx, W, H, W, F, 1.0)
st = torch.cuda.Stream(device=dev)
out = {}
# the 5x5 Sobel answers a step edge 12 x as strongly as the 3x3 one (16 * 3 against 4 * 1): (708, 1428) are the benchmark's thresholds at the same edge density
for ks, tl, th in ((3, 59.0, 119.0), (5, 59.0, 119.0), (5, 708.0, 1428.0)):
    for _ in range(3):
        plan.canny(d_in.data_ptr(), tl, th, d_e.data_ptr(), ksize=ks, stream=st.cuda_stream)
    torch.cuda.synchronize()
    plan.set_timing(1)
    acc = {}
    n = 10
    for _ in range(n):
        plan.canny(d_in.data_ptr(), tl, th, d_e.data_ptr(), ksize=ks, stream=st.cuda_stream)
        torch.cuda.synchronize()
        for name, ms in plan.get_timing():
            acc[name] = acc.get(name, 0.0) + ms
    plan.set_timing(0)
    key = "canny_ksize%d_t%d_%d" % (ks, int(tl), int(th))
    out[key] = {k: round(v / n, 4) for k, v in acc.items()}
    out[key]["edge_px"] = int((d_e != 0).sum().item())
print(json.dumps(out))
