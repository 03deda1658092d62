"""Sobel / Scharr / Prewitt detector on 32 resident 4K frames: time per call (HIP events around compvhip_plan_edge_dete = both passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from compv_amd import capi
from oracle_bindings import synth_frame
W, H, F = 3840, 2160, 32
dev = torch.device("cuda", 0)
frames = torch.from_numpy(np.stack([synth_frame(W, H, 12345 + f) for f in range(4)])).to(dev).repeat(F // 4, 1, 1).contiguous()
out = torch.empty_like(frames)
ctx = capi.Context(0); plan = capi.Plan(ctx, W, H, W, F, 1.0)
st = torch.cuda.Stream(device=dev)
for name, op in (("sobel", capi.OP_SOBEL), ("scharr", capi.OP_SCHARR), ("prewitt", capi.OP_PREWITT)):
    for _ in range(3): plan.edge_dete(frames.data_ptr(), op, out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st); plan.edge_dete(frames.data_ptr(), op, out.data_ptr(), st.cuda_stream); e1.record(st); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print("%-8s %.4f ms per %d x %dx%d frames = %.0f Mpixels/s, %.2f TB/s of the 3 B/px (2 reads + 1 write) = %.3f of 8 TB/s" % (name, ms, F, W, H, F * W * H / ms / 1e3, 3.0 * F * W * H / ms / 1e9, 3.0 * F * W * H / ms / 1e9 / 8.0))
plan.close(); ctx.close()
