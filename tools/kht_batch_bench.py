"""Batched KHT (compvhip_plan_houghkht) on one 32-frame 4K batch: ms per frame against the number of host workers."""
import os, sys, time, json
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
plan.canny(d_in.data_ptr(), 59.0, 119.0, d_e.data_ptr()); torch.cuda.synchronize()
out = {}
for threads in [int(a) for a in sys.argv[1:]] or [4, 8, 16, 24, 32, 48]:
    plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 1, threads=threads)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 1, threads=threads); ts.append((time.perf_counter() - t0) * 1e3 / F)
    out[threads] = {"ms_per_frame": [round(t, 3) for t in ts], "stages": plan.houghkht_stage_ms()["stages"]}
    print(threads, out[threads], flush=True)
