#!/usr/bin/env python
"""Experiment (not part of the product): does running two half-batches on two HIP streams from two host threads overlap the
VALU-bound Canny of one half with the LDS-bound Hough voting of the other?  Prints aggregate Mpixels/s for 1 x 32 frames and for
2 x 16 frames (two plans, two streams, two threads; ctypes releases the GIL during the pipeline call)."""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from compv_amd import capi
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from oracle_bindings import synth_frame

W, H = 3840, 2160
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
frames = np.stack([synth_frame(W, H, 12345 + f) for f in range(32)])


def make(n, off):
    d_in = torch.from_numpy(frames[off:off + n]).to(dev)
    d_edges = torch.empty_like(d_in)
    cap = 1 << 16
    d_lines = torch.zeros((n, cap, 5), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
    plan = capi.Plan(ctx, W, H, W, n, 1.0)
    st = torch.cuda.Stream(device=dev)
    return plan, d_in, d_edges, d_lines, d_counts, cap, st


def run(objs, steps):
    plan, d_in, d_edges, d_lines, d_counts, cap, st = objs
    for _ in range(steps):
        plan.pipeline(d_in.data_ptr(), 59.0, 119.0, 100, 0, d_edges.data_ptr(), d_lines.data_ptr(), cap, d_counts.data_ptr(), st.cuda_stream)


one = make(32, 0)
run(one, 3); torch.cuda.synchronize()
t0 = time.perf_counter(); run(one, 20); torch.cuda.synchronize(); t1 = time.perf_counter()
print("1 x 32 frames: %.1f Mpixels/s (%.3f ms per 32 frames)" % (32 * W * H * 20 / (t1 - t0) / 1e6, (t1 - t0) / 20 * 1e3))
a, b = make(16, 0), make(16, 16)
run(a, 3); run(b, 3); torch.cuda.synchronize()
t0 = time.perf_counter()
ta = threading.Thread(target=run, args=(a, 20)); tb = threading.Thread(target=run, args=(b, 20))
ta.start(); tb.start(); ta.join(); tb.join(); torch.cuda.synchronize()
t1 = time.perf_counter()
print("2 x 16 frames, two streams/threads: %.1f Mpixels/s (%.3f ms per 32 frames)" % (32 * W * H * 20 / (t1 - t0) / 1e6, (t1 - t0) / 20 * 1e3))
