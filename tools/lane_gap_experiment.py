#!/usr/bin/env python
"""Experiment (not part of the product): where is the GPU idle between the asynchronous steps of bench.py's two lanes?

Two plans on two HIP streams take steps in turn exactly like bench.py (depth 2).  A timing event is recorded on the lane's
stream in front of and behind every step; the script prints, per lane, the step durations and the gaps between the end of a
step and the start of the lane's next one (device time), and the wall-clock time per step.

    python tools/lane_gap_experiment.py [--steps 24] [--lanes 2] [--side-flag]
"""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from compv_amd import capi, sharding   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--lanes", type=int, default=2)
ap.add_argument("--depth", type=int, default=2)
args = ap.parse_args()

W, H, F, NB = 3840, 2160, 32, 4
dev = torch.device("cuda", 0)
synth = bench.FrameSynth(torch, dev, W, H)
blocks = [synth.batch([sharding.frame_seed(b * F + f) for f in range(F)]) for b in range(NB)]
ctx = capi.Context(0)
cap = 1 << 16
lanes = []
for _ in range(args.lanes):
    lanes.append({"plan": capi.Plan(ctx, W, H, W, F, 1.0), "edges": torch.empty_like(blocks[0]),
                  "lines": torch.zeros((F, cap, 5), dtype=torch.int32, device=dev), "counts": torch.zeros(F, dtype=torch.int32, device=dev),
                  "stream": torch.cuda.Stream(device=dev), "ev": []})
torch.cuda.synchronize()


def enqueue(q, k, record):
    if record:
        a = torch.cuda.Event(enable_timing=True); a.record(q["stream"])
    t = q["plan"].pipeline_async(blocks[k % NB].data_ptr(), 59.0, 119.0, 100, 0, q["edges"].data_ptr(), q["lines"].data_ptr(), cap,
                                 q["counts"].data_ptr(), q["stream"].cuda_stream)
    if record:
        b = torch.cuda.Event(enable_timing=True); b.record(q["stream"])
        q["ev"].append((a, b))
    return t


def run(n, record):
    pend = []
    for k in range(n):
        q = lanes[k % len(lanes)]
        pend.append((q, enqueue(q, k, record)))
        if len(pend) > args.depth * len(lanes):
            q0, t0 = pend.pop(0)
            q0["plan"].wait(t0)
    for q0, t0 in pend:
        q0["plan"].wait(t0)


run(6, False)
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True); base.record(lanes[0]["stream"])
torch.cuda.synchronize()
t0 = time.perf_counter()
run(args.steps, True)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("wall clock: %.3f ms per step (%d lanes, depth %d, events on)" % ((t1 - t0) / args.steps * 1e3, len(lanes), args.depth))
for i, q in enumerate(lanes):
    ts = [(base.elapsed_time(a), base.elapsed_time(b)) for a, b in q["ev"]]
    dur = [b - a for a, b in ts]
    gap = [ts[j + 1][0] - ts[j][1] for j in range(len(ts) - 1)]
    print("lane %d: step duration median %.3f ms (min %.3f max %.3f); gap end->next start median %.3f ms (min %.3f max %.3f)" %
          (i, float(np.median(dur)), min(dur), max(dur), float(np.median(gap)), min(gap), max(gap)))
t0 = time.perf_counter()
run(args.steps, False)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("wall clock: %.3f ms per step (events off)" % ((t1 - t0) / args.steps * 1e3))
