#!/usr/bin/env python
"""do the LDS-atomic-bound Hough vote stage and the VALU-bound Canny stage gain anything from sharing CUs?
Two plans on two HIP streams, driven by two host threads: lane A only runs the SHT stage (compaction, VOTE, reduce, line stage) on fixed edge maps,
lane B only runs the Canny stage (tile kernel + hysteresis) on fixed frames.  Each is timed alone, then both run together for a fixed wall time.
    co-run gain = (iterations_A * t_A_alone + iterations_B * t_B_alone) / wall        1.0 = the stages take turns, 2.0 = they overlap completely
The vote window is set by COMPVHIP_VOTE_MAX_WINDOW (the LDS a voting workgroup leaves free decides how many Canny waves fit beside it); run once per value.
"""
import os, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from compv_amd import capi
from bench import FrameSynth

W, H, F = 3840, 2160, 32
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
synth = FrameSynth(torch, dev, W, H)
frames = synth.batch([12345 + f for f in range(F)])
del synth
cap = 1 << 16


def lane():
    plan = capi.Plan(ctx, W, H, W, F, 1.0)
    st = torch.cuda.Stream(device=dev)
    return {"plan": plan, "st": st, "edges": torch.empty_like(frames), "lines": torch.zeros((F, cap, 5), dtype=torch.int32, device=dev),
            "counts": torch.zeros(F, dtype=torch.int32, device=dev)}


A, B = lane(), lane()
A["plan"].canny(frames.data_ptr(), 59.0, 119.0, A["edges"].data_ptr(), stream=A["st"].cuda_stream)
torch.cuda.synchronize()
nx, ny, rw = capi.houghsht_vote_grid(W, H, 1.0, F) if hasattr(capi, "houghsht_vote_grid") else (0, 0, 0)


def run_a(n):
    for _ in range(n):
        A["plan"].houghsht(A["edges"].data_ptr(), 100, 0, A["lines"].data_ptr(), cap, A["counts"].data_ptr(), A["st"].cuda_stream)
    A["st"].synchronize()


def run_b(n):
    for _ in range(n):
        B["plan"].canny(frames.data_ptr(), 59.0, 119.0, B["edges"].data_ptr(), stream=B["st"].cuda_stream)
    B["st"].synchronize()


def alone(fn, n=40):
    fn(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


tA, tB = alone(run_a), alone(run_b)
stop = [False]
cnt = {"a": 0, "b": 0}


def loop(fn, key):
    while not stop[0]:
        fn(1); cnt[key] += 1


run_a(2); run_b(2); torch.cuda.synchronize()
ths = [threading.Thread(target=loop, args=(run_a, "a")), threading.Thread(target=loop, args=(run_b, "b"))]
t0 = time.perf_counter()
for t in ths: t.start()
time.sleep(0.5)
stop[0] = True
for t in ths: t.join()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
gain = (cnt["a"] * tA + cnt["b"] * tB) / wall
ref_counts = A["counts"].cpu().numpy().copy()
print("window cap %s (grid %dx%d, %d rows = %.1f KB LDS, %.1f KB free): SHT stage alone %.4f ms, Canny stage alone %.4f ms; together %.3f s: %d SHT + %d Canny iterations "
      "-> co-run gain %.3f (SHT at %.2f, Canny at %.2f of their alone rates); lines of frame 0: %d"
      % (os.environ.get("COMPVHIP_VOTE_MAX_WINDOW", "default"), nx, ny, rw, (rw * 128 + 16) / 1024.0, 160 - (rw * 128 + 16) / 1024.0, tA * 1e3, tB * 1e3, wall, cnt["a"], cnt["b"],
         gain, cnt["a"] * tA / wall, cnt["b"] * tB / wall, int(ref_counts[0])))
