#!/usr/bin/env python
"""Interleaved A/B of vote tile grids inside ONE process (box-to-box and run-to-run noise is larger than the effect): for every value of
COMPVHIP_VOTE_MAX_WINDOW two lanes (plans made under that value), then rounds of K asynchronous two-lane steps per variant, variants taken in turn.
Prints the median / min step time per variant.  usage: grid_ab.py "<caps>" [W H [rounds [steps]]]   (cap 0 = the product's own choice)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from compv_amd import capi
from bench import FrameSynth

caps = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0 1072").split()]
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 12
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 128
F, NB = 32, 4
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
synth = FrameSynth(torch, dev, W, H)
blocks = [synth.batch([12345 + b * F + f for f in range(F)]) for b in range(NB)]
del synth
cap_lines = 1 << 16


def lane():
    return {"plan": capi.Plan(ctx, W, H, W, F, 1.0), "st": torch.cuda.Stream(device=dev), "edges": torch.empty_like(blocks[0]),
            "lines": torch.zeros((F, cap_lines, 5), dtype=torch.int32, device=dev), "counts": torch.zeros(F, dtype=torch.int32, device=dev)}


variants = {}
for c in caps:
    if c:
        os.environ["COMPVHIP_VOTE_MAX_WINDOW"] = str(c)
    else:
        os.environ.pop("COMPVHIP_VOTE_MAX_WINDOW", None)
    variants[c] = {"lanes": [lane(), lane()], "grid": capi.houghsht_vote_grid(W, H, 1.0, F), "t": []}
os.environ.pop("COMPVHIP_VOTE_MAX_WINDOW", None)


def run(v, n):
    lanes = v["lanes"]
    pend = []
    for k in range(n):
        q = lanes[k % 2]
        t = q["plan"].pipeline_async(blocks[k % NB].data_ptr(), 59.0, 119.0, 100, 0, q["edges"].data_ptr(), q["lines"].data_ptr(), cap_lines, q["counts"].data_ptr(), q["st"].cuda_stream)
        pend.append((q, t))
        if len(pend) > 3:
            qq, tt = pend.pop(0); qq["plan"].wait(tt)
    for qq, tt in pend:
        qq["plan"].wait(tt)
    torch.cuda.synchronize()


ref = None
for c, v in variants.items():
    run(v, 8)
    cnt = v["lanes"][(8 - 1) % 2]["counts"].cpu().numpy().copy()
    if ref is None: ref = cnt
    assert (cnt == ref).all(), "line counts differ between grids"
for r in range(rounds):
    for c, v in variants.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run(v, steps); v["t"].append((time.perf_counter() - t0) / steps * 1e3)
for c, v in variants.items():
    t = sorted(v["t"])
    print("cap %5d grid %dx%d window %4d rows: step median %.4f ms  min %.4f  max %.4f  (%d rounds of %d steps, %dx%d x %d frames)"
          % ((c,) + tuple(v["grid"]) + (t[len(t) // 2], t[0], t[-1], rounds, steps, W, H, F)))
