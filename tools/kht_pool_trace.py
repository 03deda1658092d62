#!/usr/bin/env python
"""Lab: who does what when in one compvhip_plan_houghkht call (32 x 4K frames) -- the pool's item spans (COMPVHIP_KHT_TRACE) as a per-worker text timeline.
usage: kht_pool_trace.py [threads]   stage tags: P bit plane, L link, K kernels / prune, S sort + sweep"""
import os, sys, subprocess, re
if os.environ.get("_KHT_CHILD"):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch, bench
    from compv_amd import capi
    W, H, F = 3840, 2160, 32
    dev = torch.device("cuda:0")
    synth = bench.FrameSynth(torch, dev, W, H)
    d_in = synth.batch([12345 + f for f in range(F)])
    d_e = torch.empty_like(d_in)
    ctx = capi.Context(0); plan = capi.Plan(ctx, W, H, W, F, 1.0)
    plan.canny(d_in.data_ptr(), 59.0, 119.0, d_e.data_ptr()); torch.cuda.synchronize()
    th = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for _ in range(3):
        plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 1, threads=th)
    os.environ["COMPVHIP_KHT_TRACE"] = "1"
    plan.houghkht(d_e.data_ptr(), 1.0, 1.0, 1, threads=th)
    print("WALL %.3f" % plan.houghkht_stage_ms()["wall_ms"], file=sys.stderr)
    sys.exit(0)
r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, _KHT_CHILD="1"), stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
spans = [(int(m.group(1)), m.group(2), float(m.group(3)), float(m.group(4))) for m in re.finditer(r"khtpool w(\d+) (\S) +([\d.]+) +([\d.]+)", r.stderr)]
wall = re.search(r"WALL ([\d.]+)", r.stderr)
print("wall", wall.group(1) if wall else "?", "ms;", len(spans), "items")
if not spans:
    print(r.stderr[-2000:]); sys.exit(1)
end = max(s[3] for s in spans); cols = 120
for w in sorted({s[0] for s in spans}):
    row = [" "] * cols
    busy = 0.0
    for ww, tag, t0, t1 in spans:
        if ww != w: continue
        busy += t1 - t0
        for c in range(int(t0 / end * cols), min(cols, int(t1 / end * cols) + 1)): row[c] = tag
    print("w%02d |%s| busy %.1f ms" % (w, "".join(row), busy))
print("scale: %d columns = %.2f ms" % (cols, end))
for tag in "PLKS":
    d = [t1 - t0 for _, tg, t0, t1 in spans if tg == tag]
    if d: print(tag, "items %d  mean %.3f ms  sum %.1f ms" % (len(d), sum(d) / len(d), sum(d)))
