"""Randomised parity sweep of the KHT host linker (no device): compvhip_houghkht_link_u8 against the oracle's restatement of the reference's byte walk on random
sizes and contents (noise of any density, stripes / bars / checkerboards with gaps and blockers, mixtures).   python tools/fuzz_linker.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from compv_amd import capi
from oracle_bindings import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = Oracle()
bad = 0
for k in range(cases):
    W = int(rng.choice([rng.randint(3, 70), rng.randint(60, 140), rng.randint(140, 700), rng.randint(700, 3000)]))
    H = int(rng.choice([rng.randint(3, 12), rng.randint(12, 80), rng.randint(80, 300)]))
    e = np.zeros((H, W), np.uint8)
    kinds = rng.randint(0, 6, size=rng.randint(1, 4))
    for kind in kinds:
        if kind == 0: e |= (rng.rand(H, W) < rng.choice([0.002, 0.02, 0.1, 0.3, 0.6, 0.95])).astype(np.uint8) * 255
        elif kind == 1:
            for y in range(int(rng.randint(0, 4)), H, int(rng.randint(2, 9))): e[y, int(rng.randint(0, 5)):W - int(rng.randint(0, 5))] = 255
        elif kind == 2:
            for x in range(int(rng.randint(0, 4)), W, int(rng.randint(2, 90))): e[:, x] = 255
        elif kind == 3:
            yy, xx = np.mgrid[0:H, 0:W]; p = int(rng.choice([1, 2, 3, 8, 32, 64])); e |= ((((xx // p) + (yy // p)) & 1) * 255).astype(np.uint8)
        elif kind == 4:
            for _ in range(int(rng.randint(1, 12))):       # slanted lines
                x0, y0 = rng.randint(0, W), rng.randint(0, H); dx, dy = rng.uniform(-1, 1), rng.uniform(-1, 1)
                t = np.arange(0, max(W, H)); xs = (x0 + t * dx).astype(int); ys = (y0 + t * dy).astype(int)
                ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H); e[ys[ok], xs[ok]] = 255
        else:
            m = rng.rand(H, W) < rng.choice([0.01, 0.1, 0.5]); e[m] = 0                                       # holes
    ms = int(rng.choice([1, 2, 3, 10, 40]))
    exp_xy, exp_ends = orc.kht_link(e, ms)
    got_xy, got_ends = capi.houghkht_link(e, ms)
    if not (np.array_equal(got_ends, exp_ends) and np.array_equal(got_xy, exp_xy)):
        bad += 1
        print("MISMATCH case %d: %dx%d kinds %s minSize %d" % (k, W, H, list(kinds), ms)); sys.stdout.flush()
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
