"""Randomised parity sweep (GPU box): host entry points against the oracle on random sizes, contents and parameters.  Not part of the test suite (the suite's
cases are fixed); run after kernel changes:   python tools/fuzz_parity.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from compv_amd import capi
from oracle_bindings import Oracle, synth_frame

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = capi.Context(0); orc = Oracle()
bad = 0
for k in range(cases):
    W = int(rng.choice([rng.randint(3, 40), rng.randint(40, 300), rng.randint(300, 1100)])); H = int(rng.choice([rng.randint(3, 40), rng.randint(40, 400)]))
    kind = rng.randint(0, 5)
    if kind == 0: img = synth_frame(W, H, int(rng.randint(1, 1 << 30)))
    elif kind == 1: img = rng.randint(0, 256, (H, W)).astype(np.uint8)
    elif kind == 2: img = ((rng.rand(H, W) < rng.rand()) * 255).astype(np.uint8)
    elif kind == 3:
        yy, xx = np.mgrid[0:H, 0:W]; p = int(rng.randint(1, 9)); img = ((((xx // p) + (yy // p)) & 1) * int(rng.randint(100, 256))).astype(np.uint8)
    else:
        img = np.clip(synth_frame(W, H, int(rng.randint(1, 1 << 30))).astype(np.int32) + rng.randint(-40, 40, (H, W)), 0, 255).astype(np.uint8)
    what = []
    try:
        for op, oop in ((capi.OP_SOBEL, 0), (capi.OP_SCHARR, 2), (capi.OP_PREWITT, 3)):
            exp, _ = orc.edge_dete(img, oop)
            if not (ctx.edge_dete(img, op) == exp).all(): what.append("edge_dete op %d" % op)
        ksize = 3 if rng.rand() < 0.8 else 5
        if rng.rand() < 0.75: tl, th, tt = float(rng.randint(1, 400)), float(rng.randint(2, 900)), 0
        else: tl, th, tt = float(rng.uniform(0.2, 1.2)), float(rng.uniform(1.3, 3.0)), 1
        if tt == 0 and tl >= th: th = tl + 5.0
        rc, exp = orc.canny(img, tl, th, ksize, tt)
        if rc == 0:
            got = ctx.canny(img, tl, th, ksize, tt)
            if not (got == exp).all(): what.append("canny k%d t%d (%g, %g): %d px" % (ksize, tt, tl, th, int((got != exp).sum())))
            thr = int(rng.randint(5, 120)); deg = float(rng.choice([1.0, 0.5, 2.0]))
            lines, acc = ctx.houghsht(exp, deg, thr, want_acc=True)
            if not (acc == orc.sht_acc(exp, deg)).all(): what.append("sht accumulator")
            el = orc.sht_lines_from_acc_reference_order(orc.sht_acc(exp, deg), W, H, deg, thr)
            if [(int(l["row"]), int(l["col"]), int(l["strength"])) for l in lines] != [(l[3], l[4], l[2]) for l in el]: what.append("sht lines")
            if rng.rand() < 0.3 and W >= 16 and H >= 16:
                ek, gs_e = orc.kht(exp, 1.0, 1.0, 1)
                gk, gs = ctx.houghkht(exp, 1.0, 1.0, 1)
                gt = [(float(l["rho"]), float(l["theta"]), int(l["strength"])) for l in gk]
                et = [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2])) for l in ek]
                if gs != gs_e or gt != et:
                    first = next((i for i in range(min(len(gt), len(et))) if gt[i] != et[i]), min(len(gt), len(et)))
                    what.append("kht (GS %s, %d / %d lines, same as sets: %s, first difference at %d: %r / %r)" % ("==" if gs == gs_e else "!=", len(gt), len(et), sorted(gt) == sorted(et), first,
                                                                                                                  gt[first] if first < len(gt) else None, et[first] if first < len(et) else None))
                    os.makedirs(os.path.join(ROOT, "gpurun_out", "fuzz"), exist_ok=True)
                    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fuzz", "kht_case_%d.npz" % k), edges=exp, got=np.array(gt), exp=np.array(et))
    except Exception as e:
        what.append("exception %r" % (e,))
    if what:
        bad += 1
        print("case %d (%dx%d, kind %d): %s" % (k, W, H, kind, "; ".join(what)), flush=True)
print("fuzz: %d cases, %d with mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
