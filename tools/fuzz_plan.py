"""Randomised parity sweep of the DEVICE-RESIDENT plan API (GPU box): batches of random size / stride / content through compvhip_plan_pipeline
(synchronous and asynchronous, kernel sizes 3 / 5, the three threshold modes, a line cut), compvhip_plan_houghkht with random knobs, and the host KHT
entry point with random knobs -- every frame against the oracle.   python tools/fuzz_plan.py [cases] [seed] [many]
("many": batches of 9..40 small frames, KHT on every case with 0 / 8 / 32 host threads: several groups of 8 frames and several group controllers)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from compv_amd import capi
from oracle_bindings import Oracle, synth_frame

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
many = len(sys.argv) > 3 and sys.argv[3] == "many"
dev = torch.device("cuda", 0)
ctx = capi.Context(0); orc = Oracle()
bad = 0


def image(W, H):
    kind = rng.randint(0, 4)
    if kind == 0: return synth_frame(W, H, int(rng.randint(1, 1 << 30)))
    if kind == 1: return rng.randint(0, 256, (H, W)).astype(np.uint8)
    if kind == 2:
        yy, xx = np.mgrid[0:H, 0:W]; p = int(rng.randint(2, 12)); return ((((xx // p) + (yy // p)) & 1) * int(rng.randint(100, 256))).astype(np.uint8)
    return np.clip(synth_frame(W, H, int(rng.randint(1, 1 << 30))).astype(np.int32) + rng.randint(-30, 30, (H, W)), 0, 255).astype(np.uint8)


for k in range(cases):
    big = (not many) and rng.rand() < 0.15
    W = int(rng.randint(1200, 4000)) if big else int(rng.randint(8, 1100)); H = int(rng.randint(300, 2200)) if big else int(rng.randint(8, 500))
    S = (W + 7) // 8 * 8 + 8 * int(rng.randint(0, 3)); F = 1 if big else int(rng.randint(9, 41)) if many else int(rng.randint(1, 5)); cap = 1 << 15
    if many: W = int(rng.randint(40, 420)); H = int(rng.randint(40, 300)); S = (W + 7) // 8 * 8 + 8 * int(rng.randint(0, 3)); cap = 1 << 12
    frames = np.zeros((F, H, S), np.uint8)
    imgs = [synth_frame(W, H, int(rng.randint(1, 1 << 30))) if big else image(W, H) for _ in range(F)]   # big frames: structured content only (the ORACLE needs minutes on megapixels of noise)
    for f in range(F): frames[f, :, :W] = imgs[f]; frames[f, :, W:] = rng.randint(0, 256, (H, S - W))   # garbage in the stride padding
    ksize = 3 if rng.rand() < 0.8 else 5
    mode = int(rng.choice([0, 0, 1, 2]))
    if mode == 0: tl, th = float(rng.randint(1, 300)), float(rng.randint(301, 900))
    elif mode == 1: tl, th = float(rng.uniform(0.3, 1.0)), float(rng.uniform(1.2, 2.5))
    else: tl, th = 0.5, 1.0
    thr = int(rng.randint(10, 150)); deg = float(rng.choice([1.0, 0.5])); maxl = int(rng.choice([0, 0, 20]))
    what = []
    try:
        plan = capi.Plan(ctx, W, H, S, F, deg)
        d_in = torch.from_numpy(frames).to(dev); d_e = torch.zeros_like(d_in)
        d_l = torch.zeros((F, cap, 5), dtype=torch.int32, device=dev); d_c = torch.zeros(F, dtype=torch.int32, device=dev)
        st = torch.cuda.Stream(device=dev)
        asyn = rng.rand() < 0.5
        tmode = {0: capi.THRESHOLD_COMPARE_TO_GRADIENT, 1: capi.THRESHOLD_PERCENT_OF_MEAN, 2: capi.THRESHOLD_OTSU}[mode]
        t = plan.pipeline_ex(d_in.data_ptr(), tl, th, thr, maxl, d_e.data_ptr(), d_l.data_ptr(), cap, d_c.data_ptr(), ksize=ksize, threshold_type=tmode,
                             stream=st.cuda_stream, asynchronous=asyn)
        if asyn: plan.wait(t)
        torch.cuda.synchronize()
        e = d_e.cpu().numpy(); counts = d_c.cpu().numpy(); raw = d_l.cpu().numpy().view(np.uint8).reshape(F, cap, 20)
        exp_edges = []
        for f in range(F):
            if mode == 2:
                lo, hi = orc.otsu_canny_thresholds(orc.otsu(imgs[f])); rc, ee = orc.canny(imgs[f], float(lo), float(hi), ksize, 0)
            else:
                rc, ee = orc.canny(imgs[f], tl, th, ksize, mode)
            if rc != 0: what.append("oracle rc %d" % rc); break
            exp_edges.append(ee)
            if not (e[f][:, :W] == ee).all(): what.append("frame %d edges: %d px" % (f, int((e[f][:, :W] != ee).sum()))); continue
            el = orc.sht(ee, deg, thr)
            if counts[f] != len(el): what.append("frame %d line count %d / %d" % (f, counts[f], len(el))); continue
            if len(el) > max(cap, 65536): continue   # documented: beyond max(lineCap, 65536) candidate lines the key buffer overflowed and the line set is an arbitrary subset (the count is exact: checked above)
            n = len(el) if maxl <= 0 else min(len(el), maxl)
            got = np.frombuffer(raw[f].tobytes(), dtype=capi.LINE_DTYPE)[:min(n, cap)]
            if maxl <= 0:
                gt = [(int(l["row"]), int(l["col"]), int(l["strength"])) for l in got]; et = [(l[3], l[4], l[2]) for l in el[:cap]]
                if gt != et:
                    i = next((j for j in range(min(len(gt), len(et))) if gt[j] != et[j]), -1)
                    what.append("frame %d lines: %d lines (cap %d), first difference at %d: %r / %r, same set: %s" % (f, len(el), cap, i, gt[i] if i >= 0 else None, et[i] if i >= 0 else None, sorted(gt) == sorted(et)))
            else:
                gs_ = [int(l["strength"]) for l in got]; es_ = [l[2] for l in el[:n]]
                if gs_ != es_: what.append("frame %d strongest lines: %d lines, got %r expected %r" % (f, len(el), gs_[:6], es_[:6]))
        # batched KHT on the edge maps just produced, random knobs
        if not what and not big and (many or rng.rand() < 0.5) and W >= 32 and H >= 32:
            rho = float(rng.choice([1.0, 0.5])); kd = float(rng.choice([1.0, 0.5, 2.0])); kthr = int(rng.choice([1, 1, 50])); mdev = float(rng.choice([2.0, 0.5, 4.0]))
            msz = int(rng.choice([10, 2, 5, 25])); mh = float(rng.choice([0.002, 0.0, 0.05]))
            lines, gss = plan.houghkht(d_e.data_ptr(), rho, kd, kthr, 0, mdev, msz, mh, threads=int(rng.choice([0, 8, 32] if many else [0, 1, 3])))
            for f in range(F):
                ek, gs_e = orc.kht(np.ascontiguousarray(exp_edges[f]), rho, kd, kthr, 0, mdev, msz, mh)
                gt = [(float(l["rho"]), float(l["theta"]), int(l["strength"])) for l in lines[f]]
                et = [(float(np.float32(l[0])), float(np.float32(l[1])), int(l[2])) for l in ek]
                if gt != et or (len(et) and gss[f] is not None and gss[f] != gs_e): what.append("frame %d kht (rho %g, theta %g, thr %d, dev %g, size %d, height %g): %d / %d lines" % (f, rho, kd, kthr, mdev, msz, mh, len(gt), len(et)))
        plan.close()
    except Exception as ex:
        what.append("exception %r" % (ex,))
    if what:
        bad += 1
        print("case %d (%dx%d stride %d, %d frames, ksize %d, mode %d): %s" % (k, W, H, S, F, ksize, mode, "; ".join(what[:4])), flush=True)
print("fuzz_plan: %d cases, %d with mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
