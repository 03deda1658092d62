#!/usr/bin/env python
"""would a RUN-based voting mapping issue fewer LDS atomics than the E x T of the pixel mapping?  Exact count on the benchmark frame, no GPU.

Pixel mapping (shipped): lane = theta, one ds_add_u32 wave-instruction per (edge pixel, group of 64 theta) -> E * ceil(T / 64) instructions per frame.
Run mapping (the proposal): edges as maximal horizontal runs (x0, y, len) and, from the transposed map, vertical runs; theta bins sorted by |cos| into waves of 64.
A lane adds `count` once per rho cell its run reaches (cells = |rho(last) - rho(first)| + 1 <= len |cos| + 1 for a horizontal run, len |sin| + 1 for a vertical one);
the wave's trip count for a run is the MAXIMUM over its 64 lanes.  Every wave may choose the decomposition (horizontal or vertical runs) that costs it less.
Prints the instruction counts, and the per-instruction VALU budget the run mapping may spend before it loses (the pixel mapping spends 4 VALU + 2 SALU + 1 LDS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_bindings import Oracle, synth_frame

W, H, T = (int(sys.argv[1]), int(sys.argv[2])) + (180,) if len(sys.argv) > 2 else (3840, 2160, 180)
orc = Oracle()
rc, e = orc.canny(synth_frame(W, H, 12345), 59.0, 119.0)
edges = e != 0
E = int(edges.sum())
theta = np.arange(T, dtype=np.float32) * np.float32(np.pi / 180.0)
cq = np.round(np.cos(theta.astype(np.float64)) * 65536).astype(np.int64)
sq = np.round(np.sin(theta.astype(np.float64)) * 65536).astype(np.int64)


def runs_of(mask):
    """maximal horizontal runs of a boolean map: arrays (row, first column, length)"""
    m = np.pad(mask, ((0, 0), (1, 1))).astype(np.int8)
    d = np.diff(m, axis=1)
    ys, xs = np.nonzero(d == 1)
    ye, xe = np.nonzero(d == -1)
    return ys, xs, xe - xs


def trips(mask, cx, cy):
    """per theta-sorted wave: sum over the runs of max over the wave's lanes of the cells a run reaches; cx, cy: Q16 factors of the run's running / fixed coordinate"""
    y, x0, ln = runs_of(mask)
    order = np.argsort(np.abs(cx), kind="stable")           # lanes sorted by the slope along the run
    out = []
    for w0 in range(0, T, 64):
        ids = order[w0:w0 + 64]
        first = (x0[:, None] * cx[ids][None, :] + y[:, None] * cy[ids][None, :]) >> 16
        last = ((x0 + ln - 1)[:, None] * cx[ids][None, :] + y[:, None] * cy[ids][None, :]) >> 16
        cells = np.abs(last - first) + 1
        out.append((ids, int(cells.max(axis=1).sum()), int(cells.sum())))
    return out, len(y), float(ln.mean())


hz, nh, lh = trips(edges, cq, sq)
vt, nv, lv = trips(edges.T.copy(), sq, cq)
waves = (T + 63) // 64
pixel = E * waves
print("# Voting by runs instead of pixels: instruction count on the benchmark frame (%dx%d, seed 12345, Canny(59,119))\n" % (W, H))
print("%d edge pixels; %d horizontal runs (mean length %.2f), %d vertical runs (mean length %.2f); T = %d theta bins = %d waves of 64 lanes\n" % (E, nh, lh, nv, lv, T, waves))
print("| mapping | ds_add wave-instructions per frame | of the pixel mapping |")
print("|---|---|---|")
print("| pixel (shipped): E x %d | %d | 1.000 |" % (waves, pixel))
tot_h = sum(t for _, t, _ in hz); tot_v = sum(t for _, t, _ in vt)
print("| horizontal runs, theta sorted by abs(cos) | %d | %.3f |" % (tot_h, tot_h / pixel))
print("| vertical runs, theta sorted by abs(sin) | %d | %.3f |" % (tot_v, tot_v / pixel))
# each theta bin is served by exactly one wave of one decomposition: the bins with the smallest |cos| go to horizontal-run waves, those with the smallest |sin| to vertical-run waves
best = None
for nhw in range(0, waves + 1):
    # nhw waves take horizontal runs (the 64 * nhw bins of smallest |cos|), the rest vertical runs over the remaining bins sorted by |sin|
    oh = np.argsort(np.abs(cq), kind="stable")
    hsel = oh[:min(T, 64 * nhw)]
    rest = np.setdiff1d(np.arange(T), hsel)
    cost = 0
    y, x0, ln = runs_of(edges)
    for w0 in range(0, len(hsel), 64):
        ids = hsel[w0:w0 + 64]
        c = np.abs((((x0 + ln - 1)[:, None] * cq[ids] + y[:, None] * sq[ids]) >> 16) - ((x0[:, None] * cq[ids] + y[:, None] * sq[ids]) >> 16)) + 1
        cost += int(c.max(axis=1).sum())
    y, x0, ln = runs_of(edges.T.copy())
    rest = rest[np.argsort(np.abs(sq[rest]), kind="stable")]
    for w0 in range(0, len(rest), 64):
        ids = rest[w0:w0 + 64]
        c = np.abs((((x0 + ln - 1)[:, None] * sq[ids] + y[:, None] * cq[ids]) >> 16) - ((x0[:, None] * sq[ids] + y[:, None] * cq[ids]) >> 16)) + 1
        cost += int(c.max(axis=1).sum())
    if best is None or cost < best[1]:
        best = (nhw, cost)
print("| best mix: %d wave(s) on horizontal runs, the rest on vertical runs | %d | %.3f |" % (best[0], best[1], best[1] / pixel))
lane_cells = sum(c for _, _, c in hz)
print("\nLane-level work of the horizontal-run mapping: %d (run, theta) cells = %.3f of the E x T votes -- what the 64 lanes of a wave would need if they could stop independently; "
      "the wave-level count above is what they cost in lock step." % (lane_cells, lane_cells / (E * T)))
r = best[1] / pixel
print("\nBudget: the pixel mapping issues 4 VALU + 2 SALU + 1 LDS per vote instruction and sits at 0.64 of its LDS-atomic floor / 0.65 of its VALU floor (both pipes partly overlapped).")
print("A run step needs, beyond those, the cell boundary (where does rho change along the run: a division or a DDA compare per step), the count of the cell and a per-lane exit test:")
print("at %.3f of the instructions it breaks even only if a run step costs less than %.2f x a pixel vote, i.e. fewer than %.1f extra issue slots on a 7-slot step." % (r, 1 / r, 7 * (1 / r - 1)))
