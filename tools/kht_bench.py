"""KHT on one 4K frame (BASELINE config 5): stage times of compvhip_houghkht_u8 and a workload for rocprofv3
(tools/profile_all.sh runs it under --kernel-trace --stats and under the FETCH_SIZE / WRITE_SIZE passes)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from compv_amd import capi
from oracle_bindings import synth_frame

W, H = 3840, 2160
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
img = synth_frame(W, H, 12345)
ctx = capi.Context(0)
edges = ctx.canny(img, 59.0, 119.0)
stages = []
wall = []
for _ in range(reps):
    t0 = time.perf_counter()
    lines, gs = ctx.houghkht(edges, 1.0, 1.0, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
    wall.append((time.perf_counter() - t0) * 1e3)
    stages.append(ctx.houghkht_stage_ms())
st = np.median(np.array(stages), axis=0)
names = ["link (host)", "subdivide (upload + kht_subdivide_kernel + kht_gather_clusters_kernel)", "statistics (kht_stats_kernel + download)", "prune + Gmin (host)",
         "vote + peaks (kht_vote_kernel, kht_peaks_kernel, download)", "sort + sweep (host)"]
print(json.dumps({"workload": "KHT(rho=1, theta=1deg, thr=100) on one %dx%d Canny(59,119) edge map, %d edge pixels" % (W, H, int((edges != 0).sum())),
                  "lines": int(len(lines)), "gs": gs, "ms_per_call_median": round(float(np.median(wall)), 3),
                  "stage_ms_median": {n: round(float(v), 3) for n, v in zip(names, st)}}))
