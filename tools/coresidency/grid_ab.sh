#!/bin/bash
# A/B of vote tile grids through the COMPVHIP_VOTE_MAX_WINDOW lab knob, interleaved REPS times: 4K step + the 1080p extra config, all frames verified.
TAG=${1:-grid_ab}; CAPS=${2:-"1264 1072"}; REPS=${3:-3}
O=gpurun_out/$TAG; mkdir -p $O
for rep in $(seq 1 $REPS); do for cap in $CAPS; do
  COMPVHIP_VOTE_MAX_WINDOW=$cap python bench.py --no-cpu-baseline > $O/bench_${cap}_$rep.json 2> $O/bench_${cap}_$rep.err
  python - $O/bench_${cap}_$rep.json $cap $rep <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels_ms_per_step"]; x = d.get("configs_extra", {}).get("fhd_1920x1080", {})
    print("cap %s rep %s: 4K step %.4f ms (%d verified) vote %.4f reduce %.4f lines %.4f sort %.4f | 1080p step %s ms (%s)" % (sys.argv[2], sys.argv[3], d["ms_per_step"],
          (d.get("verified") or {}).get("frames_checked", -1), k["sht_vote_kernel"], k["sht_reduce_kernel"], k["sht_lines_kernel"], k["sht_sort_lines"], x.get("ms_per_step"), (x.get("verified") or {}).get("frames_checked") if isinstance(x.get("verified"), dict) else x.get("verified")))
except Exception as e:
    print("cap", sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done; done
