#!/bin/bash
# One GPU-box visit: the GPU test suite, then the bench in its default mode and A/B variants.  Everything lands in gpurun_out/$TAG.
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -4 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extras"
$B > $O/bench_default.json 2> $O/bench_default.err
$B --inflight 1 > $O/bench_inflight1.json 2> $O/bench_inflight1.err
for r in 16 64; do
  COMPVHIP_CANNY_ROWS=$r $B --reps 3 --no-verify > $O/bench_r${r}.json 2> $O/bench_r${r}.err
  COMPVHIP_CANNY_ROWS=$r $B --inflight 1 --reps 3 --no-verify > $O/bench_r${r}_inflight1.json 2> $O/bench_r${r}_inflight1.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], (d.get("verified") or {}).get("frames_checked") if isinstance(d.get("verified"), dict) else d.get("verified"), d["kernels_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
