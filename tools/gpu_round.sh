#!/bin/bash
# One GPU-box visit: the GPU test suite, then the bench in its default mode and A/B variants.  Everything lands in gpurun_out/$TAG.
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -4 $O/pytest.log
B="python bench.py --no-cpu-baseline"
$B > $O/bench_default.json 2> $O/bench_default.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], (d.get("verified") or {}).get("frames_checked") if isinstance(d.get("verified"), dict) else d.get("verified"), d["kernels_ms_per_step"])
    print(d.get("host_api")); print(d.get("kht"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -3 $O/bench_default.err
