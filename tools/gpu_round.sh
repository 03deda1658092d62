#!/bin/bash
# One GPU-box visit: the GPU test suite, then the bench in its default mode and A/B variants.  Everything lands in gpurun_out/$TAG.
TAG=${1:-run}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --no-cpu-baseline --no-extras > $O/bench_default.json 2> $O/bench_default.err
echo "bench default exit $?"
python bench.py --no-cpu-baseline --no-extras --inflight 1 > $O/bench_inflight1.json 2> $O/bench_inflight1.err
COMPVHIP_CANNY_IMPL=ring python bench.py --no-cpu-baseline --no-extras > $O/bench_ring.json 2> $O/bench_ring.err
COMPVHIP_CANNY_IMPL=ring python bench.py --no-cpu-baseline --no-extras --inflight 1 > $O/bench_ring_inflight1.json 2> $O/bench_ring_inflight1.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], d.get("verified"), d["kernels_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -3 $O/bench_default.err
