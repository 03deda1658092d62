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
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 4 --warmup 1 --reps 1 --no-verify > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - $O/prof <<'PY'
import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for i, row in enumerate(csv.reader(open(f))):
        if "compvhip" in row[0] or "rocprim" in row[0] or i == 0: print(",".join(x[:60] for x in row[:5]))
PY
