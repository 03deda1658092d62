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
for w in 1 4; do for r in 32 64 128; do
  COMPVHIP_CANNY_WAVES=$w COMPVHIP_CANNY_ROWS=$r $B --inflight 1 --reps 3 --no-verify > $O/bench_w${w}_r${r}.json 2> $O/bench_w${w}_r${r}.err
done; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], (d.get("verified") or {}).get("frames_checked") if isinstance(d.get("verified"), dict) else d.get("verified"), d["kernels_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 4 --warmup 1 --reps 1 --no-verify > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - $O/prof <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for i, row in enumerate(csv.reader(open(f))):
        if i < 16: print(",".join(x[:70] for x in row[:6]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    for k, v in per.items():
        if "resolve" in k: print(k, ["%.1f" % x for x in v[-12:]])
PY
