#!/bin/bash
# kernel trace of the default two-lane bench, kept as a compact timeline (start, end, queue, kernel) of the middle steps.  usage: tools/two_lane_timeline.sh <tag> [bench args]
TAG=${1:-tl}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 4 --reps 1 --no-verify "$@" > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "compvhip" in n or "rocprim" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), n.replace("void ", "").replace("compvhip::", "").split("(")[0].split("<")[0]))
rows.sort()
n = len(rows)
mid = rows[n // 2 - 60: n // 2 + 60]
t0 = mid[0][0]
with open(o + "/timeline.txt", "w") as fh:
    for s, e, q, k in mid:
        fh.write("%9.1f %9.1f q%d %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, q, k))
print(open(o + "/timeline.txt").read())
PY
rm -rf $O/prof
