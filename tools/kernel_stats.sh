#!/bin/bash
# per-kernel rocprofv3 stats of a short single-lane bench run (kernels never overlap: durations are the kernels' own).  usage: tools/kernel_stats.sh <tag> [bench args]
TAG=${1:-kstats}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 8 --warmup 2 --reps 1 --no-verify "$@" > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - $O/prof <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    tot = 0.0
    for i, row in enumerate(csv.reader(open(f))):
        if i == 0: continue
        if "compvhip" in row[0] or "rocprim" in row[0]:
            n = row[0].replace("compvhip::", "").replace("void ", "").split("(")[0][:44]
            print("%-46s calls %4s avg %9.1f us" % (n, row[1], float(row[3]) / 1e3)); tot += float(row[3]) / 1e3 * (3 if "resolve" in n else 1)
    print("sum per step (resolve x 3): %.1f us" % tot)
PY
rm -rf $O/prof/*/*kernel_trace.csv
