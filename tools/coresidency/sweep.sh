#!/bin/bash
# does a smaller voting window (finer tile grid -> LDS left free beside a voting workgroup) let the other lane's Canny waves co-reside?
# For every window cap: the bench's step time (two lanes, all 256 frames verified), the vote / Canny kernel times (single-lane events) and -- with TRACE=1 --
# a kernel trace for tools/overlap_from_trace.py.  Usage: tools/coresidency/sweep.sh <tag> "<caps>"
TAG=${1:-cores}
CAPS=${2:-"1264 1072 960 848 784 736 672"}
O=gpurun_out/$TAG
mkdir -p $O
for cap in $CAPS; do
  COMPVHIP_VOTE_MAX_WINDOW=$cap python bench.py --no-cpu-baseline --no-extras --steps 256 > $O/bench_$cap.json 2> $O/bench_$cap.err
  python - $O/bench_$cap.json $cap <<'PY'
import json, sys, ctypes
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels_ms_per_step"]
    print("cap %s: step %.4f ms  value %.0f  verified %s  vote %.4f canny %.4f reduce %.4f compact %.4f" % (sys.argv[2], d["ms_per_step"], d["value"],
          (d.get("verified") or {}).get("frames_checked"), k["sht_vote_kernel"], k["canny_tile_kernel"], k["sht_reduce_kernel"], k["sht_compact_kernel"]))
except Exception as e:
    print("cap", sys.argv[2], "FAILED", e)
PY
done
if [ -n "$TRACE" ]; then
  cd /tmp && export TMPDIR=/tmp
  for cap in $TRACE; do
    COMPVHIP_VOTE_MAX_WINDOW=$cap rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$cap -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 4 --reps 1 --no-verify > $GRAFT_REPO_ROOT/$O/trace_$cap.log 2>&1
    COMPVHIP_VOTE_MAX_WINDOW=$cap rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1_$cap -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 24 --warmup 4 --reps 1 --no-verify > $GRAFT_REPO_ROOT/$O/trace1_$cap.log 2>&1
    python $GRAFT_REPO_ROOT/tools/overlap_from_trace.py $(ls $GRAFT_REPO_ROOT/$O/trace_$cap/*/*kernel_trace.csv | head -1) $(ls $GRAFT_REPO_ROOT/$O/trace1_$cap/*/*kernel_trace.csv | head -1) > $GRAFT_REPO_ROOT/$O/overlap_$cap.md 2>&1
    head -20 $GRAFT_REPO_ROOT/$O/overlap_$cap.md
    rm -rf $GRAFT_REPO_ROOT/$O/trace_$cap $GRAFT_REPO_ROOT/$O/trace1_$cap   # the traces are tens of MB; the table is what is kept
  done
fi
