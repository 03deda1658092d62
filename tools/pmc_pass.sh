#!/usr/bin/env bash
# One rocprofv3 counter pass (kernel-trace + pmc only, as the pool requires) over a short bench run.
# usage: tools/pmc_pass.sh <outdir-name> <counter> [<counter> ...]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_$NAME" -- \
  python "$R/bench.py" --steps 3 --warmup 1 --reps 1 --inflight 1 --no-cpu-baseline --no-extras --no-verify > "$R/gpurun_out/pmc_$NAME.log" 2>&1
echo "pmc $NAME exit $?" >> "$R/gpurun_out/pmc_$NAME.log"
