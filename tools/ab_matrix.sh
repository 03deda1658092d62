#!/usr/bin/env bash
# A/B evidence for DESIGN.md: the bench line (no CPU baseline) for the alternative kernels behind the environment switches and for the
# step modes.  Run on the GPU box:  tools/ab_matrix.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=${1:-$R/gpurun_out/ab}; mkdir -p "$O"
run() { name=$1; shift; env "$@" python "$R/bench.py" --no-cpu-baseline --reps 3 ${EXTRA:-} > "$O/$name.json" 2> "$O/$name.err"; python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-28s %.4f ms/step  %s" % (sys.argv[2], d["ms_per_step"], {k: v for k, v in d["kernels_ms_per_step"].items()}))
PY
}
EXTRA="" run default_2_in_flight A=1
EXTRA="--inflight 1" run one_stream A=1
EXTRA="--inflight 3" run three_in_flight A=1
EXTRA="--inflight 1 --sync-steps" run one_stream_sync_steps A=1
EXTRA="" run legacy_vote_2_in_flight COMPVHIP_SHT_VOTE=legacy
EXTRA="--inflight 1" run legacy_vote_one_stream COMPVHIP_SHT_VOTE=legacy
EXTRA="" run swar_canny_2_in_flight COMPVHIP_CANNY_IMPL=swar
EXTRA="--inflight 1" run swar_canny_one_stream COMPVHIP_CANNY_IMPL=swar
