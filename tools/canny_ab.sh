#!/usr/bin/env bash
# A/B of the two kernel-size-3 Canny tile kernels (COMPVHIP_CANNY_IMPL=ring|swar) on the bench step, plus two rocprofv3 SQ counter
# passes for each.  Run on the GPU box: tools/canny_ab.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-ab}; O=$R/gpurun_out/$TAG; mkdir -p $O
for impl in ring swar; do
  COMPVHIP_CANNY_IMPL=$impl python $R/bench.py --no-cpu-baseline --reps 3 > $O/bench_$impl.json 2> $O/bench_$impl.err
  python - <<PY
import json
d=json.load(open("$O/bench_$impl.json")); print("$impl", d["ms_per_step"], d["reps_ms_per_step"], d["kernels_ms_per_step"])
PY
done
if [ "${PMC:-1}" = "1" ]; then
cd /tmp && export TMPDIR=/tmp
for impl in ring swar; do
  COMPVHIP_CANNY_IMPL=$impl timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1_$impl -- python $R/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline > $O/pmc_sq1_$impl.log 2>&1
  COMPVHIP_CANNY_IMPL=$impl timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_$impl -- python $R/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline > $O/pmc_sq2_$impl.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_sq1_$impl $O/pmc_sq2_$impl > $O/pmc_summary_$impl.txt 2>&1
done
fi
