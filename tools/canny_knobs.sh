#!/usr/bin/env bash
# Measurement aid: kernel time of the Canny tile kernel with parts of it switched off (COMPVHIP_CANNY_DBG, results wrong), plus
# two rocprofv3 SQ counter passes over a short bench run.  Run on the GPU box: tools/canny_knobs.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-run}; O=$R/gpurun_out/$TAG; mkdir -p $O
for d in ${KNOBS:-0 4 12 28 60 44 8}; do
  COMPVHIP_CANNY_DBG=$d python $R/bench.py --no-cpu-baseline --reps 2 --steps 5 > $O/knob_$d.json 2> $O/knob_$d.err
  python - <<PY
import json
d=json.load(open("$O/knob_$d.json")); print("dbg=$d", d["ms_per_step"], {k:v for k,v in d["kernels_ms_per_step"].items() if k.startswith("canny")})
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- python $R/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -- python $R/bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline > $O/pmc_sq2.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_summary.txt 2>&1
grep -A18 "canny_swar" $O/pmc_summary.txt | head -40
