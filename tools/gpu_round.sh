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
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], (d.get("verified") or {}).get("frames_checked") if isinstance(d.get("verified"), dict) else d.get("verified"), d["kernels_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
bash tools/pmc_pass.sh ${TAG}_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
bash tools/pmc_pass.sh ${TAG}_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_sq1 gpurun_out/pmc_${TAG}_sq2 > $O/pmc_summary.txt 2>&1
grep -A18 "canny_swar\|canny_resolve" $O/pmc_summary.txt | head -60
