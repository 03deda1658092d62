#!/usr/bin/env bash
# Full profiling set for profiles/: kernel-trace stats of the bench (default = two batches in flight, and --inflight 1 = one stream,
# the kernels never overlap) + separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) over tools/calib_traffic.py + the KHT call.
# Run on the GPU box:  tools/profile_all.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-run}; O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python "$R/bench.py" --steps 24 --warmup 4 --reps 1 --no-cpu-baseline --no-extras > "$O/bench_under_rocprof.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_inflight1" -- python "$R/bench.py" --steps 24 --warmup 4 --reps 1 --no-cpu-baseline --no-extras --inflight 1 > "$O/bench_inflight1_under_rocprof.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_kht" -- python "$R/tools/kht_bench.py" 5 > "$O/kht_under_rocprof.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$c" -- python "$R/tools/calib_traffic.py" > "$O/pmc_$c.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_kht_$c" -- python "$R/tools/kht_bench.py" 3 > "$O/pmc_kht_$c.log" 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/pmc_sq1" -- python "$R/tools/calib_traffic.py" > "$O/pmc_sq1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_sq2" -- python "$R/tools/calib_traffic.py" > "$O/pmc_sq2.log" 2>&1
python "$R/bench.py" > "$O/bench_default_run.json" 2> "$O/bench_default_run.err"
echo done
