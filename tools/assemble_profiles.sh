#!/usr/bin/env bash
# Copies what tools/profile_all.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/<tag>/ (the files profiles/README.md describes)
# and derives pmc_counters_per_dispatch.txt and traffic.json.  Run in the build container after the gpurun call.
set -eu
TAG=${1:-r03}; R=$(cd "$(dirname "$0")/.." && pwd); P="$R/gpurun_out/prof_$TAG"; O="$R/profiles/$TAG"
mkdir -p "$O"
cp "$P"/stats/runc/*_kernel_stats.csv "$O/bench_kernel_stats.csv"
cp "$P"/stats_inflight1/runc/*_kernel_stats.csv "$O/bench_inflight1_kernel_stats.csv"
cp "$P"/stats_kht/runc/*_kernel_stats.csv "$O/kht_kernel_stats.csv"
for n in bench_under_rocprof bench_inflight1_under_rocprof kht_under_rocprof; do
  grep '^{' "$P/$n.log" | tail -1 > "$O/$n.json"
done
python "$R/tools/pmc_summary.py" "$P/pmc_FETCH_SIZE" "$P/pmc_WRITE_SIZE" "$P/pmc_sq1" "$P/pmc_sq2" > "$O/pmc_counters_per_dispatch.txt" 2>&1
python "$R/tools/pmc_summary.py" "$P/pmc_kht_FETCH_SIZE" "$P/pmc_kht_WRITE_SIZE" > "$O/kht_pmc_counters_per_dispatch.txt" 2>&1
python "$R/tools/traffic_from_pmc.py" "$P" "$O/traffic.json" > /dev/null
[ -f "$P/bench_default_run.json" ] && cp "$P/bench_default_run.json" "$O/bench_default_run.json"
# what overlaps what with two batches in flight: from the kernel traces of the two runs above
python "$R/tools/overlap_from_trace.py" "$P"/stats/runc/*_kernel_trace.csv "$P"/stats_inflight1/runc/*_kernel_trace.csv > "$O/two_lane_overlap_table.md" 2>&1 || true
echo "profiles/$TAG assembled"
