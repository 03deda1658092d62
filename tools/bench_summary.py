#!/usr/bin/env python
"""Prints the figures of a bench.py JSON line that DESIGN.md / profiles/README.md quote.  usage: tools/bench_summary.py <file with the JSON line>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("step", d["ms_per_step"], "ms; value", d["value"], "; reps", d["reps_ms_per_step"], "; verified", (d.get("verified") or {}).get("frames_checked"))
print("kernels", d["kernels_ms_per_step"])
r = d["roofline"]
print("roofline", {k: r.get(k) for k in ("frac", "traffic", "ms_per_launch", "ms_per_launch_timed_region", "timed_region_launches_sampled")}, "valu", (r.get("valu_issue") or {}).get("frac"),
      "lds", (r.get("lds_atomic_roofline") or {}).get("frac"))
c = d["roofline_canny"]
print("canny", {k: c.get(k) for k in ("frac", "traffic", "ms_per_launch", "frac_read_plus_write")}, "valu", (c.get("valu_issue") or {}).get("frac"))
print("stages", d["roofline_stage_canny"]["frac"], d["roofline_stage_sht"]["frac"], d["roofline_stage_canny"]["ms_per_step"], d["roofline_stage_sht"]["ms_per_step"])
x = (d.get("configs_extra") or {}).get("fhd_1920x1080") or {}
print("fhd", x.get("value"), x.get("ms_per_step"))
k = d.get("kht") or {}
if k:
    print("kht", k.get("ms_per_frame"), k.get("ms_per_frame_calls"), "threads", k.get("host_threads"), "budget", k.get("host_cpu_budget"), "thread-ms", k.get("thread_ms_per_frame"),
          k.get("ms_per_frame_by_host_threads"), (k.get("verified") or {}).get("frames_checked") if isinstance(k.get("verified"), dict) else k.get("verified"),
          "traffic", (k.get("roofline") or {}).get("traffic"), "ref x32", ((k.get("cpu_baseline") or {}).get("frame_parallel") or {}).get("ms_per_frame"))
cb = d.get("cpu_baseline") or {}
if cb:
    fp = cb.get("frame_parallel") or {}
    print("cpu", cb.get("value"), cb.get("cores"), "| x8", fp.get("value"), fp.get("cores"), "| x1", (fp.get("one_thread_per_process") or {}).get("value"))
print("host_api", d.get("host_api"))
print("extra", {k: (v.get("stage_ms") or v.get("ms")) for k, v in (d.get("kernels_extra") or {}).items() if isinstance(v, dict)})
