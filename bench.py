#!/usr/bin/env python
"""bench.py -- Mpixels/s of Sobel -> Canny -> HoughSHT on batches of synthetic 4K uint8 frames (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the whole hot path (compvhip_plan_pipeline: fused Sobel+NMS+tile hysteresis, cross-tile
resolve, edge compaction, Hough voting, Hough NMS, line sort/decode) over a batch of FRAMES_PER_GPU frames that are
already resident in HBM.  Frames are independent units: ranks process disjoint frame shards, no data-path collective
(weak scaling: per-GPU batch fixed).  Timing: barrier + synchronize, K steps, synchronize + barrier, MAX over ranks.

Extra objects on the JSON line:
  roofline      dominant kernel (by measured time): algorithmic bytes / average launch duration, durations measured
                with HIP events recorded by the library on the stream the kernels run on, during the timed steps
  roofline_canny  the fused Sobel->Canny tile kernel on the same basis (north_star's 40 % target is quoted on it)
  cpu_baseline  the REAL CompV CPU library (oracle/_ref, AVX2 intrinsics path, all host cores) -- or the C port --
                timed on a bounded sample of the same workload on rank 0 at N=1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
TIMING_MODES = {"sht_vote_kernel": 3, "canny_tile_kernel": 4}   # compvhip_plan_set_timing: HIP events around one kernel only
T_LOW, T_HIGH = 59.0, 119.0
THETA_DEG, SHT_THRESHOLD = 1.0, 100


def synth_batch(n, W, H, first_seed):
    from oracle_bindings import synth_frame
    return np.stack([synth_frame(W, H, first_seed + f) for f in range(n)])


def cpu_baseline(W, H, budget_s=20.0):
    """CompV's own CPU path (or the C port) on a bounded sample of the same workload."""
    from oracle_bindings import Oracle, RefShim, have_refshim, synth_frame
    cores = os.cpu_count() or 1
    if have_refshim():
        # CompVBase::init(numThreads): -1 = one worker per logical CPU.  The best of a short thread-count sweep is used
        # for the main sample, so that an over-subscribed thread pool does not flatter the GPU.
        probe = synth_batch(2, W, H, 12345)
        sweep = {}
        ref = None
        for t in (-1, 1, 8, 32):
            if t > cores:
                continue
            if ref is None:
                ref = RefShim(t)
            else:
                ref.reinit(t)
            ms, _, _ = ref.bench_pipeline(probe, T_LOW, T_HIGH, THETA_DEG, SHT_THRESHOLD)
            sweep[ref.threads] = round(ms / 2, 2)
        best = min(sweep, key=sweep.get)
        ref.reinit(best)
        n = int(max(2, min(64, budget_s * 1000.0 / max(sweep[best], 1e-3))))
        frames = synth_batch(n, W, H, 12345)
        ms, edges, lines = ref.bench_pipeline(frames, T_LOW, T_HIGH, THETA_DEG, SHT_THRESHOLD)
        return {"value": round(n * W * H / (ms * 1e-3) / 1e6, 2), "unit": "Mpixels/s", "cores": ref.threads, "host_cpus": cores,
                "kind": "reference",
                "sample": "%d frames %dx%d, CompV AVX2 intrinsics path (COMPV_ASM=0), %d threads (best of sweep), Canny(59,119)+SHT(1deg,100)" % (n, W, H, ref.threads),
                "ms_per_frame": round(ms / n, 3), "ms_per_frame_by_threads": sweep}
    orc = Oracle()
    img = synth_frame(W, H, 12345)
    t0 = time.time()
    n = 0
    while time.time() - t0 < budget_s or n < 1:
        rc, e = orc.canny(img, T_LOW, T_HIGH)
        orc.sht(e, THETA_DEG, SHT_THRESHOLD)
        n += 1
    dt = time.time() - t0
    return {"value": round(n * W * H / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "host_cpus": cores, "kind": "port",
            "sample": "%d frames %dx%d, scalar C restatement (oracle/compv_oracle.c)" % (n, W, H), "ms_per_frame": round(dt / n * 1e3, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=32)   # BASELINE config 4: 256 frames over 8 GPUs
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiment: no per-kernel HIP events in the timed steps")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed K-step loop; the MEDIAN repetition is reported")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight: N plans (own buffers) on N HIP streams take the steps in turn, so the latency-bound kernels of one "
                         "batch (key sort, decode, hysteresis rounds) run under the wide kernels of the other; 1 = one stream, kernels never overlap")
    ap.add_argument("--depth", type=int, default=2, help="steps enqueued per batch in flight before the host waits for the oldest one")
    ap.add_argument("--sync-steps", action="store_true", help="experiment: the synchronous step (one host round trip per step)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) even for a world of one rank: exercises init / barrier / all_reduce / "
                         "all_gather on real hardware where only one GPU is available")
    args = ap.parse_args()

    import torch
    from compv_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or (args.force_dist and "RANK" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist_on else 0)

    from compv_amd import sharding
    W, H, F = args.width, args.height, args.frames_per_gpu
    # weak scaling: the global batch is world*F frames, frame f uses seed 12345+f, rank r owns a contiguous block
    mine = sharding.shard_range(world * F, world, rank)
    assert len(mine) == F
    frames = synth_batch(F, W, H, sharding.frame_seed(mine[0]))
    d_in = torch.from_numpy(frames).to(dev)
    d_edges = torch.empty_like(d_in)
    line_cap = 1 << 16
    d_lines = torch.zeros((F, line_cap, 5), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(F, dtype=torch.int32, device=dev)

    ctx = capi.Context(local_rank if dist_on else 0)
    plan = capi.Plan(ctx, W, H, W, F, THETA_DEG)
    # a dedicated HIP stream (not the legacy null stream, whose implicit synchronisation costs ~3 % here); the library records its
    # kernel events on this same stream, and the timed region is bracketed by device-wide synchronisations
    launch_stream = torch.cuda.Stream(device=dev)
    stream = launch_stream.cuda_stream
    torch.cuda.synchronize()

    lanes = [{"plan": plan, "edges": d_edges, "lines": d_lines, "counts": d_counts, "stream": launch_stream}]
    for _ in range(1, max(1, args.inflight)):
        lanes.append({"plan": capi.Plan(ctx, W, H, W, F, THETA_DEG), "edges": torch.empty_like(d_in), "lines": torch.zeros_like(d_lines),
                      "counts": torch.zeros_like(d_counts), "stream": torch.cuda.Stream(device=dev)})
    torch.cuda.synchronize()

    def step():
        plan.pipeline(d_in.data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, d_edges.data_ptr(), d_lines.data_ptr(), line_cap,
                      d_counts.data_ptr(), stream)

    def run_steps(k):
        """k steps.  Default: each step is enqueued with compvhip_plan_pipeline_async and waited for while the NEXT one is already
        running (the hysteresis convergence flag is read one step late; a miss replays that step) -- no host round trip per step."""
        if args.sync_steps:
            for _ in range(k):
                step()
            return
        if args.inflight > 1:
            pend = []
            for i in range(k):
                q = lanes[i % len(lanes)]
                t = q["plan"].pipeline_async(d_in.data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, q["edges"].data_ptr(), q["lines"].data_ptr(),
                                             line_cap, q["counts"].data_ptr(), q["stream"].cuda_stream)
                pend.append((q, t))
                if len(pend) > max(1, min(args.depth, 3)) * len(lanes):   # the library keeps at most 4 steps of a plan in flight
                    q0, t0 = pend.pop(0)
                    q0["plan"].wait(t0)
            for q0, t0 in pend:
                q0["plan"].wait(t0)
            return
        prev = None
        for _ in range(k):
            t = plan.pipeline_async(d_in.data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, d_edges.data_ptr(), d_lines.data_ptr(), line_cap,
                                    d_counts.data_ptr(), stream)
            if prev is not None:
                plan.wait(prev)
            prev = t
        if prev is not None:
            plan.wait(prev)

    run_steps(args.warmup)
    torch.cuda.synchronize()
    # which kernel dominates a step?  One extra untimed, fully instrumented step decides which single kernel carries HIP events
    # during the timed steps (events around every launch would cost ~0.1 ms per step, around two kernels ~0.04 ms).
    dominant = "sht_vote_kernel"
    if not args.no_kernel_events:
        plan.set_timing(1)
        step()
        per = {}
        for name, ms in plan.get_timing():
            per[name] = per.get(name, 0.0) + ms
        plan.set_timing(0)
        torch.cuda.synchronize()
        if per:
            dominant = max(per.items(), key=lambda kv: kv[1])[0]

    # HIP events on the launch stream around the DOMINANT kernel during the timed steps (the roofline kernel); the full per-kernel
    # breakdown, and the Canny tile kernel's duration when it is not the dominant one, come from a second, untimed, instrumented pass.
    timed_mode = TIMING_MODES.get(dominant, 2)
    for q in lanes:
        q["plan"].set_timing(0 if args.no_kernel_events else timed_mode)
    per_kernel = {}

    def collect(dst, plans=None):
        for pl in (plans or [q["plan"] for q in lanes]):
            for name, ms in pl.get_timing():
                a = dst.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += 1

    # The timed region is EXACTLY K steps between barrier + synchronize brackets; it is repeated `reps` times inside this run and
    # the MEDIAN repetition is reported (one 20-step region lasts ~20 ms: a single one is a thin measurement).
    rep_elapsed = []
    for _ in range(max(1, args.reps)):
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        e = time.perf_counter() - t0
        rep_elapsed.append(sharding.max_over_ranks(e, dist if dist_on else None, dev))
        collect(per_kernel)   # the dominant kernel's events of this repetition's K steps (read after the closing bracket)
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]

    # second pass, untimed: ONE batch at a time on ONE stream with HIP events around every launch -- the per-kernel durations a kernel
    # has when it owns the GPU (in the timed region two batches are in flight and kernels of different batches share the CUs)
    for q in lanes:
        q["plan"].set_timing(0)
    breakdown = {}
    if rank == 0 and not args.no_kernel_events:
        plan.set_timing(1)
        for _ in range(args.steps):
            step()
            collect(breakdown, [plan])
    plan.set_timing(0)
    n_edges = int((d_edges != 0).sum().item())
    for q in lanes[1:]:   # every batch in flight produced the same result
        if not torch.equal(q["counts"], d_counts) or not torch.equal(q["edges"], d_edges):
            raise RuntimeError("the batches in flight disagree")
    if int(d_counts.max().item()) > line_cap:
        raise RuntimeError("a frame produced %d lines, more than the line capacity %d: its line set would be an arbitrary subset" % (int(d_counts.max().item()), line_cap))

    counts = d_counts.cpu().numpy()
    # the only result exchange of the job: all-gather of the tiny per-frame line counts, outside the timed region
    try:
        all_counts = sharding.gather_frame_results([int(c) for c in counts], dist if dist_on else None, dev)
    except Exception:  # reporting only: never let it hide the throughput number
        all_counts = None
    n_ranks = dist.get_world_size() if dist_on else 1
    total_px = n_ranks * F * W * H * args.steps
    value = total_px / elapsed / 1e6

    if rank == 0:
        R = 2 * (W + H) + 1
        T = 180
        nsteps_timed = args.steps * max(1, args.reps)
        kern = {k: {"ms_per_launch": v[0] / v[1], "launches_per_step": v[1] / nsteps_timed, "ms_per_step": v[0] / nsteps_timed}
                for k, v in per_kernel.items()}
        dom = max(kern.items(), key=lambda kv: kv[1]["ms_per_step"])[0] if kern else None
        # algorithmic bytes per frame: Sobel->Canny 1 B/px read (+1 B/px write reported separately);
        # SHT: W*H edge read + R*T*4 accumulator written once
        alg = {
            "canny_tile_kernel": F * W * H * 1.0,
            "sht_vote_kernel": F * (W * H + R * T * 4.0),
        }

        traffic_src = [None]

        def measured_traffic(name):
            # HBM bytes per launch from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB; see
            # tools/traffic_from_pmc.py for the calibration) -- not a counter of this run, and only valid for the workload they
            # were collected on; the JSON line says which file the number comes from
            try:
                rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "traffic.json")))
                t = json.load(open(os.path.join(ROOT, "profiles", rounds[-1], "traffic.json")))
                if t["workload"] == {"W": W, "H": H, "frames": F} and name in t["kernels"]:
                    traffic_src[0] = "committed PMC pass profiles/%s/traffic.json (rocprofv3 --pmc, separate runs; not measured in this run)" % rounds[-1]
                    return t["kernels"][name]["hbm_bytes"]
            except Exception:
                pass
            return None

        def roof(name, nbytes, ms=None):
            if ms is None and name not in kern:
                return None
            ms = kern[name]["ms_per_launch"] if ms is None else ms
            ach = nbytes / (ms * 1e-3) / 1e9
            tr = measured_traffic(name)
            return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tr, "traffic_source": traffic_src[0] if tr is not None else None,
                    "ms_per_launch": round(ms, 4), "algorithmic_bytes_per_launch": int(nbytes)}
        overlapped = len(lanes) > 1 and not args.sync_steps
        iso = {k: v[0] / v[1] for k, v in breakdown.items()}
        roofline = None
        if dom:
            if overlapped and dom in iso:
                # two batches in flight: a launch of the timed region shares the GPU with the other batch's kernels and its event-to-event
                # duration says how the GPU was shared, not how fast the kernel is.  The roofline is priced on the duration the kernel has when
                # it runs alone (second pass of this same run, K launches, HIP events on the launch stream); the timed-region figure is kept.
                roofline = roof(dom, alg.get(dom, F * W * H * 1.0), iso[dom])
                tr = roof(dom, alg.get(dom, F * W * H * 1.0))
                roofline["timing"] = ("HIP events on the launch stream around this kernel, %d launches of the single-stream pass that follows the timed "
                                      "region in this run; in the timed region %d batches are in flight and this kernel overlaps the other batch's "
                                      "kernels (ms_per_launch_timed_region, frac_timed_region)" % (breakdown[dom][1], len(lanes)))
                roofline["ms_per_launch_timed_region"] = tr["ms_per_launch"]
                roofline["frac_timed_region"] = tr["frac"]
            else:
                roofline = roof(dom, alg.get(dom, F * W * H * 1.0))
                roofline["timing"] = "HIP events on the launch stream around this kernel in every timed step"
            if dom == "sht_vote_kernel":
                # what actually bounds it: one ds_add_u32 wave-instruction (64 votes) per 4.1 LDS cycles per CU when conflict-free
                # (tools/microbench/lds_atomic_bench2), 256 CUs
                votes = float(n_edges) * T
                floor_ms = votes / 64.0 * 4.1 / 256.0 / 2.4e9 * 1e3
                roofline["lds_atomic_roofline"] = {"votes_per_launch": int(votes), "cycles_per_wave_instruction": 4.1, "cus": 256, "clock_ghz": 2.4,
                                                   "floor_ms": round(floor_ms, 4), "frac": round(floor_ms / roofline["ms_per_launch"], 4)}
                roofline["note"] = ("the voting kernel is bound by the LDS atomic pipe, not by HBM (lds_atomic_roofline); the HBM fraction is reported "
                                    "because the contract prices every kernel of this path against the HBM roofline")
        rc = None
        if "canny_tile_kernel" in iso:
            rc = roof("canny_tile_kernel", alg["canny_tile_kernel"], iso["canny_tile_kernel"])
            rc["timing"] = "HIP events in the single-stream instrumented pass after the timed steps"
        if rc:
            rc["frac_read_plus_write"] = round(2 * rc["frac"], 4)
        out = {
            "metric": "Mpixels/s Sobel->Canny->HoughSHT on 4K uint8",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "reps": len(rep_elapsed), "reps_ms_per_step": [round(e / args.steps * 1e3, 4) for e in rep_elapsed],
            "timing": "median of %d repetitions of the K-step region (each: barrier + synchronize, K steps, synchronize + barrier, MAX over ranks)" % len(rep_elapsed),
            "step_mode": "synchronous (host reads the hysteresis flag every step)" if args.sync_steps else
                         "pipelined (compvhip_plan_pipeline_async: step k's hysteresis flag is read while step k+1 runs), %d batch(es) in flight "
                         "(one plan + HIP stream each, steps dealt round-robin)" % len(lanes),
            "batches_in_flight": 1 if args.sync_steps else len(lanes),
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "dist_backend": (dist.get_backend() if dist_on else None),
            "config": {"workload": "batched %dx%d uint8 frames, Sobel3x3 -> Canny(59,119) -> HoughSHT(rho=1, theta=1deg, thr=100)" % (W, H),
                       "frames_per_gpu": F, "global_frames": world * F,
                       "parallelism": "frames sharded across %d GPU(s), no data-path collective" % world},
            "roofline": roofline, "roofline_canny": rc,
            # every kernel of a step, from the instrumented pass AFTER the timed steps (same process, same buffers)
            "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(breakdown.items())},
            "kernels_ms_per_step_source": "second pass of %d steps, one batch at a time on one stream, HIP events around every launch (not in the timed region)" % args.steps,
            "edge_pixels_all_frames": n_edges,
            "lines_frame0": int(counts[0]),
            "lines_all_frames": (int(sum(all_counts)) if all_counts is not None else None),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(W, H)
            except Exception as e:  # the baseline is reporting only; never let it hide the GPU number
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out))
    for q in lanes:
        q["plan"].close()
    ctx.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
