#!/usr/bin/env python
"""bench.py -- Mpixels/s of Sobel -> Canny -> HoughSHT on batches of synthetic 4K uint8 frames (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N > 1: one rank per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE / RANK / LOCAL_RANK) --gpus must equal the world size;
        a BARE `python bench.py --gpus N` starts its own N ranks (torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
        --master-port <free>) and prints rank 0's JSON line as its only stdout line.

One "step" = one pass of the whole hot path (compvhip_plan_pipeline_async: fused Sobel+NMS, hysteresis rounds, edge compaction, Hough
voting, Hough NMS, line sort/decode) over ONE batch of FRAMES_PER_GPU frames already resident in HBM.  The job holds BASELINE config 4's
256 distinct frames (seeds 12345 .. 12345+255) as 8 resident batches of 32 and rotates over them: step k of rank r processes batch
(r + k) mod 8, so no step re-reads the previous step's input (nothing of the 265 MB a step reads can still sit in the 256 MiB Infinity
Cache from the step before last, and the two batches in flight are always different ones).  Frames are independent units: ranks
process disjoint batches, no data-path collective (weak scaling: per-GPU batch fixed).  --scatter adds the RCCL data path of SURVEY 8(e):
the step's batch is born on rank 0 and scattered, the per-frame line counts and strongest lines are all-gathered.
Timing: barrier + synchronize, K steps, synchronize + barrier, MAX over ranks; median of --reps repetitions.

After the timed region every one of the 256 frames is checked against tests/golden/golden_batch.json (edge-map MD5, edge count, line
count, strength sum, line-set hash -- all produced by the real CompV library, tests/golden/make_golden_batch.py), in the same two-lane
asynchronous mode the timed steps use; a mismatch raises and NO number is printed.

Extra objects on the JSON line:
  roofline        dominant kernel (by measured time): algorithmic bytes / average launch duration (HIP events on the launch stream)
  roofline_canny  the fused Sobel->Canny tile kernel on the same basis (north_star's 40 % target is quoted on it) + its VALU-issue floor
  cpu_baseline    the REAL CompV CPU library (oracle/_ref, AVX2 intrinsics path) timed on a bounded sample of the same workload, at its
                  best thread count and frame-parallel (P processes x T threads over independent frames)
  host_api        latency of the host-pointer (drop-in) entry points on one 4K frame, PCIe transfers included
  kht             the batched KHT path on this run's edge maps (BASELINE config 5)
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS, CLOCK_GHZ = 1024, 2.4
TIMING_MODES = {"sht_vote_kernel": 3, "canny_tile_kernel": 4}   # compvhip_plan_set_timing: HIP events around one kernel only
T_LOW, T_HIGH = 59.0, 119.0
THETA_DEG, SHT_THRESHOLD = 1.0, 100
GOLDEN_FRAMES = 256
M64 = (1 << 64) - 1


# ------------------------------------------------------------------------------------------------------------------------------------
# synthetic frames on the device (bit-identical to tests/oracle_bindings.synth_frame / oracle/compv_oracle.c::orc_synth_frame)
# ------------------------------------------------------------------------------------------------------------------------------------
def lcg_tables(n):
    """s_k = a^k s_0 + c_k (mod 2^32) for k = 1..n: the LCG jump-ahead tables of synth_frame."""
    M = np.uint64(0xFFFFFFFF)
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    ap = np.array([1], dtype=np.uint64)
    cp = np.array([0], dtype=np.uint64)
    while len(ap) <= n:
        aL = (ap[-1] * a) & M
        cL = (cp[-1] * a + c) & M
        ap, cp = np.concatenate([ap, (ap * aL) & M]), np.concatenate([cp, (ap * cL + cp) & M])
    return ap[1:n + 1].astype(np.int64), cp[1:n + 1].astype(np.int64)


class FrameSynth:
    def __init__(self, torch, dev, W, H):
        self.torch, self.W, self.H = torch, W, H
        ap, cp = lcg_tables(W * H)
        self.ap = torch.from_numpy(ap).to(dev).reshape(H, W)
        self.cp = torch.from_numpy(cp).to(dev).reshape(H, W)
        i = torch.arange(W, dtype=torch.int64, device=dev)[None, :]
        j = torch.arange(H, dtype=torch.int64, device=dev)[:, None]
        self.base = 40 + (((i // 64 + j // 64) & 1) * 150)
        self.diag = ((i + 2 * j) % 257) < 3

    def frame(self, seed):
        t = self.torch
        s = (self.ap * int(seed & 0xFFFFFFFF) + self.cp) & 0xFFFFFFFF     # int64 wrap-around keeps the low 32 bits exact
        v = self.base + (s >> 28)
        v = t.where(self.diag, t.full_like(v, 255), v)
        return v.to(t.uint8)

    def batch(self, seeds):
        return self.torch.stack([self.frame(s) for s in seeds])


# ------------------------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------------------------------
def _ref_worker(threads, first_seed, n, work_s, W, H, barrier, queue, task="pipeline"):
    """One process of the frame-parallel baseline: its own CompV instance with `threads` workers, its own frames.  Import, library start-up and
    frame synthesis happen BEFORE the cross-process barrier; the clock runs over passes over the worker's n frames until work_s seconds are up.
    task "pipeline": Canny + SHT on the frames; task "kht": CompVHoughKht::process on the frames' Canny edge maps (made before the barrier)."""
    try:
        from oracle_bindings import RefShim, synth_frame
        ref = RefShim(threads)
        frames = np.stack([synth_frame(W, H, first_seed + f) for f in range(n)])
        if task == "kht":
            maps = np.stack([ref.canny(fr, T_LOW, T_HIGH)[1] for fr in frames])
            run = lambda i: ref.bench_kht(maps[i:i + 1], 1.0, THETA_DEG, 1)
        else:
            run = lambda i: ref.bench_pipeline(frames[i:i + 1], T_LOW, T_HIGH, THETA_DEG, SHT_THRESHOLD)
        run(0)   # warm the pool and the scratch buffers
        barrier.wait()
        t0 = time.time()
        done = 0
        while time.time() - t0 < work_s:
            run(done % n)
            done += 1
        queue.put((t0, time.time(), done))
    except Exception as e:  # a worker that dies must not leave the others at the barrier for ever
        try:
            barrier.abort()
        except Exception:
            pass
        queue.put(("error", repr(e), 0))


def host_cpu_budget():
    """What the host really offers this process: logical CPUs, the affinity mask and the cgroup CPU quota (a container may see 256 CPUs and own 64)."""
    out = {"logical_cpus": os.cpu_count() or 1}
    try:
        out["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        out["sched_affinity"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            quota = open(path).read().strip()
            out["cgroup_cpu_max_file"] = path
            break
        except Exception:
            pass
    out["cgroup_cpu_max"] = quota
    try:
        q = quota.split()
        if q[0] not in ("max", "-1"):
            period = float(q[1]) if len(q) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["cgroup_cpus"] = round(float(q[0]) / period, 2)
    except Exception:
        pass
    return out


def effective_cpus():
    """CPUs this process can really run on at once: min(affinity mask, cgroup quota)."""
    b = host_cpu_budget()
    n = b.get("sched_affinity") or b["logical_cpus"]
    if b.get("cgroup_cpus"):
        n = min(n, max(1, int(b["cgroup_cpus"])))
    return int(n)


def frame_parallel_baseline(W, H, threads_per_proc, cores, work_s=4.0, procs=None, task="pipeline"):
    """P processes x T threads over independent frames, all released together (barrier); every worker processes frames for work_s seconds."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    tt = max(1, threads_per_proc)
    procs = max(1, min(cores // tt, 32)) if procs is None else max(1, min(procs, cores // tt))
    n = 4
    barrier = ctx.Barrier(procs)
    queue = ctx.Queue()
    ps = [ctx.Process(target=_ref_worker, args=(tt, 20000 + 100 * i, n, work_s, W, H, barrier, queue, task)) for i in range(procs)]
    for pr in ps:
        pr.start()
    res = [queue.get(timeout=600) for _ in ps]
    for pr in ps:
        pr.join(timeout=60)
    bad = [r for r in res if r[0] == "error"]
    if bad:
        raise RuntimeError("frame-parallel worker failed: %s" % bad[0][1])
    span = max(r[1] for r in res) - min(r[0] for r in res)
    frames = sum(r[2] for r in res)
    return {"value": round(frames * W * H / span / 1e6, 2), "unit": "Mpixels/s", "processes": procs, "threads_per_process": tt,
            # cores = what the run could really occupy (a box shows 256 logical CPUs and grants a cgroup quota of 16): never the thread count alone
            "cores": min(procs * tt, effective_cpus()), "threads": procs * tt, "logical_cpus": os.cpu_count() or 1,
            "frames": frames, "span_s": round(span, 3), "start_skew_s": round(max(r[0] for r in res) - min(r[0] for r in res), 4),
            "ms_per_frame": round(span * 1e3 / max(frames, 1), 3), "thread_ms_per_frame": round(span * 1e3 * procs * tt / max(frames, 1), 2),
            "note": "all workers released by one cross-process barrier after start-up, each processes frames for %.0f s; throughput = frames / (last finish - first start)" % work_s}


def cpu_baseline(W, H, budget_s=20.0):
    """CompV's own CPU path (or the C port) on a bounded sample of the same workload."""
    from oracle_bindings import Oracle, RefShim, have_refshim, synth_frame
    cores = os.cpu_count() or 1
    if have_refshim():
        # CompVBase::init(numThreads): -1 = one worker per logical CPU.  The best of a short thread-count sweep is used
        # for the main sample, so that an over-subscribed thread pool does not flatter the GPU.
        probe = np.stack([synth_frame(W, H, 12345 + f) for f in range(2)])
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
        n = int(max(2, min(64, 0.5 * budget_s * 1000.0 / max(sweep[best], 1e-3))))
        frames = np.stack([synth_frame(W, H, 12345 + f) for f in range(n)])
        ms, edges, lines = ref.bench_pipeline(frames, T_LOW, T_HIGH, THETA_DEG, SHT_THRESHOLD)
        out = {"value": round(n * W * H / (ms * 1e-3) / 1e6, 2), "unit": "Mpixels/s", "cores": min(ref.threads, effective_cpus()), "threads": ref.threads, "host_cpus": cores,
               "kind": "reference",
               "sample": "%d frames %dx%d, CompV AVX2 intrinsics path (COMPV_ASM=0), %d threads (best of sweep), Canny(59,119)+SHT(1deg,100)" % (n, W, H, ref.threads),
               "ms_per_frame": round(ms / n, 3), "ms_per_frame_by_threads": sweep}
        # frame-parallel: CompV's row-band pool stops scaling at ~8 threads, independent frames do not -- P processes x T threads
        # ("all host cores via row-band / frame-parallel threads")
        out["host"] = host_cpu_budget()
        try:
            tt = max(1, min(best, 8))
            out["frame_parallel"] = frame_parallel_baseline(W, H, tt, cores)
            # the same harness with ONE thread per process: if P x 8 threads barely beat one process, is it the row-band pool or the host?
            out["frame_parallel"]["one_thread_per_process"] = frame_parallel_baseline(W, H, 1, cores, work_s=3.0, procs=32)
        except Exception as e:  # reporting only
            out["frame_parallel"] = {"error": str(e)}
        return out
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


def host_api_latency(capi, dev_index, frame):
    """Latency of the drop-in (host pointer) entry points on one frame: upload + kernels + download (+ host sorts), best of 3."""
    ctx = capi.Context(dev_index)
    out = {}
    try:
        def best(fn, n=3):
            fn()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
            return round(min(ts) * 1e3, 3), r
        out["canny_u8_ms"], edges = best(lambda: ctx.canny(frame, T_LOW, T_HIGH))
        out["houghsht_u8_ms"], lines = best(lambda: ctx.houghsht(edges, THETA_DEG, SHT_THRESHOLD))
        out["houghkht_u8_ms"], kl = best(lambda: ctx.houghkht(edges, 1.0, THETA_DEG, 1))
        out["sht_lines"] = int(len(lines[0]) if isinstance(lines, tuple) else len(lines))
        out["note"] = "one %dx%d frame per call, host pointers (what CompVEdgeDete::process / CompVHough::process hand over): PCIe both ways included" % (frame.shape[1], frame.shape[0])
    finally:
        ctx.close()
    return out


def so_sha256():
    h = hashlib.sha256()
    with open(os.path.join(ROOT, "compv_amd", "lib", "libcompv_hip.so"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def src_sha256():
    """sha256 over the sources the library is built from (compv_amd/csrc/*.{hip,hpp,cpp}, Makefile, include/compv_hip.h), names included."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "compv_amd", "csrc")
    files = sorted(f for f in os.listdir(d) if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile")
    for f in [os.path.join(d, x) for x in files] + [os.path.join(ROOT, "include", "compv_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def committed_counters(W, H, F):
    """PMC results committed under profiles/ (rocprofv3 --pmc passes, tools/pmc_pass.sh + tools/traffic_from_pmc.py).  They are only
    valid for the library build they were collected with: the file stores that build's sha256 and a stale file is ignored."""
    try:
        rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "traffic.json")))
        t = json.load(open(os.path.join(ROOT, "profiles", rounds[-1], "traffic.json")))
        if t.get("workload") != {"W": W, "H": H, "frames": F}:
            return None, "profiles/%s/traffic.json is for another workload" % rounds[-1]
        same_so, same_src = t.get("so_sha256") == so_sha256(), t.get("src_sha256") == src_sha256()
        if not (same_so or same_src):
            return None, "profiles/%s/traffic.json was collected with another build of libcompv_hip.so (neither the binary nor its sources match): traffic = null" % rounds[-1]
        return t, "committed PMC pass profiles/%s/traffic.json (rocprofv3 --pmc, separate runs; %s)" % (
            rounds[-1], ("same library binary, sha256 %s..." % t["so_sha256"][:12]) if same_so else ("same library sources, sha256 %s..." % t["src_sha256"][:12]))
    except Exception as e:
        return None, "no committed PMC pass (%s)" % e


def check_batch(torch, q, seeds, golden, W, H, line_cap, dev):
    """One lane's edge maps / line lists / counts of a whole batch against the reference-derived fixture (real CompV: edge-map MD5, edge count,
    line count, strength sum, order-independent line-set hash).  Returns (frames checked, edge pixels, lines); raises on the first mismatch."""
    F = len(seeds)
    counts = q["counts"].cpu().numpy()
    if int(counts.max()) > line_cap:
        raise RuntimeError("a frame produced %d lines, more than the line capacity %d: its line set would be an arbitrary subset" % (int(counts.max()), line_cap))
    n_e = (q["edges"] != 0).sum(dim=(1, 2)).cpu().numpy()
    if golden is None:
        return 0, int(n_e.sum()), int(counts.sum())
    ln = q["lines"].to(torch.int64)
    idx = torch.arange(line_cap, device=dev)[None, :]
    valid = idx < q["counts"].to(torch.int64)[:, None]
    st = torch.where(valid, ln[:, :, 2], torch.zeros_like(ln[:, :, 2]))
    hv = ((W + H) - ln[:, :, 3] + 32768) * 1000003 + ln[:, :, 4] * 7919 + ln[:, :, 2] * 31337
    hv = torch.where(valid, hv, torch.zeros_like(hv)).sum(dim=1).cpu().numpy()
    sums = st.sum(dim=1).cpu().numpy()
    edges_host = q["edges"].cpu().numpy()
    for f in range(F):
        g = golden[seeds[f]]
        got = {"canny_md5": hashlib.md5(edges_host[f].tobytes()).hexdigest(), "edges": int(n_e[f]), "lines": int(counts[f]),
               "sum_strength": int(sums[f]), "line_hash": "%016x" % (int(hv[f]) & M64)}
        exp = {k: g[k] for k in got}
        if got != exp:
            raise RuntimeError("frame seed %d (frame %d of its batch) differs from the CompV reference: got %r, expected %r" % (seeds[f], f, got, exp))
    return F, int(n_e.sum()), int(counts.sum())


def extra_config(torch, capi, sharding, ctx, dev, W, H, F, NB, steps, warmup, reps, fixture):
    """The same pipeline at another frame size (north_star: throughput on 1080p as well as 4K), same two-lane asynchronous mode, every frame of the
    rotating batches checked against its own reference-derived fixture after the timed region."""
    synth = FrameSynth(torch, dev, W, H)
    seeds = [[sharding.frame_seed(b * F + f) for f in range(F)] for b in range(NB)]
    blocks = [synth.batch(sd) for sd in seeds]
    del synth
    line_cap = 1 << 15
    lanes = [{"plan": capi.Plan(ctx, W, H, W, F, THETA_DEG), "edges": torch.empty_like(blocks[0]),
              "lines": torch.zeros((F, line_cap, 5), dtype=torch.int32, device=dev), "counts": torch.zeros(F, dtype=torch.int32, device=dev),
              "stream": torch.cuda.Stream(device=dev)} for _ in range(2)]
    try:
        def go(q, b):
            return q["plan"].pipeline_async(blocks[b].data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, q["edges"].data_ptr(), q["lines"].data_ptr(), line_cap,
                                            q["counts"].data_ptr(), q["stream"].cuda_stream)
        k = [0]

        def run(n):
            pend = []
            for _ in range(n):
                q = lanes[k[0] % 2]
                pend.append((q, go(q, k[0] % NB)))
                k[0] += 1
                if len(pend) > 4:
                    q0, t0 = pend.pop(0)
                    q0["plan"].wait(t0)
            for q0, t0 in pend:
                q0["plan"].wait(t0)
        run(warmup)
        torch.cuda.synchronize()
        el = []
        for _ in range(max(1, reps)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            el.append(time.perf_counter() - t0)
        e = sorted(el)[len(el) // 2]
        golden = None
        if fixture and os.path.exists(fixture):
            g = json.load(open(fixture))
            assert (g["W"], g["H"], g["tLow"], g["tHigh"], g["threshold"]) == (W, H, T_LOW, T_HIGH, SHT_THRESHOLD)
            golden = {x["seed"]: x for x in g["frames"]}
        checked = 0
        for b0 in range(0, NB, 2):
            todo = [(lanes[i], b0 + i) for i in range(2) if b0 + i < NB]
            tickets = [(q, go(q, b)) for q, b in todo]
            for q, t in tickets:
                q["plan"].wait(t)
            torch.cuda.synchronize()
            for q, b in todo:
                checked += check_batch(torch, q, seeds[b], golden, W, H, line_cap, dev)[0]
        return {"workload": "batched %dx%d uint8 frames, same pipeline and parameters; %d distinct frames as %d batches of %d" % (W, H, NB * F, NB, F),
                "value": round(F * W * H * steps / e / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(e / steps * 1e3, 4), "steps": steps, "reps": len(el),
                "frames_per_step": F,
                "verified": ({"frames_checked": checked, "against": os.path.relpath(fixture, ROOT) + " (real CompV, tests/golden/make_golden_batch.py fhd)"}
                             if golden is not None else "no fixture")}
    finally:
        for q in lanes:
            q["plan"].close()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_argv(gpus, argv, port):
    """The command a bare `python bench.py --gpus N ...` re-executes itself through: N ranks on this node, one per GPU."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    """Run the N ranks as children; stdout carries exactly what rank 0 printed (the ONE JSON line), everything else goes to stderr."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launch_argv(gpus, argv, free_port())
    print("bench.py: --gpus %d without a launcher: starting %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    json_lines = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:
        if l not in json_lines:
            print(l, file=sys.stderr)
    if p.returncode != 0:
        print("bench.py: the %d-rank launch exited with %d" % (gpus, p.returncode), file=sys.stderr)
        return p.returncode or 1
    if len(json_lines) != 1:
        print("bench.py: expected ONE JSON line from rank 0, got %d" % len(json_lines), file=sys.stderr)
        return 1
    print(json_lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512, help="timed steps per repetition (512 x 0.66 ms x 3 repetitions: a second of timed GPU work)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--frames-per-gpu", type=int, default=32)   # BASELINE config 4: 256 frames over 8 GPUs
    ap.add_argument("--batches", type=int, default=8, help="distinct resident batches the steps rotate over (8 x 32 = config 4's 256 frames)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host_api / kht objects")
    ap.add_argument("--no-verify", action="store_true", help="experiment only: skip the golden check of all frames (the JSON line says so)")
    ap.add_argument("--no-kernel-events", action="store_true", help="experiment: no per-kernel HIP events in the timed steps")
    ap.add_argument("--event-every", type=int, default=8,
                    help="HIP events around the dominant kernel on every N-th step of a lane in the timed region (1 = every step: ~1 %% slower, the events are stream operations)")
    ap.add_argument("--reps", type=int, default=3, help="repetitions of the timed K-step loop; the MEDIAN repetition is reported")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight: N plans (own buffers) on N HIP streams take the steps in turn, so the latency-bound kernels of one "
                         "batch (key sort, decode, hysteresis rounds) run under the wide kernels of the other; 1 = one stream, kernels never overlap")
    ap.add_argument("--depth", type=int, default=2, help="steps enqueued per batch in flight before the host waits for the oldest one")
    ap.add_argument("--sync-steps", action="store_true", help="experiment: the synchronous step (one host round trip per step)")
    ap.add_argument("--scatter", action="store_true",
                    help="RCCL data path (SURVEY 8e): every step's global batch is born on rank 0 and scattered (grouped send/recv), line counts "
                         "and the strongest lines of every frame are all-gathered; inside the timed region")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) even for a world of one rank: exercises init / barrier / all_reduce / "
                         "all_gather on real hardware where only one GPU is available")
    ap.add_argument("--shared-gpu", action="store_true", help="test mode: every rank uses cuda:0 (two HIP contexts + an RCCL ring on ONE leased GPU)")
    ap.add_argument("--dist-backend", default="nccl")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) and hand rank 0's JSON line through
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    from compv_amd import capi, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- the launcher's --nproc-per-node must equal --gpus "
                         "(a bare `python bench.py --gpus N` starts its own N ranks)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.shared_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or (args.force_dist and "RANK" in os.environ)
    dist = None
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d wants cuda:%d but this node shows %d GPU(s) -- launch with --nproc-per-node <= the visible GPUs"
                             % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        if args.shared_gpu:
            dist.init_process_group(backend=args.dist_backend)
        else:
            dist.init_process_group(backend=args.dist_backend, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist_on else 0)
    comm_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    W, H, F, NB = args.width, args.height, args.frames_per_gpu, max(1, args.batches)
    synth = FrameSynth(torch, dev, W, H)
    # frame f of block b has the global index b * F + f and the seed 12345 + (index mod 256); rank r runs block (r + k) % NB at step k
    block_seeds = sharding.block_seeds(F, NB, GOLDEN_FRAMES)
    blocks = [synth.batch(s) for s in block_seeds]
    if rank == 0:   # the device generator against the host one (one frame: the generators are checked exhaustively in tests/)
        from oracle_bindings import synth_frame
        if not np.array_equal(blocks[0][0].cpu().numpy(), synth_frame(W, H, block_seeds[0][0])):
            raise RuntimeError("device frame generator disagrees with tests/oracle_bindings.synth_frame")
    del synth
    line_cap = 1 << 16
    ctx = capi.Context(local_rank if dist_on else 0)

    def make_lane():
        return {"plan": capi.Plan(ctx, W, H, W, F, THETA_DEG), "edges": torch.empty_like(blocks[0]),
                "lines": torch.zeros((F, line_cap, 5), dtype=torch.int32, device=dev), "counts": torch.zeros(F, dtype=torch.int32, device=dev),
                # a dedicated HIP stream (not the legacy null stream, whose implicit synchronisation costs ~3 % here); the library records
                # its kernel events on this same stream, and the timed region is bracketed by device-wide synchronisations
                "stream": torch.cuda.Stream(device=dev), "in": torch.empty_like(blocks[0]) if args.scatter else None}
    lanes = [make_lane() for _ in range(max(1, 1 if args.sync_steps else args.inflight))]
    plan = lanes[0]["plan"]
    torch.cuda.synchronize()
    TOPK = 64

    def my_block(k):
        return sharding.block_of_step(rank, k, NB)

    def enqueue(q, k):
        """step k on lane q (asynchronous); returns the ticket"""
        d_in = blocks[my_block(k)]
        if args.scatter:
            # the global batch of step k lives on rank 0 (its resident blocks): rank r receives block (r + k) % NB over RCCL
            # (grouped send/recv), ordered on the lane's stream in front of the step's kernels
            with torch.cuda.stream(q["stream"]):
                sharding.scatter_blocks(dist if dist_on else None, lambda r: blocks[sharding.block_of_step(r, k, NB)], q["in"], src=0)
            d_in = q["in"]
        return q["plan"].pipeline_async(d_in.data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, q["edges"].data_ptr(), q["lines"].data_ptr(), line_cap,
                                        q["counts"].data_ptr(), q["stream"].cuda_stream)

    def finish(q, t):
        q["plan"].wait(t)
        if args.scatter:
            with torch.cuda.stream(q["stream"]):
                sharding.gather_lines(dist if dist_on else None, q["counts"], q["lines"][:, :TOPK].contiguous())

    step_no = [0]
    sample_every, sample_mode = [1], [0]   # set once the dominant kernel is known

    def run_steps(k):
        """k steps.  Default: each step is enqueued with compvhip_plan_pipeline_async and waited for while the NEXT one is already
        running (the hysteresis convergence flag is read one step late; a miss replays that step) -- no host round trip per step."""
        if args.sync_steps:
            q = lanes[0]
            for _ in range(k):
                q["plan"].pipeline(blocks[my_block(step_no[0])].data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, q["edges"].data_ptr(), q["lines"].data_ptr(),
                                   line_cap, q["counts"].data_ptr(), q["stream"].cuda_stream)
                step_no[0] += 1
            return
        pend = []
        for _ in range(k):
            q = lanes[step_no[0] % len(lanes)]
            if args.scatter:
                # ONE step in flight per lane: a step overwrites the lane's receive buffer and its result arrays, which must stay untouched until
                # the previous step on that lane has been waited for (a replayed step re-reads its input; the gather reads its results)
                while len(pend) >= len(lanes):
                    finish(*pend.pop(0))
            if sample_every[0] > 1:
                # HIP events ride on every `sample_every`-th step of the timed region only (per lane: every lane is sampled): an event pair is two more
                # operations on the lane's stream, ~6 us each -- around the dominant kernel of EVERY step they cost the headline ~1 %
                q["plan"].set_timing(sample_mode[0] if (step_no[0] // len(lanes)) % sample_every[0] == 0 else 0)
            pend.append((q, enqueue(q, step_no[0])))
            step_no[0] += 1
            if len(pend) > max(1, min(args.depth, 3)) * len(lanes):   # the library keeps at most 4 steps of a plan in flight
                finish(*pend.pop(0))
        for q0, t0 in pend:
            finish(q0, t0)

    run_steps(args.warmup)
    torch.cuda.synchronize()
    # which kernel dominates a step?  One extra untimed, fully instrumented step decides which single kernel carries HIP events
    # during the timed steps (events around every launch would cost ~0.1 ms per step, around two kernels ~0.04 ms).
    def one_sync_step(q, b):
        q["plan"].pipeline(blocks[b].data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, q["edges"].data_ptr(), q["lines"].data_ptr(), line_cap,
                           q["counts"].data_ptr(), q["stream"].cuda_stream)

    dominant = "sht_vote_kernel"
    if not args.no_kernel_events:
        plan.set_timing(1)
        one_sync_step(lanes[0], 0)
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
    if not args.no_kernel_events and not args.sync_steps:
        sample_every[0], sample_mode[0] = max(1, args.event_every), timed_mode
    per_kernel = {}

    def collect(dst, plans=None):
        for pl in (plans or [q["plan"] for q in lanes]):
            for name, ms in pl.get_timing():
                a = dst.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += 1

    # The timed region is EXACTLY K steps between barrier + synchronize brackets; it is repeated `reps` times inside this run and
    # the MEDIAN repetition is reported (one 16-step region lasts ~15 ms: a single one is a thin measurement).
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
        rep_elapsed.append(sharding.max_over_ranks(e, dist if dist_on else None, comm_dev))
        collect(per_kernel)   # the dominant kernel's events of this repetition's K steps (read after the closing bracket)
    elapsed = sorted(rep_elapsed)[len(rep_elapsed) // 2]

    # second pass, untimed: ONE batch at a time on ONE stream with HIP events around every launch -- the per-kernel durations a kernel
    # has when it owns the GPU (in the timed region two batches are in flight and kernels of different batches share the CUs)
    sample_every[0] = 1
    for q in lanes:
        q["plan"].set_timing(0)
    breakdown = {}
    iso_steps = min(args.steps, 64)
    if rank == 0 and not args.no_kernel_events:
        plan.set_timing(1)
        for k in range(iso_steps):
            one_sync_step(lanes[0], k % NB)
            collect(breakdown, [plan])
    plan.set_timing(0)

    # ---- verification: every frame of every resident batch against the reference-derived fixture, in the mode the timed steps use ----
    golden = None
    verify_note = None
    gpath = os.path.join(ROOT, "tests", "golden", "golden_batch.json")
    if args.no_verify:
        verify_note = "skipped (--no-verify)"
    elif (W, H) != (3840, 2160) or not os.path.exists(gpath):
        verify_note = "no reference-derived fixture for this geometry: outputs of the two lanes compared with each other only"
    else:
        golden = json.load(open(gpath))
        assert (golden["W"], golden["H"], golden["tLow"], golden["tHigh"], golden["threshold"]) == (W, H, T_LOW, T_HIGH, SHT_THRESHOLD)
        golden = {g["seed"]: g for g in golden["frames"]}
    total_edges = 0
    total_lines = 0
    lines_frame0 = None
    checked = 0
    for b0 in range(0, NB, len(lanes)):
        todo = [(lanes[i], b0 + i) for i in range(len(lanes)) if b0 + i < NB]
        if args.sync_steps:
            for q, b in todo:
                one_sync_step(q, b)
        else:
            tickets = [(q, q["plan"].pipeline_async(blocks[b].data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, q["edges"].data_ptr(), q["lines"].data_ptr(), line_cap,
                                                    q["counts"].data_ptr(), q["stream"].cuda_stream)) for q, b in todo]   # both lanes in flight, as in the timed steps
            for q, t in tickets:
                q["plan"].wait(t)
        torch.cuda.synchronize()
        for q, b in todo:
            c, ne, nl = check_batch(torch, q, block_seeds[b], golden, W, H, line_cap, dev)
            checked += c
            total_edges += ne
            total_lines += nl
            if b == 0:
                lines_frame0 = int(q["counts"][0].item())
    if golden is None and len(lanes) > 1 and not args.no_verify:
        # no fixture: at least the lanes must agree on one common batch
        for q in lanes:
            one_sync_step(q, 0)
        torch.cuda.synchronize()
        for q in lanes[1:]:
            if not torch.equal(q["counts"], lanes[0]["counts"]) or not torch.equal(q["edges"], lanes[0]["edges"]):
                raise RuntimeError("the batches in flight disagree")

    # the only result exchange of the default job: all-gather of the tiny per-frame line counts, outside the timed region
    try:
        all_counts = sharding.gather_frame_results([int(c) for c in lanes[0]["counts"].cpu().numpy()], dist if dist_on else None, comm_dev)
    except Exception:  # reporting only: never let it hide the throughput number
        all_counts = None
    # what a SCALE record can be checked against: every rank reports itself through an all-gather, and the collective library's version
    try:
        seen = sharding.ranks_seen(dist if dist_on else None, comm_dev)
    except Exception:  # reporting only
        seen = []
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist_on and args.dist_backend == "nccl" else None
    except Exception:
        rccl_version = None
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
        counters, counters_src = committed_counters(W, H, F)

        def counter(name, key):
            if counters and name in counters["kernels"] and key in counters["kernels"][name]:
                return counters["kernels"][name][key]
            return None

        def roof(name, nbytes, ms=None):
            if ms is None and name not in kern:
                return None
            ms = kern[name]["ms_per_launch"] if ms is None else ms
            ach = nbytes / (ms * 1e-3) / 1e9
            tr = counter(name, "hbm_bytes")
            return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tr, "traffic_source": counters_src,
                    "ms_per_launch": round(ms, 4), "algorithmic_bytes_per_launch": int(nbytes)}

        def valu_floor(name, ms):
            """issue floor of the kernel's own instruction mix: SQ_INSTS_VALU (PMC, per launch) x cycles per wave64 instruction
            (tools/isa_cost.py over the kernel's main loop with the microbenchmark costs 2.3 / 4.3) / SIMDs / clock"""
            n = counter(name, "SQ_INSTS_VALU")
            cpi = counter(name, "valu_cycles_per_instruction")
            if n is None or cpi is None:
                return None
            floor_ms = n * cpi / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
            return {"SQ_INSTS_VALU": int(n), "cycles_per_wave_instruction": cpi, "simds": SIMDS, "clock_ghz": CLOCK_GHZ,
                    "floor_ms": round(floor_ms, 4), "frac": round(floor_ms / ms, 4)}
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
                roofline["timed_region_launches_sampled"] = int(per_kernel[dom][1]) if dom in per_kernel else 0
                roofline["timed_region_sampling"] = "HIP events around this kernel on every %d-th step of each lane (--event-every)" % max(1, args.event_every)
            else:
                roofline = roof(dom, alg.get(dom, F * W * H * 1.0))
                roofline["timing"] = "HIP events on the launch stream around this kernel in every timed step"
            if dom == "sht_vote_kernel":
                # what actually bounds it: one ds_add_u32 wave-instruction (64 votes) per 4.1 LDS cycles per CU when conflict-free
                # (tools/microbench/lds_atomic_bench2), 256 CUs
                votes = float(total_edges) / NB * T
                floor_ms = votes / 64.0 * 4.1 / 256.0 / 2.4e9 * 1e3
                roofline["lds_atomic_roofline"] = {"votes_per_launch": int(votes), "cycles_per_wave_instruction": 4.1, "cus": 256, "clock_ghz": 2.4,
                                                   "floor_ms": round(floor_ms, 4), "frac": round(floor_ms / roofline["ms_per_launch"], 4)}
                roofline["note"] = ("the voting kernel is bound by the LDS atomic pipe, not by HBM (lds_atomic_roofline); the HBM fraction is reported "
                                    "because the contract prices every kernel of this path against the HBM roofline")
            vf = valu_floor(dom, roofline["ms_per_launch"])
            if vf:
                roofline["valu_issue"] = vf
        rc = None
        if "canny_tile_kernel" in iso:
            rc = roof("canny_tile_kernel", alg["canny_tile_kernel"], iso["canny_tile_kernel"])
            rc["timing"] = "HIP events in the single-stream instrumented pass after the timed steps"
        if rc:
            rc["frac_read_plus_write"] = round(2 * rc["frac"], 4)
            vf = valu_floor("canny_tile_kernel", rc["ms_per_launch"])
            if vf:
                rc["valu_issue"] = vf
        def stage_roof(names, nbytes, what):
            ms = sum(v[0] for k, v in breakdown.items() if k in names) / max(iso_steps, 1)
            if ms <= 0:
                return None
            ach = nbytes / (ms * 1e-3) / 1e9
            return {"stage": what, "kernels": sorted(k for k in breakdown if k in names), "bound": "hbm", "ms_per_step": round(ms, 4), "algorithmic_bytes": int(nbytes),
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "timing": "sum of the stage's launches in the single-stream instrumented pass (HIP events), one batch per step"}
        canny_stage = {"canny_tile_kernel", "canny_resolve_kernel", "canny_mean_thresholds", "otsu_kernels"}
        sht_stage = {k for k in breakdown if k.startswith("sht_")}
        out = {
            "metric": "Mpixels/s Sobel->Canny->HoughSHT on 4K uint8",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "reps": len(rep_elapsed), "reps_ms_per_step": [round(e / args.steps * 1e3, 4) for e in rep_elapsed],
            "timing": "median of %d repetitions of the K-step region (each: barrier + synchronize, K steps, synchronize + barrier, MAX over ranks)" % len(rep_elapsed),
            "step_mode": "synchronous (host reads the hysteresis flag every step)" if args.sync_steps else
                         "pipelined (compvhip_plan_pipeline_async: step k's hysteresis flag is read while step k+1 runs), %d batch(es) in flight "
                         "(one plan + HIP stream each, steps dealt round-robin)" % len(lanes),
            "batches_in_flight": len(lanes),
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "dist_backend": (dist.get_backend() if dist_on else None),
            "rccl_version": rccl_version, "n_ranks_seen": len(seen), "ranks_seen": seen,
            "config": {"workload": "batched %dx%d uint8 frames, Sobel3x3 -> Canny(59,119) -> HoughSHT(rho=1, theta=1deg, thr=100); %d distinct frames "
                                   "resident per GPU as %d batches of %d, a step processes one batch, steps rotate over the batches"
                                   % (W, H, min(NB * F, GOLDEN_FRAMES), NB, F),
                       "frames_per_gpu": F, "resident_batches": NB, "global_frames_per_step": world * F,
                       "parallelism": ("frames sharded across %d GPU(s), no data-path collective" % world) if not args.scatter else
                                      ("step batch scattered from rank 0 over %s (grouped send/recv), line counts + top-%d lines all-gathered" % (args.dist_backend, TOPK))},
            "verified": ({"frames_checked": checked, "against": "tests/golden/golden_batch.json (real CompV: edge-map MD5, edge count, line count, strength sum, line-set hash)",
                          "mode": "two lanes in flight, asynchronous steps" if overlapped else "one lane"} if golden is not None else verify_note),
            "roofline": roofline, "roofline_canny": rc,
            # the same algorithmic bytes priced against the whole STAGE (every launch of it), not one kernel
            "roofline_stage_canny": stage_roof(canny_stage, alg["canny_tile_kernel"], "Sobel -> NMS -> hysteresis (tile kernel + resolve rounds), 1 B/px read"),
            "roofline_stage_sht": stage_roof(sht_stage, alg["sht_vote_kernel"], "edge compaction -> voting -> reduce -> NMS / lines -> sort -> decode, W*H + R*T*4 bytes per frame"),
            # every kernel of a step, from the instrumented pass AFTER the timed steps (same process, same buffers)
            "kernels_ms_per_step": {k: round(v[0] / iso_steps, 4) for k, v in sorted(breakdown.items())},
            "kernels_ms_per_step_source": "second pass of %d steps, one batch at a time on one stream, HIP events around every launch (not in the timed region)" % iso_steps,
            "edge_pixels_per_batch": total_edges // NB,
            "lines_frame0": lines_frame0,
            "lines_per_batch": total_lines // NB,
            "lines_last_step_all_ranks": (int(sum(all_counts)) if all_counts is not None else None),
        }
        if world == 1 and not args.no_extras:
            try:
                out["host_api"] = host_api_latency(capi, local_rank if dist_on else 0, blocks[0][0].cpu().numpy())
            except Exception as e:
                out["host_api"] = {"error": str(e)}
            try:
                out["kht"] = kht_figure(capi, ctx, torch, lanes[0], blocks, W, H, F)
            except Exception as e:
                out["kht"] = {"error": str(e)}
            try:
                out["kernels_extra"] = kernels_extra(capi, torch, lanes[0], blocks, W, H, F)
            except Exception as e:
                out["kernels_extra"] = {"error": str(e)}
        if world == 1 and not args.no_extras and (W, H) == (3840, 2160):
            try:
                out["configs_extra"] = {"fhd_1920x1080": extra_config(torch, capi, sharding, ctx, dev, 1920, 1080, 32, 2, max(64, args.steps // 2), args.warmup,
                                                                        args.reps, os.path.join(ROOT, "tests", "golden", "golden_batch_fhd.json"))}
            except Exception as e:
                out["configs_extra"] = {"error": str(e)}
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


def kernels_extra(capi, torch, lane, blocks, W, H, F):
    """The kernels of the path that the benchmark step does not run: Canny with the 5x5 Sobel and the Sobel detector, one batch, HIP events
    of the plan's timing API, priced like roofline_canny (1 B/px read for the fused Canny tile kernel, 3 B/px -- two reads, one write -- for the detector)."""
    q = lane
    plan, st = q["plan"], q["stream"].cuda_stream
    px = F * W * H

    def timed(call, n=8):
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        before = plan.timing_mode
        plan.set_timing(1)
        acc = {}
        for _ in range(n):
            call()
            torch.cuda.synchronize()
            for name, ms in plan.get_timing():
                acc[name] = acc.get(name, 0.0) + ms
        plan.set_timing(before)   # (whatever mode the caller had set)
        return {k: v / n for k, v in acc.items()}

    out = {}
    # frame 0 of block 0 is (3840 x 2160, seed 12345) = fixture uhd_3840x2160 of tests/golden/golden.json (real CompV): what is timed here is also checked
    gold = None
    if (W, H) == (3840, 2160):
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["uhd_3840x2160"]
        except Exception:
            gold = None
    scratch = torch.empty_like(blocks[0])   # (never the lane's own edge maps)

    def md5_frame0(t):
        return hashlib.md5(np.ascontiguousarray(t[0].cpu().numpy()).tobytes()).hexdigest()

    # the 5x5 Sobel answers a step edge 12 x as strongly as the 3x3 one (16 * 3 against 4 * 1): thresholds scaled to the benchmark's edge density, and unscaled
    for name, tl, th, key in (("canny5_same_edge_density", 12 * T_LOW, 12 * T_HIGH, "canny5_x12"), ("canny5_benchmark_thresholds", T_LOW, T_HIGH, "canny5")):
        per = timed(lambda: plan.canny(blocks[0].data_ptr(), tl, th, scratch.data_ptr(), ksize=5, stream=st))
        tile, res = per.get("canny_tile_kernel", 0.0), per.get("canny_resolve_kernel", 0.0)
        verified = "no fixture"
        if gold is not None and key in gold:
            g = gold[key]
            got = (md5_frame0(scratch), int((scratch[0] != 0).sum().item()))
            if (g["tLow"], g["tHigh"]) != (tl, th) or got != (g["md5"], g["edges"]):
                raise RuntimeError("kernels_extra %s: frame 0 gives (md5 %s, %d edges), the reference fixture says (%s, %d)" % (name, got[0], got[1], g["md5"], g["edges"]))
            verified = "frame 0 MD5 + edge count = tests/golden/golden.json uhd_3840x2160/%s (real CompV)" % key
        out[name] = {"thresholds": [tl, th], "edge_pixels": int((scratch != 0).sum().item()), "verified": verified,
                     "canny_tile_kernel_ms": round(tile, 4), "canny_resolve_kernel_ms": round(res, 4),
                     "stage_ms": round(tile + res, 4), "roofline": {"bound": "hbm", "achieved": round(px / (tile * 1e-3) / 1e9, 1) if tile else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                                    "frac": round(px / (tile * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tile else None, "basis": "1 B/px read, tile kernel"}}
    d_out = scratch
    for opname, op, key in (("scharr", capi.OP_SCHARR, "scharr_md5"), ("prewitt", capi.OP_PREWITT, "prewitt_md5")):
        if gold is not None and key in gold:
            plan.edge_dete(blocks[0].data_ptr(), op, d_out.data_ptr(), st)
            torch.cuda.synchronize()
            if md5_frame0(d_out) != gold[key]:
                raise RuntimeError("kernels_extra edge_dete %s: frame 0 differs from the reference fixture" % opname)
    per = timed(lambda: plan.edge_dete(blocks[0].data_ptr(), capi.OP_SOBEL, d_out.data_ptr(), st))
    ms = sum(per.values())
    verified = "no fixture"
    if gold is not None:
        if md5_frame0(d_out) != gold["sobel_md5"]:
            raise RuntimeError("kernels_extra edge_dete sobel: frame 0 differs from the reference fixture")
        verified = "frame 0 MD5 (Sobel timed; Scharr, Prewitt run once) = tests/golden/golden.json uhd_3840x2160 (real CompV)"
    out["edge_dete_sobel"] = {"ms": round(ms, 4), "kernels": {k: round(v, 4) for k, v in per.items()}, "verified": verified,
                              "roofline": {"bound": "hbm", "achieved": round(3.0 * px / (ms * 1e-3) / 1e9, 1) if ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(3.0 * px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None, "basis": "3 B/px: gmax pass read + output pass read + write"}}
    return out


def kht_list_hash(lines):
    """tests/golden/make_golden_batch.py::kht_list_hash over a LINE_DTYPE array (order-dependent)."""
    rb = lines["rho"].astype(np.float32).view(np.uint32).tolist()
    tb = lines["theta"].astype(np.float32).view(np.uint32).tolist()
    st = lines["strength"].astype(np.int64).tolist()
    h = 0
    for r, t, sv in zip(rb, tb, st):
        h = (h * 1000003 + r * 7919 + t * 31337 + (sv & M64)) & M64
    return h


def kht_figure(capi, ctx, torch, lane, blocks, W, H, F):
    """BASELINE config 5 as a throughput figure: the batched KHT path on the edge maps of one resident batch."""
    if not hasattr(capi.Plan, "houghkht"):
        return {"error": "batched KHT not built"}
    q = lane
    q["plan"].pipeline(blocks[0].data_ptr(), T_LOW, T_HIGH, SHT_THRESHOLD, 0, q["edges"].data_ptr(), q["lines"].data_ptr(), q["lines"].shape[1],
                       q["counts"].data_ptr(), q["stream"].cuda_stream)
    torch.cuda.synchronize()
    res = q["plan"].houghkht(q["edges"].data_ptr(), 1.0, THETA_DEG, 1)          # warm-up (allocations, thread pool)
    dts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = q["plan"].houghkht(q["edges"].data_ptr(), 1.0, THETA_DEG, 1)
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]
    stages = q["plan"].houghkht_stage_ms()
    kht_verified = "no fixture"
    fixture = os.path.join(ROOT, "tests", "golden", "golden_batch_kht.json")
    if (W, H) == (3840, 2160) and os.path.exists(fixture):
        # every frame of the timed call against the real CompV KHT (tests/golden/make_golden_batch.py kht): count, strength sum, GS, and the list IN ORDER
        gk = json.load(open(fixture))
        recs = {g["seed"]: g for g in gk["frames"]}
        checked = 0
        for f in range(F):
            g = recs.get(12345 + f)
            if g is None:
                continue
            l, gs = res[0][f], res[1][f]
            got = (len(l), int(l["strength"].astype(np.int64).sum()), repr(gs), "%016x" % kht_list_hash(l))
            if got != (g["lines"], g["sum_strength"], g["gs"], g["list_hash"]):
                raise RuntimeError("kht_figure: frame %d gives %s, the reference fixture says %s" % (f, got, (g["lines"], g["sum_strength"], g["gs"], g["list_hash"])))
            checked += 1
        kht_verified = {"frames_checked": checked, "against": "tests/golden/golden_batch_kht.json (real CompV): line count, strength sum, GS, ordered list hash"}
    # the same call with fewer host threads (the default above is min(32, hardware threads / 2)): how much of the figure is the host pool
    by_threads = {}
    for nt in (1, 8, 16, 32):
        try:
            q["plan"].houghkht(q["edges"].data_ptr(), 1.0, THETA_DEG, 1, threads=nt)
            ts = []
            for _ in range(1 if nt == 1 else 3):
                t0 = time.perf_counter()
                q["plan"].houghkht(q["edges"].data_ptr(), 1.0, THETA_DEG, 1, threads=nt)
                ts.append(time.perf_counter() - t0)
            by_threads[str(nt)] = round(sorted(ts)[len(ts) // 2] * 1e3 / F, 3)
        except Exception as e:  # reporting only
            by_threads[str(nt)] = "error: %s" % e
    # BASELINE config 5 asks for HBM GB/s: SURVEY 8(d)'s algorithmic bytes of the KHT (the edge map read once + the vote map written once) against the time of the
    # stages that touch the GPU (host clocks of the issuing threads, single-frame call: upload + kernels + download + their synchronisations)
    roof = None
    try:
        ax = capi.houghkht_dims(W, H, 1.0, THETA_DEG) if hasattr(capi, "houghkht_dims") else None
        if ax:
            T_, rhoN = ax
            alg = W * H + (T_ + 2) * (rhoN + 2) * 4
            one = ctx.houghkht(q["edges"][0].cpu().numpy(), 1.0, THETA_DEG, 1)
            sm = ctx.houghkht_stage_ms()            # link, subdivide, statistics, prune, vote + peaks, sort + sweep
            gpu_ms = float(sm[1] + sm[2] + sm[4])
            traffic = None
            prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            have = sorted(d for d in os.listdir(prof) if os.path.exists(os.path.join(prof, d, "kht_traffic.json")))
            tj = os.path.join(prof, have[-1], "kht_traffic.json") if have else ""
            if tj:
                tdat = json.load(open(tj))
                traffic = tdat.get("hbm_bytes_per_call") if tdat.get("so_sha256") == so_sha256() else None
            roof = {"bound": "hbm", "algorithmic_bytes_per_frame": alg, "gpu_stage_ms_single_frame": round(gpu_ms, 4),
                    "achieved": round(alg / (gpu_ms * 1e-3) / 1e9, 2) if gpu_ms > 0 else None, "peak": 8000.0, "unit": "GB/s",
                    "frac": round(alg / (gpu_ms * 1e-3) / 1e9 / 8000.0, 5) if gpu_ms > 0 else None, "traffic": traffic,
                    "note": "the KHT is latency-bound, not bandwidth-bound: its GPU stages move a few MB per frame (profiles/r0*/kht_pmc*); W*H + (T+2)(rhoN+2)*4 bytes per frame "
                            "over the wall time of the subdivision + statistics + voting/peaks stages of one single-frame call (uploads, kernels, downloads, synchronisations)"}
    except Exception as e:  # reporting only
        roof = {"error": str(e)}
    cpu = None
    try:
        from oracle_bindings import RefShim, have_refshim
        if have_refshim():
            # CompV's own KHT (tests/image/houghkht.cxx loops process() on one object the same way) on this batch's first edge maps, at its best thread count
            maps = q["edges"][:4].cpu().numpy().copy()
            sweep = {}
            ref = None
            for t in (1, 8, -1):
                if ref is None:
                    ref = RefShim(t)
                else:
                    ref.reinit(t)
                ref.bench_kht(maps[:1], 1.0, THETA_DEG, 1)
                ms, nl = ref.bench_kht(maps, 1.0, THETA_DEG, 1)
                sweep[ref.threads] = round(ms / len(maps), 3)
            best = min(sweep, key=sweep.get)
            cpu = {"ms_per_frame": sweep[best], "cores": min(best, effective_cpus()), "threads": best, "kind": "reference", "ms_per_frame_by_threads": sweep, "lines_4_frames": int(nl),
                   "sample": "%d of the batch's %dx%d edge maps, CompVHoughKht::process (AVX2 / SSE intrinsics path, COMPV_ASM=0), best of the thread sweep; "
                             "single-frame LATENCY -- the throughput comparison at equal host resources is frame_parallel" % (len(maps), W, H)}
            # equal resources: P processes x 1 thread of the real CompVHoughKht::process on independent edge maps, behind one barrier
            # (tests/image/houghkht.cxx times process() in a loop the same way); thread-ms per frame on both sides
            try:
                cores = os.cpu_count() or 1
                fp = frame_parallel_baseline(W, H, 1, cores, work_s=3.0, procs=32, task="kht")
                cpu["frame_parallel"] = fp
            except Exception as e:
                cpu["frame_parallel"] = {"error": str(e)}
    except Exception as e:  # reporting only
        cpu = {"error": str(e)}
    nthreads = stages.get("threads") or 0
    return {"ms_per_frame": round(dt * 1e3 / F, 3), "ms_per_frame_calls": [round(d * 1e3 / F, 3) for d in dts], "frames": F, "lines_frame0": int(len(res[0][0])),
            "verified": kht_verified,
            "thread_ms_per_frame": round(dt * 1e3 / F * nthreads, 2) if nthreads else None,
            "ms_per_frame_by_host_threads": by_threads,
            "roofline": roof,
            "cpu_baseline": cpu,
            "host_threads": stages.get("threads"),
            "host_cpu_budget": capi.host_cpu_budget() if hasattr(capi, "host_cpu_budget") else None,   # what the default worker count is sized by (compvhip_host_cpu_budget)
            "host_share": stages.get("host_share"), "stages_ms_per_frame": stages.get("stages"),
            "note": "compvhip_plan_houghkht on the device edge maps of one batch: groups of 8 frames, host stages (linking, prune, sort + sweep) as parallel loops over a group's frames, "
                    "every GPU stage (subdivision, statistics, voting, peaks) ONE launch per group, up to 4 groups in flight; stages_ms_per_frame: host stages = thread time per frame, "
                    "GPU stages = wall time of the group's stage / its frames"}


if __name__ == "__main__":
    main()
