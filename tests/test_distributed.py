"""CPU tests of the N > 1 path (gloo, world_size 2): frame sharding, barrier-bracketed timing with MAX over ranks,
gather of per-frame results."""
import json
import os
import subprocess
import sys

import numpy as np

from compv_amd import sharding
from oracle_bindings import synth_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_every_frame_once():
    for total in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                seen.extend(sharding.shard_range(total, world, r))
            assert seen == list(range(total))
            sizes = [len(sharding.shard_range(total, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo(tmp_path, oracle):
    out = tmp_path / "dist.json"
    frames = 5
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), str(frames)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["world"] == 2
    # the gathered per-frame results equal a single-process run over the whole batch, in frame order
    exp = []
    for f in range(frames):
        rc, e = oracle.canny(synth_frame(160, 120, sharding.frame_seed(f)), 59.0, 119.0)
        exp.append(int((e != 0).sum()))
    assert res["counts"] == exp
    assert res["tmax"] >= res["elapsed_rank0"] - 1e-6 and res["tmax"] >= 0.1   # rank 1 sleeps longer: MAX picks it
    # --scatter data path: every rank received exactly its block; counts / top-K lines of all ranks arrive in rank order
    assert res["scatter_ok_all_ranks"] == 1
    assert res["gathered_counts"] == [0, 1, 2, 1000, 1001, 1002]
    assert res["gathered_lines_shape"] == [6, 4, 5] and res["gathered_lines_first"] == [0, 100000]
    assert res["ranks_seen"] == [0, 1]


def test_block_rotation_of_a_world_of_eight_covers_config_4():
    """bench.py --gpus 8 without a node: the rank arithmetic of the driver's scaling run.  Rank r runs resident block (r + k) mod 8 at step k
    (sharding.block_of_step); the blocks hold config 4's 256 distinct frames (sharding.block_seeds); every rank verifies all 8 of its resident
    blocks after the timed region (bench.py's verification loop runs over range(NB) on every rank)."""
    from compv_amd import sharding
    world, F, NB = 8, 32, 8
    seeds = sharding.block_seeds(F, NB, 256)
    flat = [s for b in seeds for s in b]
    assert sorted(flat) == list(range(sharding.BASE_SEED, sharding.BASE_SEED + 256))           # 256 distinct frames, the fixture's seeds
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_batch.json")))
    assert {g["seed"] for g in golden["frames"]} == set(flat)                                  # every one of them has a reference-derived record
    for k in range(3 * NB):
        blocks = [sharding.block_of_step(r, k, NB) for r in range(world)]
        assert sorted(blocks) == list(range(NB))                                               # one step of the job = all 256 frames, each exactly once
    for r in range(world):
        mine = [sharding.block_of_step(r, k, NB) for k in range(NB)]
        assert sorted(mine) == list(range(NB))                                                 # a rank meets every block within 8 steps
        assert all(mine[k] != mine[k + 1] for k in range(NB - 1))                              # and never re-reads the block it has just processed
    # fewer ranks than blocks (1, 2, 4 GPUs): the ranks' blocks of a step are still pairwise different
    for w in (1, 2, 4):
        for k in range(NB):
            bl = [sharding.block_of_step(r, k, NB) for r in range(w)]
            assert len(set(bl)) == w
    # the verification loop of bench.py: lanes take blocks b0, b0 + 1 in turn until all NB are checked -- on EVERY rank
    lanes = 2
    checked = []
    for b0 in range(0, NB, lanes):
        checked += [b0 + i for i in range(lanes) if b0 + i < NB]
    assert checked == list(range(NB))


def test_ranks_seen_single_process():
    from compv_amd import sharding
    assert sharding.ranks_seen(None) == [0]


def test_bare_bench_gpus_8_builds_the_eight_rank_launch():
    """`python bench.py --gpus 8` without a launcher must start 8 ranks itself.  The argv it re-executes through (no GPU needed):
    torch.distributed.run, one node, 8 processes, rendezvous on 127.0.0.1 and a free port, every user argument forwarded unchanged."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    user = ["--gpus", "8", "--steps", "64", "--warmup", "4", "--scatter"]
    port = bench.free_port()
    assert 1024 < port < 65536
    argv = bench.launch_argv(8, user, port)
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv
    assert argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[argv.index("--master-port") + 1] == str(port)
    script = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[script + 1:] == user                       # forwarded verbatim, --gpus 8 included (every rank checks it against WORLD_SIZE)


def test_bench_refuses_a_world_that_is_not_gpus():
    """A launcher started with --nproc-per-node 2 but `--gpus 4` on the command line is a mistake the JSON line would hide: non-zero exit naming both
    numbers, before anything touches a GPU."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr, r.stderr[-1000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
