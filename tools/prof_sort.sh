mkdir -p gpurun_out/r5d
cd /tmp && export TMPDIR=/tmp
for cfg in "3840 2160 8" "1920 1080 2"; do set -- $cfg
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5d/prof_$1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-verify --inflight 1 --steps 6 --warmup 2 --reps 1 --width $1 --height $2 --batches $3 > $GRAFT_REPO_ROOT/gpurun_out/r5d/prof_$1.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for d in ("gpurun_out/r5d/prof_3840","gpurun_out/r5d/prof_1920"):
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        print(f)
        for i, row in enumerate(csv.reader(open(f))):
            if "compvhip" in row[0] or "rocprim" in row[0] or i == 0: print(",".join(x[:70] for x in row[:4]))
PY
