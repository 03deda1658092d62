mkdir -p gpurun_out/c1
for d in 0 1 2 4; do
CDBG=$d timeout 300 python bench.py --no-cpu-baseline --no-extras --no-verify --inflight 1 --steps 8 --warmup 2 --reps 2 > gpurun_out/c1/i1_$d.json 2> gpurun_out/c1/i1_$d.err
done
python - <<'PY'
import json
for d in (0,1,2,4):
    try:
        x=json.loads(open("gpurun_out/c1/i1_%d.json"%d).read().strip().splitlines()[-1])
        print(d,x["ms_per_step"],x["kernels_ms_per_step"]["sht_compact_kernel"])
    except Exception as ex: print(d,"FAILED",ex); print(open("gpurun_out/c1/i1_%d.err"%d).read()[-600:])
PY
