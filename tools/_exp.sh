mkdir -p gpurun_out/x
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_integration_plugin.py -x -q -m gpu > gpurun_out/x/pytest.log 2>&1; tail -3 gpurun_out/x/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-extras --inflight 1 --steps 8 --warmup 2 --reps 2 > gpurun_out/x/i1.json 2> gpurun_out/x/i1.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 4 --reps 3 > gpurun_out/x/i2.json 2> gpurun_out/x/i2.err
python - <<'PY'
import json
for m in ("i1","i2"):
    try:
        d=json.loads(open("gpurun_out/x/%s.json"%m).read().strip().splitlines()[-1])
        print(m,d["ms_per_step"],d["value"],(d.get("verified") or {}).get("frames_checked"),d["kernels_ms_per_step"])
    except Exception as ex: print(m,"FAILED",ex); print(open("gpurun_out/x/%s.err"%m).read()[-1500:])
PY
