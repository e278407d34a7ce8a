#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c7_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/c7_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --end-to-end 0 > gpurun_out/c7_bench.json 2> gpurun_out/c7_bench.err; echo "bench rc=$?"
python - <<'P'
import json
try:
    b = json.loads(open("gpurun_out/c7_bench.json").read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "breakdown_ms", "counters", "roofline_finalize", "pinned_pipeline"):
        print(k, json.dumps(b.get(k))[:900])
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/c7_bench.err").read()[-1500:])
P
