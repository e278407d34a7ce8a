#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c9_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/c9_tests.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/c9_bench.json 2> gpurun_out/c9_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/c9_bench.err
python - <<'P'
import json
try:
    b = json.loads(open("gpurun_out/c9_bench.json").read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "breakdown_ms", "roofline", "roofline_em", "roofline_finalize", "cpu_baseline", "parity_check", "parity_check_tail", "pinned_pipeline", "end_to_end"):
        v = b.get(k)
        if isinstance(v, dict): v = {kk: vv for kk, vv in v.items() if kk not in ("traffic_source", "note", "launch", "reference", "tolerance", "sample")}
        print(k, json.dumps(v)[:1300])
except Exception as e:
    print("no bench line", e)
P
