#!/bin/bash
# state check after the k_fq_pack rewrite: the GPU suite, then the bench line without the CPU baseline
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c12_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/c12_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c12_bench.json 2> gpurun_out/c12_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/c12_bench.err
python - <<'P'
import json
try:
    b = json.loads(open("gpurun_out/c12_bench.json").read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "breakdown_ms", "roofline_finalize", "parity_check", "parity_check_tail", "pinned_pipeline", "end_to_end"):
        v = b.get(k)
        if isinstance(v, dict): v = {kk: vv for kk, vv in v.items() if kk not in ("traffic_source", "note", "launch", "reference", "tolerance", "sample")}
        print(k, json.dumps(v)[:1800])
except Exception as e:
    print("no bench line", e)
P
