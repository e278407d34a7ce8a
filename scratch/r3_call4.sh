#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python scratch/e2e_lab.py 16000000 2000 > gpurun_out/c4_lab.log 2>&1; echo "lab rc=$?"
grep -E "^==|device parser|wrote|written|gzip" gpurun_out/c4_lab.log | cut -c1-330
grep -E "throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null
timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err; echo "bench rc=$?"
tail -12 gpurun_out/c4_bench.err
python - <<'P'
import json
try:
    b = json.loads(open("gpurun_out/c4_bench.json").read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "breakdown_ms", "cpu_baseline", "parity_check", "parity_check_tail", "pinned_pipeline", "end_to_end"):
        print(k, json.dumps(b.get(k))[:1500])
except Exception as e:
    print("no bench line", e)
P
