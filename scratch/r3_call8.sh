#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c8_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/c8_tests.log
for v in 1 4 8 16; do
  KAMD_ALIGN_CHUNKS=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline > gpurun_out/c8_b_$v.json 2> gpurun_out/c8_b_$v.err
  python - "$v" <<'P'
import json, sys
v = sys.argv[1]
try:
    b = json.loads(open(f"gpurun_out/c8_b_{v}.json").read().strip().splitlines()[-1])
    print(f"chunks={v}: value {b['value']} step {b['ms_per_step']} {json.dumps(b['breakdown_ms'])}")
except Exception as e:
    print(v, "failed", e); print(open(f"gpurun_out/c8_b_{v}.err").read()[-800:])
P
done
