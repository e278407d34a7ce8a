#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_bus_tcc.py tests/test_gpu_cli.py -x -q > gpurun_out/c22_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c22_tests.log
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pinned-pipeline --end-to-end 0"
timeout 600 $B > gpurun_out/c22_bench.json 2> gpurun_out/c22_bench.err
python - <<'P'
import json
try:
    b=json.loads(open('gpurun_out/c22_bench.json').read().strip().splitlines()[-1]); pc=b.get("parity_check") or {}
    print(b["value"], b["ms_per_step"], json.dumps(b["breakdown_ms"]), "parity", pc.get("ok"), (b.get("parity_check_tail") or {}).get("ok"))
except Exception as ex: print('failed', ex); print(open('gpurun_out/c22_bench.err').read()[-600:])
P
