#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_bus_tcc.py tests/test_gpu_fullsize.py -x -q > gpurun_out/c21_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c21_tests.log
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pinned-pipeline --end-to-end 0"
for w in 1 0; do
  KAMD_RESOLVE_WORK=$w timeout 600 $B > gpurun_out/c21_w$w.json 2> gpurun_out/c21_w$w.err
  python - "$w" <<'P'
import json,sys
w=sys.argv[1]
try:
    b=json.loads(open(f'gpurun_out/c21_w{w}.json').read().strip().splitlines()[-1]); pc=b.get("parity_check") or {}
    print("resolve_work", w, b["value"], b["ms_per_step"], json.dumps(b["breakdown_ms"]), "parity", pc.get("ok"), (b.get("parity_check_tail") or {}).get("ok"))
except Exception as ex: print(w, 'failed', ex); print(open(f'gpurun_out/c21_w{w}.err').read()[-600:])
P
done
