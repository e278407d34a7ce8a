#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_bus_tcc.py tests/test_gpu_cli.py tests/test_gpu_func_tests.py -x -q > gpurun_out/c24_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c24_tests.log
KAMD_EM_PLAN_STEPS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "em_ or reproducible" > gpurun_out/c24_tests_steps.log 2>&1; echo "tests (step kernels) rc=$?"; tail -2 gpurun_out/c24_tests_steps.log
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pinned-pipeline --end-to-end 0"
for v in 0 1; do
  if [ $v = 1 ]; then export KAMD_EM_PLAN_STEPS=1; else unset KAMD_EM_PLAN_STEPS; fi
  timeout 600 $B > gpurun_out/c24_b$v.json 2> gpurun_out/c24_b$v.err
  python - "$v" <<'P'
import json,sys
v=sys.argv[1]
try:
    b=json.loads(open(f'gpurun_out/c24_b{v}.json').read().strip().splitlines()[-1]); pc=b.get("parity_check") or {}
    print("plan steps" if v=="1" else "group builder", b["value"], b["ms_per_step"], json.dumps(b["breakdown_ms"]), "parity", pc.get("ok"), (b.get("parity_check_tail") or {}).get("ok"))
except Exception as ex: print(v, 'failed', ex); print(open(f'gpurun_out/c24_b{v}.err').read()[-600:])
P
done
