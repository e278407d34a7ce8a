#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_func_tests.py tests/test_gpu_fastq_units.py tests/test_gpu_bus_tcc.py -x -q > gpurun_out/c27_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c27_tests.log
timeout 1200 python bench.py --steps 20 --warmup 5 --bootstraps 100 > gpurun_out/c27_bench.json 2> gpurun_out/c27_bench.log; echo "bench rc=$?"
python - <<'P'
import json
try:
    b = json.loads(open("gpurun_out/c27_bench.json").read().strip().splitlines()[-1])
    e = b.get("end_to_end") or {}
    print(b["value"], b["ms_per_step"], (b.get("parity_check") or {}).get("ok"), (b.get("parity_check_tail") or {}).get("ok"))
    print({k: (v.get("input_to_ecs_M_per_s"), v.get("whole_run_M_per_s"), v.get("index_on_device_s"), v.get("reads_done_s"), v.get("wall_s")) if isinstance(v, dict) and "input_to_ecs_M_per_s" in v else v for k, v in e.items() if k not in ("note", "host")})
except Exception as ex:
    print("failed", ex); print(open("gpurun_out/c27_bench.log").read()[-800:])
P
