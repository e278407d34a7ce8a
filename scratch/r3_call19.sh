#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fastq_units.py tests/test_gpu_cli.py tests/test_gpu_func_tests.py -x -q > gpurun_out/c19_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c19_tests.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pinned-pipeline --parity-sample 0"
timeout 900 $B > gpurun_out/c19_bench.json 2> gpurun_out/c19_bench.err; echo "bench rc=$?"
KAMD_FQ_SPARE_CPUS=3 timeout 900 $B > gpurun_out/c19_bench_spare3.json 2> gpurun_out/c19_bench_spare3.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ("c19_bench", "c19_bench_spare3"):
    try:
        b = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        e = b.get("end_to_end") or {}
        print(f, b["value"], {k: (v.get("input_to_ecs_M_per_s"), v.get("whole_run_M_per_s")) if isinstance(v, dict) and "input_to_ecs_M_per_s" in v else v for k, v in e.items() if k not in ("note", "host")})
    except Exception as ex:
        print(f, "failed", ex); print(open(f"gpurun_out/{f}.err").read()[-800:])
P
