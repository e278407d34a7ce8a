#!/bin/bash
# round 3, GPU call 1: the device FASTQ parser end to end (new GPU tests, CLI goldens, func tests), then a first end-to-end figure
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
nproc > gpurun_out/c1_host.txt; free -g >> gpurun_out/c1_host.txt
timeout 600 python -m pytest tests/test_gpu_fastq_units.py tests/test_gpu_cli.py tests/test_gpu_func_tests.py -x -q > gpurun_out/c1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c1_tests.log
tail -15 gpurun_out/c1_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --end-to-end 8000000 > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
echo "bench rc=$?"
tail -5 gpurun_out/c1_bench.err
python - <<'P'
import json
try:
    b = json.loads(open("gpurun_out/c1_bench.json").read().strip().splitlines()[-1])
    print(json.dumps(b.get("end_to_end"), indent=1))
    print("value", b["value"], b["breakdown_ms"])
except Exception as e:
    print("no bench line", e)
P
