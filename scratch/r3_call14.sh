#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_cli.py tests/test_gpu_bus_tcc.py -x -q > gpurun_out/c14_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c14_tests.log
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0"
KAMD_EM_CLK=/tmp/em_clk.bin timeout 600 $B > gpurun_out/c14_clk.json 2> gpurun_out/c14_clk.err; echo "clk bench rc=$?"
python scratch/em_clk_report.py /tmp/em_clk.bin > gpurun_out/c14_clk_report.txt 2>&1; cat gpurun_out/c14_clk_report.txt
KAMD_EM_SPLIT_LEN=8 KAMD_EM_CLK=/tmp/em_clk8.bin timeout 600 $B > gpurun_out/c14_clk8.json 2> gpurun_out/c14_clk8.err; echo "clk8 bench rc=$?"
python scratch/em_clk_report.py /tmp/em_clk8.bin > gpurun_out/c14_clk8_report.txt 2>&1; cat gpurun_out/c14_clk8_report.txt
python - <<'P'
import json
for f in ("c14_clk", "c14_clk8"):
    try:
        b = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, b["value"], b["ms_per_step"], json.dumps(b["breakdown_ms"]))
    except Exception as e:
        print(f, "no bench line", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
P
