#!/bin/bash
# async EM driver (no stream synchronisation per chunk) + per-wavefront phase clocks of one EM round; chained FLD sample over ranks
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_cli.py -x -q > gpurun_out/c13_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c13_tests.log
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0"
KAMD_EM_CLK=/tmp/em_clk.bin timeout 600 $B > gpurun_out/c13_clk.json 2> gpurun_out/c13_clk.err; echo "clk bench rc=$?"
python scratch/em_clk_report.py /tmp/em_clk.bin > gpurun_out/c13_clk_report.txt 2>&1; cat gpurun_out/c13_clk_report.txt
timeout 600 $B > gpurun_out/c13_bench.json 2> gpurun_out/c13_bench.err; echo "bench rc=$?"
for v in "16 1024 4" "32 512 8" "32 1024 8" "32 512 4"; do
  set -- $v
  KAMD_EM_SPLIT_LEN=$1 KAMD_EM_LOCAL_BLOCK=$2 KAMD_EM_GROUP_DIV=$3 timeout 300 $B > gpurun_out/c13_v.json 2> gpurun_out/c13_v.err
  python - "$v" <<'P'
import json,sys
try:
    d=json.loads(open('gpurun_out/c13_v.json').read().strip().splitlines()[-1]); print(sys.argv[1], d['breakdown_ms']['em'], d['ms_per_step'], d['roofline_em']['groups'], d['roofline_em']['lds_bytes_per_workgroup'])
except Exception as e: print(sys.argv[1], 'failed', e)
P
done
python - <<'P'
import json
for f in ("c13_clk", "c13_bench"):
    try:
        b = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, b["value"], b["ms_per_step"], json.dumps(b["breakdown_ms"]))
    except Exception as e:
        print(f, "no bench line", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
P
