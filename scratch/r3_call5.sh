#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q > gpurun_out/c5_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/c5_tests.log
for v in -1 192 384 768 1536; do
  KAMD_EM_SMALL_NNZ=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline > gpurun_out/c5_em_$v.json 2> gpurun_out/c5_em_$v.err
  python - "$v" <<'P'
import json, sys
v = sys.argv[1]
try:
    b = json.loads(open(f"gpurun_out/c5_em_{v}.json").read().strip().splitlines()[-1])
    print(f"small_nnz={v}: value {b['value']} step {b['ms_per_step']} em {b['breakdown_ms']['em']} rounds {b['breakdown_ms']['em_rounds']} groups {b['roofline_em'].get('groups')} lds {b['roofline_em'].get('lds_bytes_per_workgroup')}")
except Exception as e:
    print(v, "failed", e); print(open(f"gpurun_out/c5_em_{v}.err").read()[-800:])
P
done
