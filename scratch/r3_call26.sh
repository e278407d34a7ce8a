#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0"
for v in 4 2 3 5 6; do
  KAMD_EM_GROUP_DIV=$v timeout 300 $B > gpurun_out/c26_v.json 2> gpurun_out/c26_v.err
  python - "$v" <<'P'
import json,sys
try:
    d=json.loads(open('gpurun_out/c26_v.json').read().strip().splitlines()[-1]); print("group_div", sys.argv[1], "em", d['breakdown_ms']['em'], "step", d['ms_per_step'], "groups", d['roofline_em']['groups'], "lds", d['roofline_em']['lds_bytes_per_workgroup'])
except Exception as e: print(sys.argv[1], 'failed', e)
P
done
