#!/bin/bash
cd /root/repo; O=gpurun_out/lab; mkdir -p $O
for v in "8 1024 4" "16 1024 4" "24 1024 4" "32 1024 4" "48 1024 4" "16 512 8" "32 512 8" "16 1024 8" "32 1024 2"; do
  set -- $v
  KAMD_EM_SPLIT_LEN=$1 KAMD_EM_LOCAL_BLOCK=$2 KAMD_EM_GROUP_DIV=$3 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --parity-sample 0 > $O/v.json 2> $O/v.err
  python - "$v" <<'PY'
import json,sys
try:
    d=json.load(open('/root/repo/gpurun_out/lab/v.json')); print(sys.argv[1], d['breakdown_ms']['em'], d['ms_per_step'], d['roofline_em']['groups'], d['roofline_em']['lds_bytes_per_workgroup'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
