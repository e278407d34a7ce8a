#!/bin/bash
cd /root/repo; O=gpurun_out/r02c; mkdir -p $O
for v in 0 1; do
  KAMD_EM_DBG=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --parity-sample 0 > $O/v.json 2> $O/v.err
  python - "$v" <<'PY'
import json,sys
try:
    d=json.load(open('/root/repo/gpurun_out/r02c/v.json')); print('dbg', sys.argv[1], 'em_ms', d['breakdown_ms']['em'], 'rounds', d['breakdown_ms']['em_rounds'])
except Exception as e: print(sys.argv[1], 'failed', e, open('/root/repo/gpurun_out/r02c/v.err').read()[-400:])
PY
done
