#!/bin/bash
# EM time of config #3 against split length, workgroup size and groups per CU (KAMD_TUNE=em_split_len=..,em_local_block=..,em_group_div=..)
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 6 --warmup 2 --parity-sample 0"
for v in "16 1024 -1" "16 1024 3" "16 1024 4" "16 512 3" "16 512 4" "8 1024 -1" "32 1024 -1" "16 512 -1" "16 1024 6" "16 512 6"; do
  set -- $v
  KAMD_TUNE=em_split_len=$1,em_local_block=$2,em_group_div=$3 timeout 300 python bench.py $Q --detail-file /tmp/d.json 2>/dev/null | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); dd = json.load(open('/tmp/d.json')); r = dd.get('roofline_em', {})
    print('split/block/div $v:', 'em', d['breakdown_ms']['em'], 'step', d['ms_per_step'], 'rounds', d['breakdown_ms']['em_rounds'], 'groups', r.get('groups'), 'lds', r.get('lds_bytes_per_workgroup'), 'plan', dd['counters'].get('em_plan_ms'))
except Exception as e: print('$v failed', e)"
done
