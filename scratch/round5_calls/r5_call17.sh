#!/bin/bash
# round 5, call 17: the default bench line once more (the GENCODE-sized index built after the reference run, not beside it)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
SECONDS=0
timeout 1700 python bench.py > gpurun_out/r5c17_bench.json 2> gpurun_out/r5c17_bench.log
echo "bench rc $? in $SECONDS s"; grep "bench " gpurun_out/r5c17_bench.log | tail -14
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c17_bench.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'])
for k in ('parity_check','parity_check_tail','parity_check_full_size'): print(k,(d.get(k) or {}).get('ok'))
print('cpu', {k:d['cpu_baseline'].get(k) for k in ('value','cores','pseudoalign_seconds','em_seconds')})
print('stress', (d.get('stress') or {}).get('value'), ((d.get('stress') or {}).get('parity_check_full_size') or {}).get('ok'), 'gencode', (d.get('gencode_size') or {}).get('value'), (d.get('gencode_size') or {}).get('error'), 'config2', (d.get('config2') or {}).get('value'))
e=d['end_to_end']; print({k:(v.get('input_to_ecs_M_per_s')) for k,v in e.items() if isinstance(v,dict) and 'wall_s' in v})
PY
