#!/bin/bash
# round 5, call 9: the multi-rank tests (two ranks on the one GPU), then the DEFAULT bench line (with the stress child leg)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q > gpurun_out/r5c9_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r5c9_pytest.log
SECONDS=0
timeout 1700 python bench.py > gpurun_out/r5c9_bench.json 2> gpurun_out/r5c9_bench.log
echo "bench rc $? in $SECONDS s"; tail -30 gpurun_out/r5c9_bench.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c9_bench.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'])
print('roofline', d['roofline']['frac'], 'cpu', {k:d['cpu_baseline'].get(k) for k in ('value','cores','pseudoalign_seconds','em_seconds')})
for k in ('parity_check','parity_check_tail','parity_check_full_size'):
    print(k, (d.get(k) or {}).get('ok'))
print('stress', json.dumps(d.get('stress'))[:1500])
print('e2e', {k:(v.get('input_to_ecs_M_per_s') if isinstance(v,dict) else v) for k,v in (d.get('end_to_end') or {}).items()})
PY
