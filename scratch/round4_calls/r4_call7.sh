#!/bin/bash
# round 4, call 7: what k_tup_absorb's time is made of (timing experiments, results wrong): no count atomic / no verification read / neither
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c7; mkdir -p $O
export TMPDIR=/tmp
FAST="--steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
for v in 0 2 4 6; do KAMD_DEBUG_ABSORB=$v timeout 300 python bench.py $FAST > $O/absorb_$v.json 2> $O/absorb_$v.err; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c7/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d['breakdown_ms']
        print(os.path.basename(f), d['ms_per_step'], 'ms; dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'distinct', d['counters']['distinct_tuples'])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e, open(f.replace('.json', '.err')).read()[-300:])
PY
