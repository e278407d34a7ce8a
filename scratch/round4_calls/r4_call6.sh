#!/bin/bash
# round 4, call 6: the whole GPU suite after the round's kernel changes + the step time with the tuple list carrying the store offsets
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c6; mkdir -p $O
export TMPDIR=/tmp
FAST="--steps 10 --warmup 3 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
timeout 300 python bench.py $FAST > $O/default.json 2> $O/default.err
timeout 2400 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c6/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'cls', b.get('classify_kernel'), 'dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'em', b.get('em'), b.get('em_rounds'))
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e, open(f.replace('.json', '.err')).read()[-300:])
PY
