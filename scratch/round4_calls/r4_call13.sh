#!/bin/bash
# round 4, call 13: k_tup_absorb computes the record offsets of the main pass instead of loading them
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c13; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_bus_tcc.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
FAST="--steps 10 --warmup 3 --no-cpu-baseline --parity-sample 200000 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
for i in 1 2; do timeout 300 python bench.py $FAST > $O/run$i.json 2> $O/err.txt; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c13/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'cls', b.get('classify_kernel'), 'dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'em', b.get('em'), d['parity_check']['ok'], d['parity_check_tail']['ok'])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e)
PY
