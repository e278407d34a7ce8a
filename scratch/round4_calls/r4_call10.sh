#!/bin/bash
# round 4, call 10: the fragment-length prefetch behind kernel A (beside k_classify / k_tup_absorb) instead of underneath it
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c10; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -x -k "quant_matches or batches or cli_matches or several_ranks or fld or reference_reader" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
FAST="--steps 10 --warmup 3 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
for v in 1 0 1 0; do KAMD_FLD_AFTER_A=$v timeout 300 python bench.py $FAST > $O/fld_after_$v.$RANDOM.json 2> $O/err.txt; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c10/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'cls', b.get('classify_kernel'), 'dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'em', b.get('em'))
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e)
PY
