#!/bin/bash
# round 4, call 11: size of the fragment-length prefix (behind kernel A)
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c11; mkdir -p $O
export TMPDIR=/tmp
FAST="--steps 10 --warmup 3 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
for v in 1048576 655360 786432 1048576; do KAMD_FLD_FIRST_CHUNK=$v timeout 300 python bench.py $FAST > $O/chunk_$v.$RANDOM.json 2> $O/err.txt; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c11/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'cls', b.get('classify_kernel'), 'dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'em', b.get('em'))
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e)
PY
