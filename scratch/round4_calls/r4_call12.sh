#!/bin/bash
# round 4, call 12: run-to-run spread of kernel A on one box (same build, same flags), and table loads 0.4 / 0.45 / 0.5
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c12; mkdir -p $O
export TMPDIR=/tmp
FAST="--steps 10 --warmup 3 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
i=0
for load in 0.4 0.4 0.5 0.45 0.4 0.5 0.45 0.4; do i=$((i+1)); KAMD_TABLE_LOAD=$load timeout 300 python bench.py $FAST > $O/run${i}_load$load.json 2> $O/err.txt; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c12/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'cls', b.get('classify_kernel'), 'dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'em', b.get('em'), 'ceil', d['roofline']['random_line_ceiling'].get('GB/s_in_64B_lines'), d['config']['kmer_table']['bytes'])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e)
PY
