#!/bin/bash
# round 4, call 4: EM register form with byte addresses / masked scans / two slices per wavefront; auto group divisor; pinned read-backs
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -q -x -k "em_ or reproducible or bootstrap or two_ranks or quant_matches" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
FAST="--steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $FAST > $O/$name.json 2> $O/$name.err; }
run default X=1
run s16_d3 KAMD_EM_GROUP_DIV=3
run s32 KAMD_EM_SPLIT_LEN=32
run s8 KAMD_EM_SPLIT_LEN=8
run s8_d4 KAMD_EM_SPLIT_LEN=8 KAMD_EM_GROUP_DIV=4
run lds_only KAMD_EM_REG=0
timeout 700 python bench.py $FAST --genes 46000 > $O/gencode_default.json 2> $O/gencode_default.err
KAMD_EM_REG=0 KAMD_EM_SPLIT_LEN=32 KAMD_EM_GROUP_DIV=4 timeout 400 python bench.py $FAST --genes 46000 > $O/gencode_r3_em.json 2> $O/gencode_r3_em.err
KAMD_EM_CLK=$O/clk_default.bin timeout 300 python bench.py $FAST --steps 1 --warmup 0 > /dev/null 2> $O/clk_default.err
python scratch/em_clk_report.py $O/clk_default.bin > $O/clk_default.txt 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c4/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'cls', b.get('classify_kernel'), 'dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'em', b.get('em'), b.get('em_rounds'), 'groups', d['roofline_em'].get('groups'), 'lds', d['roofline_em'].get('lds_bytes_per_workgroup'), d['config']['kmer_table']['layout'], d['config']['kmer_table']['load'])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e, open(f.replace('.json', '.err')).read()[-300:])
PY
head -34 $O/clk_default.txt
