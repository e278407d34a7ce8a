#!/bin/bash
# round 4, call 3: register-resident EM kernel with tree sums in the wide forms: group size x split length; compact table default
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "em_ or reproducible" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
FAST="--steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $FAST > $O/$name.json 2> $O/$name.err; }
run lds_s32_d2 KAMD_EM_REG=0 KAMD_EM_GROUP_DIV=2
run reg_s32_d4 KAMD_EM_REG=1
run reg_s32_d3 KAMD_EM_REG=1 KAMD_EM_GROUP_DIV=3
run reg_s32_d2 KAMD_EM_REG=1 KAMD_EM_GROUP_DIV=2
run reg_s16_d4 KAMD_EM_REG=1 KAMD_EM_SPLIT_LEN=16
run reg_s16_d3 KAMD_EM_REG=1 KAMD_EM_SPLIT_LEN=16 KAMD_EM_GROUP_DIV=3
run reg_s16_d2 KAMD_EM_REG=1 KAMD_EM_SPLIT_LEN=16 KAMD_EM_GROUP_DIV=2
run reg_s8_d5 KAMD_EM_REG=1 KAMD_EM_SPLIT_LEN=8 KAMD_EM_GROUP_DIV=5
run reg_s24_d2 KAMD_EM_REG=1 KAMD_EM_SPLIT_LEN=24 KAMD_EM_GROUP_DIV=2
run reg_s32_d2_b512 KAMD_EM_REG=1 KAMD_EM_GROUP_DIV=2 KAMD_EM_LOCAL_BLOCK=512
for cfg in "32 2" "16 2"; do set -- $cfg
  KAMD_EM_REG=1 KAMD_EM_SPLIT_LEN=$1 KAMD_EM_GROUP_DIV=$2 KAMD_EM_CLK=$O/clk_s$1.bin timeout 300 python bench.py $FAST --steps 1 --warmup 0 > /dev/null 2> $O/clk_s$1.err
  python scratch/em_clk_report.py $O/clk_s$1.bin > $O/clk_s$1.txt 2>&1
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c3/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', d['breakdown_ms'].get('pseudoalign_kernel'), 'em', d['breakdown_ms'].get('em'), d['breakdown_ms'].get('em_rounds'), 'groups', d['roofline_em'].get('groups'), 'lds', d['roofline_em'].get('lds_bytes_per_workgroup'), d['config']['kmer_table'])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e, open(f.replace('.json', '.err')).read()[-300:])
PY
head -34 $O/clk_s32.txt; head -16 $O/clk_s16.txt
