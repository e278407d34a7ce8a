#!/bin/bash
# round 4, call 9: the four-word EM kernel in 64 registers (constants from LDS, one slice per wavefront and direction): two workgroups per CU
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c9; mkdir -p $O
export TMPDIR=/tmp
KAMD_EM_LEAN=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "em_ or reproducible or quant_matches or bootstrap" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
FAST="--steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
run() { name=$1; shift; env "$@" timeout 400 python bench.py $FAST > $O/$name.json 2> $O/$name.err; }
run default X=1
run lean_d4 KAMD_EM_LEAN=1 KAMD_EM_GROUP_DIV=4
run lean_d5 KAMD_EM_LEAN=1 KAMD_EM_GROUP_DIV=5
run lean_d6 KAMD_EM_LEAN=1 KAMD_EM_GROUP_DIV=6
run lean_d4_t4 KAMD_EM_LEAN=1 KAMD_EM_GROUP_DIV=4 KAMD_TABLE_LOAD=0.4
KAMD_EM_LEAN=1 KAMD_EM_GROUP_DIV=4 timeout 400 python bench.py $FAST --genes 46000 > $O/gencode_lean_d4.json 2> $O/gencode_lean_d4.err
KAMD_EM_LEAN=1 KAMD_EM_GROUP_DIV=4 KAMD_EM_CLK=$O/clk.bin timeout 300 python bench.py $FAST --steps 1 --warmup 0 > /dev/null 2> $O/clk.err
python scratch/em_clk_report.py $O/clk.bin > $O/clk_lean_d4.txt 2>&1; rm -f $O/clk.bin
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c9/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'em', b.get('em'), b.get('em_rounds'), 'groups', d['roofline_em'].get('groups'), 'lds', d['roofline_em'].get('lds_bytes_per_workgroup'), d['config']['kmer_table']['load'], d['counters']['bucket_reads_per_pair'])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e, open(f.replace('.json', '.err')).read()[-300:])
PY
head -16 $O/clk_lean_d4.txt
