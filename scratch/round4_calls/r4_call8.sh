#!/bin/bash
# round 4, call 8: (1) a full bench line with parity legs on the GENCODE-sized index; (2) two ranks on one GPU on the human-sized index with
# the merged result against the reference; (3) kernel A launch-shape sweep on the compact table
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c8; mkdir -p $O
export TMPDIR=/tmp
FAST="--steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
for cfg in "512 8" "2048 8" "1024 4" "1024 16" "1024 8"; do set -- $cfg
  KAMD_ITEMS_PER_WAVE=$1 KAMD_REFILL_MIN=$2 timeout 300 python bench.py $FAST > $O/shape_$1_$2.json 2> $O/shape_$1_$2.err
done
KAMD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --pairs 1000000 --steps 3 --warmup 1 --no-cpu-baseline --multi-parity > $O/two_ranks_human_index.json 2> $O/two_ranks_human_index.err; echo "two ranks rc=$?"
timeout 1500 python bench.py --genes 46000 --steps 10 --warmup 3 --no-cpu-baseline --end-to-end 0 --no-config2 --no-compact-leg --full-parity on > $O/gencode_size_line.json 2> $O/gencode_size_line.err; echo "gencode rc=$?"; tail -4 $O/gencode_size_line.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c8/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); b = d['breakdown_ms']
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', b.get('pseudoalign_kernel'), 'cls', b.get('classify_kernel'), 'dedup', b.get('tuple_dedup'), 'fin', b.get('ec_finalize'), 'em', b.get('em'), b.get('em_rounds'), 'lane util', d['counters'].get('lane_utilisation'))
        for k in ('multi_rank_parity', 'parity_check', 'parity_check_tail', 'parity_check_full_size'):
            if k in d: print('   ', k, json.dumps(d[k])[:1000])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e, open(f.replace('.json', '.err')).read()[-400:])
PY
