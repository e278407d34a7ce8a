#!/bin/bash
# round 4, call 1: (1) the new GPU tests (self-launched two-rank bench, flattened-index pick-up + slow exit, the late tests of round 3),
# (2) the default bench line with the full-size parity leg, (3) GENCODE-sized index: wide vs compact at three loads + kernel stats + FETCH_SIZE
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c1; mkdir -p $O
export TMPDIR=/tmp
df -h /tmp /dev/shm | tail -2; free -g | head -2; nproc
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_zz_late.py tests/test_gpu_cli.py -q -x -k "bench or late or picks_up or flattened or compact" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"; tail -25 $O/bench_default.log
FAST="--steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
timeout 700 python bench.py $FAST --genes 46000 > $O/gc_wide.json 2> $O/gc_wide.err; echo "gc wide rc=$?"
for load in 0.6 0.5 0.75; do
  timeout 300 python bench.py $FAST --genes 46000 --table-layout compact --table-load $load > $O/gc_compact_$load.json 2> $O/gc_compact_$load.err
done
cd /tmp
for lay in wide compact; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lay -o t -- python $R/bench.py $FAST --steps 2 --warmup 1 --genes 46000 --table-layout $lay > /dev/null 2>&1
  cp "$(find /tmp/prof_$lay -name '*kernel_stats.csv' | head -1)" $O/gencode_${lay}_kernel_stats.csv 2>/dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$lay -o f -- python $R/bench.py $FAST --steps 1 --warmup 0 --genes 46000 --table-layout $lay > /dev/null 2>&1
  python - "$(find /tmp/pmc_$lay -name '*counter_collection.csv' | head -1)" $lay > $O/gencode_${lay}_fetch_size.txt 2>&1 <<'PY'
import csv, sys, collections, re
tot = collections.defaultdict(float); calls = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name']); k = re.sub(r"^void ", "", k).split('(')[0]
    if r['Counter_Name'] != 'FETCH_SIZE': continue
    tot[k] += float(r['Counter_Value']); calls[k] += 1
print("# FETCH_SIZE per kernel (raw counter units as rocprofv3 reports them; MI355X_MICROARCH.md: x 32 B on gfx950 -- see profiles/README.md), layout", sys.argv[2])
for k in sorted(tot, key=lambda k: -tot[k])[:12]: print(k[:100], calls[k], int(tot[k]))
PY
done
cd $R
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r4c1/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        t = d['config'].get('kmer_table', {})
        print(os.path.basename(f), d['value'], 'M/s', d['ms_per_step'], 'ms; A', d['breakdown_ms'].get('pseudoalign_kernel'), 'cls', d['breakdown_ms'].get('classify_kernel'), 'dedup', d['breakdown_ms'].get('tuple_dedup'),
              'fin', d['breakdown_ms'].get('ec_finalize'), 'em', d['breakdown_ms'].get('em'), d['breakdown_ms'].get('em_rounds'), ';', t.get('layout'), t.get('bytes'), 'load', t.get('load'),
              '; lines/pair', d['counters'].get('bucket_reads_per_pair'), '; ceiling', (d['roofline'].get('random_line_ceiling') or {}).get('frac'), 'kmers', d['config'].get('kmers'))
        for k in ('parity_check', 'parity_check_tail', 'parity_check_full_size'):
            if k in d: print('   ', k, json.dumps(d[k])[:900])
        if 'end_to_end' in d: print('    e2e', json.dumps({k: v for k, v in d['end_to_end'].items() if isinstance(v, dict)})[:2500])
        if 'cpu_baseline' in d: print('    cpu', json.dumps(d['cpu_baseline'])[:800])
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e)
PY
head -12 $O/gencode_wide_kernel_stats.csv | cut -c1-160; head -12 $O/gencode_compact_kernel_stats.csv | cut -c1-160; cat $O/gencode_*_fetch_size.txt | cut -c1-200
