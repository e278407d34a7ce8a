#!/bin/bash
# The first gpurun call of the next round: time the compact k-mer table (built and parity-checked in round 3, never timed).
#   gpurun --timeout 1500 -- 'bash scratch/next_round/first_call.sh'
# 1. parity on hardware of what was added without GPU time (compact-table CLI tests, the reference-bound harness)
# 2. config #3: wide vs compact at loads 0.5 / 0.6 / 0.65 (full bench lines, no CPU legs)
# 3. GENCODE size (--genes 46000: ~2.5 min of `kallisto index` first): wide vs compact 0.6 / 0.5
# 4. kernel stats of the compact run
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/nr1; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_zz_late.py tests/test_gpu_cli.py -q -k "late or bound" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
FAST="--steps 5 --warmup 2 --no-cpu-baseline --parity-sample 0 --end-to-end 0 --no-pinned-pipeline --no-compact-leg"
timeout 200 python bench.py $FAST > $O/c3_wide.json 2> $O/c3_wide.err
for load in 0.5 0.6 0.7; do
  timeout 200 python bench.py $FAST --table-layout compact --table-load $load > $O/c3_compact_$load.json 2> $O/c3_compact_$load.err
done
timeout 600 python bench.py $FAST --genes 46000 > $O/gc_wide.json 2> $O/gc_wide.err
for load in 0.6 0.5 0.7 0.75; do
  timeout 300 python bench.py $FAST --genes 46000 --table-layout compact --table-load $load > $O/gc_compact_$load.json 2> $O/gc_compact_$load.err
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_compact -- python /root/repo/bench.py $FAST --steps 2 --warmup 1 --table-layout compact > /dev/null 2>&1; cp $(find /tmp/prof_compact -name "*kernel_stats.csv" | head -1) /root/repo/$O/compact_kernel_stats.csv 2>/dev/null )
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/nr1/*.json')):
    try:
        d = json.load(open(f))
        t = d['config'].get('kmer_table', {})
        print(os.path.basename(f), d['value'], 'M pairs/s', d['ms_per_step'], 'ms; kernel A', d['breakdown_ms'].get('pseudoalign_kernel'), 'ms;', t.get('layout'), t.get('bytes'), 'B load', t.get('load'),
              '; bucket lines/pair', d['counters'].get('bucket_reads_per_pair'), '; ceiling', (d['roofline'].get('random_line_ceiling') or {}).get('frac'))
    except Exception as e:
        print(os.path.basename(f), 'unreadable:', e)
PY
