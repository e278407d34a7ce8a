#!/bin/bash
# round 5, call 16: the final build: whole GPU suite, the bench line of the round (20 steps, bootstraps, all child legs), kernel stats + timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r05b
SECONDS=0
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05b/pytest.log 2>&1
echo "pytest rc $? in $SECONDS s"; tail -3 gpurun_out/r05b/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0
timeout 1700 python bench.py --steps 20 --warmup 5 --bootstraps 100 > gpurun_out/r05b/r05_bench.json 2> gpurun_out/r05b/r05_bench.log
echo "bench rc $? in $SECONDS s"
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0 --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity off > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
S=$(find /tmp/trace -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('gpurun_out/r05b/r05_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:140],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
PY
K=$(find /tmp/trace -name '*kernel_trace.csv' | head -1)
python scratch/timeline.py $K > gpurun_out/r05b/r05_step_timeline.txt 2>&1
tail -3 gpurun_out/r05b/r05_step_timeline.txt | head -1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05b/r05_bench.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'])
for k in ('parity_check','parity_check_tail','parity_check_full_size'): print(k,(d.get(k) or {}).get('ok'))
print('cpu', d['cpu_baseline']['value'], 'stress', (d.get('stress') or {}).get('value'), ((d.get('stress') or {}).get('parity_check_full_size') or {}).get('ok'), 'gencode', (d.get('gencode_size') or {}).get('value'), 'config2', (d.get('config2') or {}).get('value'))
PY
