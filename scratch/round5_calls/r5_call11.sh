#!/bin/bash
# round 5, call 11: overflow items with their set list in LDS; kernel stats of the stress workload at full size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r5c11_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r5c11_pytest.log
ARGS="--workload stress --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0"
timeout 900 python bench.py $ARGS --pairs 4000000 --full-parity on --parity-sample 0 > gpurun_out/r5c11_p4m.json 2> gpurun_out/r5c11_p4m.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c11_p4m.json')); b=d['breakdown_ms']; c=d['counters']
print('4M', d['value'], d['ms_per_step'], b, 'ovf', c['overflow_kernel_ms'], c['overflow_share'], 'full parity ok', d['parity_check_full_size']['ok'])
PY
cd /tmp && KAMD_DEBUG_FIN=1 timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stress -o t -- python $GRAFT_REPO_ROOT/bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5c11_full.json 2> /tmp/prof_stress.log
cd "$GRAFT_REPO_ROOT"
grep "kamd\] finalize" /tmp/prof_stress.log | tail -1
S=$(find /tmp/prof_stress -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys,json
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('gpurun_out/r5c11_stress_full_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:110],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
for r in rows[:22]: print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.2f} us")
d=json.load(open('gpurun_out/r5c11_full.json')); print('30M', d['value'], d['ms_per_step'], d['breakdown_ms'], d['counters']['overflow_kernel_ms'])
PY
