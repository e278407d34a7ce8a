#!/bin/bash
# round 5, call 7: k_resolve_big in two size classes with the sets' metadata in one round; wave-cooperative k_cand_singles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r5c7_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r5c7_pytest.log
ARGS="--workload stress --pairs 4000000 --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0"
KAMD_DEBUG_FIN=1 timeout 900 python bench.py $ARGS --full-parity on --parity-sample 0 > gpurun_out/r5c7_stress.json 2> gpurun_out/r5c7_stress.log
echo "stress rc $?"; grep "kamd\] finalize" gpurun_out/r5c7_stress.log | tail -2
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c7_stress.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'])
p=d['parity_check_full_size']; print({k:p.get(k) for k in ('ok','ec_multiset_equal','eff_length_equal','em_rounds','est_counts_max_rel_err_tpm_ge_1e-3','reference_stage_seconds')})
print(d.get('cpu_baseline'))
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stress -o t -- python $GRAFT_REPO_ROOT/bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > /tmp/prof_stress.json 2> /tmp/prof_stress.log
cd "$GRAFT_REPO_ROOT"
S=$(find /tmp/prof_stress -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('gpurun_out/r5c7_stress_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:110],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
for r in rows[:14]: print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.2f} us")
PY
