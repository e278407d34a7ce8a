#!/bin/bash
# round 5, call 6: no fix-up launches (long heads), stop rule once per chunk, cheaper plan; the EM forms; kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "em_forms or stress or bit_repro" > gpurun_out/r5c6_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r5c6_pytest.log
ARGS="--workload stress --pairs 4000000 --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0 --full-parity off --parity-sample 0 --no-cpu-baseline"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $ARGS > gpurun_out/r5c6_$name.json 2> gpurun_out/r5c6_$name.log
  python - "$name" <<'PY'
import json,sys
try:
    d=json.load(open(f'gpurun_out/r5c6_{sys.argv[1]}.json'))
    b=d['breakdown_ms']; c=d['counters']
    print(f"{sys.argv[1]:14s} step {d['ms_per_step']:8.2f} ms  em {b['em']:7.2f}  fin {b['ec_finalize']:6.2f}  A {b['pseudoalign_kernel']:5.2f}  ovf {c['overflow_kernel_ms']:5.2f}  plan {c['em_plan_ms']:5.2f}  rounds {b['em_rounds']} cus {c['em_oversized']['compute_units_reserved'] if c['em_oversized'] else None} chunks {c['em_oversized']['chunks_per_direction'] if c['em_oversized'] else None}")
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
run base X=1
run nomask KAMD_EM_CUMASK=0
run nograph KAMD_EM_GRAPH=0
run k16 KAMD_EM_K=16
run streamed KAMD_EM_HYBRID=0
run streamed_fix KAMD_EM_HYBRID=0 KAMD_EM_NO_LONG_HEADS=1
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stress -o t -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/prof_stress.json 2> /tmp/prof_stress.log
cd "$GRAFT_REPO_ROOT"
S=$(find /tmp/prof_stress -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('gpurun_out/r5c6_stress_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:110],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
for r in rows[:24]: print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} ms  avg {float(r['AverageNs'])/1e3:10.2f} us")
PY
