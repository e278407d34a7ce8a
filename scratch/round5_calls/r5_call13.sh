#!/bin/bash
# round 5, call 13: the oversized side with hot targets in LDS: forms test, stress 4 M and 30 M with and without
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "em_forms or stress" > gpurun_out/r5c13_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r5c13_pytest.log
ARGS="--workload stress --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0"
show() {
python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(f'gpurun_out/r5c13_{sys.argv[1]}.json'))
    b=d['breakdown_ms']; c=d['counters']
    pf=d.get('parity_check_full_size') or {}
    print(f"{sys.argv[1]:14s} value {d['value']:8.2f} step {d['ms_per_step']:8.2f} ms  em {b['em']:7.2f}  fin {b['ec_finalize']:6.2f} plan {c['em_plan_ms']:5.2f} rounds {b['em_rounds']} parity {pf.get('ok')} {pf.get('em_rounds')} {pf.get('est_counts_max_rel_err_tpm_ge_1e-3')}")
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
timeout 900 python bench.py $ARGS --pairs 4000000 --full-parity on --parity-sample 0 > gpurun_out/r5c13_p4m_hot.json 2> gpurun_out/r5c13_p4m_hot.log; show p4m_hot
KAMD_EM_HOT=0 timeout 900 python bench.py $ARGS --pairs 4000000 --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c13_p4m_cold.json 2> gpurun_out/r5c13_p4m_cold.log; show p4m_cold
timeout 900 python bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c13_full_hot.json 2> gpurun_out/r5c13_full_hot.log; show full_hot
KAMD_EM_HOT=0 timeout 900 python bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c13_full_cold.json 2> gpurun_out/r5c13_full_cold.log; show full_cold
KAMD_EM_K=16 timeout 900 python bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c13_full_hot_k16.json 2> gpurun_out/r5c13_full_hot_k16.log; show full_hot_k16
