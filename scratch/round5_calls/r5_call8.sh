#!/bin/bash
# round 5, call 8: k_resolve_big with up to 256 sets in LDS; then the stress workload at FULL size (30 M pairs): forms compared, then all pairs through the reference
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "matches_reference or stress or em_forms" > gpurun_out/r5c8_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r5c8_pytest.log
ARGS="--workload stress --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0"
show() {
python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(f'gpurun_out/r5c8_{sys.argv[1]}.json'))
    b=d['breakdown_ms']; c=d['counters']
    print(f"{sys.argv[1]:14s} value {d['value']:8.2f} step {d['ms_per_step']:8.2f} ms  em {b['em']:7.2f}  fin {b['ec_finalize']:6.2f} dedup {b['tuple_dedup']:6.2f} A {b['pseudoalign_kernel']:5.2f} cls {b['classify_kernel']:5.2f} ovf {c['overflow_kernel_ms']:5.2f}  plan {c['em_plan_ms']:5.2f}  rounds {b['em_rounds']} probes {c['probes_per_pair']} ovf_share {c['overflow_share']} form {c['em_form'][:12]} over {c['em_oversized']}")
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
KAMD_DEBUG_FIN=1 timeout 900 python bench.py $ARGS --pairs 4000000 --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c8_p4m.json 2> gpurun_out/r5c8_p4m.log; show p4m; grep "kamd\] finalize" gpurun_out/r5c8_p4m.log | tail -1
KAMD_DEBUG_FIN=1 KAMD_EM_CUMASK=0 timeout 900 python bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c8_full_nomask.json 2> gpurun_out/r5c8_full_nomask.log; show full_nomask; grep "kamd\] finalize" gpurun_out/r5c8_full_nomask.log | tail -1
KAMD_EM_HYBRID=0 timeout 900 python bench.py $ARGS --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c8_full_streamed.json 2> gpurun_out/r5c8_full_streamed.log; show full_streamed
timeout 1500 python bench.py $ARGS > gpurun_out/r5c8_full.json 2> gpurun_out/r5c8_full.log; show full
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c8_full.json'))
for k in ('parity_check','parity_check_tail','parity_check_full_size'):
    p=d.get(k) or {}
    print(k, {x:p.get(x) for x in ('ok','ec_multiset_equal','eff_length_equal','em_rounds','n_ecs','n_pseudoaligned','est_counts_max_rel_err_tpm_ge_1e-3','reference_stage_seconds','error')})
print(d.get('cpu_baseline'))
PY
