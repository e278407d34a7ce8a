#!/bin/bash
# round 5, call 15: kernel A's second pass over the items with long class lists: parity suite, then config #3 and stress timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r5c15_pytest.log 2>&1
echo "pytest rc $?"; tail -6 gpurun_out/r5c15_pytest.log
show() {
python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(f'gpurun_out/r5c15_{sys.argv[1]}.json'))
    b=d['breakdown_ms']; c=d['counters']
    pf=d.get('parity_check_full_size') or {}
    print(f"{sys.argv[1]:14s} value {d['value']:8.2f} step {d['ms_per_step']:8.2f} ms  A {b['pseudoalign_kernel']:6.2f} ovf {c['overflow_kernel_ms']:6.2f} ({c['overflow_items']} items) em {b['em']:7.2f} fin {b['ec_finalize']:6.2f} parity {pf.get('ok')} {pf.get('n_ecs')} {pf.get('em_rounds')}")
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
ARGS="--steps 5 --warmup 2 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0"
timeout 900 python bench.py $ARGS --full-parity off --parity-sample 200000 --no-cpu-baseline > gpurun_out/r5c15_c3.json 2> gpurun_out/r5c15_c3.log; show c3
python -c "
import json; d=json.load(open('gpurun_out/r5c15_c3.json')); print('c3 parity', d['parity_check']['ok'], d['parity_check_tail']['ok'])"
SARGS="--workload stress --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0"
timeout 900 python bench.py $SARGS --pairs 4000000 --full-parity on --parity-sample 0 > gpurun_out/r5c15_s4m.json 2> gpurun_out/r5c15_s4m.log; show s4m
timeout 900 python bench.py $SARGS --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c15_s30m.json 2> gpurun_out/r5c15_s30m.log; show s30m
KAMD_OVERFLOW_STRAIGHT=1 timeout 900 python bench.py $SARGS --full-parity off --parity-sample 0 --no-cpu-baseline > gpurun_out/r5c15_s30m_straight.json 2> gpurun_out/r5c15_s30m_straight.log; show s30m_straight
