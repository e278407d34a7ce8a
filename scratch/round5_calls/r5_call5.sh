#!/bin/bash
# round 5, call 5: k_resolve_big with the pair-parallel tail; EM shape experiments on the stress workload (graph on/off, entries per lane, CU split)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stress or human_pe or mosaic" > gpurun_out/r5c5_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r5c5_pytest.log
ARGS="--workload stress --pairs 4000000 --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0 --full-parity off --parity-sample 0 --no-cpu-baseline"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py $ARGS > gpurun_out/r5c5_$name.json 2> gpurun_out/r5c5_$name.log
  python - "$name" <<'PY'
import json,sys
try:
    d=json.load(open(f'gpurun_out/r5c5_{sys.argv[1]}.json'))
    b=d['breakdown_ms']; c=d['counters']
    print(f"{sys.argv[1]:14s} step {d['ms_per_step']:8.2f} ms  em {b['em']:7.2f}  fin {b['ec_finalize']:6.2f}  A {b['pseudoalign_kernel']:5.2f}  ovf {c['overflow_kernel_ms']:5.2f}  plan {c['em_plan_ms']:5.2f}  rounds {b['em_rounds']} cus {c['em_oversized']['compute_units_reserved'] if c['em_oversized'] else None} chunks {c['em_oversized']['chunks_per_direction'] if c['em_oversized'] else None}")
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
run base X=1
run nograph KAMD_EM_GRAPH=0
run k16 KAMD_EM_K=16
run k8 KAMD_EM_K=8
run k16nograph KAMD_EM_K=16 KAMD_EM_GRAPH=0
run cus192 KAMD_EM_GIANT_CUS=192
run cus64 KAMD_EM_GIANT_CUS=64
run nomask KAMD_EM_CUMASK=0
run streamed KAMD_EM_HYBRID=0
