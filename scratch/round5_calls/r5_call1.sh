#!/bin/bash
# round 5, call 1: the hybrid EM's first run on hardware + the stress workload against the reference
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5c1_build.log 2>&1 || { tail -20 gpurun_out/r5c1_build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "em_forms or stress_pe or bit_reproducible or smoke" > gpurun_out/r5c1_pytest.log 2>&1
echo "pytest rc $?"; tail -15 gpurun_out/r5c1_pytest.log
timeout 1500 python bench.py --workload stress --pairs 4000000 --steps 3 --warmup 1 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --bootstraps 0 --full-parity on > gpurun_out/r5c1_stress_4m.json 2> gpurun_out/r5c1_stress_4m.log
echo "stress rc $?"; tail -25 gpurun_out/r5c1_stress_4m.log; head -c 6000 gpurun_out/r5c1_stress_4m.json
