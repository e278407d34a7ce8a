# profile of the stress bench at 8 M pairs (kernel stats), used while tuning the blocked EM
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "em_forms or bit_reproducible" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
KAMD_DEBUG_FIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gb -- python $GRAFT_REPO_ROOT/bench.py --workload stress --pairs ${PAIRS:-8000000} --steps 2 --warmup 1 --full-parity off --no-cpu-baseline --bootstraps 0 --parity-sample 200000 --detail-file $GRAFT_REPO_ROOT/gpurun_out/gb.json > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/gb.log
grep "blocked EM" $GRAFT_REPO_ROOT/gpurun_out/gb.log | tail -1
