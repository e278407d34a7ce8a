cd /tmp && export TMPDIR=/tmp
for e in 4 10 11 12; do
KAMD_GB_EXP=$e rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$e -- python $GRAFT_REPO_ROOT/bench.py --workload stress --pairs 8000000 --steps 1 --warmup 1 --full-parity off --no-cpu-baseline --bootstraps 0 --parity-sample 0 --detail-file /tmp/x.json > /dev/null 2> /tmp/x.log
echo "exp $e"; grep -h "k_gb_pass\|k_gb_finish" $GRAFT_REPO_ROOT/gpurun_out/prof_exp_$e/*/*kernel_stats.csv | sed 's/(anonymous namespace):://g' | awk -F'","' '{print substr($1,1,40), $2, $4}'
done
