# k_classify with the slots through LDS: parity, then the step
python -m pytest tests/test_gpu_parity.py tests/test_gpu_func_tests.py tests/test_gpu_bus_tcc.py -q -x 2>&1 | tail -3
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 10 --warmup 3"
for d in 1 2; do
  python bench.py $Q --detail-file gpurun_out/ov/c_$d.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('human', d['value'], d['ms_per_step'], d['breakdown_ms'], d['parity']['prefix_ok'], d['parity']['tail_ok'])"
done
python bench.py --workload stress --pairs 30000000 $Q --detail-file gpurun_out/ov/cs.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stress', d['value'], d['ms_per_step'], d['breakdown_ms'], d['parity']['prefix_ok'], d['parity']['tail_ok'])"
