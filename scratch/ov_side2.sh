# k_classify_long with its loads up front; items per wavefront of the second pass (divisor 8 / 16 / 32)
mkdir -p gpurun_out/ov
python -m pytest tests/test_gpu_parity.py -q -k "long_class_lists or bit_reproducible" 2>&1 | tail -2
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 10 --warmup 3"
for d in 8 16 32 8 32; do
  KAMD_EXP_IPW_DIV=$d python bench.py $Q --detail-file gpurun_out/ov/h_$d.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); dd = json.load(open('gpurun_out/ov/h_$d.json'))
print('human div=$d', d['value'], d['ms_per_step'], d['breakdown_ms'], dd['counters'].get('overflow_kernel_ms'), d['parity']['prefix_ok'], d['parity']['tail_ok'])"
done
for d in 8 32; do
  KAMD_EXP_IPW_DIV=$d python bench.py --workload stress --pairs 30000000 $Q --detail-file gpurun_out/ov/s_$d.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); dd = json.load(open('gpurun_out/ov/s_$d.json'))
print('stress div=$d', d['value'], d['ms_per_step'], d['breakdown_ms'], dd['counters'].get('overflow_kernel_ms'), d['parity']['prefix_ok'], d['parity']['tail_ok'])"
done
