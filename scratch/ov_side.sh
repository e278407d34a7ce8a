# the second pass of kernel A beside the absorption (overflow_second_pass=1, default) against after it (=3): parity tests, then config #3 and stress A/B
set -x
mkdir -p gpurun_out/ov
python -m pytest tests/test_gpu_parity.py -q -k "long_class_lists or bit_reproducible or func or quant" 2>&1 | tail -5
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 10 --warmup 3"
for v in 1 3 1 3; do
  KAMD_TUNE=overflow_second_pass=$v python bench.py $Q --detail-file gpurun_out/ov/human_$v.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('human second_pass=$v', d['value'], d['ms_per_step'], d['breakdown_ms'], d['parity'])"
done
for v in 1 3; do
  KAMD_TUNE=overflow_second_pass=$v python bench.py --workload stress --pairs 30000000 $Q --detail-file gpurun_out/ov/stress_$v.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stress second_pass=$v', d['value'], d['ms_per_step'], d['breakdown_ms'], d['parity'])"
done
