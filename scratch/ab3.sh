# same-box A/B: lib_new.so (HEAD) against lib_new2.so (kernel A's raw record in 8-byte stores)
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 10 --warmup 3"
for v in new new2 new new2 new new2; do
  cp scratch/ab/lib_$v.so kallisto_amd/libkallisto_amd.so
  python bench.py $Q --detail-file gpurun_out/ov/ab3_$v.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('human $v', d['value'], d['ms_per_step'], d['breakdown_ms'], d['parity']['prefix_ok'], d['parity']['tail_ok'])"
done
cp scratch/ab/lib_new2.so kallisto_amd/libkallisto_amd.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_func_tests.py -q -x 2>&1 | tail -2
