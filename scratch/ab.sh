# same-box A/B of two builds of the library (scratch/ab/lib_old.so, lib_new.so): config #3 steps, alternating
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 10 --warmup 3 --parity-sample 0"
for v in old new old new old new; do
  cp scratch/ab/lib_$v.so kallisto_amd/libkallisto_amd.so
  python bench.py $Q --detail-file gpurun_out/ov/ab_$v.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('human $v', d['value'], d['ms_per_step'], d['breakdown_ms'])"
done
for v in old new; do
  cp scratch/ab/lib_$v.so kallisto_amd/libkallisto_amd.so
  python bench.py --workload stress --pairs 30000000 $Q --detail-file gpurun_out/ov/abs_$v.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stress $v', d['value'], d['ms_per_step'], d['breakdown_ms'])"
done
cp scratch/ab/lib_new.so kallisto_amd/libkallisto_amd.so
