# what the fragment-length kernels cost a step of config #3: an experiment build that skips them (a fixed sample instead: the EM differs, compare step - em)
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 10 --warmup 3 --parity-sample 0"
cp scratch/ab/lib_exp.so kallisto_amd/libkallisto_amd.so
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export KAMD_EXP_NOFLD=1; else unset KAMD_EXP_NOFLD; fi
  python bench.py $Q --detail-file gpurun_out/ov/fld_$v.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['breakdown_ms']
print('nofld=$v', d['ms_per_step'], 'step-em', round(b['step_total'] - b['em'], 3), b)"
done
cp scratch/ab/lib_new.so kallisto_amd/libkallisto_amd.so
