# same-box A/B: threshold between k_resolve and k_resolve_big (smallest set of a tuple: 16 / 32 / 64 members)
Q="--no-gencode-leg --no-stress-leg --no-cpu-baseline --full-parity off --no-config2 --bootstraps 0 --steps 10 --warmup 3"
for v in 16 32 64 16 32 64; do
  cp scratch/ab/lib_$v.so kallisto_amd/libkallisto_amd.so
  KAMD_DEBUG_FIN=1 python bench.py $Q --detail-file /tmp/d.json 2>/tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('human $v', d['value'], d['ms_per_step'], d['breakdown_ms']['ec_finalize'], d['parity']['prefix_ok'], d['parity']['tail_ok'])"
  grep "finalize:" /tmp/err.txt | sort | uniq -c | sort -rn | head -1
done
for v in 16 32 64; do
  cp scratch/ab/lib_$v.so kallisto_amd/libkallisto_amd.so
  KAMD_DEBUG_FIN=1 python bench.py --workload stress --pairs 30000000 $Q --detail-file /tmp/d.json 2>/tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stress $v', d['value'], d['ms_per_step'], d['breakdown_ms']['ec_finalize'], d['parity']['prefix_ok'], d['parity']['tail_ok'])"
  grep "finalize:" /tmp/err.txt | sort | uniq -c | sort -rn | head -1
done
cp scratch/ab/lib_16.so kallisto_amd/libkallisto_amd.so
