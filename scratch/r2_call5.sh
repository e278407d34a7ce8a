#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/c5
echo "=== GPU tests"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "=== kernel A v3 counters"
PAIRS=30000000 VARIANTS="v3" REPS=2 DIGEST=0 timeout 600 python scratch/ka_bench.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "=== bench"
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.log; tail -2 gpurun_out/c5/bench.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/c5/bench.json'))
print(d['value'], d['ms_per_step'], d['breakdown_ms'], d.get('parity_check',{}).get('ok'))
PY
echo "=== PMC: instruction mix of kernel A"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  PAIRS=30000000 VARIANTS="v3" REPS=1 DIGEST=0 timeout 600 rocprofv3 --pmc $set --output-format csv -d /root/repo/gpurun_out/c5/pmc_$tag -- python /root/repo/scratch/ka_bench.py > /root/repo/gpurun_out/c5/pmc_$tag.log 2>&1
done
cd /root/repo
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/c5/pmc_*/**/*counter_collection.csv',recursive=True)):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_match_v3' in k or 'k_classify' in k or 'k_em_sell' in k or 'k_resolve' in k:
            acc[(k[:40],r['Counter_Name'])]+=float(r['Counter_Value']); n[(k[:40],r['Counter_Name'])]+=1
    for k in sorted(acc): print(f, k, 'sum', acc[k], 'launches', n[k])
PY
find gpurun_out/c5 -name "*.csv" -size +20M -delete
