#!/bin/bash
# round 4, call 5: the whole GPU suite (regression check of the round's changes so far) + PMC counters of the register-resident EM kernel
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c5; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -x -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
cd /tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0 --no-compact-leg --no-config2 --bootstraps 0 --full-parity off"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$tag -o p -- $B > /dev/null 2> $O/err_$tag.txt
done
python - <<'PY' > /root/repo/gpurun_out/r4c5/pmc_em.txt 2>&1
import csv,glob,collections,re
rows=collections.defaultdict(dict); calls=collections.defaultdict(int)
for f in glob.glob('/tmp/pmc_*/**/*counter_collection.csv',recursive=True):
    seen=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::","",r['Kernel_Name']); k=re.sub(r"^void ","",k).split('(')[0]
        if k.startswith(('at::','rocprim','hipcub','__amd')) or 'at::native' in k: continue
        rows[k][r['Counter_Name']]=rows[k].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
        seen[(k,r['Counter_Name'])]+=1
    for (k,c),n in seen.items(): calls[k]=max(calls[k],n)
for k in sorted(rows,key=lambda k:-rows[k].get('SQ_WAVE_CYCLES',0))[:14]:
    print(k[:90], 'launches', calls[k], {c:int(v) for c,v in sorted(rows[k].items())})
PY
cat $O/pmc_em.txt | cut -c1-900
