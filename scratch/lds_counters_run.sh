#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/lab; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i "lds" | cut -c1-200 | head -40 > $O/lds_counters.txt
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o p -- $B > /dev/null 2> $O/err_$tag.txt
done
python - <<'PY'
import csv,glob,collections,re
rows=collections.defaultdict(dict); calls=collections.defaultdict(int)
for f in glob.glob('/root/repo/gpurun_out/lab/pmc_*/**/*counter_collection.csv',recursive=True):
    seen=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::","",r['Kernel_Name']); k=re.sub(r"^void ","",k).split('(')[0]
        if not (k.startswith('k_em_sell') or k.startswith('k_match_v3') or k.startswith('k_rec_dedup') or k.startswith('k_resolve') or k.startswith('k_classify')): continue
        rows[k][r['Counter_Name']]=rows[k].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
        seen[(k,r['Counter_Name'])]+=1
    for (k,c),n in seen.items(): calls[k]=max(calls[k],n)
with open('/root/repo/gpurun_out/lab/lds_pmc.txt','w') as fo:
    for k in rows:
        fo.write("%s launches=%d %s\n"%(k,calls[k],{c:int(v) for c,v in sorted(rows[k].items())}))
PY
rm -rf $O/pmc_*
cat $O/lds_counters.txt | head -30; cat $O/lds_pmc.txt; tail -3 $O/err_*.txt | cut -c1-300
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python - <<'PY'
import json; d=json.load(open('/root/repo/gpurun_out/lab/bench.json')); print(d['breakdown_ms'])
PY
