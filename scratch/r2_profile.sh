#!/bin/bash
# round 2 evidence for profiles/: PMC traffic passes -> traffic.json, kernel-trace stats, SQ counters, then the bench lines
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r02; mkdir -p $O
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B > /dev/null 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B > /dev/null 2>&1
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_traffic.py $F $W $R/profiles r02 30000000 1437 > $O/traffic.log 2>&1
cp $R/profiles/traffic.json $R/profiles/r02_pmc_fetch_size_per_kernel.csv $R/profiles/r02_pmc_write_size_per_kernel.csv $O/
rm -rf $O/pmc_fetch $O/pmc_write
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 500 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o p -- $B > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,collections,re
rows=collections.defaultdict(dict); calls=collections.defaultdict(int)
for f in glob.glob('/root/repo/gpurun_out/r02/pmc_*/**/*counter_collection.csv',recursive=True):
    seen=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::","",r['Kernel_Name']); k=re.sub(r"^void ","",k).split('(')[0]
        if k.startswith(('at::','rocprim','hipcub','__amd','void at')) or 'at::native' in k: continue
        rows[k][r['Counter_Name']]=rows[k].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
        seen[(k,r['Counter_Name'])]+=1
    for (k,c),n in seen.items(): calls[k]=max(calls[k],n)
names=sorted({c for k in rows for c in rows[k]})
with open('/root/repo/gpurun_out/r02/r02_pmc_sq_tcc_per_kernel.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['kernel','launches']+names)
    for k in sorted(rows,key=lambda k:-rows[k].get('SQ_WAVE_CYCLES',0))[:24]: w.writerow([k,calls[k]]+[int(rows[k].get(c,0)) for c in names])
PY
rm -rf $O/pmc_SQ_WAVE_CYCLES $O/pmc_SQ_INSTS_VALU $O/pmc_TCC_HIT_sum
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> /dev/null
S=$(find $O/trace -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('/root/repo/gpurun_out/r02/r02_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:140],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
PY
rm -rf $O/trace
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --end-to-end 4000000 --bootstraps 100 > $O/r02_bench.json 2> $O/r02_bench.log
timeout 600 python bench.py --workload yeast --steps 10 --warmup 3 --end-to-end 4000000 > $O/r02_bench_config2_yeast.json 2> $O/r02_bench_config2_yeast.log
cat $O/traffic.log | cut -c1-400; head -12 $O/r02_kernel_stats.csv | cut -c1-200; cut -c1-600 $O/r02_bench.json; echo; cut -c1-600 $O/r02_bench_config2_yeast.json; tail -3 $O/r02_bench_config2_yeast.log
