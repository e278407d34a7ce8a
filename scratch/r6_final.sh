#!/bin/bash
# round 6 evidence for profiles/: PMC traffic passes -> traffic.json, LDS counters, kernel-trace stats + step timeline of config #3, the probe
# break-down, then the bench lines (default line, config #2, GENCODE size with full parity, stress at full size with full parity)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --parity-sample 0 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity off --detail-file /tmp/d.json"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B > /dev/null 2>&1
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
ROUNDS=$(python -c "import json;print(json.load(open('/tmp/d.json'))['breakdown_ms']['em_rounds'])" 2>/dev/null || echo 1437)
python $R/tools/pmc_traffic.py $F $W $O r06 30000000 $ROUNDS > $O/traffic.log 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o p -- $B > /dev/null 2> $O/err_$tag.txt
done
python - <<PY
import csv,glob,collections,re
rows=collections.defaultdict(dict); calls=collections.defaultdict(int)
for f in glob.glob('$O/pmc_*/**/*counter_collection.csv',recursive=True):
    seen=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::|kamdi::","",r['Kernel_Name']); k=re.sub(r"^void ","",k).split('(')[0]
        if k.startswith(('at::','rocprim','hipcub','__amd','void at')) or 'at::native' in k: continue
        rows[k][r['Counter_Name']]=rows[k].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
        seen[(k,r['Counter_Name'])]+=1
    for (k,c),n in seen.items(): calls[k]=max(calls[k],n)
lds=('SQ_LDS_BANK_CONFLICT','SQ_LDS_ADDR_CONFLICT','SQ_ACTIVE_INST_LDS','SQ_LDS_IDX_ACTIVE')
with open('$O/r06_lds_counters.txt','w') as fo:
    fo.write("# LDS counters per kernel, summed over the launches of one bench step (bench.py --steps 1 --warmup 0)\n")
    fo.write("# conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent on replays / cycles it was active)\n")
    for k in sorted(rows,key=lambda k:-rows[k].get('SQ_LDS_IDX_ACTIVE',0))[:8]:
        d={c:int(rows[k].get(c,0)) for c in lds if c in rows[k]}
        act=d.get('SQ_LDS_IDX_ACTIVE',0); bc=d.get('SQ_LDS_BANK_CONFLICT',0)
        fo.write("%s launches=%d %s conflict_share=%.3f\n"%(k,calls[k],d,(bc/act if act else 0.0)))
PY
rm -rf $O/pmc_*
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --parity-sample 0 --no-stress-leg --no-gencode-leg --full-parity off --bootstraps 0 --detail-file /tmp/d2.json > /dev/null 2>&1
S=$(find $O/trace -name '*kernel_stats.csv' | head -1)
python - "$S" <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('$O/r06_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:140],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
PY
K=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/scratch/timeline.py $K > $O/r06_step_timeline.txt 2>&1
rm -rf $O/trace
cd $R
python scratch/probe_breakdown.py > $O/r06_probe_breakdown.txt 2> $O/probe.log
timeout 1700 python bench.py --steps 20 --warmup 5 --detail-file $O/r06_bench_detail.json > $O/r06_bench_line.json 2> $O/r06_bench.log
timeout 900 python bench.py --workload yeast --steps 10 --warmup 3 --no-stress-leg --no-gencode-leg --detail-file $O/r06_bench_config2_yeast.json > $O/r06_bench_config2_yeast.line 2> $O/r06_bench_config2_yeast.log
timeout 1500 python bench.py --genes 46000 --steps 5 --warmup 2 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity on --detail-file $O/r06_bench_gencode_size.json > $O/r06_bench_gencode_size.line 2> $O/r06_bench_gencode_size.log
timeout 1700 python bench.py --workload stress --steps 3 --warmup 1 --bootstraps 0 --full-parity on --detail-file $O/r06_bench_stress_full_size.json > $O/r06_bench_stress_full_size.line 2> $O/r06_bench_stress_full_size.log
cat $O/traffic.log | cut -c1-400; cat $O/r06_lds_counters.txt | cut -c1-300; tail -12 $O/r06_step_timeline.txt; cat $O/r06_probe_breakdown.txt; cut -c1-700 $O/r06_bench_line.json; echo; cut -c1-300 $O/r06_bench_config2_yeast.line; echo; cut -c1-300 $O/r06_bench_gencode_size.line; echo; cut -c1-300 $O/r06_bench_stress_full_size.line
