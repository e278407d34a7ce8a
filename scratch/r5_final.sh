#!/bin/bash
# round 5 evidence for profiles/: PMC traffic passes -> traffic.json, SQ/TCC and LDS counters, kernel-trace stats + step timeline,
# the front-end under the kernel trace, then the bench lines (config #3 with bootstraps, config #2)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r05; mkdir -p $O
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0 --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity off"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B > /dev/null 2>&1
F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python $R/tools/pmc_traffic.py $F $W $R/profiles r05 30000000 1437 > $O/traffic.log 2>&1
cp $R/profiles/traffic.json $R/profiles/r05_traffic.json 2>/dev/null
cp $R/profiles/traffic.json $R/profiles/r05_traffic.json $R/profiles/r05_pmc_fetch_size_per_kernel.csv $R/profiles/r05_pmc_write_size_per_kernel.csv $O/ 2>/dev/null
rm -rf $O/pmc_fetch $O/pmc_write
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o p -- $B > /dev/null 2> $O/err_$tag.txt
done
python - <<'PY'
import csv,glob,collections,re
rows=collections.defaultdict(dict); calls=collections.defaultdict(int)
for f in glob.glob('/root/repo/gpurun_out/r05/pmc_*/**/*counter_collection.csv',recursive=True):
    seen=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::","",r['Kernel_Name']); k=re.sub(r"^void ","",k).split('(')[0]
        if k.startswith(('at::','rocprim','hipcub','__amd','void at')) or 'at::native' in k: continue
        rows[k][r['Counter_Name']]=rows[k].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
        seen[(k,r['Counter_Name'])]+=1
    for (k,c),n in seen.items(): calls[k]=max(calls[k],n)
lds=('SQ_LDS_BANK_CONFLICT','SQ_LDS_ADDR_CONFLICT','SQ_ACTIVE_INST_LDS','SQ_LDS_IDX_ACTIVE','SQ_LDS_UNALIGNED_STALL','SQ_BUSY_CYCLES','SQ_BUSY_CU_CYCLES')
names=sorted({c for k in rows for c in rows[k] if c not in lds})
with open('/root/repo/gpurun_out/r05/r05_pmc_sq_tcc_per_kernel.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['kernel','launches']+names)
    for k in sorted(rows,key=lambda k:-rows[k].get('SQ_WAVE_CYCLES',0))[:28]: w.writerow([k,calls[k]]+[int(rows[k].get(c,0)) for c in names])
with open('/root/repo/gpurun_out/r05/r05_lds_counters.txt','w') as fo:
    fo.write("# LDS counters per kernel, summed over the launches of one bench step (bench.py --steps 1 --warmup 0); two PMC passes\n")
    fo.write("# conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent on replays / cycles it was active)\n")
    for k in sorted(rows,key=lambda k:-rows[k].get('SQ_LDS_IDX_ACTIVE',0))[:8]:
        d={c:int(rows[k].get(c,0)) for c in lds if c in rows[k]}
        act=d.get('SQ_LDS_IDX_ACTIVE',0); bc=d.get('SQ_LDS_BANK_CONFLICT',0)
        fo.write("%s launches=%d %s conflict_share=%.3f\n"%(k,calls[k],d,(bc/act if act else 0.0)))
PY
rm -rf $O/pmc_*
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --end-to-end 0 --no-pinned-pipeline --parity-sample 0 --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity off > $O/bench_trace.json 2> /dev/null
S=$(find $O/trace -name '*kernel_stats.csv' | head -1)
python - "$S" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if not any(x in r['Name'] for x in ('at::','rocprim','hipcub'))]
with open('/root/repo/gpurun_out/r05/r05_kernel_stats.csv','w',newline='') as fo:
    w=csv.writer(fo); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs'])
    for r in rows: w.writerow([r['Name'][:140],r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['MinNs'],r['MaxNs']])
PY
K=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python $R/scratch/timeline.py $K > $O/r05_step_timeline.txt 2>&1
rm -rf $O/trace
KAMD_EM_CLK=$O/em_clk.bin timeout 300 $B > /dev/null 2>&1; python $R/scratch/em_clk_report.py $O/em_clk.bin > $O/r05_em_phase_clocks.txt 2>&1; rm -f $O/em_clk.bin
cd $R
timeout 1700 python bench.py --steps 20 --warmup 5 --bootstraps 100 > $O/r05_bench.json 2> $O/r05_bench.log
timeout 900 python bench.py --workload yeast --steps 10 --warmup 3 --no-stress-leg --no-gencode-leg > $O/r05_bench_config2_yeast.json 2> $O/r05_bench_config2_yeast.log
timeout 1500 python bench.py --genes 46000 --steps 5 --warmup 2 --end-to-end 0 --no-pinned-pipeline --no-compact-leg --no-config2 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity on > $O/r05_bench_gencode_size.json 2> $O/r05_bench_gencode_size.log
cat $O/traffic.log | cut -c1-400; head -30 $O/r05_kernel_stats.csv | cut -c1-170; cat $O/r05_lds_counters.txt | cut -c1-400; tail -20 $O/r05_step_timeline.txt; cut -c1-600 $O/r05_bench.json; echo; cut -c1-400 $O/r05_bench_config2_yeast.json; head -12 $O/r05_em_phase_clocks.txt
