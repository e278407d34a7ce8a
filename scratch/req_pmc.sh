# L2 requests per kernel of one step of config #3 (the request-rate reading of kernel A, k_classify, the absorption): TCC / TCP counters, one launch each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/req; mkdir -p $O
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --parity-sample 0 --no-stress-leg --no-gencode-leg --bootstraps 0 --full-parity off --detail-file /tmp/d.json"
for set in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o p -- $B > /dev/null 2> $O/err_$tag.txt
done
python - <<PY
import csv,glob,collections,re
rows=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$O/pmc_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::|kamdi::","",r['Kernel_Name']); k=re.sub(r"^void ","",k).split('(')[0]
        if k.startswith(('at::','rocprim','hipcub','__amd')) or 'at::native' in k: continue
        rows[k][r['Counter_Name']]+=float(r['Counter_Value']); calls[k][r['Counter_Name']]+=1
with open('$O/r06_l2_requests.txt','w') as fo:
    fo.write("# L2-side requests per kernel, one step of config #3 (bench.py --steps 1 --warmup 0), summed over the kernel's launches; rocprofv3 --pmc, one pass per line of counters\n")
    for k in sorted(rows,key=lambda k:-rows[k].get('TCC_REQ_sum',0))[:14]:
        n=max(calls[k].values())
        fo.write("%s launches=%d %s\n"%(k,n,{c:int(v) for c,v in sorted(rows[k].items())}))
print(open('$O/r06_l2_requests.txt').read())
PY
rm -rf $O/pmc_*
tail -3 $O/err_TCP_TOTAL_CACHE_ACCESSES_sum.txt | cut -c1-300
