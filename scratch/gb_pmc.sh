# PMC counters of the blocked EM kernels (stress, 8 M pairs, one step)
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload stress --pairs 8000000 --steps 1 --warmup 0 --full-parity off --no-cpu-baseline --bootstraps 0 --parity-sample 0 --detail-file /tmp/x.json"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gbpmc_$tag -o p -- $B > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections,re
rows=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(int)
for f in glob.glob('$GRAFT_REPO_ROOT/gpurun_out/gbpmc_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'k_gb_pass<\d>|k_gb_finish|k_gi_rows|k_gi_cols',r['Kernel_Name'])
        if not m: continue
        rows[m.group(0)][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVES','SQ_LDS_IDX_ACTIVE','SQ_INSTS_VALU','TCC_REQ_sum'): calls[(m.group(0),r['Counter_Name'])]+=1
for k in rows:
    n=max([v for (kk,c),v in calls.items() if kk==k] or [1])
    print(k,'launches',n,{c:round(v/n,1) for c,v in sorted(rows[k].items())})
PY
