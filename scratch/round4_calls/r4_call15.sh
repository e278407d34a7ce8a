#!/bin/bash
# round 4, call 15 (host side only): kamd_index_load of the human-sized index with and without transparent huge pages for the big tables
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); O=$R/gpurun_out/r4c15; mkdir -p $O
export TMPDIR=/tmp
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
python - <<'PY'
import sys, os, time, subprocess
sys.path.insert(0, os.getcwd())
import bench
cat, tl, idx = bench.prepare_workload("human", 20000, True)
code = "import sys,time; sys.path.insert(0,'.'); import kallisto_amd.api as A; t0=time.time(); ix=A.Index(sys.argv[1]); print('load_s', round(time.time()-t0,3))"
for rep in range(3):
    for thp in (0, 1):
        env = dict(os.environ, KAMD_INDEX_TIMING="1")
        if not thp: env["KAMD_NO_THP"] = "1"
        p = subprocess.run([sys.executable, "-c", code, idx], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        ph = {l.split()[1] + ' ' + l.split()[2]: l.split()[-2] for l in p.stderr.decode().splitlines() if l.startswith('[index]')}
        print('thp', thp, p.stdout.decode().strip(), {k: v for k, v in ph.items() if k.startswith(('table:', 'node'))})
PY
