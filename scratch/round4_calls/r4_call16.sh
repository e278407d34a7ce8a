#!/bin/bash
# round 4, call 16 (host side): the front-end's index load with the node records parsed in parallel (before: 0.86-0.89 s in r04_bench.json's end_to_end legs)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
python - <<'PY'
import sys, os, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
import bench
cat, tl, idx = bench.prepare_workload("human", 20000, True)
r = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(1).integers(0, 4, (2000, 100))]
bench.write_fastq_fast("/tmp/t_1.fq", r); bench.write_fastq_fast("/tmp/t_2.fq", r)
for rep in range(4):
    p = subprocess.run(["kallisto_amd/kallisto_amd_quant", "quant", "-i", idx, "-o", "/tmp/t_out", "-t", "16", "--plaintext", "--verbose", "/tmp/t_1.fq", "/tmp/t_2.fq"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KAMD_INDEX_TIMING="1"))
    e = p.stderr.decode()
    print(rep, [l for l in e.splitlines() if l.startswith("[timing] index")], {l.split()[1] + ' ' + l.split()[2]: l.split()[-2] for l in e.splitlines() if l.startswith('[index]') and l.split()[-1] == 's'})
PY
