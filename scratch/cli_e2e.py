import sys, os, time, subprocess
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench, kallisto_amd as ka
from kallisto_amd.synth_gpu import ReadSimulator
cat, tlens, idx = bench.prepare_workload("human", 20000, True)
dev = torch.device("cuda", 0)
sim = ReadSimulator(cat, tlens, dev, seed=1000, read_len=100)
n = int(os.environ.get("PAIRS", 8_000_000))
r1, r2 = sim.draw(n)
t = time.time(); bench.write_fastq_fast("/tmp/e2e_1.fq", r1.cpu().numpy()); bench.write_fastq_fast("/tmp/e2e_2.fq", r2.cpu().numpy()); print("fastq written", time.time() - t)
del sim, r1, r2; torch.cuda.empty_cache()
t = time.time()
p = subprocess.run(["/root/repo/kallisto_amd/kallisto_amd_quant", "quant", "-i", idx, "-o", "/tmp/e2e_out", "-t", os.environ.get("THREADS", "32"), "--verbose", "/tmp/e2e_1.fq", "/tmp/e2e_2.fq"], stderr=subprocess.PIPE)
dt = time.time() - t
print("\n".join(l for l in p.stderr.decode().splitlines() if "processed" not in l)[-900:])
print("CLI end-to-end: %.1f s for %d pairs (%.2f M pairs/s incl. index load)" % (dt, n, n / dt / 1e6))
# gzip input (threaded readers: one inflate thread per file)
m = 2_000_000
os.system(f"head -n {4*m} /tmp/e2e_1.fq | gzip -1 > /tmp/e2e_1.fq.gz; head -n {4*m} /tmp/e2e_2.fq | gzip -1 > /tmp/e2e_2.fq.gz")
t = time.time()
p = subprocess.run(["/root/repo/kallisto_amd/kallisto_amd_quant", "quant", "-i", idx, "-o", "/tmp/e2e_out_gz", "-t", os.environ.get("THREADS", "32"), "--verbose", "/tmp/e2e_1.fq.gz", "/tmp/e2e_2.fq.gz"], stderr=subprocess.PIPE)
dt = time.time() - t
print("\n".join(l for l in p.stderr.decode().splitlines() if "host packing" in l or "rror" in l))
print("CLI end-to-end, gzip input: %.1f s for %d pairs" % (dt, m))
