import sys, os
sys.path.insert(0, '/root/repo')
import bench, kallisto_amd as ka
cat, tlens, idx = bench.prepare_workload("human", 20000, True)
index = ka.Index(idx); ctx = ka.Context(0); ctx.upload(index)
for waves_per_cu in (4, 8, 12, 16, 24, 32):
    nb = 256 * waves_per_cu // 4
    g, m = ctx.random_lines(nb, 256, 512)
    print(f"waves/CU {waves_per_cu:3d}: {g:8.1f} GB/s  {m:9.1f} M lines/s")
