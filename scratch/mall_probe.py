"""Rate of dependent random reads as a function of the footprint: is a 64-128 MB filter in front of the 2.4 GB k-mer table cheaper per probe?"""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
import kallisto_amd as ka
cat, lens, idx = bench.prepare_workload("human", 20000, True)
index = ka.Index(idx, threads=32)
ctx = ka.Context(0); ctx.upload(index)
for waves_per_cu in (8, 16, 24, 32):
    nb = 256 * waves_per_cu // 4
    row = []
    for span in (16, 64, 128, 256, 512, 0):
        for acc in (64, 8):
            g, m = ctx.random_lines(nb, 256, 512, span, acc)
            row.append("%s/%dB %.0fG/s" % (span or "all", acc, m / 1e3))
    print("waves/CU", waves_per_cu, " | ".join(row), flush=True)
