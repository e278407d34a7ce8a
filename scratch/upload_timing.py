"""kamd_index_load and kamd_index_upload of config #3's index, three times (host box timing: run on a GPU box).  usage: python scratch/upload_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, kallisto_amd as ka
cat, tl, idx = bench.prepare_workload("human", 20000, True)
for rep in range(3):
    t = time.time(); ix = ka.Index(idx); t1 = time.time() - t
    ctx = ka.Context(0)
    torch.cuda.synchronize(); t = time.time(); ctx.upload(ix); torch.cuda.synchronize(); t2 = time.time() - t
    print(f"kamd_index_load {t1:.3f} s   kamd_index_upload {t2:.3f} s", flush=True)
    del ctx, ix
