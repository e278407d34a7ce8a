"""First call against later calls of the step's kernels (a fresh context): classify_ms / align_kernel_ms / absorb_ms of five quants of 8 M pairs.
usage: python scratch/first_call.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import kallisto_amd as ka
from kallisto_amd.synth_gpu import ReadSimulator

cat, tlens, idx_path = bench.prepare_workload("human", 20000, True)
index = ka.Index(idx_path)
dev = torch.device("cuda", 0)
n, L = 8_000_000, 100
rec = ka.packed_record_words(L)
opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)
ctx0 = ka.Context(0); ctx0.upload(index)
sim = ReadSimulator(cat, tlens, dev, seed=1000, read_len=L)
words = torch.empty(n * 2 * rec, dtype=torch.int32, device=dev); lens = torch.empty(n * 2, dtype=torch.int16, device=dev)
for s in range(0, n, 2_000_000):
    r1, r2 = sim.draw(2_000_000)
    w, l = ctx0.pack_reads(torch.stack([r1, r2], 1).reshape(-1, L), L)
    words[s * 2 * rec:(s + 2_000_000) * 2 * rec] = w; lens[2 * s:2 * (s + 2_000_000)] = l
del ctx0
for trial in range(2):
    ctx = ka.Context(0); ctx.upload(index)
    for i in range(4):
        ctx.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        ka.quant(ctx, opts, [(words, lens, n, L)], download_ecs=False)
        torch.cuda.synchronize(); el = (time.perf_counter() - t) * 1e3
        p = ctx.profile()
        print(f"context {trial} quant {i}: {el:8.2f} ms  kernel A {p['align_kernel_ms']:.3f}  classify {p['classify_ms']:.3f}  absorb {p['absorb_ms']:.3f}  finalize {p['finalize_ms']:.3f}  em {p['em_ms']:.3f}", flush=True)
    del ctx
