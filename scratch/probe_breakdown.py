"""Where kernel A's bucket reads come from: config #3's index, 4 M pairs drawn with and without substitution errors / N bases.
Prints probes, bucket reads and text answers per pair for both.  usage: python scratch/probe_breakdown.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import kallisto_amd as ka
from kallisto_amd.synth_gpu import ReadSimulator

cat, tlens, idx_path = bench.prepare_workload("human", 20000, True)
index = ka.Index(idx_path)
ctx = ka.Context(0)
ctx.upload(index)
dev = torch.device("cuda", 0)
n, L = 4_000_000, 100
rec = ka.packed_record_words(L)
opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)
for name, kw in (("with errors (0.5 % substitutions, 0.1 % of the reads with an N): the bench's reads", {}), ("error-free reads", {"err": 0.0, "n_frac": 0.0})):
    sim = ReadSimulator(cat, tlens, dev, seed=1000, read_len=L, **kw)
    words = torch.empty(n * 2 * rec, dtype=torch.int32, device=dev); lens = torch.empty(n * 2, dtype=torch.int16, device=dev)
    for s in range(0, n, 2_000_000):
        r1, r2 = sim.draw(2_000_000)
        w, l = ctx.pack_reads(torch.stack([r1, r2], 1).reshape(-1, L), L)
        words[s * 2 * rec:(s + 2_000_000) * 2 * rec] = w; lens[2 * s:2 * (s + 2_000_000)] = l
    ctx.reset()
    res = ka.quant(ctx, opts, [(words, lens, n, L)], download_ecs=False)
    st = res.stats
    print(f"{name}: probes/pair {st['n_probes'] / n:.3f}  bucket reads/pair {st['n_bucket_reads'] / n:.3f}  text answers/pair {st['n_text_hits'] / n:.3f}  "
          f"kernel A {ctx.profile()['align_kernel_ms']:.3f} ms for {n} pairs  overflow items {ctx.profile()['n_overflow_items']}")
