"""Bring-up of the component-local EM on the bench workload (GPU box only): EM time and result for KAMD_EM_LOCAL = 0 (streamed
form, the reference point), 1 (LDS kernel, plan built on the host) and 2 (plan built on the device).
PAIRS=8000000 python scratch/next_round/em_local_real.py"""
import os
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
import bench
import kallisto_amd as ka
import kallisto_amd.api as A
from kallisto_amd.synth_gpu import ReadSimulator

cat, tlens, idx = bench.prepare_workload("human", 20000, True)
index = ka.Index(idx); ctx = ka.Context(0); ctx.upload(index)
dev = torch.device("cuda", 0); L = 100; n = int(os.environ.get("PAIRS", 30_000_000))
sim = ReadSimulator(cat, tlens, dev, seed=1000, read_len=L)
rec = ka.packed_record_words(L)
words = torch.empty(n * 2 * rec, dtype=torch.int32, device=dev); lens = torch.empty(n * 2, dtype=torch.int16, device=dev)
step = 2_000_000
for s in range(0, n, step):
    m = min(step, n - s)
    r1, r2 = sim.draw(m); inter = torch.stack([r1, r2], 1).reshape(-1, L); w, l = ctx.pack_reads(inter, L)
    words[s * 2 * rec:(s + m) * 2 * rec] = w; lens[2 * s:2 * (s + m)] = l
opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)
ctx.pseudoalign(opts, words, lens, n, L)
flens, _ = ctx.fld_from_batch(opts, words, lens, n, L)
ctx.finalize(download=False)
eff = A.eff_lens(index.target_lens, A.mean_frag_lens_trunc(flens))
ref = None
variants = os.environ.get("VARIANTS", "streamed;local:em_local_kernel=2;local;local:em_local_block=512;local:em_local_block=128;local:em_group_div=2;local:em_group_div=8;local:em_group_div=16;local:em_group_div=8,em_local_block=128").split(";")
for var in variants:
    parts = var.split(":")
    kw = {"em_form": parts[0], "em_local_kernel": 3, "em_local_block": 256, "em_group_div": 4}
    if len(parts) > 1:
        for p in parts[1].split(","):
            k, v = p.split("="); kw[k] = int(v)
    ctx.tune(**kw)
    best = None
    for rep in range(2):
        try:
            a, z, r = ctx.em_run(eff)
        except Exception as e:
            print(f"{var}: FAILED {e}", flush=True)
            a = None
            break
        p = ctx.profile()
        best = p["em_ms"] if best is None else min(best, p["em_ms"])
    if a is None:
        continue
    if ref is None:
        ref = (a, r)
    rel = np.max(np.abs(a - ref[0]) / np.maximum(np.abs(ref[0]), 1e-6))
    print(f"{var:48s} rounds {r} (first {ref[1]}) em_ms {best:7.2f} k={p['em_k']} groups {p['em_grid']} lds {p['em_lds']} max rel diff {rel:.2e}", flush=True)
