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
for level, kern in ((0, 2), (1, 1), (2, 1), (2, 2), (2, 2), (0, 2)):
    os.environ["KAMD_EM_LOCAL"] = str(level)
    os.environ["KAMD_EML_KERNEL"] = str(kern)
    try:
        a, z, r = ctx.em_run(eff)
    except Exception as e:                      # keep going: the other levels are still informative
        print(f"KAMD_EM_LOCAL={level}: FAILED {e}", flush=True)
        continue
    p = ctx.profile()
    if ref is None:
        ref = (a, r)
    rel = np.max(np.abs(a - ref[0]) / np.maximum(np.abs(ref[0]), 1e-6))
    print(f"KAMD_EM_LOCAL={level} kernel={kern}: rounds {r} (streamed {ref[1]}) em_ms {p['em_ms']:.2f} form k={p['em_k']} groups/grid {p['em_grid']} "
          f"max rel diff vs streamed {rel:.2e}", flush=True)
