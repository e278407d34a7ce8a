"""Kernel A variants on the bench workload (GPU box only): time of the pseudoalignment kernel, probe / bucket-read / text-hit
counters and the EC multiset of every variant (must be identical).
PAIRS=30000000 VARIANTS="v2;v3;v3:text_verify=2;v3:refill_min=4" python scratch/ka_bench.py"""
import os
import sys
import hashlib
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
import bench
import kallisto_amd as ka
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
del sim
opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)
variants = os.environ.get("VARIANTS", "v2;v3;v3:text_verify=2").split(";")
reps = int(os.environ.get("REPS", 3))
digests = {}
for var in variants:
    parts = var.split(":")
    kw = {"kernel_a": int(parts[0][1:]), "text_verify": 1, "refill_min": 8, "items_per_wave": 1024, "lds_pad": -1}
    for p in parts[1:]:
        k, v = p.split("="); kw[k] = int(v)
    ctx.tune(**kw)
    ts = []
    for _ in range(reps):
        ctx.reset()
        ctx.pseudoalign(opts, words, lens, n, L)
        pr = ctx.profile(); ts.append(pr["align_kernel_ms"])
    st = ctx.stats()
    ecs = ctx.finalize()
    order = np.lexsort((ecs.counts,))  # digest of the multiset: sort records by content
    items = sorted((tuple(ecs.ec_ids[ecs.ec_off[i]:ecs.ec_off[i + 1]].tolist()), int(ecs.counts[i])) for i in range(len(ecs.counts))) if len(ecs.counts) < 2_000_000 and os.environ.get("DIGEST", "1") == "1" else []
    dg = hashlib.md5(repr(items).encode()).hexdigest()[:12] if items else "-"
    digests[var] = dg
    print(f"{var:32s} kernel_ms {min(ts):7.3f} (runs {' '.join(f'{t:.2f}' for t in ts)}) classify {pr['classify_ms']:.2f} probes/pair {st['n_probes']/n:.3f} "
          f"bucket_reads/pair {st['n_bucket_reads']/n:.3f} text_hits/pair {st['n_text_hits']/n:.3f} wave_iters {st['n_wave_iters']} lane_util {st['n_lane_iters']/max(64*st['n_wave_iters'],1):.3f} ECs {len(ecs.counts)} digest {dg}", flush=True)
ok = len(set(digests.values())) == 1
print("EC multisets identical across variants:", ok)
