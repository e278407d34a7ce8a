"""EM on the bench workload's real EC matrix: sweep of the streamed form's K and the CSR form.  GPU box only."""
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench, kallisto_amd as ka
import kallisto_amd.api as A
from kallisto_amd.synth_gpu import ReadSimulator
cat, tlens, idx = bench.prepare_workload("human", 20000, True)
index = ka.Index(idx); ctx = ka.Context(0); ctx.upload(index)
dev = torch.device("cuda", 0); L = 100; n = int(os.environ.get("PAIRS", 30_000_000))
sim = ReadSimulator(cat, tlens, dev, seed=1000, read_len=L)
rec = ka.packed_record_words(L)
words = torch.empty(n * 2 * rec, dtype=torch.int32, device=dev); lens = torch.empty(n * 2, dtype=torch.int16, device=dev)
for s in range(0, n, 2_000_000):
    r1, r2 = sim.draw(2_000_000); inter = torch.stack([r1, r2], 1).reshape(-1, L); w, l = ctx.pack_reads(inter, L)
    words[s*2*rec:(s+2_000_000)*2*rec] = w; lens[2*s:2*(s+2_000_000)] = l
opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)
ctx.pseudoalign(opts, words, lens, n, L)
flens, _ = ctx.fld_from_batch(opts, words, lens, n, L)
ctx.finalize(download=False)
eff = A.eff_lens(index.target_lens, A.mean_frag_lens_trunc(flens))
ref = None
for mode in [(0, 24), (0, 20), (0, 16), (0, 28), (1, 24), (0, 24), (0, 20)]:
    os.environ["KAMD_EM_WINDOWED"] = str(mode[0]); os.environ["KAMD_EM_K"] = str(mode[1])
    for rep in range(2):
        a, z, r = ctx.em_run(eff)
    p = ctx.profile()
    if ref is None: ref = a
    rel = np.max(np.abs(a - ref) / np.maximum(np.abs(ref), 1e-6))
    print(f"{str(mode):9s} rounds {r} em_ms {p['em_ms']:.2f} -> {1e3*p['em_ms']/max(p['em_iters'],1):.2f} us/round K {p['em_k']} chunks {p['em_nseg']} max rel vs csr {rel:.2e}", flush=True)
