import sys, os, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench, kallisto_amd as ka
import kallisto_amd.api as A
from kallisto_amd.synth_gpu import ReadSimulator
cat, tlens, idx = bench.prepare_workload("human", 20000, True)
index = ka.Index(idx); ctx = ka.Context(0); ctx.upload(index)
dev = torch.device("cuda", 0); L = 100; n = 30_000_000
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
t = time.perf_counter(); a1, z1, r1_ = ctx.em_run(eff); t1 = time.perf_counter() - t
print("single-GPU EM: %.1f ms, rounds %d" % (t1 * 1e3, r1_))
lib = A.load_library()
CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32)
for world in (2, 4, 8):
    acc = np.zeros_like(a1); tt = []
    for rank in range(world):
        cb = CB(lambda u, p, k: 0)   # no other ranks: the local history stands in for the global one (timing only)
        alpha = np.zeros(len(eff)); abz = np.zeros(len(eff)); rounds = C.c_int32(0)
        t = time.perf_counter()
        rc = lib.kamd_em_run_partitioned(ctx._h, rank, world, cb, None, eff.ctypes.data, len(eff), 10000, 50, alpha.ctypes.data, abz.ctypes.data, C.byref(rounds))
        tt.append((time.perf_counter() - t) * 1e3); assert rc == 0
        acc += alpha
    print("world %d: per-rank EM ms %s (each rank stops on its own history here)" % (world, [round(x, 1) for x in tt]))
