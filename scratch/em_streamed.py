"""EM only: persistent cooperative kernel vs the three-kernel form (and the oracle on a small case) on a synthetic
gene-family CSR of the bench workload's shape.  Run on the GPU box: python scratch/em_persist.py [n_genes]"""
import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import kallisto_amd as ka
from oracle import oracle as O


def make_csr(n_genes, seed, heavy=True):
    rng = np.random.default_rng(seed)
    iso = np.minimum(rng.geometric(0.1, n_genes), 60)
    t0 = np.concatenate([[0], np.cumsum(iso)])
    T = int(t0[-1])
    rows = []
    cnts = []
    per_gene = rng.poisson(70, n_genes)
    for g in range(n_genes):
        k = iso[g]
        for _ in range(per_gene[g]):
            m = min(k, max(1, rng.geometric(0.08)))
            rows.append(np.sort(rng.choice(k, m, replace=False)) + t0[g])
            cnts.append(int(rng.pareto(1.2) * 3) + (1 if rng.random() < 0.9 else 0))
    if heavy:
        for _ in range(6):   # very long rows and (through them + the hub transcripts) heavy columns
            rows.append(np.sort(rng.choice(T, min(3000, T // 2), replace=False))); cnts.append(50)
        hubs = rng.choice(T, 4, replace=False)
        for h in hubs:
            for _ in range(4000):
                o = rng.integers(0, T)
                if o != h:
                    rows.append(np.sort(np.array([h, o]))); cnts.append(int(rng.integers(0, 5)))
    # de-duplicate rows (an EC list has distinct sets)
    seen = {}
    for r, c in zip(rows, cnts):
        key = r.tobytes()
        seen[key] = seen.get(key, 0) + c
    keys = list(seen)
    rng.shuffle(keys)
    rows = [np.frombuffer(k, dtype=rows[0].dtype) for k in keys]
    cnts = np.array([seen[k] for k in keys], np.uint32)
    off = np.zeros(len(rows) + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in rows])
    ids = np.concatenate(rows).astype(np.uint32)
    eff = rng.uniform(200, 3000, T)
    return off, ids, cnts, eff, T


def run(ctx, csr_dev, eff, persist, n_iter, k=None):
    os.environ["KAMD_EM_STREAMED"] = "1" if persist else "0"
    if k is not None: os.environ["KAMD_EM_K"] = str(k)
    else: os.environ.pop("KAMD_EM_K", None)
    a, z, r = ctx.em_run(eff, n_iter=n_iter, csr=csr_dev)
    p = ctx.profile()
    return a, z, r, p


def main():
    n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    ctx = ka.Context(0)
    dev = torch.device("cuda", 0)
    # small case against the oracle
    off, ids, cnts, eff, T = make_csr(300, 1)
    d = (torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ids.astype(np.int32)).to(dev), torch.from_numpy(cnts.astype(np.int32)).to(dev))
    ao, zo, ro = O.em_run(off, ids, cnts, eff, T)
    for persist in (0, 1):
        for k in ((None,) if not persist else (None, 8, 20, 32)):
            a, z, r, p = run(ctx, d, eff, persist, 10000, k)
            rel = np.max(np.abs(a - ao) / np.maximum(np.abs(ao), 1e-6))
            relz = np.max(np.abs(z - zo) / np.maximum(np.abs(zo), 1e-6))
            print(f"small T={T} rows={len(cnts)} streamed={persist} K={k}: rounds {r} (oracle {ro}) max rel {rel:.2e} abz {relz:.2e} zeros equal {np.array_equal(a == 0, ao == 0)} em_ms {p['em_ms']:.2f}", flush=True)
    # bench-shaped case: persistent vs three-kernel
    t = time.time()
    off, ids, cnts, eff, T = make_csr(n_genes, 2)
    print(f"big: T={T} rows={len(cnts)} nnz={len(ids)} (built in {time.time()-t:.1f}s)", flush=True)
    d = (torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ids.astype(np.int32)).to(dev), torch.from_numpy(cnts.astype(np.int32)).to(dev))
    a0, z0, r0, p0 = run(ctx, d, eff, 0, 3000)
    a0, z0, r0, p0 = run(ctx, d, eff, 0, 3000)
    print(f"three-kernel: rounds {r0} iters {p0['em_iters']} em_ms {p0['em_ms']:.2f} -> {1e3*p0['em_ms']/max(p0['em_iters'],1):.1f} us/round", flush=True)
    for k in (None, 8, 12, 16, 20, 24, 32):
        for rep in range(2):
            a, z, r, p = run(ctx, d, eff, 1, 3000, k)
        rel = np.max(np.abs(a - a0) / np.maximum(np.abs(a0), 1e-6))
        print(f"streamed K={k}: rounds {r} iters {p['em_iters']} em_ms {p['em_ms']:.2f} -> {1e3*p['em_ms']/max(p['em_iters'],1):.1f} us/round; vs three-kernel max rel {rel:.2e} zeros equal {np.array_equal(a == 0, a0 == 0)} nchunks {p['em_nseg']} K {p['em_k']} grid {p['em_grid']}", flush=True)


main()
