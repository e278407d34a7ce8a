import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import kallisto_amd as ka
exec(open('/root/repo/scratch/em_streamed.py').read().split("def run(")[0].split("import kallisto_amd as ka")[1].replace("from oracle import oracle as O", ""))
ctx = ka.Context(0); dev = torch.device("cuda", 0)
heavy = os.environ.get("HEAVY", "1") == "1"
off, ids, cnts, eff, T = make_csr(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 2, heavy)
d = (torch.from_numpy(off.astype(np.int64)).to(dev), torch.from_numpy(ids.astype(np.int32)).to(dev), torch.from_numpy(cnts.astype(np.int32)).to(dev))
for rep in range(2):
    a, z, r = ctx.em_run(eff, n_iter=3000, csr=d)
    p = ctx.profile()
    print(f"heavy={heavy} rounds {r} em_ms {p['em_ms']:.2f} -> {1e3*p['em_ms']/max(p['em_iters'],1):.1f} us/round K {p['em_k']} chunks {p['em_nseg']}", flush=True)
