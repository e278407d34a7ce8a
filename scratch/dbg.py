import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
import kallisto_amd as ka
from tests import common
case, variant = sys.argv[1], sys.argv[2]
meta, idx_path, r1, r2 = common.load_case(case)
o = common.parse_variant(meta["variants"][variant]); exp = common.load_expected(case, variant)
index = ka.Index(idx_path); ctx = ka.Context(0); ctx.upload(index)
reads = common.interleave(r1, r2 if o["paired"] else None)
words, lens, max_len = ctx.pack_reads_host(reads)
opts = ka.QuantOpts(o["paired"], o["fld"], o["sd"], o["single_overhang"], o["strand"], o["no_jump"])
ctx.pseudoalign(opts, words, lens, len(r1), max_len)
print(ctx.stats())
ecs = ctx.finalize()
ms = ecs.multiset()
print('equal', ms == exp["ecs"], sum(ms.values()), sum(exp["ecs"].values()))
