"""Two ranks on ONE GPU (gloo for the collectives, since RCCL needs one device per rank): each rank pseudoaligns its
shard on the device, the EC states are exported from the device, merged with kallisto_amd.exchange (all-reduce of the dense
vector + all-gather of tuple / explicit records) and installed back; both ranks must then finalize to the reference's EC
multiset for the WHOLE input and run the same EM.  This exercises every device-side piece of the multi-GPU path
(kamd_ec_dense_counts, kamd_ec_tuples_{export,copy,replace}, kamd_ec_explicit_*) that the 8-GPU bench relies on."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, variant, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kallisto_amd as ka
        from kallisto_amd.exchange import gather_records, merge_ec_state
        meta, idx_path, r1, r2 = common.load_case(case)
        o = common.parse_variant(meta["variants"][variant])
        paired = bool(o["paired"])
        n = len(r1)
        lo, hi = rank * n // world, (rank + 1) * n // world
        index = ka.Index(idx_path)
        ctx = ka.Context(0)
        ctx.upload(index)
        words, lens, max_len = ctx.pack_reads_host(common.interleave(r1[lo:hi], r2[lo:hi] if paired else None), 100)
        opts = ka.QuantOpts(o["paired"], o["fld"], o["sd"], o["single_overhang"], o["strand"], o["no_jump"])
        ctx.pseudoalign(opts, words, lens, hi - lo, max_len)
        # device -> host -> gloo -> device (kallisto_amd.Context.allreduce_ec_counts does the same on RCCL without the hops)
        dense = ctx.dense_counts()
        d_cpu = dense.cpu()
        w, offs = ctx.tuples_export()
        w, offs = merge_ec_state(d_cpu, w.cpu(), offs.cpu())
        dense.copy_(d_cpu)
        ctx.tuples_replace(w.cuda(), offs.cuda())
        ew, eo = ctx.explicit_export()
        ew, eo = gather_records(ew.cpu(), eo.cpu())
        ctx.explicit_replace(ew.cuda(), eo.cuda())
        ecs = ctx.finalize()
        # EM partitioned over the ranks by connected component (gloo all-reduce on the device tensors)
        import kallisto_amd.api as A
        exp = common.load_expected(case, variant)
        alpha, abz, rounds = ctx.em_run_partitioned(exp["eff"])
        alpha1, abz1, rounds1 = ctx.em_run(exp["eff"])          # the single-GPU EM on the same ECs
        q.put((rank, ecs.multiset(), alpha, abz, rounds, alpha1, abz1, rounds1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,variant", [("human_pe", "pe"), ("human_pe", "pe_rf"), ("yeast_se", "se")])
def test_two_ranks_one_gpu(case, variant):
    exp = common.load_expected(case, variant)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, variant, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = {g[0]: g for g in got}
    assert results[0][1] == results[1][1] == exp["ecs"]
    for r in (0, 1):
        _, _, alpha, abz, rounds, alpha1, abz1, rounds1 = results[r]
        assert rounds == rounds1                                    # the partitioned EM stops at the same round
        common.assert_abundance_close(alpha, exp["alpha"], "partitioned EM vs reference")
        common.assert_abundance_close(alpha, alpha1, "partitioned EM vs single-GPU EM", rel=1e-9)
        common.assert_abundance_close(abz, abz1, "alpha_before_zeroes", rel=1e-9, floor=1e-12)
    assert np.array_equal(results[0][2], results[1][2])             # every rank ends with the same vector
