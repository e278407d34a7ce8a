"""Two ranks on ONE GPU (gloo for the collectives, since RCCL needs one device per rank): each rank pseudoaligns its
shard on the device, the library merges the EC states (kamd_ec_allreduce over a communicator whose collectives are gloo's:
all-reduce of the dense vector + all-gathers of the tuple / explicit-set records), both ranks finalize to the reference's EC
multiset for the WHOLE input and run the EM partitioned over the ranks (kamd_em_run_comm).  Every device-side piece of the
multi-GPU path that the 8-GPU bench relies on runs here; only the transport differs (RCCL there).  test_rccl_world_of_one
drives the RCCL backend itself (ncclCommInitRank, all-reduce, all-gather, broadcast) on the one GPU this box has."""
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
        # the library's own merge (kamd_ec_allreduce: all-reduce of the dense vector + all-gathers of the tuple / explicit-set
        # records) over a communicator whose collectives are gloo's -- on a node with one GPU per rank the same calls run on RCCL
        comm = ka.Comm.over_process_group(ctx)
        ctx.ec_allreduce(comm)
        ecs = ctx.finalize()
        # EM partitioned over the ranks by connected component (kamd_em_run_comm)
        exp = common.load_expected(case, variant)
        alpha, abz, rounds = ctx.em_run_comm(comm, exp["eff"])
        # human_pe: the partitioned run used the component-local kernel (em_k == -2) on this rank's components; yeast_se's matrix
        # is (almost) all singleton rows, where every rank agrees on the streamed form
        form_k = ctx.profile()["em_k"]
        assert form_k == -2 or case != "human_pe", form_k
        ctx.tune(em_form="streamed")
        alpha_s, abz_s, rounds_s = ctx.em_run_comm(comm, exp["eff"])     # the same partitioned run, streamed kernels
        assert rounds_s == rounds
        common.assert_abundance_close(alpha_s, alpha, "partitioned EM: streamed vs component-local", rel=1e-9)
        alpha1, abz1, rounds1 = ctx.em_run(exp["eff"])          # the single-GPU EM on the same ECs (streamed form)
        ctx.tune(em_form="local")
        alpha2, abz2, rounds2 = ctx.em_run(exp["eff"])          # ... and the component-local form
        assert rounds2 == rounds1
        common.assert_abundance_close(alpha2, alpha1, "component-local EM vs streamed EM", rel=1e-9)
        comm.close()
        q.put((rank, ecs.multiset(), alpha, abz, rounds, alpha1, abz1, rounds1))
    finally:
        dist.destroy_process_group()


def _hybrid_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kallisto_amd as ka
        from tests.test_gpu_parity import _hybrid_csr, _distinct_rows
        off, ids, cnt, eff, T = _distinct_rows(*_hybrid_csr(3000, 11))
        ctx = ka.Context(0)
        ctx.tune(em_form="local")
        ctx.ec_upload(off, ids, cnt)
        comm = ka.Comm.over_process_group(ctx)
        alpha, abz, rounds = ctx.em_run_comm(comm, eff)
        prof = ctx.profile()
        alpha1, abz1, rounds1 = ctx.em_run(eff)       # one rank, the hybrid on the whole matrix
        prof1 = ctx.profile()
        comm.close()
        q.put((rank, alpha, abz, rounds, prof["em_k"], prof["em_giant_nnz"], alpha1, abz1, rounds1, prof1["em_giant_nnz"]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_hybrid_em():
    """a matrix with ONE component beyond a workgroup's LDS under kamd_em_run_comm.  Components are dealt to the ranks by hash(label): the
    rank that owns the oversized one runs the hybrid (LDS groups + blocked / streamed kernels for that component) on its rows, the other rank the plain
    component-local form; until round 5 every rank fell back to the whole-matrix streamed form.  Same rounds and abundances as one rank and as the oracle."""
    from oracle import oracle as O
    from tests.test_gpu_parity import _hybrid_csr, _distinct_rows
    off, ids, cnt, eff, T = _distinct_rows(*_hybrid_csr(3000, 11))
    alpha_o, abz_o, rounds_o = O.em_run(off, ids, cnt, eff, T)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hybrid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = {g[0]: g for g in got}
    assert sum(1 for r in (0, 1) if res[r][5] > 20000) == 1, [res[r][5] for r in (0, 1)]   # exactly one rank iterated the oversized component beside its groups
    for r in (0, 1):
        _, alpha, abz, rounds, em_k, giant, alpha1, abz1, rounds1, giant1 = res[r]
        assert em_k == -2 and giant1 > 20000
        assert rounds == rounds1 == rounds_o
        common.assert_abundance_close(alpha, alpha_o, "two ranks vs oracle", rel=1e-9)
        common.assert_abundance_close(alpha, alpha1, "two ranks vs one", rel=1e-9)
        tiny = lambda x: np.where(np.abs(x) < 1e-200, 0.0, x)
        common.assert_abundance_close(tiny(abz), tiny(abz1), "alpha_before_zeroes", rel=1e-9, floor=1e-12)
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("case,variant", [("human_pe", "pe"), ("human_pe", "pe_rf"), ("yeast_se", "se"), ("stress_pe", "pe")])
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


def test_rccl_world_of_one():
    """The RCCL backend on one rank: the library finds librccl, creates a communicator from its own unique id and runs every
    collective it uses (a world of one leaves the data unchanged)."""
    import kallisto_amd as ka
    meta, idx_path, r1, r2 = common.load_case("human_pe")
    index = ka.Index(idx_path)
    ctx = ka.Context(0)
    try:
        ctx.upload(index)
        uid = ka.Comm.unique_id()
        assert len(uid) == 128 and any(uid)
        comm = ka.Comm.rccl(ctx, 0, 1, uid)
        words, lens, max_len = ctx.pack_reads_host(common.interleave(r1, r2), 100)
        opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)
        res = ka.quant(ctx, opts, [(words, lens, len(r1), max_len)], comm=comm)
        exp = common.load_expected("human_pe", "pe")
        assert res.ecs.multiset() == exp["ecs"] and res.n_processed == len(r1)
        assert np.array_equal(res.flens, exp["flens"])
        common.assert_abundance_close(res.est_counts, exp["alpha"], "est_counts")
        assert comm.sum_int(7) == 7
        assert np.array_equal(comm.broadcast_np(np.arange(5, dtype=np.uint32)), np.arange(5, dtype=np.uint32))
        comm.close()
    finally:
        ctx.close()


def test_bench_two_ranks_on_one_gpu_with_parity(tmp_path):
    """`python bench.py --gpus 2` from a bare shell (no launcher around it: the script starts its own ranks under torch.distributed.run) with
    both ranks on the one GPU of the box (KAMD_BENCH_SHARE_GPU=1, collectives as callbacks over gloo): the multi-rank flow of the bench end
    to end -- reads sharded, kamd_ec_allreduce, kamd_em_run_comm, both scaling modes -- and its parity leg: the merged result must be what
    ONE rank computes from all ranks' reads (EC multiset and flens identical, same EM rounds).  Exactly one line comes back."""
    import json
    import subprocess
    import sys
    import bench
    if not os.path.exists(bench.REF_BIN):
        pytest.skip("oracle/_ref/kallisto not built: cannot create the index of the bench workload")
    env = dict(os.environ, KAMD_BENCH_SHARE_GPU="1", KAMD_BENCH_CACHE=str(tmp_path / "cache"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "KAMD_BENCH_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genes", "1500", "--pairs", "400000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--multi-parity", "--detail-file", str(tmp_path / "detail.json")], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.CONTRACT_MAX_BYTES, lines
    short = json.loads(lines[0])   # the contract line: figures only
    assert short["n_gpus"] == 2 and short["value"] > 0 and short["config"]["n_ranks_seen"] == 2 and short["parity"]["multi_rank_ok"] is True
    assert short["roofline"]["frac"] > 0 and short["other_scaling"]["scaling"] == "strong"
    line = json.load(open(tmp_path / "detail.json"))   # the detail document of the same run
    assert line["n_gpus"] == 2 and line["value"] == short["value"] and line["scaling"] == "weak" and line["other_scaling"]["scaling"] == "strong"
    assert "callbacks" in line["config"]["collective_backend"] and "gloo" in line["config"]["collective_backend"]     # the transport actually used
    assert line["config"]["n_ranks_seen"] == 2 and line["config"]["launcher"].startswith("self")
    assert line["breakdown_ms"]["collective_ms"] > 0 and line["breakdown_ms"]["em_collectives_n"] >= 1   # the collectives' share of a step is in the line
    mp = line["multi_rank_parity"]
    assert mp["ok"], mp
    assert mp["ranks"] == 2 and mp["ec_multiset_equal"] and mp["flens_equal"] and mp["em_rounds"][0] == mp["em_rounds"][1]
    # the merged result of the two ranks against the unmodified reference on both ranks' reads (800 k pairs, 9 k transcripts)
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")):
        mr = mp["merged_vs_reference"]
        assert mr["ok"] and mr["ec_multiset_equal"] and mr["flens_equal"] and mr["eff_length_equal"], mr


def test_bench_two_ranks_without_flags_carries_parity_and_cpu_baseline(tmp_path):
    """`KAMD_BENCH_SHARE_GPU=1 python bench.py --gpus 2` with no parity / baseline flag at all -- what the driver's scaling run passes -- must print a
    line with `roofline`, a `cpu_baseline` and a parity verdict: the merged result of the ranks on a bounded sample (here 100 k pairs per rank) against the
    unmodified reference on those pairs in rank order."""
    import json
    import subprocess
    import sys
    import bench
    if not (os.path.exists(bench.REF_BIN) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec"))):
        pytest.skip("oracle/_ref not built")
    env = dict(os.environ, KAMD_BENCH_SHARE_GPU="1", KAMD_BENCH_CACHE=str(tmp_path / "cache"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "KAMD_BENCH_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genes", "1500", "--pairs", "300000", "--steps", "2", "--warmup", "1",
                        "--parity-sample", "100000", "--detail-file", str(tmp_path / "detail.json")], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    text = [ln for ln in p.stdout.decode().splitlines() if ln.strip()][-1]
    short = json.loads(text)
    assert len(text) < bench.CONTRACT_MAX_BYTES and short["n_gpus"] == 2 and short["roofline"]["frac"] > 0 and short["parity"]["multi_rank_ok"] is True
    assert short["cpu_baseline"]["kind"] == "reference" and short["cpu_baseline"]["value"] > 0 and short["cpu_baseline"]["cores"] == 1 and short["config"]["n_ranks_seen"] == 2
    line = json.load(open(tmp_path / "detail.json"))
    assert line["n_gpus"] == 2 and line["roofline"]["frac"] > 0
    mp = line["multi_rank_parity"]
    assert mp["ok"] and mp["ec_multiset_equal"] and mp["flens_equal"] and mp["eff_length_equal"] and mp["sample"] == 200000, mp
    cb = line["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["value"] > 0 and cb["cores"] == 1, cb


def test_bench_refuses_more_ranks_than_gpus():
    """without the share-one-GPU switch the self-launcher must say why it cannot start instead of hanging in a rendezvous"""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KAMD_BENCH_SHARE_GPU")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 2 and b"GPU(s)" in p.stderr and not p.stdout.strip()
