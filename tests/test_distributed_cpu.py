"""World-size-2 test of the multi-GPU merge on CPU (gloo): each rank turns its shard of the reads into the EC state
kernel A would produce (dense counts + tuple records, via the CPU emulation of the per-item logic), the states are
merged with tests/exchange_gloo.py merge_ec_state (all-reduce + all-gather), resolved, and the result must equal the
reference's EC multiset for the WHOLE input -- i.e. sharding reads over ranks does not change EC counts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.exchange_gloo import merge_ec_state
        from tests import emu_binding as E
        meta, idx_path, r1, r2 = common.load_case(case)
        paired = r2 is not None
        n = len(r1)
        lo, hi = rank * n // world, (rank + 1) * n // world  # contiguous shard of the reads
        ix = E.EmuIndex(idx_path)
        words, l16, max_len = E.pack(common.interleave(r1[lo:hi], r2[lo:hi] if paired else None), 100)
        dense, stream, rec_off = E.ec_state(ix, words, l16, hi - lo, paired, max_len)
        d = torch.from_numpy(dense.view(np.int32).copy())
        w, o = merge_ec_state(d, torch.from_numpy(stream.view(np.int32).copy()), torch.from_numpy(rec_off.astype(np.int64)))
        ms = E.resolve(ix, d.numpy().view(np.uint32), w.numpy().view(np.uint32), o.numpy().astype(np.uint64))
        q.put((rank, ms))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,variant", [("human_pe", "pe"), ("ref_test_pe", "pe"), ("yeast_se", "se_overhang")])
def test_sharded_ec_counts_equal_reference(case, variant):
    exp = common.load_expected(case, variant)
    from tests import emu_binding
    emu_binding.lib()  # build the emulation library once, before the ranks race for it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert results[0] == results[1] == exp["ecs"]


def test_merge_is_identity_without_process_group():
    from tests.exchange_gloo import merge_ec_state
    d = torch.arange(5, dtype=torch.int32)
    w = torch.tensor([1, 2, 3, 4], dtype=torch.int32)
    o = torch.tensor([0], dtype=torch.int64)
    w2, o2 = merge_ec_state(d, w, o)
    assert torch.equal(w, w2) and torch.equal(o, o2) and d.tolist() == [0, 1, 2, 3, 4]


def _bare_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "KAMD_BENCH_SHARE_GPU",
                                                           "KAMD_BENCH_BACKEND", "KAMD_COMM")}
    env.update(extra)
    return env


def test_bench_self_launch_refuses_without_enough_gpus():
    """`python bench.py --gpus 2` from a bare shell is its own launcher; on a box with fewer GPUs than ranks it says so (exit code 2, nothing
    on stdout) instead of exiting with 'launch with torch.distributed.run' as it did through round 3."""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=_bare_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300)
    assert p.returncode == 2, p.stderr.decode()[-2000:]
    assert b"GPU(s)" in p.stderr and not p.stdout.strip()


@pytest.mark.skipif(torch.cuda.is_available(), reason="GPU present: test_gpu_multirank runs the real thing")
def test_bench_self_launch_starts_ranks_and_reports_their_failure():
    """no GPU here: the launcher must start both ranks under torch.distributed.run (they die at the device), find no result line, and
    return a non-zero exit code with nothing on stdout -- never hang, never print a half line"""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genes", "50", "--pairs", "1000"], cwd=ROOT,
                       env=_bare_env(KAMD_BENCH_SHARE_GPU="1", KAMD_BENCH_LAUNCH_TIMEOUT_S="240"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and not p.stdout.strip()
    err = p.stderr.decode()
    assert "launching 2 ranks" in err and "no result line from the ranks" in err


def _stub(tmp_path, body):
    p = tmp_path / "stub_rank.py"
    p.write_text("import json, os, sys\nrank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n" + body)
    return str(p)


def test_bench_self_launch_passes_rank_zeros_line_through(tmp_path):
    """the launcher half of `python bench.py --gpus N` with a stand-in for the ranks: N processes get RANK / WORLD_SIZE / MASTER_* from
    torch.distributed.run on 127.0.0.1, rank 0's ONE line reaches stdout and nothing else does, exit code 0"""
    import json
    import subprocess
    stub = _stub(tmp_path, "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ.get('KAMD_BENCH_LAUNCHER') == 'self'\n"
                           "print('noise from rank', rank)\n"
                           "if rank == 0: print(json.dumps({'metric': 'stub', 'n_gpus': world, 'argv': sys.argv[1:]}))\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2"], cwd=ROOT, env=_bare_env(KAMD_BENCH_LAUNCH_SCRIPT=stub),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["argv"] == ["--gpus", "3", "--steps", "2"]


def test_bench_self_launch_retries_with_the_process_groups_collectives(tmp_path):
    """ranks that die with the library's own RCCL communicator (no KAMD_COMM in their environment) are started once more with
    KAMD_COMM=callbacks, and that attempt's line is the one that comes back"""
    import json
    import subprocess
    stub = _stub(tmp_path, "if os.environ.get('KAMD_COMM') != 'callbacks': sys.exit(7)\n"
                           "if rank == 0: print(json.dumps({'metric': 'stub', 'transport': os.environ['KAMD_COMM']}))\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=_bare_env(KAMD_BENCH_LAUNCH_SCRIPT=stub),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["transport"] == "callbacks"
    assert "no result line from the ranks" in p.stderr.decode()
