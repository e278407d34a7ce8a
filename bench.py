#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric on MI355X: M paired-end reads/s pseudoaligned + quantified against a
human-transcriptome-sized index (BASELINE.json configs[2]; configs[3] is the same workload on N GPUs).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One *step* = one full `kallisto quant` pass over this rank's batch of synthetic read pairs already resident in HBM in
the 2-bit packed layout: k-mer pseudoalignment (kernel A) -> EC counts -> [N>1: RCCL all-reduce of the dense EC count
vector + all-gather of the tuple records] -> EC resolution/merge -> fragment-length sample -> EM (kernel B) -> TPM.
Scaling is weak: every rank processes its own `--pairs` read pairs (different seeds), so the job is N x pairs per step.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel by time, one EM round (k_pm_rows_pass + k_pm_cols_pass): SURVEY.md 8(d)'s algorithmic bytes of
                  an EM iteration / the round's HIP-event duration vs the 8 TB/s HBM peak; roofline_kernel_a: the same for k_match_v2
  cpu_baseline -- the unmodified reference (oracle/_ref/kallisto quant, built from /root/reference) on this box's host
                  cores over a bounded sample of the same reads, same index file; plus a parity check of the GPU path
                  against that run on the same sample.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
CACHE = os.environ.get("KAMD_BENCH_CACHE", "/tmp/kallisto_amd_cache")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "kallisto")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def wait_for(path, timeout=3600):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError(f"timed out waiting for {path}")
        time.sleep(1.0)


def prepare_workload(name: str, genes: int, is_builder: bool):
    """Synthetic transcriptome + index built by the reference binary; cached under CACHE."""
    from kallisto_amd import synth
    os.makedirs(CACHE, exist_ok=True)
    tag = f"{name}_g{genes}_v1"
    npz, idx, done = (os.path.join(CACHE, tag + ext) for ext in (".npz", ".idx", ".done"))
    if not os.path.exists(done):
        if is_builder:
            t0 = time.time()
            seqs = synth.human_like(n_genes=genes, seed=2) if name == "human" else synth.yeast_like(n_tr=genes, seed=1)
            lens = np.array([len(s) for s in seqs], np.int64)
            cat = np.concatenate(seqs)
            np.savez(npz, cat=cat, lens=lens)
            log(f"synthetic transcriptome: {len(seqs)} transcripts, {cat.size/1e6:.1f} Mbp in {time.time()-t0:.0f}s")
            if not os.path.exists(REF_BIN):
                raise RuntimeError(f"{REF_BIN} missing: the index is built by the reference binary (make -C oracle ref)")
            fa = os.path.join(CACHE, tag + ".fa")
            synth.write_fasta(fa, seqs)
            t0 = time.time()
            threads = min(os.cpu_count() or 8, 32)
            subprocess.check_call([REF_BIN, "index", "-t", str(threads), "-i", idx + ".tmp", fa], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
            os.replace(idx + ".tmp", idx)
            os.remove(fa)
            log(f"reference `kallisto index -t {threads}`: {time.time()-t0:.0f}s, {os.path.getsize(idx)/1e6:.0f} MB")
            open(done, "w").write("ok")
        else:
            wait_for(done)
    z = np.load(npz)
    return z["cat"], z["lens"], idx


def write_fastq_fast(path, reads: np.ndarray):
    n, L = reads.shape
    ids = np.char.zfill(np.arange(n).astype("U9"), 9).astype("S9")
    line = np.zeros((n, 2 + 9 + 1 + L + 1 + 2 + L + 1), np.uint8)
    line[:, 0] = ord("@"); line[:, 1] = ord("r")
    line[:, 2:11] = np.frombuffer(ids.tobytes(), np.uint8).reshape(n, 9)
    line[:, 11] = 10
    line[:, 12:12 + L] = reads
    line[:, 12 + L] = 10
    line[:, 13 + L] = ord("+"); line[:, 14 + L] = 10
    line[:, 15 + L:15 + 2 * L] = ord("I")
    line[:, 15 + 2 * L] = 10
    with open(path, "wb") as f:
        f.write(line.tobytes())


def cpu_reference_baseline(idx_path, r1: np.ndarray, r2: np.ndarray, threads: int):
    """Time the unmodified reference on the host cores: `kallisto quant -t threads` on the sample.  The clock starts when
    the index has been loaded (the '[quant] running in' line) and stops at process exit; the stage markers the reference
    prints on stderr split it into pseudoalignment (until the 'finding pseudoalignments ... done' line completes) and EM
    (until 'the Expectation-Maximization algorithm ran for')."""
    tmp = os.path.join(CACHE, f"cpu_baseline_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    f1, f2 = os.path.join(tmp, "s_1.fq"), os.path.join(tmp, "s_2.fq")
    write_fastq_fast(f1, r1)
    write_fastq_fast(f2, r2)
    out = os.path.join(tmp, "out")
    cmd = [REF_BIN, "quant", "-i", idx_path, "-o", out, "-t", str(threads), "--plaintext", f1, f2]
    t_start = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    t_loaded = t_aligned = t_em = None
    for raw in p.stderr:
        line = raw.decode(errors="replace")
        now = time.time()
        if t_loaded is None and line.startswith("[quant] running in"):
            t_loaded = now
        elif t_aligned is None and line.startswith("[quant] finding pseudoalignments"):
            t_aligned = now
        elif t_em is None and "Expectation-Maximization algorithm ran for" in line:
            t_em = now
    p.wait()
    t_end = time.time()
    if p.returncode != 0:
        raise RuntimeError("reference kallisto quant failed")
    info = json.load(open(os.path.join(out, "run_info.json")))
    shutil.rmtree(tmp, ignore_errors=True)
    t_loaded = t_loaded or t_start
    t_aligned = t_aligned or t_loaded
    t_em = t_em or t_end
    return {"seconds": t_end - t_loaded, "index_load_s": t_loaded - t_start, "pseudoalign_s": t_aligned - t_loaded,
            "em_s": t_em - t_aligned, "n_processed": info["n_processed"], "n_pseudoaligned": info["n_pseudoaligned"],
            "n_unique": info["n_unique"]}


def reference_parity(idx_path, r1: np.ndarray, r2: np.ndarray, res):
    """The GPU result `res` (kallisto_amd.quant on exactly these pairs, ECs downloaded) against the unmodified reference run
    deterministically (`-t 1`: the fragment-length sample is the first 10 000 qualifying pairs in input order,
    src/ProcessReads.cpp:981-1017,1174-1181) through oracle/_ref/dump_ec, which prints what the CLI never does: the EC
    multiset, flens, eff_lens and alpha of src/main.cpp:2632-2689."""
    from oracle import oracle as O
    tmp = os.path.join(CACHE, f"ref_parity_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    try:
        f1, f2 = os.path.join(tmp, "p_1.fq"), os.path.join(tmp, "p_2.fq")
        write_fastq_fast(f1, r1)
        write_fastq_fast(f2, r2)
        t0 = time.time()
        ref = O.ref_dump_quant(idx_path, [f1, f2], threads=1)
        rep = O.ref_parity_report(ref, res.ecs.multiset(), res.flens, res.eff_lens, res.est_counts, res.alpha_before_zeroes)
        rep["sample_pairs"] = int(r1.shape[0])
        rep["reference_seconds"] = round(time.time() - t0, 1)
        rep["n_pseudoaligned"] = [int(res.n_pseudoaligned), int(sum(ref["ecs"].values()))]
        rep["em_rounds_gpu"] = int(res.em_rounds)
        return rep
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=30_000_000, help="read pairs per GPU per step (BASELINE config #3: 30 M)")
    ap.add_argument("--workload", default="human", choices=["human", "yeast"])
    ap.add_argument("--genes", type=int, default=None, help="scale of the synthetic transcriptome (default: full config)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="pairs given to the CPU reference (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=200_000,
                    help="pairs of the CPU sample that also go through the reference at -t 1 for the parity gate")
    ap.add_argument("--bootstraps", type=int, default=0,
                    help="BASELINE config #5: also time B bootstrap replicates (multinomial resample + EM), split over the ranks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import kallisto_amd as ka
    from kallisto_amd.synth_gpu import ReadSimulator

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # KAMD_BENCH_SHARE_GPU=1 + KAMD_BENCH_BACKEND=gloo: several ranks on ONE GPU -- a smoke test of the multi-rank flow on a
    # single-GPU box (RCCL needs one device per rank); never used for reported numbers
    if os.environ.get("KAMD_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("KAMD_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    genes = args.genes or (20000 if args.workload == "human" else 6000)
    cat, tlens, idx_path = prepare_workload(args.workload, genes, is_builder=(rank == 0))
    t0 = time.time()
    index = ka.Index(idx_path)
    log(f"index flattened in {time.time()-t0:.1f}s: k={index.k} targets={index.num_targets} k-mers={index.num_kmers} "
        f"unitigs={index.num_unitigs} index ECs={index.num_ecs}")
    ctx = ka.Context(local)
    ctx.upload(index)

    # ---- this rank's reads, generated on the device and packed into the 2-bit layout (resident in HBM) ----
    L = 100
    sim = ReadSimulator(cat, tlens, dev, seed=1000 + rank, read_len=L)
    n = args.pairs
    rec = ka.packed_record_words(L)
    words = torch.empty(n * 2 * rec, dtype=torch.int32, device=dev)
    lens = torch.empty(n * 2, dtype=torch.int16, device=dev)
    chunk = 2_000_000
    sample = None
    t0 = time.time()
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        r1, r2 = sim.draw(m)
        inter = torch.stack([r1, r2], 1).reshape(2 * m, L)  # mate 1, mate 2 interleaved (ProcessReads.cpp:1034-1041)
        w, l = ctx.pack_reads(inter, L)
        words[s * 2 * rec:(s + m) * 2 * rec] = w
        lens[2 * s:2 * (s + m)] = l
        if s == 0 and rank == 0 and args.cpu_sample and not args.no_cpu_baseline:
            k = min(args.cpu_sample, m)
            sample = (r1[:k].cpu().numpy(), r2[:k].cpu().numpy())
        del r1, r2, inter, w, l
    torch.cuda.synchronize()
    log(f"{n} synthetic PE-{L} pairs generated + packed on the device in {time.time()-t0:.1f}s ({words.numel()*4/1e9:.2f} GB in HBM)")

    opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0)

    def step():
        ctx.reset()
        return ka.quant(ctx, opts, [(words, lens, n, L)], download_ecs=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    fence()
    t0 = time.perf_counter()
    align_ms, em_ms, em_iters, cls_ms = [], [], [], []
    for _ in range(args.steps):
        res = step()
        pr = ctx.profile()
        align_ms.append(pr["align_kernel_ms"]); em_ms.append(pr["em_ms"]); em_iters.append(pr["em_iters"]); cls_ms.append(pr["classify_ms"])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = res.stats
    total_pairs = n * world * args.steps
    # ---- BASELINE config #5 (optional): B bootstrap replicates of the last step's ECs, replicate b on rank b % world ----
    boot = None
    if args.bootstraps > 0:
        import kallisto_amd.api as A
        seeds = A.bootstrap_seeds(42, args.bootstraps)
        mine = [b for b in range(args.bootstraps) if b % world == rank]
        fence()
        tb = time.perf_counter()
        rounds_b = []
        if mine:
            _, rb = ctx.bootstrap_batch(seeds[mine], res.eff_lens)   # one multinomial launch, EMs on the cached plan
            rounds_b = [int(x) for x in rb]
        fence()
        tb = time.perf_counter() - tb
        if world > 1:
            t = torch.tensor([tb], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tb = float(t.item())
        boot = {"replicates": args.bootstraps, "seconds": round(tb, 4), "replicates_per_s": round(args.bootstraps / tb, 3),
                "ms_per_replicate_per_gpu": round(tb / max(len(mine), 1) * 1e3, 2),
                "em_rounds_first": rounds_b[:3], "note": "Bootstrap::run_em per replicate: multinomial resample of the EC counts "
                "(N = pseudoaligned pairs draws, libstdc++ semantics; all replicates of a rank drawn in one launch) + EM run(10000, 50) on "
                "the cached plan of the EC matrix; replicate b runs on rank b % world (every rank holds the merged ECs)"}

    out = None
    if rank == 0:
        # roofline of kernel A: algorithmic bytes of ONE launch: packed reads in + 16 B (key+payload)
        # per k-mer probe + what the launch writes (4 B per single-set count, the tuple record and its 8-byte offset)
        # (the counters are reset every step, so `st` describes exactly one launch)
        rec_bytes = 2 * rec * 4 + 2 * 2
        if pr["kernel_a_version"] == 2:   # k_match_v2 writes one raw record per item: header + distinct (unitig,set) classes
            alg_bytes = n * rec_bytes + 16 * st["n_probes"] + 4 * st["n_raw_words"]
        else:
            alg_bytes = n * rec_bytes + 16 * st["n_probes"] + 4 * st["n_single"] + 4 * st["n_stream_words"] + 8 * st["n_multi"]
        a_ms = float(np.mean(align_ms))
        achieved = alg_bytes / (a_ms * 1e-3) / 1e9
        # EM round = the two launches k_pm_rows_pass + k_pm_cols_pass (streamed form; CSR form: k_em_rows + k_em_seg + k_em_final).
        # `achieved` uses SURVEY.md section 8(d)'s ALGORITHMIC bytes of one EM iteration, independent of the layout:
        #   B_B = nnz*(4 id + 8 alpha gather + 8 next accumulate) + N_ec*(4 count + 8 offsets) + T*(8 alpha + 8 next + 8 eff_len)
        # `layout_bytes_per_round` is what the streamed layout actually has to move per round:
        #   2 passes * nnz_multi*(4 index + 8 gather) + rows*(8 count word + 8 g) + transcripts*(4 reads + 3 writes)*8
        T = int(index.num_targets)
        em_bytes = pr["em_nnz"] * 20 + pr["em_necs"] * 12 + T * 24
        if pr["em_k"]:
            layout_bytes = pr["em_nnz_multi"] * 24 + pr["em_necs"] * 16 + T * 56
            em_kernel = "EM round (k_pm_rows_pass + k_pm_cols_pass)"
        else:
            layout_bytes = pr["em_nnz_multi"] * 24 + pr["em_necs"] * 24 + pr["em_nseg"] * 28 + T * 64
            em_kernel = "EM round (k_em_rows + k_em_seg + k_em_final)"
        em_round_ms = float(np.mean(em_ms)) / max(int(em_iters[-1]), 1)
        em_ach = em_bytes / (em_round_ms * 1e-3) / 1e9
        em_roof = {"kernel": em_kernel, "bound": "hbm", "achieved": round(em_ach, 2),
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(em_ach / HBM_PEAK_GBS, 5), "traffic": None,
                   "algorithmic_bytes_per_launch": int(em_bytes), "layout_bytes_per_round": int(layout_bytes),
                   "launch_ms": round(em_round_ms, 5), "launch": "one EM round (all its launches; kamd_em_run's HIP-event time / rounds, set-up included)",
                   "rounds": int(em_iters[-1]), "nnz": pr["em_nnz"], "nnz_multi": pr["em_nnz_multi"], "rows": pr["em_necs"],
                   "entries_per_lane": pr["em_k"], "chunks": pr["em_nseg"] if pr["em_k"] else None}
        out = {
            "metric": "M paired-end reads/sec quantified (human txome index)",
            "value": round(total_pairs / elapsed / 1e6, 4),
            "unit": "M read pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64+f64", "data": "synthetic",
            "config": {
                "workload": (f"BASELINE config #3: synthetic human-like transcriptome ({index.num_targets} transcripts, "
                             f"{index.num_kmers} k-mers, k={index.k}; index built by the reference `kallisto index`), "
                             f"{n} PE-{L} read pairs per GPU resident in HBM (2-bit packed), full quant per step"
                             if args.workload == "human" and genes == 20000 else
                             f"REDUCED {args.workload} genes={genes} pairs={n} (not the BASELINE configuration)"),
                "pairs_per_gpu": n, "read_len": L, "paired": True, "targets": int(index.num_targets),
                "kmers": int(index.num_kmers),
                "parallelism": (f"{world} ranks, one per GPU: reads sharded, EC counts all-reduced + tuple records all-gathered (RCCL), "
                                f"EM partitioned over the ranks by connected component" if world > 1 else "1 GPU"),
                "collective_backend": backend if world > 1 else None,
            },
            "breakdown_ms": {"pseudoalign_kernel": round(a_ms, 3), "classify_kernel": round(float(np.mean(cls_ms)), 3),
                             "kernel_a_version": pr["kernel_a_version"], "em": round(float(np.mean(em_ms)), 3),
                             "em_rounds": int(em_iters[-1]), "step_total": round(elapsed / args.steps * 1e3, 3)},
            "counters": {"probes_per_pair": round(st["n_probes"] / n, 3),
                         "bucket_reads_per_probe": round(st["n_bucket_reads"] / max(st["n_probes"], 1), 4),
                         "single_set_pairs": st["n_single"], "multi_set_pairs": st["n_multi"],
                         "distinct_tuples": st["n_distinct_tuples"], "final_ecs": int(ctx.ec_result.n_ecs),
                         "em_rounds": res.em_rounds},
            # dominant kernel by time: one EM round
            "roofline": em_roof,
            "roofline_kernel_a": {"kernel": "k_match_v2" if pr["kernel_a_version"] == 2 else "k_pseudoalign", "bound": "hbm",
                                  "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                                  "algorithmic_bytes_per_launch": int(alg_bytes), "launch_ms": round(a_ms, 3),
                                  "bucket_line_bytes_per_launch": int(64 * st["n_bucket_reads"])},
        }
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                tj = json.load(open(prof))
                if tj.get("pairs") == n and tj.get("workload") == args.workload and tj.get("genes") == genes:
                    out["roofline"]["traffic"] = tj.get("em_round_hbm_bytes")
                    out["roofline_kernel_a"]["traffic"] = tj.get("kernel_a_hbm_bytes")
                    out["roofline"]["traffic_source"] = out["roofline_kernel_a"]["traffic_source"] = tj.get("source")
            except Exception:
                pass
    # ---- CPU baseline (rank 0, N=1 only): the reference at -t <cores> for the timing; parity against the reference at -t 1 ----
    if rank == 0 and world == 1 and sample is not None:
        threads = min(os.cpu_count() or 1, 64)
        k = sample[0].shape[0]
        log(f"CPU baseline: reference `kallisto quant -t {threads}` on the first {k} pairs ...")
        try:
            cb = cpu_reference_baseline(idx_path, sample[0], sample[1], threads)
            out["cpu_baseline"] = {"value": round(k / cb["seconds"] / 1e6, 4), "unit": "M read pairs/s", "cores": threads,
                                   "kind": "reference",
                                   "sample": f"first {k} pairs of rank 0's reads as uncompressed FASTQ, `kallisto quant -t {threads} "
                                             f"--plaintext`, clock from index-loaded to exit ({cb['seconds']:.1f}s; index load "
                                             f"{cb['index_load_s']:.1f}s excluded)",
                                   "pseudoalign_seconds": round(cb["pseudoalign_s"], 2), "em_seconds": round(cb["em_s"], 2),
                                   "pseudoalign_only_value": round(k / max(cb["pseudoalign_s"], 1e-9) / 1e6, 4),
                                   "note": "the reference's EM is single-threaded and independent of the read count (it dominates "
                                           "small samples); pseudoalign_only_value is the rate of the threaded stage alone"}
        except Exception as e:  # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "M read pairs/s", "cores": threads, "kind": "reference",
                                   "sample": f"failed: {e}"}
        # parity gate: the same pairs through the HIP path and through the unmodified reference at -t 1
        ks = min(k, args.parity_sample)
        log(f"parity: reference `dump_ec quant -t 1` on the first {ks} pairs ...")
        try:
            ctx.reset()
            gres = ka.quant(ctx, opts, [(words[:ks * 2 * rec], lens[:2 * ks], ks, L)], download_ecs=True)
            out["parity_check"] = reference_parity(idx_path, sample[0][:ks], sample[1][:ks], gres)
        except Exception as e:
            out["parity_check"] = {"ok": False, "error": str(e)}
    if rank == 0:
        if boot is not None:
            out["bootstrap"] = boot
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
