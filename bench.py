#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric on MI355X: M paired-end reads/s pseudoaligned + quantified against a
human-transcriptome-sized index (BASELINE.json configs[2]; configs[3] is the same workload on N GPUs).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1 from a bare shell: the script launches its own N ranks, self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One *step* = one full `kallisto quant` pass over this rank's batch of synthetic read pairs already resident in HBM in
the 2-bit packed layout: k-mer pseudoalignment (kernel A) -> EC counts -> [N>1, inside the library over RCCL: all-reduce of
the dense EC count vector + all-gathers of the tuple records] -> EC resolution/merge -> fragment-length sample -> EM
(kernel B; N>1: partitioned over the ranks by connected component) -> TPM.
Scaling: weak by default (every rank processes its own `--pairs` read pairs, different seeds); `--scaling strong` shards
`--pairs` over the ranks (BASELINE config #4).  With N > 1 the other mode is measured in the same run and reported as
`other_scaling`.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline           -- the dominant kernel by time, kernel A (k_match_v3): SURVEY.md 8(d)'s algorithmic bytes of the launch / its
                        HIP-event duration vs the 8 TB/s HBM peak; roofline_em: one EM round (LDS-resident: against the LDS pipe,
                        with the HBM-equivalent figure next to it); roofline_finalize: the EC resolution kernels
  cpu_baseline       -- the unmodified reference (oracle/_ref, built from /root/reference) on this box's host cores: the stage clocks of the
                        full-size parity leg's own run (all pairs of the configuration, pseudoalignment and EM seconds); a run on a bounded sample of
                        the same reads when that leg is off
  parity_check       -- the GPU path against the unmodified reference run deterministically (-t 1, oracle/_ref/dump_ec) on a
                        prefix of the same reads: EC multiset, flens, eff_length identical, est_counts / TPM <= 1e-4; `ok` gates on all
  parity_check_tail  -- the last pairs of the input, pseudoaligned as the final batch of a run over ALL pairs (record stream grown and
                        reallocated, de-duplication table regrown): EC counts of the whole run minus those of the run without them must
                        equal the reference's EC multiset of those pairs
  parity_check_full_size -- ALL pairs of the run through the unmodified reference on all cores (written as FASTQ while they are generated):
                        EC multiset of the whole run, effective lengths and EM round count identical, est_counts / TPM <= 1e-4; the oracle's
                        EM on the GPU's own equivalence classes <= 1e-9 (FullSizeParity; runs in the background of the other legs)
  pinned_pipeline    -- packed reads in pinned host memory -> H2D on a copy stream -> pseudoalignment, double buffered
  end_to_end         -- the C++ front-end from FASTQ files (plain / BGZF / gzip; 8 M pairs, and plain_full_size*: all 30 M), input -> ECs
                        and whole-run rates, index load stated
  bootstrap          -- (--bootstraps B) BASELINE config #5: B replicates of multinomial resample + EM
  stress             -- a child run of `--workload stress` (a transcriptome with repeat families, paralog families and poly-A tails; 12 % off-transcriptome
                        pairs; a 3' quality tail) on 8 M pairs with ALL of them through the reference: its figures, parity verdict and CPU baseline
  gencode_size       -- a child run on a GENCODE-sized index (46 000 genes: ~444 k transcripts), prefix and tail parity
`--workload yeast` is BASELINE config #2 (10 M single-end reads, ~6 k transcripts); `--workload stress` the stress workload on its own.
N > 1 (no flags needed): `multi_rank_parity` (the merged result of the ranks on 200 k pairs per rank against the reference), a `cpu_baseline`,
`breakdown_ms.collective_ms`.
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


def effective_cpus():
    """CPUs this process may keep busy: the cgroup's CPU quota (containers show all host processors in os.cpu_count()), the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                n = min(n, max(1, int(q / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


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
            bg = np.zeros(0, np.uint8)
            if name == "stress":
                seqs, bg = synth.human_stress(n_genes=genes, seed=7)
            else:
                seqs = synth.human_like(n_genes=genes, seed=2) if name == "human" else synth.yeast_like(n_tr=genes, seed=1)
            lens = np.array([len(s) for s in seqs], np.int64)
            cat = np.concatenate(seqs)
            np.savez(npz, cat=cat, lens=lens, bg=bg)
            log(f"synthetic transcriptome: {len(seqs)} transcripts, {cat.size/1e6:.1f} Mbp in {time.time()-t0:.0f}s")
            if not os.path.exists(REF_BIN):
                raise RuntimeError(f"{REF_BIN} missing: the index is built by the reference binary (make -C oracle ref)")
            fa = os.path.join(CACHE, tag + ".fa")
            synth.write_fasta(fa, seqs)
            t0 = time.time()
            threads = min(effective_cpus(), 32)
            subprocess.check_call([REF_BIN, "index", "-t", str(threads), "-i", idx + ".tmp", fa], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
            os.replace(idx + ".tmp", idx)
            os.remove(fa)
            log(f"reference `kallisto index -t {threads}`: {time.time()-t0:.0f}s, {os.path.getsize(idx)/1e6:.0f} MB")
            open(done, "w").write("ok")
        else:
            wait_for(done)
    z = np.load(npz)
    prepare_workload.background = z["bg"] if "bg" in z.files and z["bg"].size else None   # (stress: what off-transcriptome fragments are drawn from)
    return z["cat"], z["lens"], idx


def write_fastq_fast(path, reads: np.ndarray, mate: int = 0):
    """the reads as FASTQ.  Default: records as a sequencer writes them (kallisto_amd/synth_fastq.py: variable-length Illumina headers of 63-65 bytes, binned
    quality strings that fall off towards the 3' end) -- rounds 1-4 wrote fixed 216-byte records with 11-byte headers and 100 x `I`, which flatters
    every byte-bound rate (20 % fewer bytes per pair) and the inflaters (constant quality lines are one long match); KAMD_BENCH_FASTQ=regular
    brings that form back for comparisons"""
    n, L = reads.shape
    if os.environ.get("KAMD_BENCH_FASTQ", "realistic") != "regular":
        from kallisto_amd import synth_fastq
        synth_fastq.write_fastq(path, reads, mate)
        return
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


class FastqSpool:
    """The WHOLE input of a run as FASTQ files, written while the reads are generated on the device: the text of a chunk is assembled
    there (`@r<9 digits>`, the read, `+`, L x `I`: what write_fastq_fast writes), copied to the host and appended by one writer thread per
    file.  These files are what the reference reads in the full-size parity leg and what the front-end reads in the full-size end-to-end legs."""

    def __init__(self, directory: str, paired: bool, L: int):
        import queue
        import threading
        os.makedirs(directory, exist_ok=True)
        self.dir, self.L, self.n = directory, L, 0
        self.files = [os.path.join(directory, f"full_{i + 1}.fq") for i in range(2 if paired else 1)]
        self.rec_bytes = 2 + 9 + 1 + L + 1 + 2 + L + 1
        self.realistic = os.environ.get("KAMD_BENCH_FASTQ", "realistic") != "regular"   # (write_fastq_fast: records as a sequencer writes them)
        self._ft = None
        self.error = None
        self._q = [queue.Queue(maxsize=2) for _ in self.files]
        self._th = [threading.Thread(target=self._writer, args=(i,), daemon=True) for i in range(len(self.files))]
        for t in self._th:
            t.start()
        self._tmpl = None

    def _writer(self, i):
        try:
            with open(self.files[i], "wb") as f:
                while True:
                    a = self._q[i].get()
                    if a is None:
                        return
                    f.write(memoryview(a.numpy()).cast("B"))
        except Exception as e:   # noqa: BLE001
            self.error = str(e)
            while self._q[i].get() is not None:
                pass

    def add(self, mates):
        """mates: the (m, L) uint8 device tensors of this chunk, one per file"""
        import torch
        m, L, dev = mates[0].shape[0], self.L, mates[0].device
        if self.realistic:
            if self._ft is None:
                from kallisto_amd import synth_fastq
                self._ft = [synth_fastq.FastqText(L, i, dev) for i in range(len(mates))]
            for i, r in enumerate(mates):
                self._q[i].put(self._ft[i].text(r, self.n).cpu())
            self.n += m
            return
        if self._tmpl is None or self._tmpl.shape[0] < m:
            t = torch.empty((m, self.rec_bytes), dtype=torch.uint8, device=dev)
            t[:, 0] = ord("@"); t[:, 1] = ord("r"); t[:, 11] = 10; t[:, 12 + L] = 10; t[:, 13 + L] = ord("+"); t[:, 14 + L] = 10
            t[:, 15 + L:15 + 2 * L] = ord("I"); t[:, 15 + 2 * L] = 10
            self._tmpl = t
            self._pow = (10 ** torch.arange(8, -1, -1, device=dev, dtype=torch.int64))[None, :]
        ids = torch.arange(self.n, self.n + m, device=dev, dtype=torch.int64)[:, None]
        digits = ((ids // self._pow) % 10 + 48).to(torch.uint8)
        for i, r in enumerate(mates):
            t = self._tmpl[:m]
            t[:, 2:11] = digits
            t[:, 12:12 + L] = r
            host = t.cpu()              # (a copy: the template is reused for the next mate / chunk ...
            if host.data_ptr() == t.data_ptr():
                host = host.clone()     #  ... also when the reads were generated on the CPU, where .cpu() is the tensor itself)
            self._q[i].put(host)
        self.n += m

    def close(self):
        for q in self._q:
            q.put(None)
        for t in self._th:
            t.join()
        self._tmpl = None
        self._ft = None
        if self.error:
            raise RuntimeError("writing the full-size FASTQ files failed: " + self.error)
        return self.files

    def remove(self):
        shutil.rmtree(self.dir, ignore_errors=True)


class FullSizeParity:
    """BASELINE config #3 at its FULL size against the unmodified reference: every pair of the run goes through
    `oracle/_ref/dump_ec quant -t <cores>` (the reference's ProcessReads on all cores, then its single-threaded EMAlgorithm::run), and
      * the EC multiset of the WHOLE run must equal the GPU's (EC counts do not depend on the number of threads: MinCollector::increaseCount
        under the mutex of MasterProcessor::update, src/ProcessReads.cpp:424-481, src/MinCollector.cpp:251-269);
      * the reference's EM on ITS ECs (src/EMAlgorithm.h:112-223; effective lengths from the fragment-length sample of the run in input
        order = the -t 1 sample, which the prefix parity check pins, handed over with --flens because a multi-threaded run's own sample
        depends on the schedule) must stop in the same round as the GPU's and agree on est_counts / TPM within BASELINE.json's 1e-4.
    start() launches the reference in the background (about 0.5 min of all cores + 1.5 min of one), finish() waits and compares."""

    def __init__(self, idx_path, files, res, threads, extra=()):
        from oracle import oracle as O
        self.res, self.threads, self.t0 = res, threads, time.time()
        self.tmp = os.path.join(CACHE, f"full_parity_{os.getpid()}")
        os.makedirs(self.tmp, exist_ok=True)
        fl = os.path.join(self.tmp, "gpu.flens")
        with open(fl, "w") as f:
            for i, c in enumerate(np.asarray(res.flens).tolist()):
                if c:
                    f.write(f"{i} {c}\n")
        self.out, self.err = open(os.path.join(self.tmp, "ref.out"), "wb"), open(os.path.join(self.tmp, "ref.err"), "wb")
        cmd = [os.path.join(O.REF_DIR, "dump_ec"), "quant", idx_path, str(threads), *extra, "--flens", fl, *files]
        self.proc = subprocess.Popen(cmd, stdout=self.out, stderr=subprocess.PIPE)
        # the reference's own stage markers on stderr, time-stamped as they arrive: index loaded = ProcessReads announces itself
        # ("[quant] finding pseudoalignments ..."), pseudoalignment finished = that line's " done", EM finished = "... ran for N rounds"
        self.clock = {"start": self.t0}
        import threading

        def _pump():
            seen = b""
            while True:
                chunk = self.proc.stderr.read1(65536) if hasattr(self.proc.stderr, "read1") else self.proc.stderr.read(4096)
                if not chunk:
                    break
                now = time.time()
                self.err.write(chunk)
                seen = (seen + chunk)[-1 << 16:]
                if "loaded" not in self.clock and b"[quant] finding pseudoalignments" in seen:
                    self.clock["loaded"] = now
                if "loaded" in self.clock and "aligned" not in self.clock and b" done" in seen[seen.find(b"[quant] finding pseudoalignments"):]:
                    self.clock["aligned"] = now
                if "em" not in self.clock and b"Expectation-Maximization algorithm ran for" in seen:
                    self.clock["em"] = now
            self.clock["end"] = time.time()
        self._pump = threading.Thread(target=_pump, daemon=True)
        self._pump.start()
        # beside it, on one more core: the oracle's restatement of EMAlgorithm::run (oracle/kallisto_oracle.c ko_em_run) on the GPU's OWN
        # equivalence classes and effective lengths -- same matrix, so the tolerance is the association of FP64 sums: 1e-9
        import threading
        self.oracle_em = {}

        def _oracle_em():
            try:
                t0 = time.time()
                alpha, _, rounds = O.em_run(res.ecs.ec_off, res.ecs.ec_ids, res.ecs.counts, res.eff_lens, len(res.eff_lens))
                big = alpha > 1e-6
                rel = float(np.max(np.abs(res.est_counts[big] - alpha[big]) / alpha[big])) if big.any() else 0.0
                small = float(np.max(np.abs(res.est_counts[~big] - alpha[~big]))) if (~big).any() else 0.0
                self.oracle_em = {"rounds": [int(res.em_rounds), int(rounds)], "est_counts_max_rel_err": rel, "max_abs_err_below_1e-6": small,
                                  "seconds": round(time.time() - t0, 1), "ok": bool(int(rounds) == int(res.em_rounds) and rel <= 1e-9 and small <= 1e-12)}
            except Exception as e:   # noqa: BLE001
                self.oracle_em = {"ok": False, "error": str(e)[:200]}
        self._th = threading.Thread(target=_oracle_em, daemon=True)
        self._th.start()

    def finish(self, timeout_s=1200.0):
        from oracle import oracle as O
        try:
            try:
                rc = self.proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                self.proc.kill()
                self.proc.wait()
                return {"ok": False, "error": f"the reference did not finish within {timeout_s:.0f} s"}
            ref_s = time.time() - self.t0
            self._pump.join(timeout=30.0)
            self.out.close(); self.err.close()
            if rc != 0:
                return {"ok": False, "error": f"dump_ec exit code {rc}: " + open(self.err.name, errors="replace").read()[-300:]}
            ref = O.parse_ref_dump(open(self.out.name).read(), open(self.err.name, errors="replace").read())
            res = self.res
            gm = res.ecs.multiset()
            rep = O.ref_parity_report(ref, gm, res.flens, res.eff_lens, res.est_counts, res.alpha_before_zeroes)
            # (flens_equal compares with the sample the reference drew itself at -t > 1: schedule-dependent, reported, not gated)
            out = {"pairs_or_reads": int(res.n_processed), "reference": f"oracle/_ref/dump_ec quant -t {self.threads} --flens <the run's sample in input order> "
                   "(unmodified reference: ProcessReads on all cores, EMAlgorithm::run on its own ECs)",
                   "n_processed_ref": int(ref["nproc"]), "n_ecs": [len(gm), len(ref["ecs"])], "ec_multiset_equal": bool(gm == ref["ecs"]),
                   "n_pseudoaligned": [int(res.n_pseudoaligned), int(sum(ref["ecs"].values()))],
                   "eff_length_equal": rep["eff_length_equal"], "em_rounds": [int(res.em_rounds), ref["rounds"]],
                   "est_counts_max_rel_err_tpm_ge_1e-3": rep["est_counts_max_rel_err_tpm_ge_1e-3"],
                   "tpm_max_rel_err_tpm_ge_1e-3": rep["tpm_max_rel_err_tpm_ge_1e-3"], "tpm_max_abs_err_below_floor": rep["tpm_max_abs_err_below_floor"],
                   "zero_pattern_equal": rep["zero_pattern_equal"], "own_sample_of_the_threaded_reference_equal": rep["flens_equal"],
                   "reference_seconds": round(ref_s, 1), "tolerance": "EC multiset and eff_length identical, same EM round count, est_counts / tpm <= 1e-4"}
            ck = self.clock
            if all(k in ck for k in ("loaded", "aligned", "em", "end")):
                out["reference_stage_seconds"] = {"index_load": round(ck["loaded"] - ck["start"], 2), "pseudoalign": round(ck["aligned"] - ck["loaded"], 2),
                                                  "em": round(ck["em"] - ck["aligned"], 2), "output": round(ck["end"] - ck["em"], 2), "threads": self.threads}
            self._th.join(timeout=max(1.0, timeout_s - (time.time() - self.t0)))
            out["oracle_em_on_the_gpus_ecs"] = self.oracle_em or {"ok": False, "error": "did not finish"}
            out["ok"] = bool(out["oracle_em_on_the_gpus_ecs"].get("ok") and
                             out["ec_multiset_equal"] and out["eff_length_equal"] and out["em_rounds"][0] == out["em_rounds"][1] and
                             int(ref["nproc"]) == int(res.n_processed) and rep["est_counts_max_rel_err_tpm_ge_1e-3"] <= 1e-4 and
                             rep["tpm_max_rel_err_tpm_ge_1e-3"] <= 1e-4 and rep["tpm_max_abs_err_below_floor"] <= 1e-7 and rep["zero_pattern_equal"])
            return out
        except Exception as e:   # noqa: BLE001
            return {"ok": False, "error": str(e)[:300]}
        finally:
            shutil.rmtree(self.tmp, ignore_errors=True)


def cpu_reference_baseline(idx_path, r1: np.ndarray, r2, threads: int, extra=()):
    """Time the unmodified reference on the host cores: `kallisto quant -t threads` on the sample.  The clock starts when
    the index has been loaded (the '[quant] running in' line) and stops at process exit; the stage markers the reference
    prints on stderr split it into pseudoalignment (until the 'finding pseudoalignments ... done' line completes) and EM
    (until 'the Expectation-Maximization algorithm ran for')."""
    tmp = os.path.join(CACHE, f"cpu_baseline_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    files = [os.path.join(tmp, "s_1.fq")]
    write_fastq_fast(files[0], r1)
    if r2 is not None:
        files.append(os.path.join(tmp, "s_2.fq"))
        write_fastq_fast(files[1], r2, 1)
    out = os.path.join(tmp, "out")
    cmd = [REF_BIN, "quant", "-i", idx_path, "-o", out, "-t", str(threads), "--plaintext", *extra, *files]
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


def _ref_dump(idx_path, r1, r2, extra=()):
    """oracle/_ref/dump_ec quant -t 1 on these reads (FASTQ written to the cache directory)"""
    from oracle import oracle as O
    tmp = os.path.join(CACHE, f"ref_parity_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    try:
        files = [os.path.join(tmp, "p_1.fq")]
        write_fastq_fast(files[0], r1)
        if r2 is not None:
            files.append(os.path.join(tmp, "p_2.fq"))
            write_fastq_fast(files[1], r2, 1)
        return O.ref_dump_quant(idx_path, files, threads=1, extra=extra)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def reference_parity(idx_path, r1: np.ndarray, r2, res, extra=()):
    """The GPU result `res` (kallisto_amd.quant on exactly these reads, ECs downloaded) against the unmodified reference run
    deterministically (`-t 1`: the fragment-length sample is the first 10 000 qualifying pairs in input order,
    src/ProcessReads.cpp:981-1017,1174-1181) through oracle/_ref/dump_ec, which prints what the CLI never does: the EC
    multiset, flens, eff_lens and alpha of src/main.cpp:2632-2689."""
    from oracle import oracle as O
    t0 = time.time()
    ref = _ref_dump(idx_path, r1, r2, extra)
    rep = O.ref_parity_report(ref, res.ecs.multiset(), res.flens, res.eff_lens, res.est_counts, res.alpha_before_zeroes)
    rep["sample"] = int(r1.shape[0])
    rep["reference_seconds"] = round(time.time() - t0, 1)
    rep["n_pseudoaligned"] = [int(res.n_pseudoaligned), int(sum(ref["ecs"].values()))]
    rep["em_rounds_gpu"] = int(res.em_rounds)
    return rep


def tail_parity(ctx, opts, idx_path, words, lens, n, per, rec, L, t1, t2, extra=()):
    """The LAST pairs of the input against the reference, in the state a full-size run leaves the context in: the whole input is
    pseudoaligned with those pairs as the final batch (record stream grown and reallocated, de-duplication tables regrown, overflow
    items redirected), then the input without them; the difference of the two EC multisets must be the reference's EC multiset of
    those pairs alone (EC counts are additive over reads: MinCollector::increaseCount, src/MinCollector.cpp:251-269)."""
    k = int(t1.shape[0])
    t0 = time.time()

    def ec_multiset(parts):
        ctx.reset()
        for a, b in parts:
            if b > a:
                ctx.pseudoalign(opts, words[a * per * rec:b * per * rec], lens[per * a:per * b], b - a, L)
        return ctx.finalize(download=True).multiset()
    whole = ec_multiset([(0, n - k), (n - k, n)])
    head = ec_multiset([(0, n - k)])
    diff = {}
    bad = 0
    for key, c in whole.items():
        d = c - head.get(key, 0)
        if d < 0:
            bad += 1
        elif d > 0:
            diff[key] = d
    bad += sum(1 for key in head if key not in whole)
    ref = _ref_dump(idx_path, t1, t2, extra)
    return {"pairs_or_reads": k, "position": f"items [{n - k}, {n}) of {n}, pseudoaligned as the final batch of a run over all of them",
            "n_ecs_ref": len(ref["ecs"]), "n_ecs": len(diff), "counts_never_decrease": bad == 0,
            "ec_multiset_equal": bool(diff == ref["ecs"]), "n_pseudoaligned": [int(sum(diff.values())), int(sum(ref["ecs"].values()))],
            "seconds": round(time.time() - t0, 1), "ok": bool(bad == 0 and diff == ref["ecs"])}


def pinned_pipeline(ka, ctx, opts, words, lens, n, per, rec, L, chunk_items=4_000_000):
    """Packed reads in pinned host memory -> ECs: H2D copies on a side stream, double buffered, kamd_pseudoalign per chunk on the
    context stream, then finalize ('pipeline reads/s from pinned host memory').  Never `value`."""
    import torch
    dev = words.device
    hw = torch.empty(words.numel(), dtype=words.dtype, pin_memory=True)
    hl = torch.empty(lens.numel(), dtype=lens.dtype, pin_memory=True)
    hw.copy_(words); hl.copy_(lens)
    torch.cuda.synchronize()
    copy = torch.cuda.Stream(device=dev)
    bufs = [(torch.empty(chunk_items * per * rec, dtype=words.dtype, device=dev), torch.empty(chunk_items * per, dtype=lens.dtype, device=dev),
             torch.cuda.Event()) for _ in range(2)]
    chunks = [(a, min(a + chunk_items, n)) for a in range(0, n, chunk_items)]

    def issue(i):
        a, b = chunks[i]
        w, l, ev = bufs[i % 2]
        with torch.cuda.stream(copy):
            w[:(b - a) * per * rec].copy_(hw[a * per * rec:b * per * rec], non_blocking=True)
            l[:(b - a) * per].copy_(hl[a * per:b * per], non_blocking=True)
            ev.record(copy)
    best = None
    for _ in range(2):   # one warm-up pass
        ctx.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        issue(0)
        for i, (a, b) in enumerate(chunks):
            bufs[i % 2][2].synchronize()
            if i + 1 < len(chunks):
                issue(i + 1)            # (the other buffer: its chunk was consumed by the previous, synchronous kamd_pseudoalign)
            w, l, _ = bufs[i % 2]
            ctx.pseudoalign(opts, w[:(b - a) * per * rec], l[:(b - a) * per], b - a, L)
        ctx.finalize(download=False)
        torch.cuda.synchronize()
        best = time.perf_counter() - t0
    gb = (hw.numel() * 4 + hl.numel() * 2) / 1e9
    return {"items": n, "seconds": round(best, 4), "M_per_s": round(n / best / 1e6, 2), "host_GB": round(gb, 3), "h2d_GB_per_s": round(gb / best, 2),
            "chunk_items": chunk_items, "note": "2-bit packed reads in pinned host memory, copies of chunk i+1 under the pseudoalignment of chunk i, "
            "EC finalize included, no EM; bound by the host-to-device link (packed PE-100: 104 B per pair)"}


def _bgzf_block(chunk: bytes) -> bytes:
    import struct
    import zlib
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    comp = co.compress(chunk) + co.flush()
    return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1)
            + comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def _bgzf_part(args):
    src, a, b = args
    with open(src, "rb") as f:
        f.seek(a)
        data = f.read(b - a)
    return b"".join(_bgzf_block(data[i:i + 65280]) for i in range(0, len(data), 65280))


def write_bgzf(src, dst, procs):
    """what `bgzip` writes: independent gzip members of <= 64 KiB (written by a pool of processes: zlib is slow)"""
    import multiprocessing as mp
    size = os.path.getsize(src)
    step = 65280 * 256
    parts = [(src, a, min(a + step, size)) for a in range(0, size, step)]
    with mp.get_context("fork").Pool(procs) as pool, open(dst, "wb") as fo:
        for blob in pool.imap(_bgzf_part, parts):
            fo.write(blob)
        fo.write(_bgzf_block(b""))


def end_to_end(idx_path, r1: np.ndarray, r2, paired: bool, threads: int, extra, gz_items=4_000_000, full_files=None, full_items=0):
    """The C++ front-end (kallisto_amd/kallisto_amd_quant) from FASTQ files on disk: plain text, BGZF and gzip.  Wall-clock per stage
    from its --verbose timing lines; never part of `value`."""
    exe = os.path.join(ROOT, "kallisto_amd", "kallisto_amd_quant")
    if not os.path.exists(exe):
        return {"error": "kallisto_amd_quant not built"}
    tmp = os.path.join(CACHE, f"e2e_{os.getpid()}")
    os.makedirs(tmp, exist_ok=True)
    out = {}
    unit = "pairs" if paired else "reads"
    try:
        n = int(r1.shape[0])
        f1, f2 = os.path.join(tmp, "e_1.fq"), os.path.join(tmp, "e_2.fq")
        t0 = time.time()
        write_fastq_fast(f1, r1)
        if paired:
            write_fastq_fast(f2, r2, 1)
        plain = [f1, f2] if paired else [f1]
        # the same reads as BGZF (block-parallel inflate) and, a subset, as one ordinary gzip member per file (`gzip -1`)
        bg = [os.path.join(tmp, f"b_{i + 1}.fq.gz") for i in range(len(plain))]
        for src, dst in zip(plain, bg):
            write_bgzf(src, dst, max(2, min(threads, 48)))
        ngz = min(n, gz_items)
        gz = []
        procs = []
        for i, src in enumerate(plain):
            g = os.path.join(tmp, f"g_{i + 1}.fq")
            if ngz == n:
                shutil.copyfile(src, g)
            else:   # (records have variable lengths: the first ngz of them written again)
                write_fastq_fast(g, (r1, r2)[i][:ngz], i)
            procs.append(subprocess.Popen(["gzip", "-1", "-f", g]))
            gz.append(g + ".gz")
        for p in procs:
            p.wait()
        prep_s = time.time() - t0
        # the device tables written once as a file (`kallisto_amd_quant flatten`): what a multi-sample front-end would load
        flat = os.path.join(tmp, "index.kamd")
        t0 = time.time()
        have_flat = subprocess.run([exe, "flatten", "-i", idx_path, "-o", flat, "-t", str(threads)], stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL).returncode == 0
        flatten_s = time.time() - t0
        # the rate legs measure the input path alone (KAMD_FQ_NO_OVERLAP: the input waits for the index, as the bulk of a large input does);
        # the *_overlapped legs are the front-end as it runs by default: input read, copied and parsed while the index loads
        seq = {"KAMD_FQ_NO_OVERLAP": "1"}
        runs = [("plain", idx_path, plain, n, seq), ("bgzf", idx_path, bg, n, seq), ("gzip", idx_path, gz, ngz, seq),
                ("plain_host_parsed", idx_path, plain, n, {"KAMD_HOST_PARSE": "1", **seq}), ("plain_overlapped", idx_path, plain, n, {})]
        if have_flat:
            runs.append(("plain_flattened_index", flat, plain, n, seq))
            runs.append(("plain_flattened_index_overlapped", flat, plain, n, {}))
        if full_files:   # the configuration the metric names, all of it, from files (the bench's FastqSpool wrote them)
            runs.append(("plain_full_size", idx_path, full_files, full_items, seq))
            runs.append(("plain_full_size_overlapped", idx_path, full_files, full_items, {}))
            if have_flat:
                runs.append(("plain_full_size_flattened_index_overlapped", flat, full_files, full_items, {}))
        for kind, ipath, files, cnt, env in runs:
            cmd = [exe, "quant", "-i", ipath, "-o", os.path.join(tmp, "out_" + kind), "-t", str(threads), "--plaintext", "--verbose", *extra, *files]
            t0 = time.time()
            p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            wall = time.time() - t0
            if p.returncode != 0:
                out[kind] = {"error": p.stderr.decode(errors="replace")[-300:]}
                continue
            tm = {}
            text = p.stderr.decode(errors="replace")
            for line in text.splitlines():
                if line.startswith("[timing] index file read"):
                    w = line.split()
                    tm["index_load_s"] = float(w[7]); tm["index_on_device_s"] = float(w[-2])
                elif line.startswith("[timing] reads parsed"):
                    tm["reads_done_s"] = float(line.split()[-2])
                elif line.startswith("[timing] total"):
                    tm["total_s"] = float(line.split()[-2])
            reads_s = tm.get("reads_done_s", wall) - tm.get("index_on_device_s", 0.0)
            if kind.endswith("_overlapped"):   # (the window behind the index says nothing here: most of the input was handled under the index load)
                out[kind] = {unit: cnt, "wall_s": round(wall, 2), **{k: round(v, 3) for k, v in tm.items()},
                             "after_index_s": round(reads_s, 3), "whole_run_M_per_s": round(cnt / wall / 1e6, 3)}
                continue
            out[kind] = {unit: cnt, "file_GB": round(sum(os.path.getsize(f) for f in files) / 1e9, 3), "wall_s": round(wall, 2),
                         **{k: round(v, 3) for k, v in tm.items()},
                         "input_to_ecs_M_per_s": round(cnt / max(reads_s, 1e-9) / 1e6, 3),
                         "whole_run_M_per_s": round(cnt / wall / 1e6, 3), "host_threads": threads,
                         "parser": "host (general reader)" if "device parser: 0 units" in text else "device (kamd_fastq_unit_parse)"}
        if have_flat:
            out["flatten_s"] = round(flatten_s, 2)
        out["file_preparation_s"] = round(prep_s, 1)
        # the outputs of the device-parsed and the host-parsed run must be the same files
        try:
            a = open(os.path.join(tmp, "out_plain", "abundance.tsv"), "rb").read()
            out["device_parser_equals_host_parser"] = all(a == open(os.path.join(tmp, "out_" + k, "abundance.tsv"), "rb").read()
                                                          for k in ("plain_host_parsed", "bgzf"))
        except OSError:
            out["device_parser_equals_host_parser"] = None
        out["note"] = ("kallisto_amd_quant from FASTQ on local disk (page cache warm).  plain / bgzf / gzip: host threads move bytes into pinned rings "
                       "(copies out of a mapping of the file; block-parallel inflate of BGZF and of ordinary gzip) and count newlines, lines / record check / 2-bit packing on the GPU; "
                       "plain_host_parsed: the general reader (KAMD_HOST_PARSE=1).  input_to_ecs = from index-on-device to the last pseudoalignment "
                       "(reading, H2D, parsing, packing, pseudoalignment) with the input held back until the index is on the device (KAMD_FQ_NO_OVERLAP: "
                       "the input path alone); whole_run = process start to exit (index load, EM, output files).  *_overlapped: the default front-end, "
                       "which reads, copies and parses the input while the index loads (after_index_s = what is left to do once the index is there)")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def compact_table_leg(idx_path, compact_dir, n, L, paired, steps, warmup, device, res, script=None, loads="0.6,0.5,0.7", timeout_s=600):
    """The steps of the headline on the compact k-mer table, in a child process (tools/compact_table_leg.py) that finds the packed reads in
    compact_dir; `res` = the headline's result, which every leg must reproduce (counts, fragment lengths, est_counts to the bit, EM rounds).
    Returns the list of legs; never raises; removes compact_dir."""
    try:
        cmd = [sys.executable, script or os.path.join(ROOT, "tools", "compact_table_leg.py"), "--index", idx_path, "--dir", compact_dir, "--items", str(n),
               "--read-len", str(L), "--paired", "1" if paired else "0", "--steps", str(steps), "--warmup", str(warmup), "--device", str(device), "--loads", loads]
        pc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        if pc.returncode != 0:
            return [{"error": f"rc {pc.returncode}: " + pc.stderr.decode(errors="replace")[-400:]}]
        legs = json.loads(pc.stdout.decode().strip().splitlines()[-1])
        for i, entry in enumerate(legs):   # the layout must change nothing
            f = os.path.join(compact_dir, f"result_{i}.npz")
            if "error" in entry or not os.path.exists(f):
                continue
            z = np.load(f)
            entry["identical_to_headline"] = bool(int(z["n_pseudoaligned"]) == res.n_pseudoaligned and int(z["n_unique"]) == res.n_unique and
                                              np.array_equal(z["flens"], res.flens) and np.array_equal(z["est_counts"], res.est_counts) and
                                              int(z["em_rounds"]) == res.em_rounds)
        return legs
    except Exception as e:   # noqa: BLE001  (a side leg must not take the line down)
        return [{"error": str(e)[:300]}]
    finally:
        shutil.rmtree(compact_dir, ignore_errors=True)


def bench_line_digest(d):
    """the figures of a bench line (value, step time, break-down, roofline fraction, CPU baseline, parity verdicts) without its prose"""
    keep = {k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "breakdown_ms", "counters")}
    keep["workload"] = (d.get("config") or {}).get("workload")
    keep["roofline"] = {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "peak", "frac", "launch_ms")}
    cb = d.get("cpu_baseline") or {}
    keep["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
    for k in ("parity_check", "parity_check_tail", "parity_check_full_size"):
        pcheck = d.get(k)
        if isinstance(pcheck, dict):
            keep[k] = {kk: vv for kk, vv in pcheck.items() if not isinstance(vv, (list, dict)) or kk == "n_pseudoaligned"}
    return keep


CONTRACT_MAX_BYTES = 4096   # the driver keeps a bounded tail of stdout: the round-5 line (24.9 KB) did not survive it


def _num(x, nd=5):
    """a figure for the contract line: ints and bools as they are, floats rounded, everything else dropped"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        return float(f"{float(x):.{nd + 2}g}") if abs(float(x)) < 1 else round(float(x), nd)
    return None


def _figures(d, keys):
    """the named keys of a block, numbers and short strings only (no prose: `note`, `launch`, nested tables stay in the detail file)"""
    o = {}
    for k in keys:
        v = (d or {}).get(k)
        if isinstance(v, str):
            o[k] = v[:96]
        elif v is not None or k in ("traffic", "value"):
            o[k] = _num(v)
    return o


def _ok(block):
    """verdict of a parity block: True / False, None when the leg did not run"""
    if not isinstance(block, dict) or "ok" not in block:
        return None
    return bool(block["ok"])


ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "launch_ms")


def contract_line(out: dict, detail_path: str | None = None) -> str:
    """THE line rank 0 prints: the contract's fields + `roofline` + `cpu_baseline` + verdicts, figures only, at most CONTRACT_MAX_BYTES bytes.
    Everything else bench.py measures (notes, child legs, end-to-end table, layouts, counters) is the *detail* -- bench_detail.json beside this
    file and stderr --, never this line (VERDICT r5 "next round" #1).  `out` is the detail dictionary."""
    cfg = out.get("config") or {}
    unit_key = next((k for k in cfg if k.endswith("_per_gpu")), "pairs_per_gpu")
    line = {k: (out.get(k)[:120] if isinstance(out.get(k), str) else out.get(k))
            for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": str(cfg.get("workload_short") or cfg.get("workload") or "")[:200], unit_key: cfg.get(unit_key), "read_len": cfg.get("read_len"),
                      "targets": cfg.get("targets"), "kmers": cfg.get("kmers"), "parallelism": str(cfg.get("parallelism_short") or cfg.get("parallelism") or "")[:80],
                      "n_ranks_seen": cfg.get("n_ranks_seen"), "collective_backend": cfg.get("collective_backend")}
    rf = out.get("roofline") or {}
    line["roofline"] = _figures(rf, ROOF_KEYS)
    if isinstance(rf.get("random_line_ceiling"), dict) and "frac" in rf["random_line_ceiling"]:
        line["roofline"]["frac_of_random_line_rate"] = _num(rf["random_line_ceiling"]["frac"])
    for k in ("roofline_em", "roofline_finalize"):
        if isinstance(out.get(k), dict):
            line[k] = _figures(out[k], ROOF_KEYS + (("rounds", "nnz") if k == "roofline_em" else ()))
            line[k]["kernel"] = str(out[k].get("kernel_short") or out[k].get("kernel") or "")[:48]
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _figures(cb, ("value", "unit", "cores", "kind", "pseudoalign_seconds", "em_seconds", "index_load_seconds"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb.get("sample") or "")[:160]
    bd = out.get("breakdown_ms")
    if isinstance(bd, dict):
        line["breakdown_ms"] = {k: _num(v, 3) for k, v in list(bd.items())[:16] if isinstance(v, (int, float)) and not isinstance(v, bool)}
    ct = out.get("counters") or {}
    line["counters"] = {k: _num(ct.get(k), 4) for k in ("probes_per_pair", "bucket_reads_per_pair", "text_answers_per_pair", "lane_utilisation", "final_ecs", "em_rounds",
                                                        "overflow_share") if k in ct}
    st, gc, mp = out.get("stress"), out.get("gencode_size"), out.get("multi_rank_parity")
    line["parity"] = {"prefix_ok": _ok(out.get("parity_check")), "tail_ok": _ok(out.get("parity_check_tail")), "full_size_ok": _ok(out.get("parity_check_full_size")),
                      "stress_ok": (None if not isinstance(st, dict) or "value" not in st else
                                    bool(all(_ok(st.get(k)) is not False for k in ("parity_check", "parity_check_tail", "parity_check_full_size")) and
                                         any(_ok(st.get(k)) for k in ("parity_check", "parity_check_full_size")))),
                      "gencode_ok": (None if not isinstance(gc, dict) or "value" not in gc else
                                     bool(_ok(gc.get("parity_check")) and _ok(gc.get("parity_check_tail")) is not False)),
                      **({"multi_rank_ok": _ok(mp)} if isinstance(mp, dict) else {})}
    fp = out.get("parity_check_full_size")
    if isinstance(fp, dict) and fp.get("ok") is not None:
        line["parity"]["full_size"] = {k: fp.get(k) for k in ("pairs_or_reads", "n_ecs", "em_rounds") if k in fp}
        line["parity"]["full_size"]["est_counts_max_rel_err"] = _num(fp.get("est_counts_max_rel_err_tpm_ge_1e-3"))
    for k, leg in (("stress", st), ("gencode_size", gc), ("config2", out.get("config2"))):
        if isinstance(leg, dict):
            line[k] = ({"value": _num(leg.get("value")), "ms_per_step": _num(leg.get("ms_per_step"), 3), "unit": str(leg.get("unit") or "")[:24],
                        "roofline_frac": _num((leg.get("roofline") or {}).get("frac")), "cpu_baseline": _num((leg.get("cpu_baseline") or {}).get("value"))}
                       if "value" in leg else {"skipped": str(leg.get("skipped") or leg.get("error") or "")[:100]})
    bs = out.get("bootstrap")
    if isinstance(bs, dict):
        line["bootstrap"] = _figures(bs, ("replicates", "seconds", "replicates_per_s")) if "seconds" in bs else {"error": str(bs.get("error"))[:100]}
    if isinstance(out.get("other_scaling"), dict):
        line["other_scaling"] = _figures(out["other_scaling"], ("scaling", "pairs_per_gpu", "value", "unit", "ms_per_step"))
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("counters", "config2", "bootstrap", "other_scaling", "roofline_finalize", "breakdown_ms", "roofline_em"):   # never reached with today's fields; a bound, not a plan
        if len(text) <= CONTRACT_MAX_BYTES:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > CONTRACT_MAX_BYTES:
        raise RuntimeError(f"contract line of {len(text)} bytes")
    return text


def write_detail(out: dict, path: str | None = None):
    """the lab notebook: everything the run measured, as one JSON document in a file (and nothing of it on stdout).  Returns the path, relative to ROOT."""
    path = path or os.environ.get("KAMD_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path + ".tmp", "w") as f:
            json.dump(out, f, indent=1)
        os.replace(path + ".tmp", path)
        side = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(side) and os.path.dirname(os.path.abspath(path)) == os.path.abspath(ROOT):   # (the default place only: a child leg's temporary file is not evidence)
            shutil.copyfile(path, os.path.join(side, os.path.basename(path)))
        return os.path.relpath(path, ROOT)
    except OSError as e:
        log(f"detail file not written: {e}")
        return None


def run_child(cmd, timeout_s):
    """a child run of this script: its DETAIL document (the child writes it to a file of its own; its stdout carries only the short line)"""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="kamd_bench_child_", suffix=".json")
    os.close(fd)
    try:
        pc = subprocess.run(cmd + ["--detail-file", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        if pc.returncode != 0:
            raise RuntimeError(f"rc {pc.returncode}: " + pc.stderr.decode(errors="replace")[-400:])
        return json.load(open(path))
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


def config2_leg(timeout_s=900):
    """BASELINE config #2 (yeast-like transcriptome, 10 M single-end reads, --single -l 200 -s 20) as a child run of this script: its line, cut
    down to the figures, for the line of config #3.  Never raises."""
    try:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "yeast", "--steps", "5", "--warmup", "2", "--end-to-end", "0", "--no-pinned-pipeline",
               "--no-compact-leg", "--no-config2", "--bootstraps", "0"]
        keep = bench_line_digest(run_child(cmd, timeout_s))
        keep["command"] = "python bench.py " + " ".join(cmd[2:])
        return keep
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:300]}


def stress_leg(timeout_s=900, pairs=8_000_000):
    """The stress workload (synth.human_stress + off-transcriptome reads + quality tails) as a child run of this script with ALL its pairs through
    the unmodified reference (FullSizeParity): its line, cut down to the figures, for the line of config #3.  Never raises."""
    try:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "stress", "--pairs", str(pairs), "--steps", "6", "--warmup", "2", "--end-to-end", "0",
               "--no-pinned-pipeline", "--no-compact-leg", "--no-config2", "--no-stress-leg", "--bootstraps", "0", "--full-parity", "on"]
        d = run_child(cmd, timeout_s)
        keep = bench_line_digest(d)
        keep["roofline_em"] = {k: (d.get("roofline_em") or {}).get(k) for k in ("kernel", "bound", "achieved", "peak", "frac", "launch_ms", "rounds", "nnz", "rows")}
        keep["command"] = "python bench.py " + " ".join(cmd[2:])
        return keep
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:300]}


GENCODE_GENES = 46000   # synth.human_like(46 000 genes): ~444 k transcripts, ~131 M k-mers -- the size of a GENCODE transcriptome index


def gencode_leg(prep, timeout_s=600):
    """Config #3's reads on a GENCODE-SIZED index (VERDICT r4 "missing" #4: the figure existed builder-side only): a child run of this script, its line cut
    down to the figures.  prep: the background process that built and cached the index (or None).  Never raises."""
    try:
        if prep is not None:
            prep.wait(timeout=timeout_s)
        cmd = [sys.executable, os.path.abspath(__file__), "--genes", str(GENCODE_GENES), "--steps", "6", "--warmup", "2", "--end-to-end", "0", "--no-pinned-pipeline",
               "--no-compact-leg", "--no-config2", "--no-stress-leg", "--no-gencode-leg", "--bootstraps", "0", "--full-parity", "off", "--no-cpu-baseline"]
        d = run_child(cmd, timeout_s)
        keep = bench_line_digest(d)
        keep["config"] = {k: (d.get("config") or {}).get(k) for k in ("targets", "kmers", "kmer_table")}
        keep["roofline_em"] = {k: (d.get("roofline_em") or {}).get(k) for k in ("kernel", "bound", "achieved", "peak", "frac", "launch_ms", "rounds", "nnz", "rows", "groups")}
        keep["command"] = "python bench.py " + " ".join(cmd[2:])
        keep["note"] = ("the human-like generator at 46 000 genes (2.3 x config #3's index: the size of a real GENCODE transcriptome index); same 30 M PE-100 pairs per step; "
                        "parity on the first and the last 200 k pairs against the reference at -t 1 (the full-size leg on this index ran builder-side: profiles/)")
        return keep
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:300]}


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N ...` without a launcher around it: run the same command line as N ranks of one node under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <a free one>` (one process per
    GPU; the ranks find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), pass rank 0's ONE JSON line through and return the
    launcher's exit code.  If the attempt with the library's own RCCL communicator dies or exceeds KAMD_BENCH_LAUNCH_TIMEOUT_S, the ranks are
    started once more with the same collectives carried by torch.distributed's process group (KAMD_COMM=callbacks) -- the line names the
    transport that was actually used (`config.collective_backend`).  KAMD_BENCH_SHARE_GPU=1: all ranks on GPU 0 over gloo (1-GPU boxes:
    a smoke test of the flow, never a reported number)."""
    import socket
    share = os.environ.get("KAMD_BENCH_SHARE_GPU") == "1"
    script = os.environ.get("KAMD_BENCH_LAUNCH_SCRIPT")   # (tests: a stand-in for this file, started the same way; no GPUs asked for)
    if not share and not script:
        import torch
        have = torch.cuda.device_count()
        if have < n_gpus:
            print(f"bench.py: --gpus {n_gpus} but this node shows {have} GPU(s) (KAMD_BENCH_SHARE_GPU=1 puts all ranks on GPU 0 over gloo: a "
                  f"smoke test only)", file=sys.stderr)
            return 2
    # two attempts must fit the driver's 1 800 s: 850 s each by default (an N-GPU line is about a minute of index building, a minute of work per
    # rank and, at N = 8, two to three minutes of the reference at -t 1 on the 1.6 M pairs of the default parity leg)
    limit = float(os.environ.get("KAMD_BENCH_LAUNCH_TIMEOUT_S", "850"))
    attempts = [{}] if (share or os.environ.get("KAMD_COMM") == "callbacks") else [{}, {"KAMD_COMM": "callbacks"}]
    rc = 1
    for extra in attempts:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ)
        env.update(extra)
        env["KAMD_BENCH_LAUNCHER"] = "self"
        env.setdefault("KAMD_COMM_INIT_TIMEOUT_S", "180")   # (the library's watchdog over ncclCommInitRank is opt-in: only ranks started here may end their process)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            env.setdefault("KAMD_BENCH_BACKEND", "gloo")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), script or os.path.abspath(__file__)] + sys.argv[1:]
        log(f"launching {n_gpus} ranks: {' '.join(cmd[1:8])} bench.py {' '.join(sys.argv[1:])}" + (f"  [{extra}]" if extra else ""))
        pc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
        try:
            so_, _ = pc.communicate(timeout=limit)
            rc = pc.returncode
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(pc.pid, signal.SIGKILL)   # the process group this call started, nothing else
            except ProcessLookupError:
                pass
            so_, _ = pc.communicate()
            rc = 124
            log(f"the ranks did not finish within {limit:.0f} s (KAMD_BENCH_LAUNCH_TIMEOUT_S)")
        line = None
        for ln in so_.decode(errors="replace").splitlines():
            if ln.startswith("{") and ln.rstrip().endswith("}"):
                try:
                    json.loads(ln)
                    line = ln
                except ValueError:
                    pass
        if line is not None:
            print(line, flush=True)
            return 0 if rc in (0, None) else rc
        log(f"no result line from the ranks (exit code {rc})")
    return rc or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed steps (a step of config #3 is 27 ms: ten of them are a quarter of a second of a run that takes minutes)")
    ap.add_argument("--warmup", type=int, default=3, help="untimed steps in front (the first step of a context allocates; the next two still settle by 1-2 %%)")
    ap.add_argument("--pairs", type=int, default=None, help="read pairs (reads) per GPU per step; default: BASELINE config #3: 30 M "
                    "(human), config #2: 10 M single-end reads (yeast)")
    ap.add_argument("--workload", default="human", choices=["human", "yeast", "stress"],
                    help="human = BASELINE configs #3/#4/#5 (paired-end); yeast = config #2 (single-end, -l 200 -s 20); stress = the human-sized "
                         "workload with the structure of a real transcriptome (synth.human_stress: repeat families in the UTRs, paralog families, poly-A "
                         "tails -> one connected component with half of the EC matrix) and the reads the other workloads lack: 12 %% off-transcriptome "
                         "pairs, a 3' quality tail of errors")
    ap.add_argument("--prepare-only", action="store_true", help="build and cache the workload's transcriptome and index, then exit (the default run prepares the "
                    "GENCODE-sized index in the background this way)")
    ap.add_argument("--no-gencode-leg", action="store_true", help="skip the child run on a GENCODE-sized index (46 000 genes, ~444 k transcripts) that the default one-GPU run of config #3 appends")
    ap.add_argument("--no-stress-leg", action="store_true", help="skip the child run of the stress workload that the default one-GPU run of config #3 appends")
    ap.add_argument("--genes", type=int, default=None, help="scale of the synthetic transcriptome (default: full config)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --pairs per GPU (default); strong = --pairs in total, sharded over the GPUs (BASELINE config #4). "
                         "The other mode is measured too and reported in the same line.")
    ap.add_argument("--cpu-sample", type=int, default=8_000_000, help="pairs (reads) given to the CPU reference (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=200_000,
                    help="pairs of the CPU sample that also go through the reference at -t 1 for the parity gate")
    ap.add_argument("--bootstraps", type=int, default=None,
                    help="BASELINE config #5: also time B bootstrap replicates (multinomial resample + EM), split over the ranks "
                         "(default: 100 on one GPU -- about a second --, 0 on several)")
    ap.add_argument("--in-flight", type=int, default=1, help="2: also measure two samples in flight on one GPU (two contexts / streams); reported "
                    "beside the headline value, never as it")
    ap.add_argument("--end-to-end", type=int, default=None, help="run the C++ front-end from FASTQ files with this many pairs / reads (N = 1 only; 0 = skip; default: 0, "
                    "8 M with --extras)")
    ap.add_argument("--no-pinned-pipeline", action="store_true")
    ap.add_argument("--no-multi-sample-parity", action="store_true", help="N > 1: skip the default parity leg (the merged result of the ranks on the first --parity-sample "
                    "pairs of every rank against the reference at -t 1) and the cpu_baseline it yields")
    ap.add_argument("--multi-parity", action="store_true", help="N > 1: rank 0 also runs the reads of ALL ranks on its own (one context, no communicator) "
                    "and the merged result of the ranks must equal it (EC multiset, flens identical; est_counts 1e-9; same EM rounds)")
    ap.add_argument("--table-layout", default=None, choices=["wide", "compact", "auto"],
                    help="layout of the k-mer table (KAMD_TABLE_LAYOUT; default: the library's, auto = compact when it fits): compact = four quotiented 16-byte slots per "
                         "line instead of three 20-byte ones (kamd_core.h)")
    ap.add_argument("--no-config2", action="store_true", help="skip the child run of BASELINE config #2 (yeast, single-end) that the default one-GPU run of config #3 appends")
    ap.add_argument("--no-compact-leg", action="store_true", help="skip the side leg that repeats the steps on the compact k-mer table (N = 1 only)")
    ap.add_argument("--table-load", type=float, default=None, help="load factor of the compact table (KAMD_TABLE_LOAD; default: chosen from the table's size, config.kmer_table.load reports it)")
    ap.add_argument("--full-parity", default="auto", choices=["auto", "on", "off"],
                    help="N = 1: the WHOLE input also goes through the unmodified reference (written as FASTQ while it is generated; oracle/_ref/dump_ec on all "
                         "cores in the background): EC multiset of the whole run, EM round count and abundances at full size; the same files feed the full-size "
                         "end-to-end legs.  auto = on for the full BASELINE configuration, off for reduced ones")
    ap.add_argument("--extras", action="store_true", help="N = 1: also run the side measurements that are not part of the line (they go to the detail file): the pinned pipeline, "
                    "the other k-mer table layouts (tools/compact_table_leg.py), BASELINE config #2 as a child run, the C++ front-end from FASTQ files (--end-to-end, 8 M pairs)")
    ap.add_argument("--detail-file", default=None, help="where the detail document goes (default: bench_detail.json beside this script, or $KAMD_BENCH_DETAIL)")
    args = ap.parse_args()
    if not args.extras:
        args.no_compact_leg = args.no_config2 = args.no_pinned_pipeline = True
    if args.end_to_end is None:
        args.end_to_end = 8_000_000 if args.extras else 0
    if args.gpus > 1 and "RANK" not in os.environ and "LOCAL_RANK" not in os.environ:
        # `python bench.py --gpus N` from a bare shell: this process becomes the launcher of its own N ranks
        raise SystemExit(self_launch(args.gpus))
    if args.prepare_only:
        prepare_workload(args.workload, args.genes or (6000 if args.workload == "yeast" else 20000), is_builder=True)
        return
    t_start = time.time()
    # the two appended child runs (compact-table legs, config #2) only start while the whole run is inside this many seconds (KAMD_BENCH_BUDGET_S)
    budget_s = float(os.environ.get("KAMD_BENCH_BUDGET_S", "420"))
    if args.table_layout:
        os.environ["KAMD_TABLE_LAYOUT"] = args.table_layout
    if args.table_load:
        os.environ["KAMD_TABLE_LOAD"] = str(args.table_load)

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
        # (a bench policy, not the library's: a rank that never comes back from ncclCommInitRank ends its process after this long instead of holding
        # the launcher until its own time limit; the library arms the watchdog only when the variable is set)
        os.environ.setdefault("KAMD_COMM_INIT_TIMEOUT_S", "300")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.bootstraps is None:
        args.bootstraps = 100 if world == 1 else 0
    paired = args.workload != "yeast"
    stress = args.workload == "stress"
    genes = args.genes or (20000 if paired else 6000)
    n_default = 30_000_000 if paired else 10_000_000
    n_arg = args.pairs or n_default
    # weak: n_arg per GPU; strong: n_arg in total.  Each rank generates max(weak, strong) share once and both modes run on prefixes.
    n_weak, n_strong = n_arg, max(n_arg // world, 1)
    n = n_weak if args.scaling == "weak" else n_strong
    n_gen = max(n_weak, n_strong) if world > 1 else n_arg
    cat, tlens, idx_path = prepare_workload(args.workload, genes, is_builder=(rank == 0))
    t0 = time.time()
    index = ka.Index(idx_path)
    log(f"index flattened in {time.time()-t0:.1f}s: k={index.k} targets={index.num_targets} k-mers={index.num_kmers} "
        f"unitigs={index.num_unitigs} index ECs={index.num_ecs}")
    ctx = ka.Context(local)
    ctx.upload(index)

    # ---- this rank's reads, generated on the device and packed into the 2-bit layout (resident in HBM) ----
    L = 100
    def make_sim(seed):
        if stress:   # 12 % of the pairs from the background (random sequence + copies of the repeat families), errors 0.2 % rising to 5 % (7.5 % on mate 2) at the 3' end
            return ReadSimulator(cat, tlens, dev, seed=seed, read_len=L, err=0.002, background=prepare_workload.background, off_frac=0.12, tail_err=0.05)
        return ReadSimulator(cat, tlens, dev, seed=seed, read_len=L)
    sim = make_sim(1000 + rank)
    rec = ka.packed_record_words(L)
    per = 2 if paired else 1
    words = torch.empty(n_gen * per * rec, dtype=torch.int32, device=dev)
    lens = torch.empty(n_gen * per, dtype=torch.int16, device=dev)
    chunk = 2_000_000
    sample = None      # first pairs: CPU baseline
    psample = None     # first pairs: parity against the reference (independent of the CPU baseline)
    e2e_sample = None  # first pairs: front-end from FASTQ
    tail_sample = None # last pairs: parity of the tail
    spool = None
    full_size = genes == (20000 if paired else 6000) and n_arg == n_default
    if rank == 0 and world == 1 and (args.full_parity == "on" or (args.full_parity == "auto" and full_size)) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")):
        need = n_gen * per * (2 * L + 70) + (8 << 30)
        os.makedirs(CACHE, exist_ok=True)
        if shutil.disk_usage(CACHE).free > need:
            spool = FastqSpool(os.path.join(CACHE, f"full_{os.getpid()}"), paired, L)
        else:
            log(f"full-size parity skipped: {shutil.disk_usage(CACHE).free / 1e9:.0f} GB free under {CACHE}, {need / 1e9:.0f} GB wanted")
    # the CPU baseline: the reference's own stage clocks of the full-size run when there is one (it runs anyway, in the background of the
    # parity legs); a run of its own on a sample of the first pairs otherwise
    cpu_from_full = spool is not None and not args.no_cpu_baseline
    want_head = 0
    if rank == 0 and world == 1:
        if args.cpu_sample and not args.no_cpu_baseline and not cpu_from_full:
            want_head = max(want_head, min(args.cpu_sample, n_gen))
        if args.end_to_end:
            want_head = max(want_head, min(args.end_to_end, n_gen))
        if args.parity_sample:
            want_head = max(want_head, min(args.parity_sample, n_gen))
    want_tail = min(args.parity_sample, n_gen) if (rank == 0 and world == 1 and args.parity_sample) else 0
    head1, head2, have_head = [], [], 0
    t0 = time.time()
    for s in range(0, n_gen, chunk):
        m = min(chunk, n_gen - s)
        r1, r2 = sim.draw(m)
        if spool is not None:
            spool.add([r1, r2] if paired else [r1])
        inter = torch.stack([r1, r2], 1).reshape(2 * m, L) if paired else r1  # mate 1, mate 2 interleaved (ProcessReads.cpp:1034-1041)
        w, l = ctx.pack_reads(inter, L)
        words[s * per * rec:(s + m) * per * rec] = w
        lens[per * s:per * (s + m)] = l
        if have_head < want_head:
            k = min(want_head - have_head, m)
            head1.append(r1[:k].cpu().numpy()); head2.append(r2[:k].cpu().numpy())
            have_head += k
        if want_tail and s + m == n_gen:
            k = min(want_tail, m)   # (the tail lies inside the last chunk: parity_sample <= chunk)
            tail_sample = (r1[m - k:].cpu().numpy(), r2[m - k:].cpu().numpy() if paired else None)
        del r1, r2, inter, w, l
    if want_head:
        h1, h2 = np.concatenate(head1), np.concatenate(head2)
        if args.cpu_sample and not args.no_cpu_baseline and not cpu_from_full:
            k = min(args.cpu_sample, n_gen)
            sample = (h1[:k], h2[:k] if paired else None)
        if args.parity_sample:
            k = min(args.parity_sample, n_gen)
            psample = (h1[:k], h2[:k] if paired else None)
        if args.end_to_end:
            k = min(args.end_to_end, n_gen)
            e2e_sample = (h1[:k], h2[:k] if paired else None)
        del head1, head2
    torch.cuda.synchronize()
    full_files = None
    if spool is not None:
        try:
            full_files = spool.close()
        except Exception as e:   # noqa: BLE001
            log(str(e))
            spool.remove(); spool = None
    log(f"{n_gen} synthetic {'PE' if paired else 'SE'}-{L} {'pairs' if paired else 'reads'} generated + packed on the device in "
        f"{time.time()-t0:.1f}s ({words.numel()*4/1e9:.2f} GB in HBM)" + (f"; the whole input also written as FASTQ ({sum(os.path.getsize(f) for f in full_files)/1e9:.1f} GB)" if full_files else ""))

    opts = ka.QuantOpts(1, 0.0, 0.0, 0, 0) if paired else ka.QuantOpts(0, 200.0, 20.0, 0, 0)
    cli_extra = [] if paired else ["--single", "-l", "200", "-s", "20"]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_items):
        """W untimed + K timed steps on the first n_items items of this rank; returns (seconds max over ranks, last result, per-step profiles)"""
        def step():
            ctx.reset()
            return ka.quant(ctx, opts, [(words[:n_items * per * rec], lens[:per * n_items], n_items, L)], download_ecs=False)
        for _ in range(args.warmup):
            res = step()
        fence()
        t0 = time.perf_counter()
        profs = []
        for _ in range(args.steps):
            res = step()
            profs.append(ctx.profile())
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res, profs

    elapsed, res, profs = timed(n)
    pr = profs[-1]
    align_ms = [p["align_kernel_ms"] for p in profs]; em_ms = [p["em_ms"] for p in profs]
    cls_ms = [p["classify_ms"] for p in profs]; fin_ms = [p["finalize_ms"] for p in profs]
    abs_ms = [p["absorb_ms"] for p in profs]
    em_iters = [p["em_iters"] for p in profs]
    st = res.stats
    total_items = n * world * args.steps
    other = None
    if world > 1:   # the other scaling mode, same run
        n_o = n_strong if args.scaling == "weak" else n_weak
        el_o, _, _ = timed(n_o)
        other = {"scaling": "strong" if args.scaling == "weak" else "weak", "pairs_per_gpu": n_o, "pairs_total_per_step": n_o * world,
                 "value": round(n_o * world * args.steps / el_o / 1e6, 4), "unit": "M read pairs/s", "ms_per_step": round(el_o / args.steps * 1e3, 3),
                 "note": "BASELINE config #4 is the strong case: the same 30 M pairs sharded over the GPUs; the EC merge (one all-reduce + "
                         "all-gathers) and the EM (partitioned by connected component, stop rule summed over the ranks) do not shrink with "
                         "1/N, so strong scaling is bounded by them" }
    # ---- N > 1, no flags: the MERGED result of the ranks on a bounded sample (the first --parity-sample pairs of every rank, in rank order = the
    # input order of one process reading all of them) against the unmodified reference at -t 1 (MasterProcessor::update, src/ProcessReads.cpp:424-481,
    # is the merge kamd_ec_allreduce replaces), and the CPU baseline from that very run -- so that a scaling line carries a parity verdict and a
    # cpu_baseline like the one-GPU line does.  --multi-parity is the heavier leg below (every read of every rank).
    multi_parity = None
    sample_cpu = None
    if world > 1 and not args.multi_parity and args.parity_sample and not args.no_multi_sample_parity:
        n_par = min(n, args.parity_sample)
        ctx.reset()
        mres = ka.quant(ctx, opts, [(words[:n_par * per * rec], lens[:per * n_par], n_par, L)], download_ecs=True)   # collective: every rank calls it
        if rank == 0:
            try:
                if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")):
                    raise RuntimeError("oracle/_ref/dump_ec not built")
                host1, host2 = [], []
                for r in range(world):   # rank r's first pairs: the same generator, the same seed, the same first chunk
                    q1, q2 = make_sim(1000 + r).draw(min(chunk, n_gen))
                    host1.append(q1[:n_par].cpu().numpy()); host2.append(q2[:n_par].cpu().numpy())
                    del q1, q2
                rp = reference_parity(idx_path, np.concatenate(host1), np.concatenate(host2) if paired else None, mres, cli_extra)
                multi_parity = {"ranks": world, "items_per_rank": n_par, "what": "the merged result of the ranks on the first pairs of every rank against the unmodified "
                                "reference (dump_ec quant -t 1) on those pairs in rank order; --multi-parity also compares all reads of all ranks with one rank that sees them",
                                **{k: rp[k] for k in ("ok", "ec_multiset_equal", "flens_equal", "eff_length_equal", "est_counts_max_rel_err_tpm_ge_1e-3",
                                                      "tpm_max_rel_err_tpm_ge_1e-3", "zero_pattern_equal", "sample", "reference_seconds", "n_pseudoaligned", "em_rounds_gpu") if k in rp}}
                sample_cpu = {"value": round(rp["sample"] / max(rp["reference_seconds"], 1e-9) / 1e6, 4), "unit": "M read pairs/s" if paired else "M reads/s", "cores": 1,
                              "sample_short": f"{rp['sample']} {'pairs' if paired else 'reads'} ({n_par}/rank), unmodified reference `quant -t 1` incl. index load, rank 0's host",
                              "kind": "reference", "processors_visible": os.cpu_count(),
                              "sample": f"the multi_rank_parity sample: {rp['sample']} {'pairs' if paired else 'reads'} ({n_par} per rank) as uncompressed FASTQ through the unmodified reference at "
                                        f"-t 1 (oracle/_ref/dump_ec quant: index load, ProcessReads, EMAlgorithm::run; {rp['reference_seconds']:.1f} s in all), on rank 0's host while the other "
                                        f"ranks wait -- one thread, because the parity leg needs the deterministic fragment-length sample; the one-GPU line reports the reference on all cores"}
                del host1, host2
            except Exception as e:   # noqa: BLE001
                multi_parity = {"ok": False, "error": str(e)[:300]}
        fence()
    if world > 1 and args.multi_parity:
        ctx.reset()
        mres = ka.quant(ctx, opts, [(words[:n * per * rec], lens[:per * n], n, L)], download_ecs=True)   # collective: every rank calls it
        if rank == 0:
            try:
                ctx1 = ka.Context(local)
                ctx1.upload(index)
                batches, keep = [], []
                want_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dump_ec")) and n * world <= 2_000_000
                host1, host2 = [], []   # every rank's reads on the host too, in input order, for the reference (small runs only)
                for r in range(world):   # rank r's reads: the same generator, the same seed, the same chunks
                    simr = make_sim(1000 + r)
                    wr = torch.empty(n * per * rec, dtype=torch.int32, device=dev); lr = torch.empty(n * per, dtype=torch.int16, device=dev)
                    for s0 in range(0, n_gen, chunk):
                        m = min(chunk, n_gen - s0)
                        q1, q2 = simr.draw(m)
                        if s0 >= n:
                            break
                        m = min(m, n - s0)
                        inter = torch.stack([q1[:m], q2[:m]], 1).reshape(2 * m, L) if paired else q1[:m]
                        w, l = ctx1.pack_reads(inter, L)
                        wr[s0 * per * rec:(s0 + m) * per * rec] = w; lr[per * s0:per * (s0 + m)] = l
                        if want_ref:
                            host1.append(q1[:m].cpu().numpy()); host2.append(q2[:m].cpu().numpy())
                    keep.append((wr, lr)); batches.append((wr, lr, n, L))
                sres = ka.quant(ctx1, opts, batches, download_ecs=True, comm=False)
                big = sres.est_counts > 1e-6
                rel = float(np.max(np.abs(mres.est_counts[big] - sres.est_counts[big]) / sres.est_counts[big])) if big.any() else 0.0
                multi_parity = {"ranks": world, "items_per_rank": n, "ec_multiset_equal": bool(mres.ecs.multiset() == sres.ecs.multiset()),
                                "flens_equal": bool(np.array_equal(mres.flens, sres.flens)), "n_processed": [int(mres.n_processed), int(sres.n_processed)],
                                "est_counts_max_rel_err": rel, "em_rounds": [int(mres.em_rounds), int(sres.em_rounds)]}
                multi_parity["ok"] = bool(multi_parity["ec_multiset_equal"] and multi_parity["flens_equal"] and rel <= 1e-9 and
                                          mres.em_rounds == sres.em_rounds and mres.n_processed == sres.n_processed)
                if want_ref:
                    # ... and the MERGED result of the ranks against the unmodified reference run on all ranks' reads in input order
                    # (dump_ec quant -t 1: MasterProcessor::update, src/ProcessReads.cpp:424-481, is the merge kamd_ec_allreduce replaces)
                    try:
                        rp = reference_parity(idx_path, np.concatenate(host1), np.concatenate(host2) if paired else None, mres, cli_extra)
                        multi_parity["merged_vs_reference"] = {k: rp[k] for k in ("ok", "ec_multiset_equal", "flens_equal", "eff_length_equal", "est_counts_max_rel_err_tpm_ge_1e-3",
                                                                                  "tpm_max_rel_err_tpm_ge_1e-3", "zero_pattern_equal", "sample", "reference_seconds") if k in rp}
                        multi_parity["ok"] = bool(multi_parity["ok"] and rp["ok"])
                    except Exception as e:   # noqa: BLE001
                        multi_parity["merged_vs_reference"] = {"ok": False, "error": str(e)[:300]}
                        multi_parity["ok"] = False
                    del host1, host2
                ctx1.close()
                del keep, batches
            except Exception as e:   # noqa: BLE001
                multi_parity = {"ok": False, "error": str(e)}
        fence()
    # ---- BASELINE config #5 (optional): B bootstrap replicates of the last step's ECs, replicate b on rank b % world ----
    boot = None
    if args.bootstraps > 0:
        try:
            import kallisto_amd.api as A
            ctx.reset()
            res = ka.quant(ctx, opts, [(words[:n * per * rec], lens[:per * n], n, L)], download_ecs=False)
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
        except Exception as e:   # noqa: BLE001  (world == 1: a side leg; with several ranks every rank must reach the collectives, so it is left to fail there)
            if world > 1:
                raise
            boot = {"error": str(e)[:300]}

    # ---- optional: two samples in flight on one GPU (the EM of one is LDS-bound, the pseudoalignment of the next is bound by
    # memory requests) -- a second context on its own stream, two host threads, each runs full quants; NOT the headline value ----
    in_flight = None
    if args.in_flight == 2 and world == 1:
        import threading
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            ctx2 = ka.Context(local)
            ctx2.upload(index)
        steps_each = max(args.steps // 2, 1)
        errs = []

        def worker(cx, stream, delay):
            try:
                time.sleep(delay)   # half a quant out of phase: the EM of one sample meets the pseudoalignment of the other
                with torch.cuda.stream(stream):
                    for _ in range(steps_each):
                        cx.reset()
                        ka.quant(cx, opts, [(words[:n * per * rec], lens[:per * n], n, L)], download_ecs=False)
            except Exception as e:   # noqa: BLE001
                errs.append(str(e))
        for _ in range(2):   # one untimed pass, one timed
            fence()
            t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(ctx, torch.cuda.current_stream(dev), 0.0)),
                  threading.Thread(target=worker, args=(ctx2, side, 0.5 * elapsed / args.steps))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            fence()
            el2 = time.perf_counter() - t0
        in_flight = {"samples_in_flight": 2, "quants": 2 * steps_each, "seconds": round(el2, 4),
                     "value": round(2 * steps_each * n / el2 / 1e6, 4), "unit": "M read pairs/s" if paired else "M reads/s",
                     "ms_per_quant": round(el2 / (2 * steps_each) * 1e3, 3), "errors": errs,
                     "note": "two contexts on two streams, one host thread each: every quant is complete (pseudoalignment, EC resolution, "
                             "FLD, full EM); the throughput of a multi-sample pipeline, not the latency of one quant -- `value` above stays "
                             "the one-sample-at-a-time figure"}
        ctx2.close()

    # ---- the same steps on the COMPACT layout of the k-mer table (kamd_core.h: four quotiented 16-byte slots per line): reported beside
    # the headline, never as the headline (the library's default layout is the wide one until this leg says otherwise) ----
    compact_leg = None
    compact_dir = None
    if rank == 0 and world == 1 and not args.no_compact_leg:
        # The leg runs in a process of its own (tools/compact_table_leg.py), as the very last thing before the line is printed: a fault in a
        # side leg must not take the line down.  The packed reads go through /dev/shm (3 GB for config #3) -- written here, while they exist.
        need = n * per * (rec * 4 + 2)
        base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need + (1 << 30) else CACHE
        compact_dir = os.path.join(base, f"kamd_bench_{os.getpid()}")
        try:
            os.makedirs(compact_dir, exist_ok=True)
            words[:n * per * rec].cpu().numpy().tofile(os.path.join(compact_dir, "words.i32"))
            lens[:per * n].cpu().numpy().tofile(os.path.join(compact_dir, "lens.i16"))
        except Exception as e:   # noqa: BLE001
            compact_leg = [{"error": "reads not handed over: " + str(e)[:300]}]
            shutil.rmtree(compact_dir, ignore_errors=True)
            compact_dir = None

    out = None
    if rank == 0:
        # roofline of kernel A (the dominant kernel by time): ALGORITHMIC bytes of ONE launch (DESIGN.md
        # section 3): packed reads in + 16 B (key + payload) per k-mer probe (dbg.find call of the reference) + the raw record out
        rec_bytes = per * rec * 4 + per * 2
        alg_bytes = n * rec_bytes + 16 * st["n_probes"] + 4 * st["n_raw_words"]
        # the stricter count of SURVEY.md section 8(d): L / 4 bytes of sequence per read (the kernel leaves the non-ACGT plane alone unless a
        # read's flag says it has one) instead of the whole packed record
        alg_bytes_strict = n * per * (L // 4 + 2) + 16 * st["n_probes"] + 4 * st["n_raw_words"]
        a_ms = float(np.mean(align_ms))
        achieved = alg_bytes / (a_ms * 1e-3) / 1e9
        T = int(index.num_targets)
        # EM round: SURVEY 8(d)'s algorithmic bytes of one iteration, independent of the layout:
        #   B_B = nnz*(4 id + 8 alpha gather + 8 next accumulate) + N_ec*(4 count + 8 offsets) + T*(8 alpha + 8 next + 8 eff_len)
        em_bytes = pr["em_nnz"] * 20 + pr["em_necs"] * 12 + T * 24
        em_round_ms = float(np.mean(em_ms)) / max(int(em_iters[-1]), 1)
        em_ach = em_bytes / (em_round_ms * 1e-3) / 1e9
        local_form = pr["em_k"] < 0 and not pr["em_giant_nnz"]   # (the hybrid streams half of the matrix from HBM / MALL every round: priced against HBM)
        if local_form:
            # component-local form: the matrix sits in LDS for the rounds of a launch; a round gathers one FP64 value and reads one
            # 16-bit index per entry and direction out of LDS -- the bound is the LDS pipe, HBM only sees the per-launch load / store
            lds_bytes = 2 * pr["em_nnz"] * (8 + 2)
            lds_peak = 256 * 256 * 2.4   # GB/s: 256 CUs x 256 B/clk (ds_read_b64: 64 banks x 4 B, MI355X_MICROARCH.md LDS section) x 2.4 GHz
            em_roof = {"kernel": "EM round inside k_em_sell (component-local, sliced ELLPACK in LDS)", "kernel_short": "k_em_sell (one EM round)", "bound": "lds",
                       "achieved": round(lds_bytes / (em_round_ms * 1e-3) / 1e9, 2), "peak": round(lds_peak, 1), "unit": "GB/s",
                       "frac": round(lds_bytes / (em_round_ms * 1e-3) / 1e9 / lds_peak, 5), "traffic": None,
                       "algorithmic_lds_bytes_per_round": int(lds_bytes),
                       "hbm_algorithmic_bytes_per_round": int(em_bytes), "hbm_equivalent_GBps": round(em_ach, 2),
                       "hbm_equivalent_frac": round(em_ach / HBM_PEAK_GBS, 5),
                       "launch_ms": round(em_round_ms, 5), "launch": "one EM round = kamd_em_run's HIP-event time / rounds (plan set-up, "
                       "64-round launches, speculative chunk + replay included)",
                       "rounds": int(em_iters[-1]), "nnz": pr["em_nnz"], "rows": pr["em_necs"], "groups": pr["em_grid"],
                       "lds_bytes_per_workgroup": pr["em_lds"]}
        else:
            layout_bytes = pr["em_nnz_multi"] * 24 + pr["em_necs"] * 16 + T * 56
            em_roof = {"kernel": "EM round, hybrid: k_em_sell on the components that fit, k_gi_rows + k_gi_cols (+ fix-ups) on the oversized ones beside it" if pr["em_giant_nnz"] else
                                 "EM round (k_pm_rows_pass + k_pm_cols_pass)" if pr["em_k"] else "EM round (k_em_rows + k_em_seg + k_em_final)",
                       "kernel_short": "hybrid EM round (k_em_sell + oversized side)" if pr["em_giant_nnz"] else "k_pm_rows_pass+k_pm_cols_pass" if pr["em_k"] else "k_em_rows+k_em_seg+k_em_final",
                       "bound": "hbm", "achieved": round(em_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(em_ach / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": int(em_bytes),
                       "layout_bytes_per_round": int(layout_bytes), "launch_ms": round(em_round_ms, 5), "rounds": int(em_iters[-1]),
                       "nnz": pr["em_nnz"], "rows": pr["em_necs"], "entries_per_lane": pr["em_k"]}
        # EC resolution: the batch's tuple records absorbed into the persistent tuple table (inside kamd_pseudoalign: one record read +
        # one 32-byte table slot per record) + kamd_ec_finalize (the distinct tuples resolved, candidate sets written, merged, emitted)
        f_ms = float(np.mean(fin_ms)) + float(np.mean(abs_ms))
        fin_bytes = (4 * st["n_stream_words"] + 8 * n + 32 * st["n_multi"]) + (2 * 4 * pr["tuple_store_words"] + 32 * pr["fin_records"] + 3 * 4 * pr["fin_cand_words"])
        fin_roof = {"kernel": "tuple de-duplication (k_tup_absorb, k_tup_store) + kamd_ec_finalize (k_bound_tuples, k_resolve, k_cand_singles, merge, CSR)",
                    "kernel_short": "tuple dedup + kamd_ec_finalize", "bound": "hbm",
                    "achieved": round(fin_bytes / (f_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(fin_bytes / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                    "algorithmic_bytes_per_launch": int(fin_bytes), "launch_ms": round(f_ms, 3),
                    "note": "dependent gathers (table slot -> owner record -> set offsets -> members): latency-, not bandwidth-bound"}
        unit_name = "pairs" if paired else "reads"
        # what the memory system gives the kernel's access pattern: dependent random 64-byte reads of the k-mer table (measured live;
        # the rate does not depend on the footprint or the access size -- profiles/README.md); the kernel's requests are its
        # bucket lines + text reads + the packed reads
        ceiling = None
        try:
            g_s, _ = ctx.random_lines(256 * 6, 256, 256)
            req_bytes = 64.0 * (st["n_bucket_reads"] + st["n_text_hits"])
            line_gbs = (req_bytes + n * per * rec * 4.0) / (a_ms * 1e-3) / 1e9
            ceiling = {"GB/s_in_64B_lines": round(g_s, 1), "kernel_GB/s_in_64B_lines": round(line_gbs, 1), "frac": round(line_gbs / g_s, 4),
                       "note": "kamd_debug_random_lines at 24 wavefronts/CU; a request-rate ceiling (same for 64 MB .. 2.4 GB footprints, 8 .. 64 B accesses)"}
        except Exception as e:   # diagnostic only
            ceiling = {"error": str(e)}
        out = {
            "metric": ("M paired-end reads/sec quantified (human txome index, stress workload)" if stress else "M paired-end reads/sec quantified (human txome index)") if paired
                      else "M single-end reads/sec quantified (yeast-sized index)",
            "value": round(total_items / elapsed / 1e6, 4),
            "unit": "M read pairs/s" if paired else "M reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u64+f64", "data": "synthetic",
            "config": {
                "workload": ((f"STRESS (not a BASELINE configuration; VERDICT r4 #1): human-sized synthetic transcriptome with real structure ({index.num_targets} transcripts, "
                              f"{index.num_kmers} k-mers, k={index.k}: 4 repeat families in the terminal exons of 30 % of the genes, 3 x 200 paralog genes, poly-A tails on 3 % "
                              f"of the transcripts; index built by the reference `kallisto index`), {n} PE-{L} read pairs per GPU resident in HBM, 12 % of them "
                              f"off-transcriptome, errors 0.2 % rising to 5 / 7.5 % at the 3' end, full quant per step" if stress else
                              f"BASELINE config #{3 if world == 1 else 4}: synthetic human-like transcriptome ({index.num_targets} transcripts, "
                              f"{index.num_kmers} k-mers, k={index.k}; index built by the reference `kallisto index`), "
                              f"{n} PE-{L} read pairs per GPU resident in HBM (2-bit packed), full quant per step"
                              if paired else
                              f"BASELINE config #2: synthetic yeast-like transcriptome ({index.num_targets} transcripts, {index.num_kmers} k-mers, "
                              f"k={index.k}), {n} SE-{L} reads per GPU resident in HBM, --single -l 200 -s 20, full quant per step")
                             if (genes == (20000 if paired else 6000) and n_arg == n_default) or stress else
                             f"REDUCED {args.workload} genes={genes} {unit_name}={n} (not the BASELINE configuration)"),
                "workload_short": ((f"stress (not a BASELINE config): {n} PE-{L} pairs/GPU in HBM, human-sized index with repeat/paralog/poly-A structure, 12% off-transcriptome, full quant/step" if stress else
                                    f"BASELINE config #{3 if world == 1 else 4}: {n} PE-{L} pairs/GPU resident in HBM (2-bit packed), human-like index by the reference's `kallisto index`, full quant/step" if paired else
                                    f"BASELINE config #2: {n} SE-{L} reads/GPU resident in HBM, yeast-like index, --single -l 200 -s 20, full quant/step")
                                   if (genes == (20000 if paired else 6000) and n_arg == n_default) or stress else
                                   f"REDUCED {args.workload} genes={genes} {unit_name}={n} (not the BASELINE configuration)"),
                "parallelism_short": f"{world} ranks (1/GPU): reads sharded, RCCL EC merge, EM by component" if world > 1 else "1 GPU",
                f"{unit_name}_per_gpu": n, "read_len": L, "paired": paired, "targets": int(index.num_targets),
                "kmers": int(index.num_kmers),
                "kmer_table": {"layout": "compact" if index.view.table_layout else "wide", "slots_per_line": int(index.view.slots_per_bucket),
                               "bytes": int((index.view.n_buckets + index.view.pad_buckets) * 64),
                               "load": round(index.num_kmers / float(index.view.slots_per_bucket * index.view.n_buckets), 3)},
                "parallelism": (f"{world} ranks, one per GPU: reads sharded; in the library (RCCL): one all-reduce of the dense EC count vector + "
                                f"all-gathers of the tuple records, then the EM partitioned over the ranks by connected component"
                                if world > 1 else "1 GPU"),
                "collective_backend": (getattr(getattr(ctx, "_comm", None), "transport", None) or f"unknown ({backend})") if world > 1 else None,
                "n_ranks_seen": (ctx._comm.info()["ranks_seen"] if getattr(ctx, "_comm", None) is not None else None) if world > 1 else 1,
                "launcher": "self (python bench.py --gpus N -> torch.distributed.run)" if os.environ.get("KAMD_BENCH_LAUNCHER") == "self" else
                            ("external (torch.distributed.run)" if world > 1 else "none"),
            },
            "breakdown_ms": {"pseudoalign_kernel": round(a_ms, 3), "classify_kernel": round(float(np.mean(cls_ms)), 3),
                             "kernel_a_version": pr["kernel_a_version"], "tuple_dedup": round(float(np.mean(abs_ms)), 3),
                             "ec_finalize": round(float(np.mean(fin_ms)), 3), "em": round(float(np.mean(em_ms)), 3),
                             "em_rounds": int(em_iters[-1]), "step_total": round(elapsed / args.steps * 1e3, 3),
                             **({"ec_merge": round(float(np.mean([p_["merge_ms"] for p_ in profs])), 3),
                                 "em_collectives": round(float(np.mean([p_["em_collective_ms"] for p_ in profs])), 3),
                                 "em_collectives_n": int(profs[-1]["em_collectives"]),
                                 "collective_ms": round(float(np.mean([p_["merge_ms"] + p_["em_collective_ms"] for p_ in profs])), 3),
                                 "collective_note": "ec_merge = kamd_ec_allreduce (one all-gather of sizes, the all-reduce of the dense counts, the all-gathers of the records and "
                                                    "their de-duplication; HIP events on rank 0); em_collectives = host wall time inside the EM's all-reduces (one per chunk of 64 rounds + the "
                                                    "final sum)"} if world > 1 else {})},
            "counters": {"probes_per_pair": round(st["n_probes"] / n, 3),
                         "bucket_reads_per_pair": round(st["n_bucket_reads"] / n, 3), "text_answers_per_pair": round(st["n_text_hits"] / n, 3),
                         "lane_utilisation": round(st["n_lane_iters"] / max(64 * st["n_wave_iters"], 1), 4),
                         "single_set_pairs": st["n_single"], "multi_set_pairs": st["n_multi"],
                         "distinct_tuples": st["n_distinct_tuples"], "final_ecs": int(ctx.ec_result.n_ecs),
                         "ec_state_bytes": int(32 * pr["tuple_table_slots"] + 4 * pr["tuple_store_words"] + 12 * index.num_ecs),
                         "em_rounds": res.em_rounds,
                         "pseudoaligned_share": round(res.n_pseudoaligned / max(res.n_processed, 1), 4),
                         "overflow_items": pr["n_overflow_items"], "overflow_share": round(pr["n_overflow_items"] / max(n, 1), 5), "overflow_kernel_ms": round(pr["overflow_ms"], 3),
                         "em_largest_component_nnz": pr["em_max_comp_nnz"],
                         "em_form": ("hybrid: k_em_sell on the components that fit + streamed kernels on the oversized ones" if pr["em_giant_nnz"] else
                                     "component-local (k_em_sell)") if pr["em_k"] < 0 else ("streamed" if pr["em_k"] > 0 else "csr"),
                         "em_oversized": {"nnz": pr["em_giant_nnz"], "rows": pr["em_giant_rows"], "transcripts": pr["em_giant_tr"], "chunks_per_direction": pr["em_giant_chunks"]} if pr["em_giant_nnz"] else None,
                         "em_plan_ms": round(pr["em_plan_ms"], 3)},
            # dominant kernel by time: kernel A
            "roofline": {"kernel": {3: "k_match_v3"}[pr["kernel_a_version"]], "bound": "hbm",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "launch_ms": round(a_ms, 3),
                         "sequence_plane_only": {"algorithmic_bytes_per_launch": int(alg_bytes_strict),
                                                 "frac": round(alg_bytes_strict / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                                 "note": "reads counted as L/4 bytes of sequence (+ length) instead of the packed record with its non-ACGT plane"},
                         "launch": "one k_match_v3 launch over the step's batch, HIP events on the context stream (the FLD kernel of the first "
                                   "prefix runs underneath it on a side stream)",
                         "table_line_bytes_per_launch": int(64 * st["n_bucket_reads"]), "text_bytes_per_launch": int(12 * st["n_text_hits"]),
                         "random_line_ceiling": ceiling},
            "roofline_em": em_roof,
            "roofline_finalize": fin_roof,
        }
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                tj = json.load(open(prof))
                if tj.get("pairs") == n and tj.get("workload") == args.workload and tj.get("genes") == genes and world == 1:
                    out["roofline"]["traffic"] = tj.get("kernel_a_hbm_bytes")
                    out["roofline_em"]["traffic"] = tj.get("em_round_hbm_bytes")
                    out["roofline_finalize"]["traffic"] = tj.get("finalize_hbm_bytes")
                    for k in ("roofline", "roofline_em", "roofline_finalize"):
                        out[k]["traffic_source"] = "static: " + str(tj.get("source"))
            except Exception:
                pass
        if other is not None:
            out["other_scaling"] = other
    # ---- CPU baseline (rank 0, N=1 only): the reference at -t <cores> for the timing; parity against the reference at -t 1 ----
    unit_name = "pairs" if paired else "reads"
    rate_unit = "M read pairs/s" if paired else "M reads/s"
    if rank == 0 and world == 1 and sample is not None:
        threads = min(effective_cpus(), 64)   # (the cgroup's CPU quota, not the host's processor count: more busy threads than that get throttled)
        k = sample[0].shape[0]
        log(f"CPU baseline: reference `kallisto quant -t {threads}` on the first {k} {unit_name} ...")
        try:
            cb = cpu_reference_baseline(idx_path, sample[0], sample[1], threads, cli_extra)
            out["cpu_baseline"] = {"value": round(k / cb["seconds"] / 1e6, 4), "unit": rate_unit, "cores": threads,
                                   "kind": "reference", "processors_visible": os.cpu_count(),
                                   "sample": f"first {k} {unit_name} of rank 0's reads as uncompressed FASTQ, `kallisto quant -t {threads} "
                                             f"--plaintext {' '.join(cli_extra)}`, clock from index-loaded to exit ({cb['seconds']:.1f}s; index load "
                                             f"{cb['index_load_s']:.1f}s excluded)",
                                   "sample_short": f"first {k} {unit_name} of rank 0's reads, uncompressed FASTQ, unmodified reference `quant -t {threads}`, index load excluded",
                                   "index_load_seconds": round(cb["index_load_s"], 2),
                                   "whole_run_value_including_index_load": round(k / (cb["seconds"] + cb["index_load_s"]) / 1e6, 4),
                                   "pseudoalign_seconds": round(cb["pseudoalign_s"], 2), "em_seconds": round(cb["em_s"], 2),
                                   "pseudoalign_only_value": round(k / max(cb["pseudoalign_s"], 1e-9) / 1e6, 4),
                                   "projected_value_at_full_size": round(n / (n / k * cb["pseudoalign_s"] + cb["em_s"] + (cb["seconds"] - cb["pseudoalign_s"] - cb["em_s"])) / 1e6, 4),
                                   "projected_whole_run_value_at_full_size": round(n / (n / k * cb["pseudoalign_s"] + cb["em_s"] + (cb["seconds"] - cb["pseudoalign_s"] - cb["em_s"]) + cb["index_load_s"]) / 1e6, 4),
                                   "note": "the reference's EM is single-threaded and independent of the read count; projected_value_at_full_size "
                                           f"scales the threaded pseudoalignment stage to the {n} {unit_name} of the GPU workload and keeps the EM and output time"}
        except Exception as e:  # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"value": None, "unit": rate_unit, "cores": threads, "kind": "reference",
                                   "sample": f"failed: {e}"}
    full_parity = None
    if rank == 0 and world == 1 and full_files is not None:
        # the whole input through the unmodified reference, in the background from here on (all cores for about half a minute, then one)
        log(f"full-size parity: reference `dump_ec quant -t {min(effective_cpus(), 64)}` on all {n} {unit_name}, in the background ...")
        try:
            ctx.reset()
            fres = ka.quant(ctx, opts, [(words[:n * per * rec], lens[:per * n], n, L)], download_ecs=True)
            full_parity = FullSizeParity(idx_path, full_files, fres, min(effective_cpus(), 64), cli_extra)
        except Exception as e:   # noqa: BLE001
            out["parity_check_full_size"] = {"ok": False, "error": str(e)[:300]}
    if rank == 0 and world == 1 and psample is not None:
        # parity gate: the same reads through the HIP path and through the unmodified reference at -t 1
        ks = psample[0].shape[0]
        log(f"parity: reference `dump_ec quant -t 1` on the first {ks} {unit_name} ...")
        try:
            ctx.reset()
            gres = ka.quant(ctx, opts, [(words[:ks * per * rec], lens[:per * ks], ks, L)], download_ecs=True)
            out["parity_check"] = reference_parity(idx_path, psample[0], psample[1] if paired else None, gres, cli_extra)
        except Exception as e:
            out["parity_check"] = {"ok": False, "error": str(e)}
        if tail_sample is not None:
            log(f"parity: the last {tail_sample[0].shape[0]} {unit_name}, pseudoaligned as the final batch of a run over all {n} ...")
            try:
                out["parity_check_tail"] = tail_parity(ctx, opts, idx_path, words, lens, n, per, rec, L, tail_sample[0], tail_sample[1], cli_extra)
            except Exception as e:
                out["parity_check_tail"] = {"ok": False, "error": str(e)}
    if rank == 0 and world == 1 and not args.no_pinned_pipeline:
        log("pinned pipeline: packed reads in pinned host memory -> ECs ...")
        try:
            out["pinned_pipeline"] = pinned_pipeline(ka, ctx, opts, words, lens, n, per, rec, L)
            out["pinned_pipeline"]["unit"] = rate_unit
        except Exception as e:
            out["pinned_pipeline"] = {"error": str(e)}
    if rank == 0 and compact_dir is not None:
        if time.time() - t_start > budget_s - 60:
            compact_leg = [{"skipped": f"{time.time() - t_start:.0f} s into the run (budget {budget_s:.0f} s, KAMD_BENCH_BUDGET_S)"}]
            shutil.rmtree(compact_dir, ignore_errors=True)
        else:
            log("compact k-mer table: the same steps in a child process ...")
            compact_leg = compact_table_leg(idx_path, compact_dir, n, L, paired, args.steps, max(args.warmup, 1), local, res,
                                            loads="0.6,0.5,0.7" if index.view.table_layout == 0 else "wide,0.6",
                                            timeout_s=max(60.0, budget_s + 120 - (time.time() - t_start)))
    if rank == 0 and world == 1 and args.workload == "human" and not args.no_config2 and genes == 20000 and n_arg == n_default:
        if time.time() - t_start > budget_s - 45:
            out["config2"] = {"skipped": f"{time.time() - t_start:.0f} s into the run (budget {budget_s:.0f} s, KAMD_BENCH_BUDGET_S)"}
        else:
            log("BASELINE config #2 (yeast, single-end) as a child run ...")
            out["config2"] = config2_leg(timeout_s=max(60.0, budget_s + 120 - (time.time() - t_start)))
    def attach_full(fp):
        """the full-size leg's verdict, and the CPU baseline from the clocks of that very run"""
        out["parity_check_full_size"] = fp
        st_ = fp.get("reference_stage_seconds")
        if cpu_from_full and st_ and "cpu_baseline" not in out:
            work = st_["pseudoalign"] + st_["em"] + st_["output"]
            out["cpu_baseline"] = {"value": round(n / work / 1e6, 4), "unit": rate_unit, "cores": st_["threads"], "kind": "reference", "processors_visible": os.cpu_count(),
                                   "sample": f"ALL {n} {unit_name} of the run as uncompressed FASTQ through the unmodified reference (oracle/_ref/dump_ec quant -t {st_['threads']} = "
                                             f"KmerIndex::load, ProcessReads on all cores, EMAlgorithm::run single-threaded; the run of parity_check_full_size, in the background of the "
                                             f"other parity legs, which keep one or two cores busy), clock from index-loaded to exit ({work:.1f}s; index load {st_['index_load']:.1f}s excluded)",
                                   "sample_short": f"all {n} {unit_name} of the run, uncompressed FASTQ, unmodified reference `quant -t {st_['threads']}`, index load excluded",
                                   "index_load_seconds": st_["index_load"], "pseudoalign_seconds": st_["pseudoalign"], "em_seconds": st_["em"],
                                   "whole_run_value_including_index_load": round(n / (work + st_["index_load"]) / 1e6, 4),
                                   "pseudoalign_only_value": round(n / max(st_["pseudoalign"], 1e-9) / 1e6, 4),
                                   "note": "measured at the full size of the configuration, not projected from a sample (VERDICT r4); the reference's EM is single-threaded"}
        elif cpu_from_full and "cpu_baseline" not in out:
            out["cpu_baseline"] = {"value": None, "unit": rate_unit, "cores": min(effective_cpus(), 64), "kind": "reference", "sample": "failed: " + str(fp.get("error", "no stage clocks"))[:200]}
    want_gencode = rank == 0 and world == 1 and args.workload == "human" and not args.no_gencode_leg and genes == 20000 and n_arg == n_default
    if rank == 0 and world == 1 and args.workload == "human" and not args.no_stress_leg and genes == 20000 and n_arg == n_default:
        if time.time() - t_start > budget_s - 30:
            out["stress"] = {"skipped": f"{time.time() - t_start:.0f} s into the run (budget {budget_s:.0f} s, KAMD_BENCH_BUDGET_S)"}
        else:
            log("stress workload (real transcriptome structure, off-transcriptome reads) as a child run ...")
            if full_parity is not None:   # (both want every core: the stress leg runs the reference on all of its pairs too)
                log("full-size parity: waiting for the reference ...")
                attach_full(full_parity.finish())
                full_parity = None
            out["stress"] = stress_leg(timeout_s=max(120.0, budget_s + 360 - (time.time() - t_start)))
    if want_gencode:
        if time.time() - t_start > budget_s + 60:
            out["gencode_size"] = {"skipped": f"{time.time() - t_start:.0f} s into the run (budget {budget_s:.0f} s, KAMD_BENCH_BUDGET_S); builder-side figures: profiles/r05_bench_gencode_size.json"}
        else:
            # (its index is built by the reference inside the child, ~70 s on all cores: only once the full-size reference run -- whose stage clocks are
            # the CPU baseline -- has finished; a first version prepared it in the background of that run's single-threaded EM and slowed it by 40 %)
            if full_parity is not None:
                log("full-size parity: waiting for the reference ...")
                attach_full(full_parity.finish())
                full_parity = None
            log("GENCODE-sized index (46 000 genes) as a child run ...")
            out["gencode_size"] = gencode_leg(None, timeout_s=max(120.0, budget_s + 480 - (time.time() - t_start)))
    if rank == 0 and world == 1 and e2e_sample is not None:
        log(f"end to end: kallisto_amd_quant from FASTQ ({e2e_sample[0].shape[0]} {unit_name}) ...")
        if full_parity is not None:   # (the reference's single-threaded EM may still be running: the end-to-end legs want the host to themselves)
            log("full-size parity: waiting for the reference ...")
            attach_full(full_parity.finish())
            full_parity = None
        ctx.close()   # the front-end is its own process on the same GPU
        del words, lens
        torch.cuda.empty_cache()
        try:
            out["end_to_end"] = end_to_end(idx_path, e2e_sample[0], e2e_sample[1], paired, min(effective_cpus(), 64), cli_extra, full_files=full_files, full_items=n)
            out["end_to_end"]["host"] = {"cpus_available": effective_cpus(), "processors_visible": os.cpu_count()}
        except Exception as e:
            out["end_to_end"] = {"error": str(e)}
    if full_parity is not None:
        attach_full(full_parity.finish())
    if spool is not None:
        spool.remove()
    if rank == 0:
        if sample_cpu is not None and "cpu_baseline" not in out:
            out["cpu_baseline"] = sample_cpu
        if multi_parity is not None:
            out["multi_rank_parity"] = multi_parity
        if boot is not None:
            out["bootstrap"] = boot
        if in_flight is not None:
            out["two_samples_in_flight"] = in_flight
        if compact_leg is not None:
            out["kmer_table_layouts"] = {"legs": compact_leg,
                                         "note": "the same steps on the other layouts of the k-mer table (wide = three 20-byte slots per 64-byte line at a load of 0.5; "
                                                 "compact = four exact 16-byte slots by quotienting, DESIGN.md section 2, at other load factors); a side measurement -- "
                                                 "`value` above is the library's default (auto: compact when its fields fit, at the load factor kamd_index.cpp picks from the table's size -- config.kmer_table says which)"}
        out["seconds_total"] = round(time.time() - t_start, 1)
        detail = write_detail(out, args.detail_file)
        log("detail: " + json.dumps({k: out[k] for k in ("value", "ms_per_step", "breakdown_ms") if k in out}) + (f"; everything else in {detail}" if detail else ""))
        print(contract_line(out, detail), flush=True)
    if world > 1:
        # tear down in order while everything is alive: the library's communicator (ncclCommDestroy), the context, then torch's
        # process group; the interpreter's own shutdown order is not one a C++ runtime survives reliably, and the line is printed
        try:
            comm = getattr(ctx, "_comm", None)
            if comm is not None:
                comm.close()
            ctx.close()
            dist.barrier()
            dist.destroy_process_group()
        finally:
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
