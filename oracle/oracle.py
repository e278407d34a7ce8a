"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(kallisto_amd/) must never do so.  See oracle/kallisto_oracle.h for what is restated and how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")
MAX_FRAG_LEN = 1000


def build(force: bool = False) -> None:
    """Compile the C restatement (and, when /root/reference is present, the unmodified reference into oracle/_ref)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(HERE, "kallisto_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)


class Hit(C.Structure):
    _fields_ = [("unitig", C.c_uint32), ("dist", C.c_uint32), ("size", C.c_uint32), ("strand", C.c_uint32),
                ("lb", C.c_uint32), ("ub", C.c_uint32), ("block", C.c_uint32), ("ec", C.c_uint32), ("pos", C.c_int32)]


class Opts(C.Structure):
    _fields_ = [("paired", C.c_int), ("fld", C.c_double), ("sd", C.c_double), ("single_overhang", C.c_int),
                ("strand", C.c_int), ("no_jump", C.c_int), ("do_union", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        u64, u32p, u64p, i32p, dp, vp = (C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p)
        L.ko_index_load.restype = vp
        L.ko_index_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.ko_index_free.argtypes = [vp]
        for f in ("ko_index_num_unitigs", "ko_index_num_kmers", "ko_index_num_blocks", "ko_index_num_ecs",
                  "ko_index_num_targets", "ko_index_dlist_size"):
            getattr(L, f).restype = u64
            getattr(L, f).argtypes = [vp]
        L.ko_index_k.argtypes = [vp]
        L.ko_index_target_lens.restype = i32p
        L.ko_index_target_lens.argtypes = [vp]
        L.ko_index_target_name.restype = C.c_char_p
        L.ko_index_target_name.argtypes = [vp, u64]
        L.ko_index_ec.restype = u64
        L.ko_index_ec.argtypes = [vp, C.c_uint32, C.POINTER(u32p)]
        L.ko_index_find.argtypes = [vp, u64, C.POINTER(Hit)]
        L.ko_match.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(Hit), C.c_int, C.POINTER(C.c_int)]
        L.ko_pseudoalign.argtypes = [vp, C.POINTER(Opts), C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_double,
                                     C.c_int, u32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ko_map_pair.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.ko_find_position.argtypes = [vp, C.c_uint32, C.POINTER(Hit), C.c_int, C.POINTER(C.c_int)]
        L.ko_result_new.restype = vp
        L.ko_result_free.argtypes = [vp]
        L.ko_process_reads.restype = C.c_int64
        L.ko_process_reads.argtypes = [vp, C.POINTER(Opts), C.c_char_p, u64p, i32p, u64, vp]
        for f in ("ko_result_num_ecs", "ko_result_nnz", "ko_result_num_processed", "ko_result_num_probes",
                  "ko_result_num_hits"):
            getattr(L, f).restype = u64
            getattr(L, f).argtypes = [vp]
        L.ko_result_export.argtypes = [vp, u64p, u32p, u32p]
        L.ko_result_flens.restype = u32p
        L.ko_result_flens.argtypes = [vp]
        L.ko_mean_frag_lens_trunc.argtypes = [u32p, dp]
        L.ko_trunc_gaussian_fld.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, dp]
        L.ko_trunc_gaussian_counts.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, u32p]
        L.ko_frag_len_means.argtypes = [i32p, u64, dp, dp]
        L.ko_calc_eff_lens.argtypes = [i32p, u64, dp, dp]
        L.ko_em_run.argtypes = [u64p, u32p, u32p, u32p, u64, dp, u64, u64, u64, dp, dp]
        L.ko_counts_to_tpm.argtypes = [dp, dp, u64, dp]
        L.ko_bootstrap_seeds.argtypes = [u64, C.c_int, u64p]
        L.ko_multinomial_sample.argtypes = [u32p, u64, u64, u32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Index:
    def __init__(self, path: str):
        err = C.create_string_buffer(256)
        self.h = lib().ko_index_load(path.encode(), err, 256)
        if not self.h:
            raise RuntimeError("oracle index load failed: " + err.value.decode())
        L = lib()
        self.k = L.ko_index_k(self.h)
        self.num_unitigs = L.ko_index_num_unitigs(self.h)
        self.num_kmers = L.ko_index_num_kmers(self.h)
        self.num_blocks = L.ko_index_num_blocks(self.h)
        self.num_ecs = L.ko_index_num_ecs(self.h)
        self.num_targets = L.ko_index_num_targets(self.h)
        self.dlist_size = L.ko_index_dlist_size(self.h)
        lens = L.ko_index_target_lens(self.h)
        self.target_lens = np.ctypeslib.as_array(lens, shape=(self.num_targets,)).copy() if self.num_targets else \
            np.zeros(0, np.int32)

    def target_names(self):
        return [lib().ko_index_target_name(self.h, i).decode() for i in range(self.num_targets)]

    def ec(self, ecid: int):
        p = C.POINTER(C.c_uint32)()
        n = lib().ko_index_ec(self.h, ecid, C.byref(p))
        return [p[i] for i in range(n)]

    def match(self, seq: bytes, partial: bool = False, max_hits: int = 4096):
        hits = (Hit * max_hits)()
        probes = C.c_int(0)
        n = lib().ko_match(self.h, seq, len(seq), int(partial), hits, max_hits, C.byref(probes))
        return [hits[i] for i in range(n)], probes.value

    def pseudoalign(self, opts: Opts, s1: bytes, s2: bytes | None = None, mean_fl: float = 0.0, has_mean_fl=False):
        out = np.zeros(max(self.num_targets, 1), np.uint32)
        n1, n2 = C.c_int(0), C.c_int(0)
        n = lib().ko_pseudoalign(self.h, C.byref(opts), s1, len(s1), s2 if s2 is not None else None,
                                 len(s2) if s2 is not None else 0, mean_fl, int(has_mean_fl), _p(out, C.c_uint32),
                                 len(out), C.byref(n1), C.byref(n2))
        return out[:n].tolist(), n1.value, n2.value

    def map_pair(self, s1: bytes, s2: bytes) -> int:
        return lib().ko_map_pair(self.h, s1, len(s1), s2, len(s2))

    def close(self):
        if self.h:
            lib().ko_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class ECResult:
    n_processed: int
    ec_off: np.ndarray
    ec_ids: np.ndarray
    counts: np.ndarray
    flens: np.ndarray
    n_probes: int
    n_hits: int

    def multiset(self):
        """{sorted transcript tuple: count} -- EC ids are discovery-order and carry no meaning (SURVEY key fact 3)."""
        return {tuple(self.ec_ids[self.ec_off[i]:self.ec_off[i + 1]].tolist()): int(self.counts[i])
                for i in range(len(self.counts))}


def pack_reads(seqs):
    """list of bytes -> (buffer, offsets, lengths) with NUL terminators (what fetchSequences hands to processBuffer)."""
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    off = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
    buf = b"\0".join(seqs) + b"\0"
    return buf, off, lens


def pack_read_matrix(r1: np.ndarray, r2: np.ndarray | None = None):
    """(n, L) ASCII matrices -> interleaved NUL-terminated buffer without Python loops."""
    n, L = r1.shape
    if r2 is None:
        m = np.zeros((n, L + 1), np.uint8)
        m[:, :L] = r1
        nseq = n
        lens = np.full(nseq, L, np.int32)
        off = (np.arange(nseq, dtype=np.uint64) * np.uint64(L + 1))
    else:
        L2 = r2.shape[1]
        m = np.zeros((n, L + 1 + L2 + 1), np.uint8)
        m[:, :L] = r1
        m[:, L + 1:L + 1 + L2] = r2
        nseq = 2 * n
        lens = np.empty(nseq, np.int32)
        lens[0::2] = L
        lens[1::2] = L2
        off = np.empty(nseq, np.uint64)
        base = np.arange(n, dtype=np.uint64) * np.uint64(L + L2 + 2)
        off[0::2] = base
        off[1::2] = base + np.uint64(L + 1)
    return m.tobytes(), off, lens


def read_fastq(path: str):
    op = gzip.open if path.endswith(".gz") else open
    seqs = []
    with op(path, "rb") as f:
        for i, line in enumerate(f):
            if i % 4 == 1:
                seqs.append(line.rstrip(b"\r\n"))
    return seqs


def process_reads(index: Index, opts: Opts, buf: bytes, off: np.ndarray, lens: np.ndarray) -> ECResult:
    L = lib()
    res = L.ko_result_new()
    try:
        off = np.ascontiguousarray(off, np.uint64)
        lens = np.ascontiguousarray(lens, np.int32)
        n = L.ko_process_reads(index.h, C.byref(opts), buf, _p(off, C.c_uint64), _p(lens, C.c_int32), len(off), res)
        ne, nnz = L.ko_result_num_ecs(res), L.ko_result_nnz(res)
        ec_off = np.zeros(ne + 1, np.uint64)
        ec_ids = np.zeros(max(nnz, 1), np.uint32)
        counts = np.zeros(max(ne, 1), np.uint32)
        L.ko_result_export(res, _p(ec_off, C.c_uint64), _p(ec_ids, C.c_uint32), _p(counts, C.c_uint32))
        flens = np.ctypeslib.as_array(L.ko_result_flens(res), shape=(MAX_FRAG_LEN,)).copy()
        return ECResult(int(n), ec_off, ec_ids[:nnz], counts[:ne], flens, int(L.ko_result_num_probes(res)),
                        int(L.ko_result_num_hits(res)))
    finally:
        L.ko_result_free(res)


def mean_frag_lens_trunc(flens: np.ndarray) -> np.ndarray:
    out = np.zeros(MAX_FRAG_LEN, np.float64)
    fl = np.ascontiguousarray(flens, np.uint32)
    lib().ko_mean_frag_lens_trunc(_p(fl, C.c_uint32), _p(out, C.c_double))
    return out


def trunc_gaussian_fld(mean: float, sd: float) -> np.ndarray:
    out = np.zeros(MAX_FRAG_LEN, np.float64)
    lib().ko_trunc_gaussian_fld(0, MAX_FRAG_LEN, mean, sd, _p(out, C.c_double))
    return out


def eff_lens(target_lens: np.ndarray, mean_fl_trunc: np.ndarray):
    tl = np.ascontiguousarray(target_lens, np.int32)
    means = np.zeros(len(tl), np.float64)
    out = np.zeros(len(tl), np.float64)
    t = np.ascontiguousarray(mean_fl_trunc, np.float64)
    lib().ko_frag_len_means(_p(tl, C.c_int32), len(tl), _p(t, C.c_double), _p(means, C.c_double))
    lib().ko_calc_eff_lens(_p(tl, C.c_int32), len(tl), _p(means, C.c_double), _p(out, C.c_double))
    return out, means


def em_run(ec_off, ec_ids, counts, eff, n_tr: int, weight_counts=None, n_iter=10000, min_rounds=50):
    ec_off = np.ascontiguousarray(ec_off, np.uint64)
    ec_ids = np.ascontiguousarray(ec_ids, np.uint32)
    counts = np.ascontiguousarray(counts, np.uint32)
    wc = counts if weight_counts is None else np.ascontiguousarray(weight_counts, np.uint32)
    eff = np.ascontiguousarray(eff, np.float64)
    alpha = np.zeros(n_tr, np.float64)
    abz = np.zeros(n_tr, np.float64)
    rounds = lib().ko_em_run(_p(ec_off, C.c_uint64), _p(ec_ids, C.c_uint32), _p(counts, C.c_uint32),
                             _p(wc, C.c_uint32), len(counts), _p(eff, C.c_double), n_tr, n_iter, min_rounds,
                             _p(alpha, C.c_double), _p(abz, C.c_double))
    return alpha, abz, rounds


def counts_to_tpm(est, eff):
    est = np.ascontiguousarray(est, np.float64)
    eff = np.ascontiguousarray(eff, np.float64)
    out = np.zeros(len(est), np.float64)
    lib().ko_counts_to_tpm(_p(est, C.c_double), _p(eff, C.c_double), len(est), _p(out, C.c_double))
    return out


def bootstrap_seeds(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint64)
    lib().ko_bootstrap_seeds(seed, n, _p(out, C.c_uint64))
    return out


def multinomial_sample(counts, seed: int) -> np.ndarray:
    counts = np.ascontiguousarray(counts, np.uint32)
    out = np.zeros(len(counts), np.uint32)
    lib().ko_multinomial_sample(_p(counts, C.c_uint32), len(counts), int(seed), _p(out, C.c_uint32))
    return out


# ---- the unmodified reference (oracle/_ref), when it has been built ----------------------------------------------

def ref_available() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "dump_ec")) and os.path.exists(os.path.join(REF_DIR, "kallisto"))


def ref_index(fasta: str, out_idx: str, threads: int = 8, k: int = 31) -> None:
    subprocess.check_call([os.path.join(REF_DIR, "kallisto"), "index", "-t", str(threads), "-k", str(k), "-i", out_idx,
                           fasta], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def ref_dump_quant(idx: str, files, threads: int = 1, extra=(), no_em: bool = False, flens=None):
    """Run the reference through the dump_ec harness; returns dict(nproc, ecs, flens, tr=[(len, eff, alpha, abz)], bs, rounds).
    no_em: stop behind ProcessReads (ECs and flens only).  flens (array of MAX_FRAG_LEN counts): the reference's EM runs on its own
    ECs with this fragment-length sample instead of the one it drew (a multi-threaded run's own sample is schedule-dependent)."""
    import re
    import tempfile
    cmd = [os.path.join(REF_DIR, "dump_ec"), "quant", idx, str(threads), *extra]
    tmp = None
    if no_em:
        cmd.append("--no-em")
    if flens is not None:
        tmp = tempfile.NamedTemporaryFile("w", suffix=".flens", delete=False)
        for i, c in enumerate(np.asarray(flens).tolist()):
            if c:
                tmp.write(f"{i} {c}\n")
        tmp.close()
        cmd += ["--flens", tmp.name]
    try:
        pc = subprocess.run([*cmd, *files], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    finally:
        if tmp is not None:
            os.unlink(tmp.name)
    return parse_ref_dump(pc.stdout.decode(), pc.stderr.decode(errors="replace"))


def parse_ref_dump(out: str, err: str = "") -> dict:
    import re
    res = {"nproc": 0, "ecs": {}, "flens": np.zeros(MAX_FRAG_LEN, np.uint32), "tr": [], "bs": {}, "rounds": None}
    for line in out.splitlines():
        f = line.split()
        if f[0] == "NPROC":
            res["nproc"] = int(f[1])
        elif f[0] == "EC":
            res["ecs"][tuple(int(x) for x in f[1].split(","))] = int(f[2])
        elif f[0] == "FLEN":
            res["flens"][int(f[1])] = int(f[2])
        elif f[0] == "TR":
            res["tr"].append((int(f[2]), float(f[3]), float(f[4]), float(f[5])))
        elif f[0] == "BS":
            res["bs"].setdefault(int(f[1]), []).append(float(f[3]))
    m = re.search(r"Expectation-Maximization algorithm ran for ([0-9,]+) rounds", err)
    if m:
        res["rounds"] = int(m.group(1).replace(",", ""))
    return res


def ref_parity_report(ref: dict, ecs_multiset: dict, flens, eff, est_counts, abz=None) -> dict:
    """Compare one quant result with the unmodified reference's (`ref` = ref_dump_quant(..., threads=1) on the same reads)
    under the tolerances of SURVEY.md section 8(d) / BASELINE.json: EC multiset identical, fragment-length sample identical,
    effective lengths identical, est_counts and TPM within 1e-4 relative for TPM >= 1e-3 (below that: 1e-7 absolute), same
    zero pattern except transcripts whose alpha_before_zeroes is within 1 % of the 1e-8 clamp (EMAlgorithm.h:217-219).
    `ok` requires all of it."""
    r_eff = np.array([t[1] for t in ref["tr"]], np.float64)
    r_alpha = np.array([t[2] for t in ref["tr"]], np.float64)
    r_abz = np.array([t[3] for t in ref["tr"]], np.float64)
    eff = np.asarray(eff, np.float64)
    est = np.asarray(est_counts, np.float64)

    def tpm_of(a, e):
        w = a / e
        s = w.sum()
        return w / s * 1e6 if s > 0 else w
    r_tpm, g_tpm = tpm_of(r_alpha, r_eff), tpm_of(est, eff)
    big = r_tpm >= 1e-3
    rel_cnt = float(np.max(np.abs(est[big] - r_alpha[big]) / r_alpha[big])) if big.any() else 0.0
    rel_tpm = float(np.max(np.abs(g_tpm[big] - r_tpm[big]) / r_tpm[big])) if big.any() else 0.0
    abs_small = float(np.max(np.abs(g_tpm[~big] - r_tpm[~big]))) if (~big).any() else 0.0
    near_clamp = np.abs(r_abz - 1e-8) <= 1e-10
    zero_ok = bool(np.array_equal((est == 0)[~near_clamp], (r_alpha == 0)[~near_clamp]))
    rep = {
        "reference": "oracle/_ref/dump_ec quant -t 1 (unmodified reference: ProcessReads -> FLD -> EMAlgorithm::run)",
        "n_processed_ref": int(ref["nproc"]),
        "n_ecs_ref": len(ref["ecs"]), "n_ecs": len(ecs_multiset),
        "ec_multiset_equal": bool(ecs_multiset == ref["ecs"]),
        "flens_equal": bool(np.array_equal(np.asarray(flens, np.uint32), ref["flens"])),
        "eff_length_equal": bool(np.array_equal(eff, r_eff)),
        "est_counts_max_rel_err_tpm_ge_1e-3": rel_cnt,
        "tpm_max_rel_err_tpm_ge_1e-3": rel_tpm,
        "tpm_max_abs_err_below_floor": abs_small,
        "zero_pattern_equal": zero_ok,
        "tolerance": "EC multiset / flens / eff_length identical; est_counts and tpm <= 1e-4 relative for tpm >= 1e-3, "
                     "<= 1e-7 absolute below; same zero pattern",
    }
    rep["ok"] = bool(rep["ec_multiset_equal"] and rep["flens_equal"] and rep["eff_length_equal"] and rel_cnt <= 1e-4 and
                     rel_tpm <= 1e-4 and abs_small <= 1e-7 and zero_ok)
    return rep
