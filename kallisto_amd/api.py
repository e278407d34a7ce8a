"""ctypes mirror of include/kallisto_amd.h plus the `quant` driver flow of the reference (src/main.cpp:2620-2798).

Names follow the reference: an *index* (KmerIndex), *pseudoalignment* of read batches into equivalence-class (EC)
counts (ProcessReads / MinCollector), the fragment-length distribution (FLD), effective lengths, the EM
(EMAlgorithm::run) and TPM.  Everything that computes goes through libkallisto_amd.so; if the library or a GPU is
missing the calls raise KallistoAmdError -- there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass, field

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_FRAG_LEN = 1000


class KallistoAmdError(RuntimeError):
    pass


def library_path() -> str:
    return os.path.join(HERE, "libkallisto_amd.so")


class _View(C.Structure):
    _fields_ = ([("k", C.c_int32)] +
                [(n, C.c_uint64) for n in ("n_kmers", "n_unitigs", "n_blocks", "n_uec", "n_ecs", "ec_nnz", "n_targets",
                                           "dlist_size", "n_buckets", "pad_buckets")] +
                [(n, C.c_void_p) for n in ("table", "slot_block", "slot_dist", "uec_ec", "ec_off", "ec_ids",
                                           "unitig_blk_off", "unitig_len", "blk_unitig", "blk_lb", "blk_ub", "blk_ec",
                                           "blk_pos_off", "blk_posw", "blk_sense", "target_lens", "onlist_bits")] +
                [("onlist_words", C.c_uint64), ("dtable", C.c_void_p), ("n_dbuckets", C.c_uint64), ("dpad_buckets", C.c_uint64),
                 ("dummy_slot", C.c_uint64), ("dummy_uec", C.c_uint32), ("dummy_strand", C.c_uint32),
                 ("utext", C.c_void_p), ("utext_words", C.c_uint64), ("text_bases", C.c_uint64), ("unitig_gpos", C.c_void_p)] +
                [(n, C.c_uint32) for n in ("table_layout", "slots_per_bucket", "tag_q", "tag_dsh", "tag_w")])


class QuantOpts(C.Structure):
    """kamd_quant_opts: the subset of ProgramOptions (src/common.h:93-209) the hot path reads."""
    _fields_ = [("paired", C.c_int32), ("fld", C.c_double), ("sd", C.c_double), ("single_overhang", C.c_int32),
                ("strand", C.c_int32), ("no_jump", C.c_int32), ("do_union", C.c_int32)]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_processed", "n_single", "n_multi", "n_probes", "n_bucket_reads",
                                          "n_distinct_tuples", "n_stream_words", "n_raw_words", "n_text_hits", "n_wave_iters",
                                          "n_lane_iters")]


class Tuning(C.Structure):
    """kamd_tuning: which of the equivalent kernels / EM forms run (0 = keep the current value; on/off fields: 1 on, 2 off)."""
    _fields_ = [(n, C.c_int32) for n in ("text_verify", "items_per_wave", "refill_min", "lds_pad", "em_form", "em_entries_per_lane", "em_windowed", "em_graph", "em_row_lanes",
                                         "em_fin_blocks", "em_local_block", "em_group_div", "em_split_len", "dedup_form", "align_chunks", "em_small_nnz", "em_reg_slices",
                                         "em_hybrid", "overflow_second_pass", "em_giant_nnz", "em_blocked")]


EM_FORMS = {"streamed": 1, "csr": 2, "local": 3}
ABI_VERSION = 6   # KAMD_ABI_VERSION of include/kallisto_amd.h


class _Profile(C.Structure):
    _fields_ = [("last_align_kernel_ms", C.c_float), ("last_em_ms", C.c_float), ("last_em_iters", C.c_uint64),
                ("last_classify_ms", C.c_float), ("kernel_a_version", C.c_int32), ("last_em_nnz", C.c_uint64),
                ("last_em_nnz_multi", C.c_uint64), ("last_em_nseg", C.c_uint64), ("last_em_necs", C.c_uint64),
                ("last_em_k", C.c_int32), ("last_em_grid", C.c_uint32), ("last_em_lds", C.c_uint32),
                ("last_em_plan_cached", C.c_int32), ("last_finalize_ms", C.c_float), ("last_fin_records", C.c_uint64),
                ("last_fin_stream_words", C.c_uint64), ("last_fin_cand_words", C.c_uint64), ("absorb_ms", C.c_float),
                ("n_distinct_tuples", C.c_uint64), ("tuple_store_words", C.c_uint64), ("tuple_table_slots", C.c_uint64),
                ("last_em_max_comp_nnz", C.c_uint64), ("last_em_giant_nnz", C.c_uint64), ("last_em_giant_rows", C.c_uint64),
                ("last_em_giant_tr", C.c_uint64), ("last_em_giant_chunks", C.c_uint32), ("last_em_graph_fallback", C.c_int32),
                ("last_em_plan_ms", C.c_float), ("n_overflow_items", C.c_uint64), ("overflow_ms", C.c_float),
                ("last_merge_ms", C.c_float), ("em_collective_ms", C.c_float), ("em_collectives", C.c_uint32), ("n_overflow_second_pass", C.c_uint64), ("last_em_giant_pieces", C.c_uint64)]


class _FastqUnit(C.Structure):
    _fields_ = [("d_words", C.c_void_p), ("d_len", C.c_void_p), ("n_items", C.c_uint64), ("max_len", C.c_int32), ("status", C.c_int32),
                ("first_bad_record", C.c_uint64)]


class _Batch(C.Structure):
    _fields_ = [("d_words", C.c_void_p), ("d_len", C.c_void_p), ("n_items", C.c_uint64), ("max_len", C.c_int32)]


class _QuantOut(C.Structure):
    _fields_ = [("n_processed", C.c_uint64), ("em_rounds", C.c_int32), ("flens", C.c_void_p), ("eff_lens", C.c_void_p), ("est_counts", C.c_void_p),
                ("alpha_before_zeroes", C.c_void_p), ("tpm", C.c_void_p)]


class _EcResult(C.Structure):
    _fields_ = [("n_ecs", C.c_uint64), ("nnz", C.c_uint64), ("n_pseudoaligned", C.c_uint64), ("d_ec_off", C.c_void_p),
                ("d_ec_ids", C.c_void_p), ("d_counts", C.c_void_p)]


_LIB = None

_SYMBOLS = {
    # name: (restype, argtypes)
    "kamd_last_error": (C.c_char_p, []),
    "kamd_abi_version": (C.c_uint32, []),
    "kamd_index_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "kamd_index_load_layout": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p)]),
    "kamd_index_free": (None, [C.c_void_p]),
    "kamd_index_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "kamd_index_get_view": (C.c_int, [C.c_void_p, C.POINTER(_View)]),
    "kamd_index_target_name": (C.c_char_p, [C.c_void_p, C.c_uint64]),
    "kamd_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "kamd_ctx_destroy": (None, [C.c_void_p]),
    "kamd_ctx_tune": (C.c_int, [C.c_void_p, C.POINTER(Tuning)]),
    "kamd_ctx_get_tuning": (C.c_int, [C.c_void_p, C.POINTER(Tuning)]),
    "kamd_index_upload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kamd_ec_reset": (C.c_int, [C.c_void_p]),
    "kamd_ec_track_order": (C.c_int, [C.c_void_p, C.c_int]),
    "kamd_packed_record_words": (C.c_uint64, [C.c_int32]),
    "kamd_pack_reads_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "kamd_pack_reads_host_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.c_uint64, C.c_uint64]),
    "kamd_pack_reads_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p,
                                         C.c_void_p]),
    "kamd_fastq_unit_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(_FastqUnit)]),
    "kamd_fastq_unit_parse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(_FastqUnit)]),
    "kamd_fastq_batch_pack": (C.c_int, [C.c_void_p, C.POINTER(_FastqUnit)]),
    "kamd_pseudoalign": (C.c_int, [C.c_void_p, C.POINTER(QuantOpts), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    "kamd_fld_prefetch": (C.c_int, [C.c_void_p, C.POINTER(QuantOpts), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    "kamd_fld_from_batch": (C.c_int, [C.c_void_p, C.POINTER(QuantOpts), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p,
                                      C.POINTER(C.c_uint64)]),
    "kamd_align_stats_get": (C.c_int, [C.c_void_p, C.POINTER(_Stats)]),
    "kamd_profile_get": (C.c_int, [C.c_void_p, C.POINTER(_Profile)]),
    "kamd_debug_random_lines": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "kamd_debug_random_lines_span": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double)]),
    "kamd_ec_dense_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "kamd_ec_tuples_export": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kamd_ec_tuples_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kamd_ec_tuples_replace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "kamd_ec_explicit_export": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kamd_ec_explicit_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kamd_ec_explicit_replace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "kamd_flat_index_matches": (C.c_int, [C.c_char_p, C.c_char_p]),
    "kamd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "kamd_comm_create_rccl": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "kamd_comm_create_callbacks": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "kamd_comm_destroy": (None, [C.c_void_p]),
    "kamd_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "kamd_ec_allreduce": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kamd_comm_broadcast_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]),
    "kamd_comm_sum_u64_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kamd_em_run_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_int32)]),
    "kamd_ec_finalize": (C.c_int, [C.c_void_p, C.POINTER(_EcResult)]),
    "kamd_ec_finalize_result": (C.c_int, [C.c_void_p, C.POINTER(_EcResult)]),
    "kamd_ec_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kamd_ec_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kamd_ec_set_counts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kamd_em_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                              C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "kamd_em_run_partitioned": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "kamd_bootstrap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                 C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "kamd_bootstrap_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "kamd_bootstrap_seeds": (None, [C.c_uint64, C.c_int32, C.c_void_p]),
    "kamd_quant_batches": (C.c_int, [C.c_void_p, C.POINTER(QuantOpts), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(_QuantOut)]),
    "kamd_mean_frag_lens_trunc": (None, [C.c_void_p, C.c_void_p]),
    "kamd_trunc_gaussian_fld": (None, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p]),
    "kamd_eff_lens": (None, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "kamd_counts_to_tpm": (None, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
}


def exported_symbols():
    """Every entry point include/kallisto_amd.h declares (used by the symbol-presence test)."""
    return sorted(_SYMBOLS)


def _load_hip_runtime():
    """libkallisto_amd.so carries no DT_NEEDED for the HIP runtime (see csrc/Makefile): bind it to the runtime this
    process uses -- PyTorch's bundled libamdhip64.so when torch is importable, else the ROCm installation's."""
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    raise KallistoAmdError("no HIP runtime (libamdhip64.so) found")


def load_library():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise KallistoAmdError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(hipcc --offload-arch=gfx950); kallisto_amd has no CPU path")
        _load_hip_runtime()
        lib = C.CDLL(path)
        lib.kamd_abi_version.restype = C.c_uint32
        if lib.kamd_abi_version() != ABI_VERSION:   # (the structures of include/kallisto_amd.h are mirrored here field by field)
            raise KallistoAmdError(f"{path} was built with KAMD_ABI_VERSION {lib.kamd_abi_version()}, this binding mirrors version {ABI_VERSION}: rebuild the library")
        for name, (res, args) in _SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def _check(rc: int, what: str):
    if rc != 0:
        raise KallistoAmdError(f"{what} failed ({rc}): {load_library().kamd_last_error().decode(errors='replace')}")


def packed_record_words(max_len: int) -> int:
    return int(load_library().kamd_packed_record_words(max_len))


def _np(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=int(n))


class Index:
    """Flattened kallisto index (format v13).  Mirrors KmerIndex::load (src/KmerIndex.cpp:1330)."""

    TABLE_LAYOUTS = {"wide": 0, "compact": 1, "auto": 2}

    def __init__(self, path: str, threads: int = 0, table_layout: str | None = None, table_load: float = 0.0):
        """table_layout: None = kamd_index_load (the environment's KAMD_TABLE_LAYOUT, default auto: compact when it fits); "wide" / "compact" / "auto" =
        kamd_index_load_layout with that layout of the k-mer table (and table_load as the compact table's load factor, 0 = the library's choice)."""
        lib = load_library()
        self._h = C.c_void_p()
        if table_layout is None:
            _check(lib.kamd_index_load(os.fsencode(path), threads, C.byref(self._h)), "kamd_index_load")
        else:
            _check(lib.kamd_index_load_layout(os.fsencode(path), threads, self.TABLE_LAYOUTS[table_layout], float(table_load), C.byref(self._h)),
                   "kamd_index_load_layout")
        self.view = _View()
        _check(lib.kamd_index_get_view(self._h, C.byref(self.view)), "kamd_index_get_view")
        v = self.view
        self.k = v.k
        self.num_kmers, self.num_unitigs, self.num_blocks = v.n_kmers, v.n_unitigs, v.n_blocks
        self.num_ecs, self.num_targets = v.n_ecs, v.n_targets
        self.target_lens = _np(v.target_lens, v.n_targets, np.int32).copy()

    @property
    def handle(self):
        return self._h

    def target_names(self):
        lib = load_library()
        return [lib.kamd_index_target_name(self._h, i).decode() for i in range(self.num_targets)]

    def ec_sets(self):
        """(ec_off, ec_ids) of the de-duplicated index transcript sets (host views)."""
        v = self.view
        return _np(v.ec_off, v.n_ecs + 1, np.uint64), _np(v.ec_ids, v.ec_nnz, np.uint32)

    def save(self, path: str):
        """kamd_index_save: the flattened tables as a file that Index(path) / kamd_index_load reads back without rebuilding them."""
        _check(load_library().kamd_index_save(self._h, path.encode()), "kamd_index_save")

    def close(self):
        if getattr(self, "_h", None):
            load_library().kamd_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class ECs:
    """EC multiset {sorted transcript set -> count} as CSR (index.ecmapinv x MinCollector::counts)."""
    ec_off: np.ndarray
    ec_ids: np.ndarray
    counts: np.ndarray

    def multiset(self):
        return {tuple(self.ec_ids[self.ec_off[i]:self.ec_off[i + 1]].tolist()): int(self.counts[i])
                for i in range(len(self.counts))}


class Context:
    """One GPU: device copy of the index and the EC state.  All work runs on torch's current stream of `device`."""

    def __init__(self, device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise KallistoAmdError("no GPU visible: kallisto_amd has no CPU path")
        self.torch = torch
        self.device = device
        torch.cuda.set_device(device)
        self.stream = torch.cuda.current_stream(device)
        self._h = C.c_void_p()
        _check(load_library().kamd_ctx_create(device, C.c_void_p(self.stream.cuda_stream), C.byref(self._h)), "kamd_ctx_create")
        self.index = None

    def upload(self, index: Index):
        _check(load_library().kamd_index_upload(self._h, index.handle), "kamd_index_upload")
        self.index = index

    def reset(self):
        _check(load_library().kamd_ec_reset(self._h), "kamd_ec_reset")

    def tune(self, **kw):
        """kamd_ctx_tune: e.g. tune(text_verify=False), tune(em_form="streamed", em_entries_per_lane=16).  Returns the tuning in force."""
        t = Tuning()
        for k, v in kw.items():
            if k == "em_form" and isinstance(v, str):
                v = EM_FORMS[v]
            if isinstance(v, bool):
                v = 1 if v else 2
            setattr(t, k, int(v))
        _check(load_library().kamd_ctx_tune(self._h, C.byref(t)), "kamd_ctx_tune")
        cur = Tuning()
        _check(load_library().kamd_ctx_get_tuning(self._h, C.byref(cur)), "kamd_ctx_get_tuning")
        return {n: int(getattr(cur, n)) for n, _ in Tuning._fields_}

    def track_order(self, on: bool = True):
        """finalize() then emits the sets in first-occurrence order (the reference's ids at -t 1); call before the first batch."""
        _check(load_library().kamd_ec_track_order(self._h, 1 if on else 0), "kamd_ec_track_order")

    # ---- reads ----
    def pack_reads(self, seqs_u8, max_len: int | None = None):
        """(n_reads, L) uint8 ASCII tensor on the device -> (words, lens) packed records (kamd_pack_reads_device)."""
        torch = self.torch
        assert seqs_u8.dtype == torch.uint8 and seqs_u8.is_cuda and seqs_u8.dim() == 2
        seqs_u8 = seqs_u8.contiguous()
        n, L = seqs_u8.shape
        max_len = max_len or L
        rec = packed_record_words(max_len)
        off = torch.arange(n, device=seqs_u8.device, dtype=torch.int64) * L
        ln = torch.full((n,), L, device=seqs_u8.device, dtype=torch.int32)
        words = torch.empty(n * rec, device=seqs_u8.device, dtype=torch.int32)
        lens = torch.empty(n, device=seqs_u8.device, dtype=torch.int16)
        _check(load_library().kamd_pack_reads_device(self._h, seqs_u8.data_ptr(), off.data_ptr(), ln.data_ptr(), n, max_len,
                                                     words.data_ptr(), lens.data_ptr()), "kamd_pack_reads_device")
        return words, lens

    def pack_reads_host(self, seqs, max_len: int | None = None):
        """list of bytes -> packed device tensors through the host packer (kamd_pack_reads_host)."""
        torch = self.torch
        n = len(seqs)
        lens = np.array([len(s) for s in seqs], np.int32)
        max_len = max_len or int(lens.max(initial=1))
        off = np.zeros(n, np.uint64)
        if n:
            off[1:] = np.cumsum(lens[:-1].astype(np.uint64))
        buf = b"".join(seqs)
        rec = packed_record_words(max_len)
        words = np.zeros(n * rec, np.uint32)
        l16 = np.zeros(n, np.uint16)
        _check(load_library().kamd_pack_reads_host(buf, off.ctypes.data, lens.ctypes.data, n, max_len, words.ctypes.data,
                                                   l16.ctypes.data), "kamd_pack_reads_host")
        dw = torch.from_numpy(words.view(np.int32)).to(f"cuda:{self.device}")
        dl = torch.from_numpy(l16.view(np.int16)).to(f"cuda:{self.device}")
        return dw, dl, max_len

    def _fastq_texts(self, texts):
        torch = self.torch
        dev = f"cuda:{self.device}"
        keep = []
        for t in texts:
            if isinstance(t, (bytes, bytearray)):
                pad = torch.zeros(len(t) + 64, dtype=torch.uint8)
                pad[:len(t)] = torch.frombuffer(bytearray(t), dtype=torch.uint8)
                keep.append((pad.to(dev), len(t)))
            else:
                assert t.dtype == torch.uint8 and t.is_cuda
                # the ABI wants the text readable up to 32 bytes behind n_bytes (k_fq_pack loads whole groups of 32 bases).  A tensor that is a
                # view of a larger storage with that much behind it -- a unit cut out of a device text buffer -- is passed as it is; only an
                # exactly-sized one is copied into a padded tensor rather than read past its allocation (the copy was unconditional,
                # a device-to-device copy of every text unit)
                n = t.numel()
                flat = t.reshape(-1)
                slack = flat.untyped_storage().nbytes() - (flat.storage_offset() + n) if flat.is_contiguous() else -1
                if slack >= 32 and flat.data_ptr() % 16 == 0:
                    keep.append((flat, n))
                else:
                    pad = torch.zeros(n + 64, dtype=torch.uint8, device=t.device)
                    pad[:n] = flat
                    keep.append((pad, n))
        ptrs = (C.c_void_p * 2)(*[k[0].data_ptr() for k in keep], *([None] * (2 - len(keep))))
        nb = (C.c_uint64 * 2)(*[k[1] for k in keep], *([0] * (2 - len(keep))))
        return keep, ptrs, nb

    def _fastq_result(self, u, n_files):
        torch = self.torch
        words = lens = None
        if u.status == 0 and u.n_items and u.d_words:
            n_reads = int(u.n_items) * n_files
            words = _alias_tensor(torch, u.d_words, n_reads * packed_record_words(int(u.max_len)), torch.int32, self.device)
            lens = _alias_tensor(torch, u.d_len, n_reads, torch.int16, self.device)
        return words, lens, int(u.n_items), int(u.max_len), int(u.status), int(u.first_bad_record)

    def fastq_unit_pack(self, texts, n_records: int):
        """kamd_fastq_unit_pack: one unit of strict 4-line FASTQ text per file (bytes, or uint8 device tensors) -> the packed batch
        (views of the context's buffers, valid until the next call).  Returns (words, lens, n_items, max_len, status, first_bad)."""
        keep, ptrs, nb = self._fastq_texts(texts)
        u = _FastqUnit()
        _check(load_library().kamd_fastq_unit_pack(self._h, ptrs, nb, len(keep), int(n_records), C.byref(u)), "kamd_fastq_unit_pack")
        return self._fastq_result(u, len(keep))

    def fastq_unit_parse(self, texts, n_records: int):
        """kamd_fastq_unit_parse: adds a unit to the batch under construction; the device copies of the texts are kept alive until
        fastq_batch_pack.  Returns (status, first_bad, max_len)."""
        keep, ptrs, nb = self._fastq_texts(texts)
        u = _FastqUnit()
        _check(load_library().kamd_fastq_unit_parse(self._h, ptrs, nb, len(keep), int(n_records), C.byref(u)), "kamd_fastq_unit_parse")
        self._fq_keep = getattr(self, "_fq_keep", []) + [keep]
        self._fq_files = len(keep)
        return int(u.status), int(u.first_bad_record), int(u.max_len)

    def fastq_batch_pack(self):
        u = _FastqUnit()
        _check(load_library().kamd_fastq_batch_pack(self._h, C.byref(u)), "kamd_fastq_batch_pack")
        self.torch.cuda.synchronize(self.device)
        self._fq_keep = []
        return self._fastq_result(u, getattr(self, "_fq_files", 1))

    # ---- pseudoalignment ----
    def pseudoalign(self, opts: QuantOpts, words, lens, n_items: int, max_len: int):
        _check(load_library().kamd_pseudoalign(self._h, C.byref(opts), words.data_ptr(), lens.data_ptr(), n_items, max_len),
               "kamd_pseudoalign")

    def fld_prefetch(self, opts: QuantOpts, words, lens, n_items: int, max_len: int):
        """Start the FLD kernel for the first prefix of a batch on a side stream (kamd_fld_prefetch); fld_from_batch on the same
        batch then only waits for it."""
        _check(load_library().kamd_fld_prefetch(self._h, C.byref(opts), words.data_ptr(), lens.data_ptr(), n_items, max_len),
               "kamd_fld_prefetch")

    def fld_from_batch(self, opts: QuantOpts, words, lens, n_items: int, max_len: int, flens=None, used: int = 0):
        """Fragment-length sample of the first 10000 qualifying pairs in input order.  `flens` / `used` continue a sample
        started on earlier batches (the reference carries tlencount across batches, src/ProcessReads.cpp:981-1008)."""
        if flens is None:
            flens = np.zeros(MAX_FRAG_LEN, np.uint32)
        used = C.c_uint64(used)
        _check(load_library().kamd_fld_from_batch(self._h, C.byref(opts), words.data_ptr(), lens.data_ptr(), n_items, max_len,
                                                  flens.ctypes.data, C.byref(used)), "kamd_fld_from_batch")
        return flens, int(used.value)

    def stats(self) -> dict:
        s = _Stats()
        _check(load_library().kamd_align_stats_get(self._h, C.byref(s)), "kamd_align_stats_get")
        return {n: int(getattr(s, n)) for n, _ in _Stats._fields_}

    def random_lines(self, n_blocks: int, block_threads: int = 256, iters: int = 256, span_mb: int = 0, access_bytes: int = 64):
        """Diagnostic: (GB/s, M lines/s) of dependent random reads of the k-mer table (span_mb: only its first MiB; access_bytes 64 / 8)."""
        g, m = C.c_double(0), C.c_double(0)
        _check(load_library().kamd_debug_random_lines_span(self._h, n_blocks, block_threads, iters, span_mb, access_bytes, C.byref(g), C.byref(m)),
               "kamd_debug_random_lines_span")
        return g.value, m.value

    def profile(self) -> dict:
        p = _Profile()
        _check(load_library().kamd_profile_get(self._h, C.byref(p)), "kamd_profile_get")
        return {"align_kernel_ms": float(p.last_align_kernel_ms), "em_ms": float(p.last_em_ms), "em_iters": int(p.last_em_iters),
                "classify_ms": float(p.last_classify_ms), "kernel_a_version": int(p.kernel_a_version),
                "em_nnz": int(p.last_em_nnz), "em_nnz_multi": int(p.last_em_nnz_multi), "em_nseg": int(p.last_em_nseg),
                "em_necs": int(p.last_em_necs), "em_k": int(p.last_em_k), "em_grid": int(p.last_em_grid), "em_lds": int(p.last_em_lds),
                "em_plan_cached": int(p.last_em_plan_cached), "finalize_ms": float(p.last_finalize_ms),
                "fin_records": int(p.last_fin_records), "fin_stream_words": int(p.last_fin_stream_words),
                "fin_cand_words": int(p.last_fin_cand_words), "absorb_ms": float(p.absorb_ms), "n_distinct_tuples": int(p.n_distinct_tuples),
                "tuple_store_words": int(p.tuple_store_words), "tuple_table_slots": int(p.tuple_table_slots),
                "em_max_comp_nnz": int(p.last_em_max_comp_nnz), "em_giant_nnz": int(p.last_em_giant_nnz), "em_giant_rows": int(p.last_em_giant_rows),
                "em_giant_tr": int(p.last_em_giant_tr), "em_giant_chunks": int(p.last_em_giant_chunks), "em_graph_fallback": int(p.last_em_graph_fallback), "em_giant_pieces": int(p.last_em_giant_pieces),
                "em_plan_ms": float(p.last_em_plan_ms), "n_overflow_items": int(p.n_overflow_items), "overflow_ms": float(p.overflow_ms),
                "merge_ms": float(p.last_merge_ms), "em_collective_ms": float(p.em_collective_ms), "em_collectives": int(p.em_collectives), "n_overflow_second_pass": int(p.n_overflow_second_pass)}

    # ---- multi-GPU exchange: all-reduce of the dense EC count vector + all-gather of the tuple records ----
    def dense_counts(self):
        """The dense per-index-set count vector as a torch tensor aliasing the context's device memory."""
        torch = self.torch
        p, n = C.c_void_p(), C.c_uint64(0)
        _check(load_library().kamd_ec_dense_counts(self._h, C.byref(p), C.byref(n)), "kamd_ec_dense_counts")
        return _alias_tensor(torch, p.value, int(n.value), torch.int32, self.device)

    def tuples_export(self):
        torch = self.torch
        nw, nt = C.c_uint64(0), C.c_uint64(0)
        _check(load_library().kamd_ec_tuples_export(self._h, C.byref(nw), C.byref(nt)), "kamd_ec_tuples_export")
        words = torch.zeros(max(int(nw.value), 1), dtype=torch.int32, device=f"cuda:{self.device}")
        offs = torch.zeros(max(int(nt.value), 1), dtype=torch.int64, device=f"cuda:{self.device}")
        _check(load_library().kamd_ec_tuples_copy(self._h, words.data_ptr(), offs.data_ptr()), "kamd_ec_tuples_copy")
        return words[:int(nw.value)], offs[:int(nt.value)]

    def tuples_replace(self, words, offs):
        _check(load_library().kamd_ec_tuples_replace(self._h, words.data_ptr() if words.numel() else None, words.numel(),
                                                     offs.data_ptr() if offs.numel() else None, offs.numel()),
               "kamd_ec_tuples_replace")

    def explicit_export(self):
        torch = self.torch
        nw, nr = C.c_uint64(0), C.c_uint64(0)
        _check(load_library().kamd_ec_explicit_export(self._h, C.byref(nw), C.byref(nr)), "kamd_ec_explicit_export")
        words = torch.zeros(max(int(nw.value), 1), dtype=torch.int32, device=f"cuda:{self.device}")
        offs = torch.zeros(max(int(nr.value), 1), dtype=torch.int64, device=f"cuda:{self.device}")
        _check(load_library().kamd_ec_explicit_copy(self._h, words.data_ptr(), offs.data_ptr()), "kamd_ec_explicit_copy")
        return words[:int(nw.value)], offs[:int(nr.value)]

    def explicit_replace(self, words, offs):
        _check(load_library().kamd_ec_explicit_replace(self._h, words.data_ptr() if words.numel() else None, words.numel(),
                                                       offs.data_ptr() if offs.numel() else None, offs.numel()),
               "kamd_ec_explicit_replace")

    def ec_allreduce(self, comm: "Comm"):
        """kamd_ec_allreduce: merge the EC state of all ranks inside the library (one all-reduce of the dense count vector +
        all-gathers of the tuple / explicit-set records, RCCL over xGMI)."""
        _check(load_library().kamd_ec_allreduce(self._h, comm._h), "kamd_ec_allreduce")

    def em_run_comm(self, comm: "Comm", eff_lens: np.ndarray, n_iter: int = 10000, min_rounds: int = 50):
        """kamd_em_run_comm: the EM partitioned over the communicator's ranks by connected component; every rank returns the
        same (alpha, alpha_before_zeroes, rounds)."""
        eff = np.ascontiguousarray(eff_lens, np.float64)
        T = len(eff)
        alpha = np.zeros(T, np.float64)
        abz = np.zeros(T, np.float64)
        rounds = C.c_int32(0)
        _check(load_library().kamd_em_run_comm(self._h, comm._h, eff.ctypes.data, T, n_iter, min_rounds, alpha.ctypes.data,
                                               abz.ctypes.data, C.byref(rounds)), "kamd_em_run_comm")
        return alpha, abz, int(rounds.value)

    # ---- finalize / EM ----
    def finalize(self, download: bool = True):
        res = _EcResult()
        _check(load_library().kamd_ec_finalize(self._h, C.byref(res)), "kamd_ec_finalize")
        self.ec_result = res
        if not download:
            return None
        return self.download_ecs()

    def download_ecs(self):
        """kamd_ec_download of the finalized result (self.ec_result)"""
        res = self.ec_result
        ec_off = np.zeros(res.n_ecs + 1, np.uint64)
        ec_ids = np.zeros(max(res.nnz, 1), np.uint32)
        counts = np.zeros(max(res.n_ecs, 1), np.uint32)
        _check(load_library().kamd_ec_download(self._h, ec_off.ctypes.data, ec_ids.ctypes.data, counts.ctypes.data),
               "kamd_ec_download")
        return ECs(ec_off, ec_ids[:res.nnz], counts[:res.n_ecs])

    def ec_upload(self, ec_off, ec_ids, counts=None):
        """quant-tcc: the ECs of a file become the context's EC result (KmerIndex::loadECsFromFile + the TCC counts of one sample)."""
        off = np.ascontiguousarray(ec_off, np.uint64)
        ids = np.ascontiguousarray(ec_ids, np.uint32)
        cnt = None if counts is None else np.ascontiguousarray(counts, np.uint32)
        _check(load_library().kamd_ec_upload(self._h, off.ctypes.data, ids.ctypes.data if ids.size else None,
                                             None if cnt is None else cnt.ctypes.data, len(off) - 1), "kamd_ec_upload")

    def ec_set_counts(self, counts):
        """Another sample's counts on the same EC matrix (the EM plan is reused)."""
        cnt = np.ascontiguousarray(counts, np.uint32)
        _check(load_library().kamd_ec_set_counts(self._h, cnt.ctypes.data), "kamd_ec_set_counts")

    def em_run(self, eff_lens: np.ndarray, n_iter: int = 10000, min_rounds: int = 50, csr=None):
        """EMAlgorithm(counts, ...).run(n_iter, min_rounds) on the finalized ECs (or on a device CSR triple)."""
        eff = np.ascontiguousarray(eff_lens, np.float64)
        T = len(eff)
        alpha = np.zeros(T, np.float64)
        abz = np.zeros(T, np.float64)
        rounds = C.c_int32(0)
        if csr is None:
            args = (None, None, None, None, 0)
        else:
            off, ids, cnt = csr
            args = (off.data_ptr(), ids.data_ptr(), cnt.data_ptr(), None, cnt.numel())
        _check(load_library().kamd_em_run(self._h, *args, eff.ctypes.data, T, n_iter, min_rounds, alpha.ctypes.data,
                                          abz.ctypes.data, C.byref(rounds)), "kamd_em_run")
        return alpha, abz, int(rounds.value)

    def em_run_partitioned(self, eff_lens: np.ndarray, group=None, n_iter: int = 10000, min_rounds: int = 50):
        """The EM over all ranks of `group`: every rank (holding the same finalized ECs) runs the connected components it
        owns; the only collective inside is the sum of the per-round change counters once per chunk of rounds, plus
        one all-reduce of the result.  Returns the same (alpha, alpha_before_zeroes, rounds) on every rank."""
        import torch.distributed as dist
        torch = self.torch
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        eff = np.ascontiguousarray(eff_lens, np.float64)
        T = len(eff)
        alpha = np.zeros(T, np.float64)
        abz = np.zeros(T, np.float64)
        rounds = C.c_int32(0)
        device = self.device

        def _sum(user, d_counts, n):
            try:
                t = _alias_tensor(torch, d_counts, int(n), torch.int32, device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                torch.cuda.synchronize(device)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32)(_sum)
        _check(load_library().kamd_em_run_partitioned(self._h, rank, world, cb, None, eff.ctypes.data, T, n_iter, min_rounds,
                                                      alpha.ctypes.data, abz.ctypes.data, C.byref(rounds)), "kamd_em_run_partitioned")
        both = torch.from_numpy(np.stack([alpha, abz])).to(f"cuda:{device}")
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)  # every transcript is non-zero on exactly one rank
        both = both.cpu().numpy()
        return both[0].copy(), both[1].copy(), int(rounds.value)

    def bootstrap(self, seed: int, eff_lens: np.ndarray, csr=None, want_sample: bool = False):
        """One bootstrap replicate (Bootstrap::run_em): multinomial resample of the EC counts + EM.  Returns
        (alpha, rounds[, resampled counts])."""
        eff = np.ascontiguousarray(eff_lens, np.float64)
        T = len(eff)
        alpha = np.zeros(T, np.float64)
        rounds = C.c_int32(0)
        if csr is None:
            args, n = (None, None, None, 0), int(self.ec_result.n_ecs)
        else:
            off, ids, cnt = csr
            args, n = (off.data_ptr(), ids.data_ptr(), cnt.data_ptr(), cnt.numel()), cnt.numel()
        samp = np.zeros(max(n, 1), np.uint32) if want_sample else None
        _check(load_library().kamd_bootstrap(self._h, *args, int(seed), eff.ctypes.data, T, alpha.ctypes.data, C.byref(rounds),
                                             samp.ctypes.data if want_sample else None), "kamd_bootstrap")
        return (alpha, int(rounds.value), samp[:n]) if want_sample else (alpha, int(rounds.value))

    def bootstrap_batch(self, seeds, eff_lens: np.ndarray):
        """kamd_bootstrap_batch: len(seeds) replicates on the finalized ECs -> (alpha [n_rep, T], rounds [n_rep])."""
        eff = np.ascontiguousarray(eff_lens, np.float64)
        sd = np.ascontiguousarray(seeds, np.uint64)
        T, n = len(eff), len(sd)
        alpha = np.zeros((n, T), np.float64)
        rounds = np.zeros(n, np.int32)
        _check(load_library().kamd_bootstrap_batch(self._h, sd.ctypes.data, n, eff.ctypes.data, T, alpha.ctypes.data, rounds.ctypes.data),
               "kamd_bootstrap_batch")
        return alpha, rounds

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def close(self):
        # a communicator cached by quant() holds this context's device and stream: it goes first (kamd_comm_destroy
        # dereferences the context)
        comm = getattr(self, "_comm", None)
        if comm is not None:
            comm.close()
            self._comm = None
        if getattr(self, "_h", None):
            load_library().kamd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CommCallbacks(C.Structure):
    _fields_ = [("allreduce_sum", C.c_void_p), ("allgather", C.c_void_p), ("broadcast", C.c_void_p)]


COMM_ID_BYTES = 128


class Comm:
    """kamd_comm: the context's place among `world` ranks (one process per GPU).

    Comm.rccl(ctx, rank, world, uid): RCCL inside the library (ncclCommInitRank; uid = Comm.unique_id() of rank 0, handed to
    the other ranks by the launcher).  Comm.over_process_group(ctx, group): the collectives of torch.distributed as callbacks
    (gloo in the tests -- two ranks on one GPU; device buffers are staged through the host there).
    Comm.for_context(ctx): RCCL when the default process group runs on nccl (the id is broadcast through it), else callbacks."""

    def __init__(self, ctx: "Context", handle, keep=None, rank: int = 0, world: int = 1, transport: str = "callbacks"):
        self.ctx, self._h, self._keep = ctx, handle, keep
        self.rank, self.world = int(rank), int(world)
        self.transport = transport   # what actually carries the collectives: "rccl (inside libkallisto_amd.so)" or "callbacks over <backend>"

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(load_library().kamd_comm_unique_id(buf), "kamd_comm_unique_id")
        return buf.raw

    @classmethod
    def rccl(cls, ctx: "Context", rank: int, world: int, uid: bytes | None):
        h = C.c_void_p()
        buf = C.create_string_buffer(uid, COMM_ID_BYTES) if uid is not None else None
        _check(load_library().kamd_comm_create_rccl(ctx._h, rank, world, buf, C.byref(h)), "kamd_comm_create_rccl")
        return cls(ctx, h, rank=rank, world=world, transport="rccl (kamd_comm, inside libkallisto_amd.so)")

    @classmethod
    def over_process_group(cls, ctx: "Context", group=None):
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        device = ctx.device
        via_host = dist.get_backend(group) == "gloo"
        tmap = {0: (torch.int32, 4), 1: (torch.int32, 4), 2: (torch.int64, 8), 3: (torch.float64, 8)}   # sums of u32 / u64 as signed: same bits

        def _wrap(fn):
            def g(*a):
                try:
                    fn(*a)
                    torch.cuda.synchronize(device)
                    return 0
                except Exception:   # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            return g

        def _allreduce(user, d_buf, count, typ):
            dt, _ = tmap[int(typ)]
            t = _alias_tensor(torch, d_buf, int(count), dt, device)
            if via_host:
                h = t.cpu(); dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group); t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

        def _allgather(user, d_send, d_recv, nbytes):
            n = int(nbytes)
            if n == 0:
                return
            src = _alias_tensor(torch, d_send, n, torch.uint8, device)
            dst = _alias_tensor(torch, d_recv, n * world, torch.uint8, device)
            if via_host:
                parts = [torch.empty(n, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, src.cpu(), group=group)
                dst.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(dst, src, group=group)

        def _broadcast(user, d_buf, nbytes, root):
            t = _alias_tensor(torch, d_buf, int(nbytes), torch.uint8, device)
            if via_host:
                h = t.cpu(); dist.broadcast(h, src=int(root), group=group); t.copy_(h)
            else:
                dist.broadcast(t, src=int(root), group=group)

        f1 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32)(_wrap(_allreduce))
        f2 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)(_wrap(_allgather))
        f3 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32)(_wrap(_broadcast))
        cbs = _CommCallbacks(C.cast(f1, C.c_void_p), C.cast(f2, C.c_void_p), C.cast(f3, C.c_void_p))
        h = C.c_void_p()
        _check(load_library().kamd_comm_create_callbacks(ctx._h, rank, world, C.byref(cbs), None, C.byref(h)), "kamd_comm_create_callbacks")
        return cls(ctx, h, keep=(f1, f2, f3, cbs), rank=rank, world=world,
                   transport=f"callbacks over torch.distributed ({dist.get_backend(group)}{', staged through the host' if via_host else ''})")

    @classmethod
    def for_context(cls, ctx: "Context", group=None):
        import torch.distributed as dist
        if dist.get_backend(group) != "nccl" or os.environ.get("KAMD_COMM") == "callbacks":
            return cls.over_process_group(ctx, group)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # RCCL inside the library; if the library cannot get at RCCL at all (no librccl to dlopen -- the same on every rank of a
        # node) the same collectives run through torch.distributed's nccl backend as callbacks
        try:
            uid = cls.unique_id() if rank == 0 else None
            ok = 1
        except KallistoAmdError as e:
            uid, ok = None, 0
            if rank == 0:
                print(f"[kallisto_amd] RCCL not reachable from the library ({e}); using torch.distributed callbacks", file=sys.stderr)
        box = [uid, ok]
        dist.broadcast_object_list(box, src=0, group=group)
        if not box[1]:
            return cls.over_process_group(ctx, group)
        # every rank runs ncclCommInitRank on its context's device (kamd_comm_create_rccl sets it first); the ranks then agree -- through
        # the process group, not through the communicator under test -- that all of them hold a communicator that counts `world` ranks
        # and sums correctly.  Otherwise all of them drop it and take the callbacks: no rank may be left alone in a collective.
        comm, err = None, ""
        # OPT-IN watchdog over ncclCommInitRank and the first collective (KAMD_COMM_INIT_TIMEOUT_S, set by bench.py's self_launch for the ranks it starts;
        # unset = no watchdog: a library user gets exceptions and the fall-back below, never a hard exit).  A rank that never comes back from those
        # calls cannot be cancelled (the call is inside the library) but it can end its process: the launcher then takes the other ranks down and
        # starts the run again with the collectives carried by torch.distributed (KAMD_COMM=callbacks) instead of waiting out its whole limit on a
        # hang.  The watchdog is disarmed on EVERY way out of the guarded region (try / finally), exceptions included.
        import threading
        done = threading.Event()
        limit = float(os.environ.get("KAMD_COMM_INIT_TIMEOUT_S", "0") or 0)
        if limit > 0:
            def _watch():
                if not done.wait(limit):
                    print(f"[kallisto_amd] rank {rank}: the RCCL communicator did not come up within {limit:.0f} s (KAMD_COMM_INIT_TIMEOUT_S); giving up "
                          f"(exit code 3)", file=sys.stderr, flush=True)
                    os._exit(3)
            threading.Thread(target=_watch, daemon=True).start()
        flags = [None] * world
        try:
            try:
                comm = cls.rccl(ctx, rank, world, box[0])
                seen = comm.info()["ranks_seen"]
                if seen != world:
                    err = f"ncclCommCount says {seen} ranks, the process group has {world}"
            except (KallistoAmdError, OSError, RuntimeError) as e:
                err = str(e) or type(e).__name__
            dist.all_gather_object(flags, err, group=group)
            if not any(flags):
                try:
                    if comm.sum_int(rank + 1) != world * (world + 1) // 2:
                        err = "the all-reduce over the new communicator returned a wrong sum"
                except (KallistoAmdError, OSError, RuntimeError) as e:
                    err = str(e) or type(e).__name__
                dist.all_gather_object(flags, err, group=group)
        finally:
            done.set()
        if any(flags):
            if rank == 0:
                print(f"[kallisto_amd] RCCL communicator inside the library unusable ({next(f for f in flags if f)}); using torch.distributed "
                      f"callbacks", file=sys.stderr)
            if comm is not None:
                comm.close()
            return cls.over_process_group(ctx, group)
        return comm

    def info(self) -> dict:
        """rank / world as created, ranks the transport itself counts (ncclCommCount), backend name"""
        r, w, seen, b = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _check(load_library().kamd_comm_info(self._h, C.byref(r), C.byref(w), C.byref(seen), C.byref(b)), "kamd_comm_info")
        return {"rank": r.value, "world": w.value, "ranks_seen": seen.value, "backend": {0: "none", 1: "rccl", 2: "callbacks"}[b.value]}

    def broadcast_np(self, arr: np.ndarray, root: int = 0) -> np.ndarray:
        a = np.ascontiguousarray(arr).copy()
        _check(load_library().kamd_comm_broadcast_host(self.ctx._h, self._h, a.ctypes.data, a.nbytes, root), "kamd_comm_broadcast_host")
        return a

    def sum_int(self, v: int) -> int:
        a = np.array([int(v)], np.uint64)
        _check(load_library().kamd_comm_sum_u64_host(self.ctx._h, self._h, a.ctypes.data, 1), "kamd_comm_sum_u64_host")
        return int(a[0])

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):   # the context is still alive (Context.close() closes its communicator first)
                load_library().kamd_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _alias_tensor(torch, ptr: int, n: int, dtype, device: int):
    """torch tensor over device memory owned by the library (no copy) via __cuda_array_interface__."""
    class _Holder:
        pass
    h = _Holder()
    typestr = {torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1", torch.float64: "<f8", torch.int16: "<i2"}[dtype]
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device=f"cuda:{device}")


# ---- host-side FP64 helpers (bit-exact with the reference) ------------------------------------------------------------

def mean_frag_lens_trunc(flens: np.ndarray) -> np.ndarray:
    fl = np.ascontiguousarray(flens, np.uint32)
    out = np.zeros(MAX_FRAG_LEN, np.float64)
    load_library().kamd_mean_frag_lens_trunc(fl.ctypes.data, out.ctypes.data)
    return out


def trunc_gaussian_fld(mean: float, sd: float) -> np.ndarray:
    out = np.zeros(MAX_FRAG_LEN, np.float64)
    load_library().kamd_trunc_gaussian_fld(0, MAX_FRAG_LEN, mean, sd, out.ctypes.data)
    return out


def eff_lens(target_lens: np.ndarray, mean_fl_trunc: np.ndarray) -> np.ndarray:
    tl = np.ascontiguousarray(target_lens, np.int32)
    t = np.ascontiguousarray(mean_fl_trunc, np.float64)
    out = np.zeros(len(tl), np.float64)
    load_library().kamd_eff_lens(tl.ctypes.data, len(tl), t.ctypes.data, out.ctypes.data)
    return out


def bootstrap_seeds(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint64)
    load_library().kamd_bootstrap_seeds(int(seed), n, out.ctypes.data)
    return out


def counts_to_tpm(est_counts: np.ndarray, eff: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(est_counts, np.float64)
    e = np.ascontiguousarray(eff, np.float64)
    out = np.zeros(len(a), np.float64)
    load_library().kamd_counts_to_tpm(a.ctypes.data, e.ctypes.data, len(a), out.ctypes.data)
    return out


@dataclass
class QuantResult:
    n_processed: int
    n_pseudoaligned: int
    n_unique: int
    ecs: ECs | None
    flens: np.ndarray
    eff_lens: np.ndarray
    est_counts: np.ndarray
    alpha_before_zeroes: np.ndarray
    tpm: np.ndarray
    em_rounds: int
    stats: dict = field(default_factory=dict)


def quant(ctx: Context, opts: QuantOpts, batches, download_ecs: bool = True, group=None, comm: Comm | None = None) -> QuantResult:
    """The `kallisto quant` flow (src/main.cpp:2654-2730) over device-resident read batches.

    batches: iterable of (words, lens, n_items, max_len).  Several ranks (torch.distributed initialised, or `comm` given):
    every rank passes its own shard of the reads; the EC counts are merged inside the library (kamd_ec_allreduce: one
    all-reduce + all-gathers over RCCL) before the EM, which runs partitioned over the ranks (kamd_em_run_comm).
    comm=False: this rank on its own, even inside an initialised process group.
    """
    index = ctx.index
    batches = list(batches)
    n_proc = 0
    if comm is False:
        comm = None
    elif comm is None and (group is not None or _dist_on()):
        comm = getattr(ctx, "_comm", None)
        if comm is None:
            comm = ctx._comm = Comm.for_context(ctx, group)
    # the flow itself runs inside the library (kamd_quant_batches); this function only marshals
    T = int(index.num_targets)
    arr = (_Batch * max(len(batches), 1))()
    for i, (words, lens, n_items, max_len) in enumerate(batches):
        arr[i] = _Batch(words.data_ptr(), lens.data_ptr(), int(n_items), int(max_len))
    flens = np.zeros(MAX_FRAG_LEN, np.uint32)
    eff = np.zeros(T, np.float64); alpha = np.zeros(T, np.float64); abz = np.zeros(T, np.float64); tpm = np.zeros(T, np.float64)
    tl = np.ascontiguousarray(index.target_lens, np.int32)
    q = _QuantOut(0, 0, flens.ctypes.data, eff.ctypes.data, alpha.ctypes.data, abz.ctypes.data, tpm.ctypes.data)
    _check(load_library().kamd_quant_batches(ctx._h, C.byref(opts), arr, len(batches), tl.ctypes.data, T, comm._h if comm is not None else None, C.byref(q)),
           "kamd_quant_batches")
    res = _EcResult()
    _check(load_library().kamd_ec_finalize_result(ctx._h, C.byref(res)), "kamd_ec_finalize_result")
    ctx.ec_result = res
    ecs = ctx.download_ecs() if download_ecs else None
    n_aln = n_uniq = 0
    if ecs is not None:
        n_aln = int(ecs.counts.sum(dtype=np.uint64))
        sizes = np.diff(ecs.ec_off.astype(np.int64))
        n_uniq = int(ecs.counts[sizes == 1].sum(dtype=np.uint64))
    return QuantResult(int(q.n_processed), n_aln, n_uniq, ecs, flens, eff, alpha, abz, tpm, int(q.em_rounds), ctx.stats())


def _dist_rank_of(comm: Comm, group=None) -> int:
    return int(comm.rank)   # the communicator knows its place (with or without torch.distributed)


def _dist_on() -> bool:
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:
        return False


def _dist_rank(group=None) -> int:
    import torch.distributed as dist
    return dist.get_rank(group)


def _sum_int(ctx: Context, v: int, group=None) -> int:
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v)], dtype=torch.int64, device=f"cuda:{ctx.device}")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def _broadcast_np(ctx: Context, arr: np.ndarray, group=None) -> np.ndarray:
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(arr.astype(np.int64)).to(f"cuda:{ctx.device}")
    dist.broadcast(t, src=0, group=group)
    return t.cpu().numpy().astype(arr.dtype)
