"""ctypes binding of tests/emu/libkamd_emu.so (TEST INFRASTRUCTURE): the product's host code (index loader, packer) and
the per-item device logic of kamd_core.h compiled for the CPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from kallisto_amd.api import _View

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "emu", "libkamd_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
        L = C.CDLL(PATH)
        L.kamd_last_error.restype = C.c_char_p
        L.kamd_packed_record_words.restype = C.c_uint64
        L.emu_pseudoalign.restype = C.c_int64
        L.emu_ec_state.restype = C.c_int64
        L.emu_resolve.restype = C.c_int64
        _lib = L
    return _lib


class EmuIndex:
    def __init__(self, path):
        L = lib()
        self.h = C.c_void_p()
        rc = L.kamd_index_load(path.encode(), 2, C.byref(self.h))
        if rc != 0:
            raise RuntimeError(L.kamd_last_error().decode())
        self.view = _View()
        L.kamd_index_get_view(self.h, C.byref(self.view))

    def close(self):
        if self.h:
            lib().kamd_index_free(self.h)
            self.h = None


def pack(seqs, max_len=None):
    L = lib()
    n = len(seqs)
    lens = np.array([len(s) for s in seqs], np.int32)
    max_len = max_len or max(int(lens.max(initial=1)), 1)
    off = np.zeros(n, np.uint64)
    if n:
        off[1:] = np.cumsum(lens[:-1].astype(np.uint64))
    rec = L.kamd_packed_record_words(max_len)
    words = np.zeros(max(n * rec, 1), np.uint32)
    l16 = np.zeros(max(n, 1), np.uint16)
    rc = L.kamd_pack_reads_host(b"".join(seqs), off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), C.c_uint64(n),
                                C.c_int32(max_len), words.ctypes.data_as(C.c_void_p), l16.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(L.kamd_last_error().decode())
    return words, l16, max_len


def pseudoalign(ix: EmuIndex, words, l16, n_items, paired, max_len):
    """per item: sorted transcript set after the on-list mask; returns (off, ids, nhits[2*n], probes, bucket_reads, tuple_sizes)"""
    L = lib()
    out_off = np.zeros(n_items + 1, np.uint64)
    cap = max(n_items * 256, 1024)
    out_ids = np.zeros(cap, np.uint32)
    nh = np.zeros(2 * max(n_items, 1), np.int32)
    ts = np.zeros(max(n_items, 1), np.uint32)
    pr, br = C.c_uint64(0), C.c_uint64(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    r = L.emu_pseudoalign(C.byref(ix.view), p(words), p(l16), C.c_uint64(n_items), int(paired), C.c_int32(max_len), p(out_off),
                          p(out_ids), C.c_uint64(cap), p(nh), C.byref(pr), C.byref(br), p(ts))
    if r < 0:
        raise RuntimeError(f"emu_pseudoalign failed {r}")
    return out_off, out_ids, nh, pr.value, br.value, ts


def pseudoalign_opts(ix: EmuIndex, words, l16, n_items, paired, max_len, single_overhang, strand, fl, has_mean_fl, no_jump=0):
    L = lib()
    L.emu_pseudoalign_opts.restype = C.c_int64
    out_off = np.zeros(n_items + 1, np.uint64)
    cap = max(n_items * 256, 1024)
    out_ids = np.zeros(cap, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    r = L.emu_pseudoalign_opts(C.byref(ix.view), p(words), p(l16), C.c_uint64(n_items), int(paired), C.c_int32(max_len),
                               int(single_overhang), int(strand), int(fl), int(has_mean_fl), int(no_jump), p(out_off), p(out_ids), C.c_uint64(cap))
    if r < 0:
        raise RuntimeError(f"emu_pseudoalign_opts failed {r}")
    return out_off, out_ids


def tuples(ix: EmuIndex, words, l16, n_items, paired, max_len, use_stepper, stride=40):
    L = lib()
    L.emu_tuples.restype = C.c_int64
    out = np.zeros(n_items * stride, np.uint32)
    pr = (C.c_uint64 * 3)(0, 0, 0)   # [0] dbg.find calls, [1] of them answered from the unitig text (use_stepper & 4), [2] appended classes (use_stepper & 8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    r = L.emu_tuples(C.byref(ix.view), p(words), p(l16), C.c_uint64(n_items), int(paired), C.c_int32(max_len), int(use_stepper),
                     p(out), C.c_uint64(stride), pr)
    assert r == 0
    tuples.last_text_hits = int(pr[1])
    tuples.last_appended = int(pr[2])
    return out.reshape(n_items, stride), int(pr[0])


def ec_state(ix: EmuIndex, words, l16, n_items, paired, max_len):
    L = lib()
    dense = np.zeros(max(ix.view.n_ecs, 1), np.uint32)
    cap = max(n_items * 64, 1024)
    stream = np.zeros(cap, np.uint32)
    rec_off = np.zeros(max(n_items, 1), np.uint64)
    nr = C.c_uint64(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    nw = L.emu_ec_state(C.byref(ix.view), p(words), p(l16), C.c_uint64(n_items), int(paired), C.c_int32(max_len), p(dense),
                        p(stream), C.c_uint64(cap), p(rec_off), C.byref(nr))
    if nw < 0:
        raise RuntimeError(f"emu_ec_state failed {nw}")
    return dense, stream[:nw].copy(), rec_off[:nr.value].copy()


def resolve(ix: EmuIndex, dense, stream, rec_off):
    L = lib()
    cap_e, cap_i = len(dense) + len(rec_off) + 1, int(ix.view.ec_nnz) + 64 * (len(rec_off) + 1)
    ec_off = np.zeros(cap_e + 1, np.uint64)
    ec_ids = np.zeros(cap_i, np.uint32)
    counts = np.zeros(cap_e, np.uint32)
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    dense = np.ascontiguousarray(dense, np.uint32); stream = np.ascontiguousarray(stream, np.uint32)
    rec_off = np.ascontiguousarray(rec_off, np.uint64)
    n = L.emu_resolve(C.byref(ix.view), p(dense), p(stream), p(rec_off), C.c_uint64(len(rec_off)), p(ec_off), p(ec_ids), p(counts),
                      C.c_uint64(cap_e), C.c_uint64(cap_i))
    if n < 0:
        raise RuntimeError("emu_resolve failed")
    return {tuple(ec_ids[ec_off[i]:ec_off[i + 1]].tolist()): int(counts[i]) for i in range(n)}
