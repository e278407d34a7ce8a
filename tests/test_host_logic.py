"""No-GPU tests of the product's host code and of the per-item device logic (compiled for the CPU in tests/emu):
index flattening vs the oracle's parse, the 2-bit packer, the match state machine + table probe per item vs the oracle,
the host FP64 helpers, and the C ABI surface of libkallisto_amd.so."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O
from tests import common, emu_binding as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_with_layout(path, layout, load=None):
    """kamd_index_load with KAMD_TABLE_LAYOUT (and KAMD_TABLE_LOAD) set for the call: the k-mer table's layout is chosen at load time."""
    old = {k: os.environ.get(k) for k in ("KAMD_TABLE_LAYOUT", "KAMD_TABLE_LOAD")}
    os.environ["KAMD_TABLE_LAYOUT"] = layout
    if load is not None:
        os.environ["KAMD_TABLE_LOAD"] = str(load)
    try:
        return E.EmuIndex(path)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# every test that takes `idx` runs on both layouts of the k-mer table (kamd_core.h): wide = 3 slots of 20 bytes per line,
# compact = 4 quotiented slots of 16 bytes (the default since round 4)
@pytest.fixture(scope="module", params=["wide", "compact"])
def idx(request):
    cache = {}

    def get(case):
        if case not in cache:
            p = common.load_case(case)[1]
            cache[case] = (_load_with_layout(p, request.param), O.Index(p))
            assert cache[case][0].view.table_layout == (1 if request.param == "compact" else 0)
        return cache[case]
    return get


@pytest.mark.parametrize("case", common.CASES)
def test_flattened_index_matches_oracle_parse(case, idx):
    e, o = idx(case)
    v = e.view
    assert (v.k, v.n_kmers, v.n_unitigs, v.n_blocks, v.n_ecs, v.n_targets) == \
        (o.k, o.num_kmers, o.num_unitigs, o.num_blocks, o.num_ecs, o.num_targets)
    from kallisto_amd.api import _np
    assert np.array_equal(_np(v.target_lens, v.n_targets, np.int32), o.target_lens)
    # every transcript set of the oracle's parse exists in the flattened index
    ec_off, ec_ids = _np(v.ec_off, v.n_ecs + 1, np.uint64), _np(v.ec_ids, v.ec_nnz, np.uint32)
    mine = {tuple(ec_ids[ec_off[i]:ec_off[i + 1]].tolist()) for i in range(v.n_ecs)}
    theirs = {tuple(o.ec(i)) for i in range(o.num_ecs)}
    assert mine == theirs
    table = _np(v.table, (v.n_buckets + v.pad_buckets) * 8, np.uint64).reshape(-1, 8)
    if v.table_layout == 1:
        _check_compact_table(case, v, table)
        return
    # table invariants: every k-mer placed once, load factor <= 0.5 over 3-slot buckets (words 0..2 of each 64-byte bucket)
    assert v.slots_per_bucket == 3
    keys = table[:, 0:3] & np.uint64((1 << 62) - 1)
    used = keys != np.uint64((1 << 62) - 1)
    assert int(used.sum()) == v.n_kmers
    assert 2 * v.n_kmers <= 3 * v.n_buckets + 6
    # text positions: every k-mer's slot points at its own bases in the unitig text
    gpos = np.ascontiguousarray(table[:, 6:8]).view(np.uint32).reshape(-1, 4)[:, 0:3]
    text = _np(v.utext, v.utext_words, np.uint32)
    assert v.text_bases == int(_np(v.unitig_len, v.n_unitigs, np.uint32).sum())
    bases = ((text[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).reshape(-1).astype(np.uint64)
    kk = int(v.k)
    sel = np.argwhere(used)[:: max(1, int(used.sum()) // 4000)]   # a sample of the slots
    for b, j in sel:
        g = int(gpos[b, j])
        fwd = 0
        for x in bases[g:g + kk]:
            fwd = (fwd << 2) | int(x)
        rc = 0
        for x in bases[g:g + kk][::-1]:
            rc = (rc << 2) | (3 - int(x))
        assert min(fwd, rc) == int(keys[b, j]), (case, b, j)


def _kmer_hash32(canon):
    """kamd_core.h kmer_hash32, restated"""
    M = 0xFFFFFFFF
    lo, hi = canon & M, canon >> 32
    h = (hi * 0x9E3779B1) & M
    h ^= h >> 15
    x = ((lo ^ h) * 0x85EBCA6B) & M
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & M
    x ^= x >> 16
    return x


def _check_compact_table(case, v, table):
    """The compact layout decoded here, independently of kamd_core.h: four {w0, w1} slots per line; w0 = tag | class << tag_w, the tag =
    low tag_q bits of the hash | key bits above 32 | displacement from the home bucket (7 = empty); w1 = rem_f | rem_b | text position |
    forward-is-canonical | continue flag.  Every k-mer once; a sample of slots re-derived from the unitig text."""
    from kallisto_amd.api import _np
    assert v.slots_per_bucket == 4
    q, dsh, tw = int(v.tag_q), int(v.tag_dsh), int(v.tag_w)
    kk = int(v.k)
    db = tw - dsh   # bits of the displacement: four when the class ids leave room for them
    assert dsh == q + max(0, 2 * kk - 32) and db == (4 if dsh + 4 + max(int(v.n_uec), 1).bit_length() <= 64 else 3)
    span = -(-(1 << 32) // int(v.n_buckets))
    assert (1 << q) >= span and (q == 0 or (1 << (q - 1)) < span)
    w0, w1 = table[:, 0::2], table[:, 1::2]
    disp = (w0 >> np.uint64(dsh)) & np.uint64((1 << db) - 1)
    used = disp != np.uint64((1 << db) - 1)
    assert np.all(w0[~used] == np.uint64(0xFFFFFFFFFFFFFFFF))
    assert int(used.sum()) == v.n_kmers
    assert int(disp[used].max(initial=0)) <= (1 << db) - 2
    assert int((w0[used] >> np.uint64(tw)).max(initial=0)) < v.n_uec
    # a bucket whose continue flag is set is full, and the flag sits in slot 0 only
    cont = (w1[:, 0] >> np.uint64(63)) != 0
    assert np.all(used[cont].all(axis=1)) and not np.any((w1[:, 1:] >> np.uint64(63)) != 0)
    text = _np(v.utext, v.utext_words, np.uint32)
    bases = ((text[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).reshape(-1).astype(np.uint64)
    sel = np.argwhere(used)[:: max(1, int(used.sum()) // 4000)]
    for b, j in sel:
        a, c = int(w0[b, j]), int(w1[b, j])
        g = (c >> 32) & 0x3FFFFFFF
        fwd = 0
        for x in bases[g:g + kk]:
            fwd = (fwd << 2) | int(x)
        rc = 0
        for x in bases[g:g + kk][::-1]:
            rc = (rc << 2) | (3 - int(x))
        canon = min(fwd, rc)
        assert ((c >> 62) & 1) == (1 if fwd < rc else 0), (case, b, j)
        h = _kmer_hash32(canon)
        home = (h * int(v.n_buckets)) >> 32
        d = int(b) - home
        assert 0 <= d <= (1 << db) - 2
        assert (a & ((1 << tw) - 1)) == ((h & ((1 << q) - 1)) | ((canon >> 32) << q) | (d << dsh)), (case, b, j)
        # reachable: every bucket from the home up to the one before carries the continue flag
        assert all(cont[home + i] for i in range(d)), (case, b, j)


@pytest.mark.parametrize("case", common.CASES)
def test_every_kmer_is_found_through_the_products_probe(case, idx):
    """All k-mers of all unitigs looked up with kamd_core.h's probe_table in the tables kamd_index_load built (parallel first touch, prefetch
    rings, continue flags set by the placement): found, own text position, own offset, a covering block of the own unitig.  (The same check
    passed on the human-sized index of BASELINE config #3: 56 799 409 k-mers, 1.087 bucket lines per probe.)"""
    import ctypes as C
    from tests import emu_binding as E
    e, _ = idx(case)
    L = E.lib()
    L.emu_verify_table.restype = C.c_int64
    lines = C.c_uint64(0)
    assert L.emu_verify_table(C.byref(e.view), C.byref(lines)) == 0
    assert e.view.n_kmers <= lines.value <= 2 * e.view.n_kmers


@pytest.mark.parametrize("case", common.CASES)
def test_per_item_logic_matches_oracle(case, idx):
    """match() jump logic, k-mer table probe and intersectKmers per read/pair: identical sets AND identical hit counts."""
    e, o = idx(case)
    meta, _, r1, r2 = common.load_case(case)
    paired = r2 is not None
    reads = common.interleave(r1, r2)
    words, l16, max_len = E.pack(reads)
    off, ids, nh, probes, breads, ts = E.pseudoalign(e, words, l16, len(r1), paired, max_len)
    opts = O.Opts(int(paired), 0.0, 0.0, 1, 0)
    tot_probes = 0
    for i in range(len(r1)):
        s, n1, n2 = o.pseudoalign(opts, r1[i], r2[i] if paired else None)
        assert ids[off[i]:off[i + 1]].tolist() == s, (case, i)
        if paired or s:  # single-end match() clears v when the running intersection empties (KmerIndex.cpp:1766-1769)
            assert (nh[2 * i], nh[2 * i + 1]) == (n1, n2), (case, i)
    assert breads >= probes > 0


@pytest.mark.parametrize("case,variant", common.all_variants())
def test_positional_filters_match_oracle(case, variant, idx):
    """findPosition fragment-length filter (single-end / orphan mates) and --fr/--rf strand filter, per read."""
    e, o = idx(case)
    meta, _, r1, r2 = common.load_case(case)
    ov = common.parse_variant(meta["variants"][variant])
    paired = bool(ov["paired"])
    words, l16, max_len = E.pack(common.interleave(r1, r2 if paired else None))
    has_fl = ov["fld"] > 0
    mean_fl = float(ov["fld"]) if has_fl else 0.0   # the -l value itself while reads are processed (MinCollector.h:38-41)
    off, ids = E.pseudoalign_opts(e, words, l16, len(r1), paired, max_len, ov["single_overhang"], ov["strand"], int(mean_fl), has_fl,
                                  ov["no_jump"] | 2 * ov["union"])   # --union, and the per-hit strand filter it / --no-jump switch on
    opts = O.Opts(ov["paired"], ov["fld"], ov["sd"], ov["single_overhang"], ov["strand"], ov["no_jump"], ov["union"])
    for i in range(len(r1)):
        s, _, _ = o.pseudoalign(opts, r1[i], r2[i] if paired else None, mean_fl, has_fl)
        assert ids[off[i]:off[i + 1]].tolist() == s, (case, variant, i)


@pytest.mark.parametrize("no_jump", [0, 1])
@pytest.mark.parametrize("case", common.CASES)
def test_resumable_state_machine_equals_match(case, no_jump, idx):
    """kernel A v2's one-probe-per-step state machine vs the straight-line match(): same sets, same probe count
    (also with --no-jump, where every k-mer is looked up: up to read_len - k + 1 hits per mate)."""
    e, _ = idx(case)
    meta, _, r1, r2 = common.load_case(case)
    paired = r2 is not None
    words, l16, max_len = E.pack(common.interleave(r1, r2))
    a, pa = E.tuples(e, words, l16, len(r1), paired, max_len, 0 | 2 * no_jump, stride=80)
    b, pb = E.tuples(e, words, l16, len(r1), paired, max_len, 1 | 2 * no_jump, stride=80)
    assert pa == pb and np.array_equal(a, b)
    # kernel A v3: jump / middle / back-off windows answered from the unitig text where they match it -- same sets, same
    # number of dbg.find calls, some of them without touching the table
    t, pt = E.tuples(e, words, l16, len(r1), paired, max_len, 1 | 2 * no_jump | 4, stride=80)
    assert pt == pa and np.array_equal(a, t)
    if not no_jump and case in ("ref_test_pe", "human_pe", "yeast_se"):
        assert E.tuples.last_text_hits > 0
    # kernel A's second pass (round 6): the class list append-only -- a class equal to the one appended last merges, anything else is appended,
    # duplicates that are not neighbours stay until the classes are mapped to sets: same sets, same probes, and never more list entries than probes
    ap, pp = E.tuples(e, words, l16, len(r1), paired, max_len, 1 | 2 * no_jump | 4 | 8, stride=80)
    assert pp == pa and np.array_equal(a, ap)
    assert 0 < E.tuples.last_appended <= pp
    if no_jump and case == "mosaic_pe":
        c, pc = E.tuples(e, words, l16, len(r1), paired, max_len, 0, stride=80)
        assert pa > pc and not np.array_equal(a, c)   # the flag does something on this case


def test_probe_counts_match_oracle(idx):
    e, o = idx("ref_test_pe")
    meta, _, r1, r2 = common.load_case("ref_test_pe")
    words, l16, max_len = E.pack(common.interleave(r1, r2))
    _, _, _, probes, _, _ = E.pseudoalign(e, words, l16, len(r1), True, max_len)
    buf, off, lens = O.pack_reads(common.interleave(r1, r2))
    res = O.process_reads(o, O.Opts(1, 0.0, 0.0, 0, 0), buf, off, lens)
    assert probes == res.n_probes == 68339  # instrumented reference, config #1


def test_packer_layout():
    words, l16, max_len = E.pack([b"ACGTN", b"acgtacgtacgtacgtT", b""], 20)
    rec = int(E.lib().kamd_packed_record_words(20))
    assert rec == (20 + 15) // 16 + 1 + (20 + 31) // 32 + 1
    w = words.reshape(3, rec)
    assert w[0, 0] == (0 | 1 << 2 | 2 << 4 | 3 << 6) and w[0, 3] == 1 << 4 and l16[0] == 5
    assert w[1, 0] == int("".join(["11100100"] * 4), 2) and w[1, 1] == 3 and w[1, 3] == 0 and l16[1] == 17
    assert not w[2].any() and l16[2] == 0


def test_host_fp64_helpers_match_oracle():
    import kallisto_amd.api as A
    rng = np.random.default_rng(0)
    flens = np.zeros(1000, np.uint32)
    flens[150:260] = rng.integers(0, 200, 110)
    assert np.array_equal(A.mean_frag_lens_trunc(flens), O.mean_frag_lens_trunc(flens))
    assert np.array_equal(A.trunc_gaussian_fld(187.3, 23.1), O.trunc_gaussian_fld(187.3, 23.1))
    lens = rng.integers(20, 5000, 500).astype(np.int32)
    t = O.mean_frag_lens_trunc(flens)
    assert np.array_equal(A.eff_lens(lens, t), O.eff_lens(lens, t)[0])
    est = rng.random(500) * 100
    assert np.array_equal(A.counts_to_tpm(est, A.eff_lens(lens, t)), O.counts_to_tpm(est, O.eff_lens(lens, t)[0]))


def test_library_loads_and_exports_every_declared_symbol():
    """libkallisto_amd.so must load on a box without a GPU and export everything include/kallisto_amd.h declares."""
    import kallisto_amd.api as A
    lib = A.load_library()
    hdr = open(os.path.join(ROOT, "include", "kallisto_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kamd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/kallisto_amd.h but not exported"
    assert declared == set(A.exported_symbols()), "api.py's symbol table is out of sync with the header"


def test_no_cpu_path():
    """Compute entry points fail loudly without a GPU (this container has none; on the GPU box this test is a no-op)."""
    import torch
    import kallisto_amd as ka
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ka.KallistoAmdError):
        ka.Context(0)
    h = C.c_void_p()
    rc = ka.load_library().kamd_ctx_create(0, None, C.byref(h))
    assert rc != 0 and b"no HIP device" in ka.load_library().kamd_last_error()


def test_product_does_not_touch_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "kallisto_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                txt = open(os.path.join(d, f)).read()
                assert "oracle" not in txt.lower().replace("oracle/_ref/kallisto", "").replace("oracle ref", "") or f in ("synth.py",), \
                    f"{f} mentions the oracle"


@pytest.mark.parametrize("case", ["ref_test_pe", "dlist_pe"])
def test_index_loader_refuses_damaged_files(case, tmp_path):
    """kamd_index_load on a truncated index returns an error (with a message) instead of reading past the end; the
    reference exits from inside Bifrost on such files."""
    data = open(common.load_case(case)[1], "rb").read()
    L = E.lib()
    rng = np.random.default_rng(3)
    cuts = [0, 1, 7, 8, 16, 24, 40, 100, 1000, len(data) - 8, len(data) - 1] + [int(x) for x in rng.integers(0, len(data), 12)]
    for c in sorted(set(cuts)):
        p = str(tmp_path / "t.idx")
        open(p, "wb").write(data[:c])
        h = C.c_void_p()
        rc = L.kamd_index_load(p.encode(), 2, C.byref(h))
        assert rc != 0 and L.kamd_last_error(), c
    h = C.c_void_p()
    assert L.kamd_index_load(str(tmp_path / "missing.idx").encode(), 2, C.byref(h)) != 0
    # counts the loader sizes its tables by come from the file: an absurd 64-bit value anywhere must end in an error code or a clean load, never
    # in an exception crossing the C ABI or an allocation of that size (the node count)
    for pos in sorted(set(int(x) for x in rng.integers(8, len(data) - 8, 80))):
        bad = bytearray(data)
        bad[pos:pos + 8] = (1 << 44).to_bytes(8, "little")
        p = str(tmp_path / "c.idx")
        open(p, "wb").write(bytes(bad))
        h = C.c_void_p()
        rc = L.kamd_index_load(p.encode(), 2, C.byref(h))
        if rc == 0:
            L.kamd_index_free(h)
        else:
            assert L.kamd_last_error(), pos


@pytest.mark.parametrize("layout", ["wide", "compact"])
@pytest.mark.parametrize("case", ["human_pe", "dlist_pe", "tiny_k7_se"])
def test_flattened_index_file_round_trip(case, layout, tmp_path, monkeypatch):
    """kamd_index_save / kamd_index_load on the saved file: every table of the view, the scalars and the target names come back
    bit-identical (the front-end's `flatten` sub-command and `-i index.kamd`)."""
    import ctypes as C
    import subprocess
    from kallisto_amd import api
    idx_path = common.load_case(case)[1]
    monkeypatch.setenv("KAMD_TABLE_LAYOUT", layout)
    a = api.Index(idx_path)
    flat = str(tmp_path / "index.kamd")
    a.save(flat)
    # a flattened file carries its layout: with no layout named it loads as what it is; naming the OTHER layout is an error, not a silent
    # hand-over of the wrong table (the compact one is asked for to fit a footprint)
    monkeypatch.delenv("KAMD_TABLE_LAYOUT")
    b = api.Index(flat)
    other = "wide" if layout == "compact" else "compact"
    monkeypatch.setenv("KAMD_TABLE_LAYOUT", other)
    with pytest.raises(api.KallistoAmdError, match="was asked for"):
        api.Index(flat)
    monkeypatch.setenv("KAMD_TABLE_LAYOUT", "auto")
    assert api.Index(flat).view.table_layout == a.view.table_layout
    monkeypatch.setenv("KAMD_TABLE_LAYOUT", layout)
    assert api.Index(flat).view.table_layout == a.view.table_layout
    # the file knows which kallisto index it was written from (what the front-end asks before it picks up <index>.kamd beside an index)
    L = api.load_library()
    assert L.kamd_flat_index_matches(flat.encode(), idx_path.encode()) == 1
    other_idx = common.load_case("yeast_se" if case != "yeast_se" else "human_pe")[1]
    assert L.kamd_flat_index_matches(flat.encode(), other_idx.encode()) == 0
    assert L.kamd_flat_index_matches(idx_path.encode(), idx_path.encode()) == 0 and L.kamd_flat_index_matches(str(tmp_path / "none").encode(), idx_path.encode()) == 0
    va, vb = a.view, b.view
    S = va.slots_per_bucket
    assert (va.table_layout, S) == ((1, 4) if layout == "compact" else (0, 3))
    sizes = {"table": (va.n_buckets + va.pad_buckets) * 8 * 8, "slot_block": (va.n_buckets + va.pad_buckets) * S * 4,
             "slot_dist": (va.n_buckets + va.pad_buckets) * S * 4, "uec_ec": va.n_uec * 4, "ec_off": (va.n_ecs + 1) * 8, "ec_ids": va.ec_nnz * 4,
             "unitig_blk_off": (va.n_unitigs + 1) * 8, "unitig_len": va.n_unitigs * 4, "blk_unitig": va.n_blocks * 4, "blk_lb": va.n_blocks * 4,
             "blk_ub": va.n_blocks * 4, "blk_ec": va.n_blocks * 4, "blk_pos_off": va.n_blocks * 8, "blk_sense": va.n_blocks,
             "onlist_bits": va.onlist_words * 4, "utext": va.utext_words * 4, "unitig_gpos": (va.n_unitigs + 1) * 8,
             "dtable": (va.n_dbuckets + va.dpad_buckets) * 8 * 8 if va.n_dbuckets else 0}
    # blk_posw: one word per (block, transcript) pair, its length is the last block's offset + size -- compared through the offsets
    n_posw = 0
    if va.n_blocks:
        po = np.frombuffer(C.string_at(C.cast(va.blk_pos_off, C.c_void_p), int(va.n_blocks) * 8), np.uint64)
        be = np.frombuffer(C.string_at(C.cast(va.blk_ec, C.c_void_p), int(va.n_blocks) * 4), np.uint32)
        eo = np.frombuffer(C.string_at(C.cast(va.ec_off, C.c_void_p), int(va.n_ecs + 1) * 8), np.uint64)
        n_posw = int((po + (eo[be.astype(np.int64) + 1] - eo[be.astype(np.int64)])).max())
    sizes["blk_posw"] = n_posw * 4
    sizes["target_lens"] = len(a.target_lens) * 4
    for name, typ in type(va)._fields_:
        x, y = getattr(va, name), getattr(vb, name)
        if typ is C.c_void_p:
            n = int(sizes[name])
            if n:
                assert C.string_at(C.cast(x, C.c_void_p), n) == C.string_at(C.cast(y, C.c_void_p), n), name
        else:
            assert x == y, name
    assert np.array_equal(a.target_lens, b.target_lens)
    lib = api.load_library()
    lib.kamd_index_target_name.restype = C.c_char_p
    assert [lib.kamd_index_target_name(a._h, i) for i in range(a.num_targets)] == [lib.kamd_index_target_name(b._h, i) for i in range(b.num_targets)]
    # the front-end's sub-command writes the same file
    exe = os.path.join(ROOT, "kallisto_amd", "kallisto_amd_quant")
    if os.path.exists(exe):
        flat2 = str(tmp_path / "cli.kamd")
        assert subprocess.run([exe, "flatten", "-i", idx_path, "-o", flat2, "-t", "3"]).returncode == 0
        c = api.Index(flat2)   # (a second build: the order of the k-mers inside a bucket depends on the builder's threads, so no byte equality)
        assert (c.num_kmers, c.num_unitigs, c.num_ecs, c.num_targets) == (a.num_kmers, a.num_unitigs, a.num_ecs, a.num_targets)
        assert all(np.array_equal(x, y) for x, y in zip(c.ec_sets(), a.ec_sets())) and np.array_equal(c.target_lens, a.target_lens)
    # a damaged file is refused, not trusted
    blob = bytearray(open(flat, "rb").read())
    open(flat, "wb").write(blob[: len(blob) // 2])
    with pytest.raises(api.KallistoAmdError):
        api.Index(flat)
    # ... and so is one whose arrays no longer fit together: an index-valued entry beyond its range (here: a transcript id of an EC
    # list set to 2^31), another layout stamp, another magic
    v = a.view
    import ctypes as C
    ec_ids = np.ctypeslib.as_array(C.cast(v.ec_ids, C.POINTER(C.c_uint32)), shape=(int(v.ec_nnz),)).copy()
    at = bytes(blob).find(ec_ids.tobytes())
    assert at > 0
    for patch_at, patch in ((at, (1 << 31).to_bytes(4, "little")), (8, b"\x07\x00\x00\x00"), (0, b"X"), (7, b"2")):   # (7: the previous format version)
        bad = bytearray(blob)
        bad[patch_at:patch_at + len(patch)] = patch
        open(flat, "wb").write(bad)
        with pytest.raises(api.KallistoAmdError):
            api.Index(flat)


@pytest.mark.parametrize("load", [0.3, 0.75, 0.9])
@pytest.mark.parametrize("case", ["human_pe", "dlist_pe"])
def test_compact_table_at_other_loads(case, load):
    """KAMD_TABLE_LOAD: a sparse table, and dense ones where keys sit several buckets from home -- beyond what the displacement field
    can say (14 buckets, or 6 when the class ids leave it three bits) the builder takes a larger table and counts again.  Every k-mer is still found with its own payload."""
    import ctypes as C
    e = _load_with_layout(common.load_case(case)[1], "compact", load)
    v = e.view
    assert v.table_layout == 1 and v.n_kmers <= 4 * v.n_buckets
    if load <= 0.75:
        assert abs(v.n_kmers / (4.0 * v.n_buckets) - load) < 0.02
    L = E.lib()
    L.emu_verify_table.restype = C.c_int64
    lines = C.c_uint64(0)
    assert L.emu_verify_table(C.byref(v), C.byref(lines)) == 0
    assert 1.0 <= lines.value / v.n_kmers < 2.5
    e.close()


def test_table_layout_variable():
    """KAMD_TABLE_LAYOUT: `auto` takes the compact layout when its fields fit (all fixtures), anything else than wide / compact / auto is an error."""
    p = common.load_case("ref_test_pe")[1]
    e = _load_with_layout(p, "auto")
    assert e.view.table_layout == 1
    e.close()
    with pytest.raises(RuntimeError, match="KAMD_TABLE_LAYOUT"):
        _load_with_layout(p, "dense")
    e = _load_with_layout(p, "wide")
    assert (e.view.table_layout, e.view.slots_per_bucket, e.view.tag_w) == (0, 3, 0)
    e.close()


def test_table_layout_through_the_abi_and_the_front_end(tmp_path):
    """kamd_index_load_layout (the layout as an argument instead of KAMD_TABLE_LAYOUT) and `kallisto_amd_quant flatten --kmer-table compact`: the
    flattened file carries the compact layout; a caller that names the other one is told so."""
    import subprocess
    from kallisto_amd import api
    p = common.load_case("human_pe")[1]
    a = api.Index(p, table_layout="compact", table_load=0.5)
    assert (a.view.table_layout, a.view.slots_per_bucket) == (1, 4) and abs(a.num_kmers / (4.0 * a.view.n_buckets) - 0.5) < 0.02
    w = api.Index(p, table_layout="wide")
    assert (w.view.table_layout, w.view.slots_per_bucket) == (0, 3)
    assert api.Index(p, table_layout="auto").view.table_layout == 1
    lib = api.load_library()
    h = C.c_void_p()
    assert lib.kamd_index_load_layout(p.encode(), 2, 7, 0.0, C.byref(h)) != 0 and b"layout" in lib.kamd_last_error()
    exe = os.path.join(ROOT, "kallisto_amd", "kallisto_amd_quant")
    if os.path.exists(exe):
        flat = str(tmp_path / "c.kamd")
        assert subprocess.run([exe, "flatten", "-i", p, "-o", flat, "-t", "3", "--kmer-table", "compact"]).returncode == 0
        c = api.Index(flat, table_layout="auto")   # (a flattened file is what it is: `auto` takes it, naming the other layout is an error)
        assert c.view.table_layout == 1 and c.num_kmers == a.num_kmers
        with pytest.raises(api.KallistoAmdError, match="was asked for"):
            api.Index(flat, table_layout="wide")
        assert subprocess.run([exe, "flatten", "-i", p, "-o", flat, "--kmer-table", "dense"], stderr=subprocess.DEVNULL).returncode != 0


@pytest.mark.parametrize("case", ["human_pe", "dlist_pe", "mosaic_pe"])
def test_dense_compact_table_through_the_kernels_stepper(case):
    """A compact table as dense as the builder lets it be (keys many buckets from home, long runs of continue flags): kernel A's
    one-line-per-step walk (stepper + unitig text, as k_match_v3 runs it) and the straight-line matcher give the sets of the wide table."""
    meta, p, r1, r2 = common.load_case(case)
    paired = r2 is not None
    words, l16, max_len = E.pack(common.interleave(r1, r2))
    w = _load_with_layout(p, "wide")
    c = _load_with_layout(p, "compact", 0.9)
    assert c.view.table_layout == 1
    for mode in (0, 1 | 4, 1 | 2 | 4):
        a, pa = E.tuples(w, words, l16, len(r1), paired, max_len, mode, stride=80)
        b, pb = E.tuples(c, words, l16, len(r1), paired, max_len, mode, stride=80)
        assert pa == pb and np.array_equal(a, b), mode
    w.close(); c.close()


def test_gene_level_outputs_of_quant_tcc(tmp_path):
    """`quant-tcc -g` on the host side (kamd_genes.h through tests/emu): the mapping file parsed like Transcriptome::parseGeneMap (genes numbered
    in order of first appearance, transcripts without a line in no gene), sums in transcript order, the reference's two writers.  Input: the
    reference's own transcript-level abundance_1.tsv of the golden case; expected: its abundance.gene_1.tsv and genes.txt."""
    gold = os.path.join(common.GOLDEN, "bus_tcc", "human_pe_genes")
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(gold, "tcc_out", "abundance_1.tsv"))][1:]
    names = [r[0] for r in rows]
    alpha = np.array([float(r[3]) for r in rows]); tpm = np.array([float(r[4]) for r in rows])
    L = E.lib()
    L.fe_gene_outputs.restype = C.c_int64
    tsv, nm = str(tmp_path / "g.tsv"), str(tmp_path / "genes.txt")
    trg = np.zeros(len(names), np.int32)
    err = C.create_string_buffer(512)
    ng = L.fe_gene_outputs(os.path.join(gold, "genemap.txt").encode(), "\n".join(names).encode(), C.c_uint64(len(names)), alpha.ctypes.data_as(C.c_void_p),
                           tpm.ctypes.data_as(C.c_void_p), tsv.encode(), nm.encode(), trg.ctypes.data_as(C.c_void_p), err, C.c_uint64(512))
    assert ng == 223, err.value
    assert open(nm).read() == open(os.path.join(gold, "tcc_out", "genes.txt")).read()
    got = [l.rstrip("\n").split("\t") for l in open(tsv)]
    want = [l.rstrip("\n").split("\t") for l in open(os.path.join(gold, "tcc_out", "abundance.gene_1.tsv"))]
    assert got[0] == want[0] and [g[:2] for g in got] == [w[:2] for w in want]
    for col in (2, 3):   # the inputs were printed with six digits: sums agree to five
        a, b = np.array([float(g[col]) for g in got[1:]]), np.array([float(w[col]) for w in want[1:]])
        assert np.all(np.abs(a - b) <= 2e-5 * np.maximum(np.abs(b), 1e-3)), col
    assert (trg[9::10] == -1).all() and (trg >= -1).all() and trg.max() == ng - 1   # every tenth transcript is in no gene
    # the reference's two errors
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write(names[0] + "\n")
    assert L.fe_gene_outputs(bad.encode(), "\n".join(names).encode(), C.c_uint64(len(names)), alpha.ctypes.data_as(C.c_void_p), tpm.ctypes.data_as(C.c_void_p),
                             tsv.encode(), nm.encode(), None, err, C.c_uint64(512)) == -1 and b"No gene associated with transcript" in err.value
    open(bad, "w").write("no_such_transcript\tG1\n")
    assert L.fe_gene_outputs(bad.encode(), "\n".join(names).encode(), C.c_uint64(len(names)), alpha.ctypes.data_as(C.c_void_p), tpm.ctypes.data_as(C.c_void_p),
                             tsv.encode(), nm.encode(), None, err, C.c_uint64(512)) == -1 and b"Invalid transcript" in err.value


def test_ctypes_mirrors_match_the_header(tmp_path):
    """The structures of include/kallisto_amd.h as the C compiler lays them out against their ctypes mirrors in kallisto_amd/api.py: same size, every field
    at the same offset under the same name.  (The library writes kamd_profile / kamd_align_stats through the caller's pointer: a mirror that is a field short
    is a write behind the end of a Python object.)"""
    import ctypes
    import subprocess
    from kallisto_amd import api
    pairs = [("kamd_index_view", api._View), ("kamd_quant_opts", api.QuantOpts), ("kamd_align_stats", api._Stats), ("kamd_tuning", api.Tuning),
             ("kamd_profile", api._Profile), ("kamd_fastq_unit", api._FastqUnit), ("kamd_batch", api._Batch), ("kamd_quant_out", api._QuantOut),
             ("kamd_ec_result", api._EcResult), ("kamd_comm_callbacks", api._CommCallbacks)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "kallisto_amd.h"', 'int main(void) {']
    for cname, mirror in pairs:
        src.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in mirror._fields_:
            src.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src += ['  return 0;', '}']
    c_file = tmp_path / "layout.c"
    c_file.write_text("\n".join(src) + "\n")
    exe = tmp_path / "layout"
    p = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(c_file), "-o", str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-3000:]       # (a field name the header does not have fails here)
    got = dict(line.split() for line in subprocess.run([str(exe)], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines())
    for cname, mirror in pairs:
        assert int(got[cname]) == ctypes.sizeof(mirror), (cname, got[cname], ctypes.sizeof(mirror))
        for fname, _ in mirror._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(mirror, fname).offset, (cname, fname)


def test_ctypes_bindings_cover_the_header_with_the_same_arity():
    """every function include/kallisto_amd.h declares has a ctypes binding in kallisto_amd/api.py with as many arguments (and no binding names a function the
    header does not declare)"""
    import re
    from kallisto_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = open(os.path.join(root, "include", "kallisto_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"//[^\n]*", "", h)
    protos = {}
    for m in re.finditer(r"\b(kamd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(re.split(r",(?![^()]*\))", args))
    assert len(protos) >= 60
    assert set(protos) == set(api._SYMBOLS)
    assert {k: len(v[1]) for k, v in api._SYMBOLS.items()} == protos
