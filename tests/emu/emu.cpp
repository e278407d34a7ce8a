// tests/emu/emu.cpp -- TEST INFRASTRUCTURE: drives the per-item device logic of kallisto_amd/csrc/kamd_core.h on the CPU
// so that the match state machine, the table probe and the packed-read iterator can be checked against the oracle on a
// box without a GPU.  Never linked into libkallisto_amd.so.
#include "../../include/kallisto_amd.h"
#include "../../kallisto_amd/csrc/kamd_core.h"

#include <algorithm>
#include <cstring>
#include <vector>

// the k-mer table(s) of the host view as the per-item logic sees them (mirrors make_table of kamd_dev.h)
static kamd::Table emu_table(const kamd_index_view* v, bool partial, bool no_jump = false) {
  kamd::Table t{v->table, v->n_buckets};
  t.layout = (uint8_t)v->table_layout; t.q = (uint8_t)v->tag_q; t.dsh = (uint8_t)v->tag_dsh; t.tagw = (uint8_t)v->tag_w;
  t.dslots = v->dtable; t.n_dbuckets = v->n_dbuckets; t.dummy_uec = v->dummy_uec; t.dummy_slot = v->dummy_slot;
  t.dummy_strand = v->dummy_strand != 0; t.partial = partial; t.no_jump = no_jump;
  return t;
}

extern "C" {
// For every item: the sorted intersection of the collected transcript sets (what the GPU resolves later per tuple) after
// the on-list mask.  out_off[n_items+1], out_ids capacity cap; also per item n_hits of each mate and probe counts.
int64_t emu_pseudoalign(const kamd_index_view* v, const uint32_t* words, const uint16_t* lens, uint64_t n_items, int paired,
                        int32_t max_len, uint64_t* out_off, uint32_t* out_ids, uint64_t cap, int32_t* nhits,
                        uint64_t* probes, uint64_t* bucket_reads, uint32_t* tuple_sizes) {
  using namespace kamd;
  const uint64_t sw = (uint64_t)(max_len + 15) / 16 + 1, rec = kamd_packed_record_words(max_len);
  const Table t = emu_table(v, !paired);
  std::vector<uint8_t> nonempty(v->n_ecs);
  for (uint64_t e = 0; e < v->n_ecs; e++) nonempty[e] = v->ec_off[e + 1] > v->ec_off[e];
  uint64_t o = 0;
  uint32_t ecbuf[64];
  for (uint64_t i = 0; i < n_items; i++) {
    out_off[i] = o;
    EcList ecs{ecbuf, 64, 0, false};
    MateInfo m[2]; memset(m, 0, sizeof m);
    for (int mate = 0; mate < (paired ? 2 : 1); mate++) {
      uint64_t r = paired ? 2 * i + mate : i;
      ReadView rv{words + r * rec, words + r * rec + sw, lens[r]};
      match_mate(t, v->uec_ec, nonempty.data(), rv, v->k, ecs, m[mate]);
      *probes += m[mate].probes; *bucket_reads += m[mate].bucket_reads;
    }
    if (ecs.overflow) return -1;
    nhits[2 * i] = m[0].n_hits; nhits[2 * i + 1] = m[1].n_hits;
    tuple_sizes[i] = (uint32_t)ecs.n;
    if (!pair_is_mapped(m[0], m[1])) continue;
    std::vector<uint32_t> cur(v->ec_ids + v->ec_off[ecs.e[0]], v->ec_ids + v->ec_off[ecs.e[0] + 1]);
    for (int j = 1; j < ecs.n; j++) {
      std::vector<uint32_t> nx;
      std::set_intersection(cur.begin(), cur.end(), v->ec_ids + v->ec_off[ecs.e[j]], v->ec_ids + v->ec_off[ecs.e[j] + 1],
                            std::back_inserter(nx));
      cur.swap(nx);
    }
    for (uint32_t tr : cur)
      if (v->onlist_bits[tr >> 5] >> (tr & 31) & 1) { if (o >= cap) return -2; out_ids[o++] = tr; }
  }
  out_off[n_items] = o;
  return (int64_t)o;
}
}

// the resumable one-probe-per-step state machine (kernel A v2): per item the sorted distinct set ids it collects, the
// mapped flag and the probe count -- must equal what match_mate produces
extern "C" int64_t emu_tuples(const kamd_index_view* v, const uint32_t* words, const uint16_t* lens, uint64_t n_items, int paired,
                              int32_t max_len, int use_stepper, uint32_t* out, uint64_t stride, uint64_t* probes) {
  using namespace kamd;
  const uint64_t sw = (uint64_t)(max_len + 15) / 16 + 1, rec = kamd_packed_record_words(max_len);
  const Table t = emu_table(v, !paired, (use_stepper & 2) != 0);   // bit 1 of use_stepper: --no-jump
  const bool use_text = (use_stepper & 4) != 0;                     // bit 2: the unitig text in front of the table (kernel A v3)
  const bool append = (use_stepper & 8) != 0;                       // bit 3: the append-only class list of kernel A's second pass (round 6): duplicates
                                                                    // stay in the list unless they are neighbours, uecs_to_ecs removes them
  use_stepper &= 1;
  std::vector<uint8_t> nonempty(v->n_ecs);
  for (uint64_t e = 0; e < v->n_ecs; e++) nonempty[e] = v->ec_off[e + 1] > v->ec_off[e];
  uint32_t ecbuf[1024], uecbuf[1024];
  for (uint64_t i = 0; i < n_items; i++) {
    EcList ecs{ecbuf, 1024, 0, false};
    bool mapped;
    if (!use_stepper) {
      MateInfo m[2]; memset(m, 0, sizeof m);
      for (int mate = 0; mate < (paired ? 2 : 1); mate++) {
        uint64_t r = paired ? 2 * i + mate : i;
        ReadView rv{words + r * rec, words + r * rec + sw, lens[r]};
        match_mate(t, v->uec_ec, nonempty.data(), rv, v->k, ecs, m[mate]);
        *probes += m[mate].probes;
      }
      mapped = pair_is_mapped(m[0], m[1]);
    } else {
      UecList ul{uecbuf, 1024, 0, false};
      ul.append = append;
      MateFirst mf[2] = {{0, 0, -1, false}, {0, 0, -1, false}};
      for (int mate = 0; mate < (paired ? 2 : 1); mate++) {
        uint64_t r = paired ? 2 * i + mate : i;
        ReadView rv{words + r * rec, words + r * rec + sw, lens[r]};
        // the packers' has-N flag word decides whether the mask plane is consulted (as in kernel A v3)
        rv.has_n = (rv.seq[sw - 1] & REC_FLAG_HAS_N) != 0;
        MatchState st; match_init(st, rv, v->k);
        while (st.phase != PH_DONE) {
          bool fc; uint64_t canon = window_canon(rv, st.w, v->k, &fc);
          Probe p; p.found = false;
          if (use_text && text_applies(st)) {
            if (text_canon(v->utext, text_pos_of(st), v->k) == canon) {
              p.found = true; p.strand = st.um_strand; p.uec = st.um_uec; p.dist = 0; p.slot = 0; p.gpos = 0;
              ++*probes; ++probes[1];
            } else { st.text_tried = true; continue; }
          } else if (use_text) {   // as kernel A v3: one bucket per step, a continue flag costs another step
            const Table pt = phase_table(t, st.phase);
            const uint32_t h = kmer_hash32(canon);
            const uint64_t b = bucket_of_hash(h, pt.n_buckets) + st.disp;
            if (pt.layout == LAYOUT_COMPACT) {   // (the D-list table is wide: phase_table hands it out with layout 0)
              if (match_bucket_compact(load_bucket(pt.slots, b), pt, compact_tag(pt, canon, h, st.disp), fc, b, p) == BUCKET_CONTINUE &&
                  st.disp < compact_max_disp(pt)) { ++st.disp; continue; }
            } else if (match_bucket(load_bucket(pt.slots, b), canon, fc, b, p) == BUCKET_CONTINUE) { ++st.disp; continue; }
            if (st.phase != PH_DLIST) ++*probes;
          } else {
            p = probe_table(phase_table(t, st.phase), canon, fc, nullptr);
            if (st.phase != PH_DLIST) ++*probes;   // dbg.find calls of match() only
          }
          if (t.n_dbuckets) match_feed<true>(st, rv, v->k, p, ul, mate, mf[mate], t); else match_feed<false>(st, rv, v->k, p, ul, mate, mf[mate], t);
        }
      }
      bool ne0, ne1;
      if (append) { probes[2] += (uint64_t)ul.n; if (ul.overflow) return -2; }
      uecs_to_ecs(ul.e, ul.n, v->uec_ec, nonempty.data(), ecs, &ne0, &ne1);
      MateInfo a, b; a.n_hits = mf[0].n_hits; a.n_nonempty = ne0; b.n_hits = mf[1].n_hits; b.n_nonempty = ne1;
      mapped = pair_is_mapped(a, b);
    }
    uint32_t* o = out + i * stride;
    o[0] = mapped ? (uint32_t)ecs.n : 0xFFFFFFFFu;
    for (int j = 0; j < ecs.n && (uint64_t)j + 1 < stride; j++) o[1 + j] = ecs.e[j];
  }
  return 0;
}

// same as emu_pseudoalign, plus the positional filters of processBuffer (fragment-length compatibility via findPosition,
// strand specificity) evaluated by kamd_core.h's keep_transcript on each member of the intersection
extern "C" int64_t emu_pseudoalign_opts(const kamd_index_view* v, const uint32_t* words, const uint16_t* lens, uint64_t n_items,
                                        int paired, int32_t max_len, int single_overhang, int strand, int fl, int has_mean_fl,
                                        int no_jump, uint64_t* out_off, uint32_t* out_ids, uint64_t cap) {
  using namespace kamd;
  const bool do_union = (no_jump & 2) != 0;   // bit 1 of `no_jump`: --union
  no_jump &= 1;
  const uint64_t sw = (uint64_t)(max_len + 15) / 16 + 1, rec = kamd_packed_record_words(max_len);
  const Table t = emu_table(v, !paired && !do_union, no_jump != 0);   // --union: match(..., partial = false) (KmerIndex.cpp:1704)
  PosTables pt{v->unitig_blk_off, v->unitig_len, v->blk_unitig, v->blk_lb, v->blk_ub, v->blk_ec, v->blk_pos_off, v->blk_posw,
               v->blk_sense, v->ec_off, v->ec_ids, v->target_lens, v->k};
  const SetTables stt{v->ec_off, v->ec_ids};
  std::vector<uint8_t> nonempty(v->n_ecs);
  for (uint64_t e = 0; e < v->n_ecs; e++) nonempty[e] = v->ec_off[e + 1] > v->ec_off[e];
  uint64_t o = 0;
  uint32_t ecbuf[1024], curbuf[1024], hbuf[1024];
  const bool comprehensive = strand != 0 && (no_jump || do_union);   // ProcessReads.cpp:1139-1140
  for (uint64_t i = 0; i < n_items; i++) {
    out_off[i] = o;
    EcList ecs{ecbuf, 1024, 0, false};
    MateInfo m[2]; memset(m, 0, sizeof m);
    HitBlocks hb{v->slot_block, hbuf, 1024, 0, false};
    for (int mate = 0; mate < (paired ? 2 : 1); mate++) {
      uint64_t r = paired ? 2 * i + mate : i;
      ReadView rv{words + r * rec, words + r * rec + sw, lens[r]};
      match_mate(t, v->uec_ec, nonempty.data(), rv, v->k, ecs, m[mate], do_union ? (mate ? EC_MATE2 : EC_MATE1) : 0u,
                 (comprehensive && mate == 0) ? &hb : nullptr);
    }
    if (ecs.overflow || hb.overflow) return -1;
    if (!pair_is_mapped(m[0], m[1])) continue;
    FirstHit h[2];
    for (int mate = 0; mate < 2; mate++) {
      h[mate].valid = m[mate].n_hits > 0;
      h[mate].block = h[mate].valid ? v->slot_block[m[mate].first_slot] : 0;
      h[mate].dist = h[mate].valid ? v->slot_dist[m[mate].first_slot] : 0;
      h[mate].strand = m[mate].first_strand; h[mate].pos = m[mate].first_pos;
    }
    FilterCfg cfg;
    cfg.fraglen = !single_overhang && has_mean_fl && (!paired || m[0].n_hits == 0 || m[1].n_hits == 0);
    cfg.fl = fl; cfg.strand = strand;
    cfg.comprehensive = comprehensive; cfg.hits1 = hbuf; cfg.n_hits1 = hb.n;
    int64_t err = 0;
    for_each_in_set(stt, ecs, do_union, curbuf, [&](uint32_t tr) {
      if (!(v->onlist_bits[tr >> 5] >> (tr & 31) & 1)) return;
      if ((cfg.fraglen || cfg.strand) && !keep_transcript(pt, cfg, h[0], h[1], tr)) return;
      if (o >= cap) { err = -2; return; }
      out_ids[o++] = tr;
    });
    if (err) return err;
  }
  out_off[n_items] = o;
  return (int64_t)o;
}

// ---- CPU stand-ins for kernel A's outputs and for kamd_ec_finalize, used to test the multi-rank merge on gloo ----
#include <map>
extern "C" {
// dense[n_ecs] += single-set items; tuple records [1, m, e0..] appended to stream (capacity cap words)
int64_t emu_ec_state(const kamd_index_view* v, const uint32_t* words, const uint16_t* lens, uint64_t n_items, int paired,
                     int32_t max_len, uint32_t* dense, uint32_t* stream, uint64_t cap, uint64_t* rec_off, uint64_t* n_recs) {
  using namespace kamd;
  const uint64_t sw = (uint64_t)(max_len + 15) / 16 + 1, rec = kamd_packed_record_words(max_len);
  const Table t = emu_table(v, !paired);
  std::vector<uint8_t> nonempty(v->n_ecs);
  for (uint64_t e = 0; e < v->n_ecs; e++) nonempty[e] = v->ec_off[e + 1] > v->ec_off[e];
  uint64_t o = 0, nr = 0;
  uint32_t ecbuf[1024];
  for (uint64_t i = 0; i < n_items; i++) {
    EcList ecs{ecbuf, 1024, 0, false};
    MateInfo m[2]; memset(m, 0, sizeof m);
    for (int mate = 0; mate < (paired ? 2 : 1); mate++) {
      uint64_t r = paired ? 2 * i + mate : i;
      ReadView rv{words + r * rec, words + r * rec + sw, lens[r]};
      match_mate(t, v->uec_ec, nonempty.data(), rv, v->k, ecs, m[mate]);
    }
    if (ecs.overflow) return -1;
    if (!pair_is_mapped(m[0], m[1])) continue;
    if (ecs.n == 1) { dense[ecs.e[0]]++; continue; }
    if (o + ecs.n + 2 > cap) return -2;
    rec_off[nr++] = o;
    stream[o++] = 1; stream[o++] = (uint32_t)ecs.n;
    for (int j = 0; j < ecs.n; j++) stream[o++] = ecs.e[j];
  }
  *n_recs = nr;
  return (int64_t)o;
}
// resolve dense counts + weighted tuple records into the EC multiset (CSR); returns number of ECs
int64_t emu_resolve(const kamd_index_view* v, const uint32_t* dense, const uint32_t* stream, const uint64_t* rec_off,
                    uint64_t n_recs, uint64_t* ec_off, uint32_t* ec_ids, uint32_t* counts, uint64_t cap_ecs, uint64_t cap_ids) {
  std::map<std::vector<uint32_t>, uint64_t> tuples, sets;
  for (uint64_t r = 0; r < n_recs; r++) {
    const uint32_t* w = stream + rec_off[r];
    tuples[std::vector<uint32_t>(w + 2, w + 2 + w[1])] += w[0];
  }
  auto onl = [&](uint32_t tr) { return (v->onlist_bits[tr >> 5] >> (tr & 31)) & 1; };
  for (uint64_t e = 0; e < v->n_ecs; e++) if (dense[e]) {
    std::vector<uint32_t> s;
    for (uint64_t j = v->ec_off[e]; j < v->ec_off[e + 1]; j++) if (onl(v->ec_ids[j])) s.push_back(v->ec_ids[j]);
    if (!s.empty()) sets[s] += dense[e];
  }
  for (auto& kv : tuples) {
    const auto& es = kv.first;
    std::vector<uint32_t> cur(v->ec_ids + v->ec_off[es[0]], v->ec_ids + v->ec_off[es[0] + 1]);
    for (size_t j = 1; j < es.size(); j++) {
      std::vector<uint32_t> nx;
      std::set_intersection(cur.begin(), cur.end(), v->ec_ids + v->ec_off[es[j]], v->ec_ids + v->ec_off[es[j] + 1], std::back_inserter(nx));
      cur.swap(nx);
    }
    std::vector<uint32_t> s;
    for (uint32_t tr : cur) if (onl(tr)) s.push_back(tr);
    if (!s.empty()) sets[s] += kv.second;
  }
  uint64_t n = 0, o = 0;
  ec_off[0] = 0;
  for (auto& kv : sets) {
    if (n >= cap_ecs || o + kv.first.size() > cap_ids) return -1;
    for (uint32_t tr : kv.first) ec_ids[o++] = tr;
    counts[n++] = (uint32_t)kv.second;
    ec_off[n] = o;
  }
  return (int64_t)n;
}
}

// every k-mer of every unitig looked up through the product's own probe (kamd_core.h probe_table): found, with its own text
// position, offset on its unitig and a block of that unitig that covers it.  Returns the number of k-mers that fail (0 = all
// good), -1 if the number of k-mers walked differs from the index's count.  *lines: bucket lines read in all.
extern "C" int64_t emu_verify_table(const kamd_index_view* v, uint64_t* lines) {
  const kamd::Table t = emu_table(v, false);
  uint64_t bad = 0, n = 0, ll = 0;
  for (uint64_t u = 0; u < v->n_unitigs; u++) {
    const uint64_t g0 = v->unitig_gpos[u], len = v->unitig_len[u];
    for (uint64_t d = 0; d + (uint64_t)v->k <= len; d++) {
      const uint32_t g = (uint32_t)(g0 + d);
      const uint64_t cn = kamd::text_canon(v->utext, g, v->k);
      uint32_t reads = 0;
      const kamd::Probe p = kamd::probe_table(t, cn, true, &reads);
      ++n; ll += reads;
      if (!p.found || p.gpos != g || v->slot_dist[p.slot] != d) { ++bad; continue; }
      const uint32_t blk = v->slot_block[p.slot];
      if (v->blk_unitig[blk] != u || !(v->blk_lb[blk] <= d && d < v->blk_ub[blk])) { ++bad; continue; }
      // the payload: strand, distance to the end of the block in either read direction (saturated at 16 bits), the block's set
      const kamd::TextWords w = kamd::load_text(v->utext, g);
      const int sh = (int)(g & 15u) * 2;
      uint64_t x = ((uint64_t)w.a | ((uint64_t)w.b << 32)) >> sh;
      if (sh + 2 * v->k > 64) x |= (uint64_t)w.c << (64 - sh);
      x &= (1ULL << (2 * v->k)) - 1;
      const uint64_t fwd = kamd::rev_bases64(x) >> (64 - 2 * v->k), rc = (~x) & ((1ULL << (2 * v->k)) - 1);
      const bool fwd_canon = fwd < rc;
      const kamd::Probe pf = kamd::probe_table(t, cn, fwd_canon, nullptr), pb = kamd::probe_table(t, cn, !fwd_canon, nullptr);
      const uint32_t rem_f = std::min<uint32_t>(v->blk_ub[blk] - 1 - (uint32_t)d, 65535u), rem_b = std::min<uint32_t>((uint32_t)d - v->blk_lb[blk], 65535u);
      if (!pf.strand || pb.strand || pf.dist != rem_f || pb.dist != rem_b || v->uec_ec[pf.uec] != v->blk_ec[blk]) ++bad;
    }
  }
  if (lines) *lines = ll;
  return n == v->n_kmers ? (int64_t)bad : -1;
}
