// tests/emu/emu.cpp -- TEST INFRASTRUCTURE: drives the per-item device logic of kallisto_amd/csrc/kamd_core.h on the CPU
// so that the match state machine, the table probe and the packed-read iterator can be checked against the oracle on a
// box without a GPU.  Never linked into libkallisto_amd.so.
#include "../../include/kallisto_amd.h"
#include "../../kallisto_amd/csrc/kamd_core.h"

#include <algorithm>
#include <cstring>
#include <vector>

extern "C" {
// For every item: the sorted intersection of the collected transcript sets (what the GPU resolves later per tuple) after
// the on-list mask.  out_off[n_items+1], out_ids capacity cap; also per item n_hits of each mate and probe counts.
int64_t emu_pseudoalign(const kamd_index_view* v, const uint32_t* words, const uint16_t* lens, uint64_t n_items, int paired,
                        int32_t max_len, uint64_t* out_off, uint32_t* out_ids, uint64_t cap, int32_t* nhits,
                        uint64_t* probes, uint64_t* bucket_reads, uint32_t* tuple_sizes) {
  using namespace kamd;
  const uint64_t sw = (uint64_t)(max_len + 15) / 16 + 1, rec = kamd_packed_record_words(max_len);
  Table t{v->table, v->n_buckets};
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
