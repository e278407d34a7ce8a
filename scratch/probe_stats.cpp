// scratch analysis: which probes of kernel A go to the k-mer table, by phase / outcome / position relative to the previous hit
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../include/kallisto_amd.h"
#include "../kallisto_amd/csrc/kamd_core.h"
using namespace kamd;
static Table mk(const kamd_index_view* v) {
  Table t{v->table, v->n_buckets};
  t.dslots = v->dtable; t.n_dbuckets = v->n_dbuckets; t.dummy_uec = v->dummy_uec; t.dummy_slot = v->dummy_slot; t.dummy_strand = v->dummy_strand != 0;
  t.partial = false; t.no_jump = false;
  return t;
}
int main(int argc, char** argv) {
  kamd_index* idx; if (kamd_index_load(argv[1], 4, &idx)) { fprintf(stderr, "%s\n", kamd_last_error()); return 1; }
  kamd_index_view v; kamd_index_get_view(idx, &v);
  std::ifstream in(argv[2]); std::string line; std::string cat; std::vector<uint64_t> off; std::vector<int32_t> len;
  while (std::getline(in, line)) { off.push_back(cat.size()); len.push_back((int)line.size()); cat += line; }
  const int max_len = 100; const uint64_t rec = kamd_packed_record_words(max_len), sw = (max_len + 15) / 16 + 1;
  std::vector<uint32_t> words(off.size() * rec); std::vector<uint16_t> lens(off.size());
  kamd_pack_reads_host(cat.data(), off.data(), len.data(), off.size(), max_len, words.data(), lens.data());
  const Table t = mk(&v);
  uint32_t uecbuf[1024];
  // counters
  uint64_t n_reads = 0, text_ok = 0, text_fail = 0;
  uint64_t tab[5][2] = {{0}};            // [phase][found]
  uint64_t scan_first = 0, scan_after_hit_found_same_unitig_next = 0, scan_after_hit_found_other = 0, scan_after_hit_miss = 0, scan_after_miss = 0;
  uint64_t scan_same_unitig_any = 0;
  for (uint64_t r = 0; r < off.size(); r++) {
    ++n_reads;
    ReadView rv{words.data() + r * rec, words.data() + r * rec + sw, lens[r]};
    rv.has_n = (rv.seq[sw - 1] & REC_FLAG_HAS_N) != 0;
    UecList ul{uecbuf, 1024, 0, false}; MateFirst mf{0, 0, -1, false};
    MatchState st; match_init(st, rv, v.k);
    bool first = true; bool prev_found = false; uint32_t prev_gpos = 0; bool prev_strand = false; int prev_w = -1; bool have_prev_hit = false;
    while (st.phase != PH_DONE) {
      bool fc; uint64_t canon = window_canon(rv, st.w, v.k, &fc);
      Probe p; p.found = false;
      const int phase = st.phase, w = st.w;
      if (text_applies(st)) {
        if (text_canon(v.utext, text_pos_of(st), v.k) == canon) { p.found = true; p.strand = st.um_strand; p.uec = st.um_uec; p.dist = 0; p.slot = 0; p.gpos = 0; ++text_ok; }
        else { st.text_tried = true; ++text_fail; continue; }
      } else {
        p = probe_table(t, canon, fc, nullptr);
        ++tab[phase < 5 ? phase : 4][p.found ? 1 : 0];
        if (phase == PH_SCAN) {
          if (first) ++scan_first;
          else if (!prev_found) ++scan_after_miss;
          else if (!p.found) ++scan_after_hit_miss;
          else {
            // is this hit the text neighbour of the last table/text hit we know the position of?
            bool same = false;
            if (have_prev_hit) { const int d = w - prev_w; const uint32_t expect = prev_strand ? prev_gpos + d : prev_gpos - d; same = (p.gpos == expect) && (p.strand == prev_strand); }
            if (same) ++scan_after_hit_found_same_unitig_next; else ++scan_after_hit_found_other;
          }
          if (p.found && have_prev_hit) { const int d = w - prev_w; const uint32_t expect = prev_strand ? prev_gpos + d : prev_gpos - d; if (p.gpos == expect && p.strand == prev_strand) ++scan_same_unitig_any; }
        }
        first = false;
      }
      prev_found = p.found;
      if (p.found && !(p.gpos == 0 && p.slot == 0)) { have_prev_hit = true; prev_gpos = p.gpos; prev_strand = p.strand; prev_w = w; }
      else if (p.found && have_prev_hit) { /* text answer: position follows from the previous hit */ const int d = w - prev_w; prev_gpos = prev_strand ? prev_gpos + d : prev_gpos - d; prev_w = w; }
      match_feed<false>(st, rv, v.k, p, ul, 0, mf, t);
    }
  }
  const char* names[5] = {"SCAN", "JUMP", "MIDDLE", "BACKOFF", "other"};
  printf("reads %llu  text ok %.3f fail %.3f per read\n", (unsigned long long)n_reads, (double)text_ok / n_reads, (double)text_fail / n_reads);
  double tot = 0;
  for (int ph = 0; ph < 5; ph++) { printf("table %-8s found %.3f  miss %.3f per read\n", names[ph], (double)tab[ph][1] / n_reads, (double)tab[ph][0] / n_reads); tot += tab[ph][0] + tab[ph][1]; }
  printf("table probes per read %.3f\n", tot / n_reads);
  printf("SCAN: first %.3f | after a miss %.3f | after a hit: miss %.3f, found at the text neighbour (same unitig) %.3f, found elsewhere %.3f  [any SCAN hit on the text continuation %.3f]\n",
         (double)scan_first / n_reads, (double)scan_after_miss / n_reads, (double)scan_after_hit_miss / n_reads, (double)scan_after_hit_found_same_unitig_next / n_reads,
         (double)scan_after_hit_found_other / n_reads, (double)scan_same_unitig_any / n_reads);
  return 0;
}
