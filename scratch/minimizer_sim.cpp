// scratch analysis (CPU): how many 64-byte table lines would kernel A's probes read if the k-mer table were bucketed by MINIMIZER
// instead of by k-mer (same 3-slot lines, same Robin-Hood order by home, load factor 0.5), with a per-lane cache of the last K lines?
// usage: minimizer_sim index reads.txt      (reads: one per line, mates interleaved)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>
#include "../include/kallisto_amd.h"
#include "../kallisto_amd/csrc/kamd_core.h"
using namespace kamd;
static int K;   // k-mer length
static uint64_t minimizer_hash(uint64_t key, int m) {   // key: canonical k-mer, MSB-first right-aligned; m == K: the k-mer itself
  if (m >= K) return mix64(key);
  uint64_t best = ~0ULL, best_x = 0;
  const uint64_t mask = (1ULL << (2 * m)) - 1;
  for (int j = 0; j + m <= K; j++) {
    const uint64_t x = (key >> (2 * (K - m - j))) & mask;
    const uint64_t rc = revcomp_msb(x, m);
    const uint64_t c = x < rc ? x : rc;
    const uint64_t h = mix64(c + 0x9e3779b97f4a7c15ULL);
    if (h < best) { best = h; best_x = c; }
  }
  return mix64(best_x ^ 0x5851f42d4c957f2dULL);   // (the minimum of several hashes is not uniform: hash the chosen m-mer again)
}
struct Sim {
  int m; uint64_t n_lines;
  std::vector<uint64_t> line_key;    // 3 per line, KEY_EMPTY if none
  std::vector<uint32_t> line_home;   // home line of each stored key
  uint64_t home_of(uint64_t key) const { return (uint64_t)(((unsigned __int128)minimizer_hash(key, m) * n_lines) >> 64); }
  void build(const std::vector<uint64_t>& keys) {
    const uint64_t N = keys.size();
    n_lines = std::max<uint64_t>(16, (2 * N + 2) / 3);
    std::vector<std::pair<uint64_t, uint64_t>> hk(N);
    for (uint64_t i = 0; i < N; i++) hk[i] = {home_of(keys[i]), keys[i]};
    std::sort(hk.begin(), hk.end());
    const uint64_t total = n_lines + N / 3 + 64;
    line_key.assign(total * 3, KEY_EMPTY); line_home.assign(total * 3, 0xFFFFFFFFu);
    uint64_t cursor = 0;
    for (uint64_t i = 0; i < N; i++) { cursor = std::max(cursor, hk[i].first * 3); line_key[cursor] = hk[i].second; line_home[cursor] = (uint32_t)hk[i].first; ++cursor; }
  }
  // lines read for a lookup, not counting those in `cache` (LRU of the last lines this mate has read)
  int lookup(uint64_t key, bool* found, std::vector<uint64_t>& cache, size_t cache_cap) const {
    const uint64_t h = home_of(key);
    int reads = 0;
    *found = false;
    for (uint64_t b = h;; b++) {
      auto it = std::find(cache.begin(), cache.end(), b);
      if (it == cache.end()) { ++reads; if (cache_cap) { if (cache.size() >= cache_cap) cache.erase(cache.begin()); cache.push_back(b); } }
      else if (cache_cap) { cache.erase(it); cache.push_back(b); }
      bool stop = false;
      for (int j = 0; j < 3; j++) {
        const uint64_t kk = line_key[3 * b + j];
        if (kk == KEY_EMPTY) { stop = true; break; }
        if (kk == key) { *found = true; return reads; }
        if (line_home[3 * b + j] > h) { stop = true; break; }
      }
      if (stop) return reads;
    }
  }
};
int main(int argc, char** argv) {
  kamd_index* idx; if (kamd_index_load(argv[1], 4, &idx)) { fprintf(stderr, "%s\n", kamd_last_error()); return 1; }
  kamd_index_view v; kamd_index_get_view(idx, &v);
  K = v.k;
  std::vector<uint64_t> keys;
  for (uint64_t b = 0; b < v.n_buckets + v.pad_buckets; b++) for (int j = 0; j < 3; j++) { const uint64_t kk = v.table[8 * b + j] & KEY_MASK; if (kk != KEY_EMPTY) keys.push_back(kk); }
  fprintf(stderr, "%zu k-mers\n", keys.size());
  std::ifstream in(argv[2]); std::string line; std::string cat; std::vector<uint64_t> off; std::vector<int32_t> len;
  while (std::getline(in, line)) { off.push_back(cat.size()); len.push_back((int)line.size()); cat += line; }
  const int max_len = 100; const uint64_t rec = kamd_packed_record_words(max_len), sw = (max_len + 15) / 16 + 1;
  std::vector<uint32_t> words(off.size() * rec); std::vector<uint16_t> lens(off.size());
  kamd_pack_reads_host(cat.data(), off.data(), len.data(), off.size(), max_len, words.data(), lens.data());
  Table t{v.table, v.n_buckets};
  t.dslots = nullptr; t.n_dbuckets = 0; t.dummy_uec = 0; t.dummy_slot = 0; t.dummy_strand = false; t.partial = false; t.no_jump = false;
  uint32_t uecbuf[1024];
  const int ms[] = {31, 27, 25, 23, 21, 19, 17, 15, 13};
  const size_t caps[] = {0, 1, 2, 4};
  for (int m : ms) {
    Sim S; S.m = m; S.build(keys);
    // cluster statistics: k-mers per distinct home value is not tracked; lines per successful lookup tells the same story
    for (size_t cap : caps) {
      uint64_t n_reads = 0, probes = 0, reqs = 0, reqs_hit = 0, hits = 0, reqs_miss = 0;
      for (uint64_t r = 0; r < off.size(); r++) {
        ++n_reads;
        ReadView rv{words.data() + r * rec, words.data() + r * rec + sw, lens[r]};
        rv.has_n = (rv.seq[sw - 1] & REC_FLAG_HAS_N) != 0;
        UecList ul{uecbuf, 1024, 0, false}; MateFirst mf{0, 0, -1, false};
        MatchState st; match_init(st, rv, v.k);
        std::vector<uint64_t> cache;
        while (st.phase != PH_DONE) {
          bool fc; uint64_t canon = window_canon(rv, st.w, v.k, &fc);
          Probe p; p.found = false;
          if (text_applies(st)) {
            if (text_canon(v.utext, text_pos_of(st), v.k) == canon) { p.found = true; p.strand = st.um_strand; p.uec = st.um_uec; p.dist = 0; p.slot = 0; p.gpos = 0; }
            else { st.text_tried = true; continue; }
          } else {
            p = probe_table(t, canon, fc, nullptr);      // the real answer (payload) from the real table
            bool f2; const int rq = S.lookup(canon, &f2, cache, cap);
            if (f2 != p.found) { fprintf(stderr, "simulated table disagrees\n"); return 2; }
            ++probes; reqs += rq; if (p.found) { ++hits; reqs_hit += rq; } else reqs_miss += rq;
          }
          match_feed<false>(st, rv, v.k, p, ul, 0, mf, t);
        }
      }
      printf("m %2d cache %zu lines: table probes/mate %.3f  line requests/mate %.3f  (per hit %.2f, per miss %.2f)\n", m, cap, (double)probes / n_reads,
             (double)reqs / n_reads, (double)reqs_hit / std::max<uint64_t>(hits, 1), (double)reqs_miss / std::max<uint64_t>(probes - hits, 1));
      fflush(stdout);
    }
  }
  return 0;
}
