// kamd_host.cpp -- host-side pieces of the quant driver that stay FP64/serial on the CPU, bit-exact with the reference:
// the fragment-length model and effective lengths (src/MinCollector.cpp:629-651, src/weights.cpp:7-79,248-271),
// counts_to_tpm (src/PlaintextWriter.cpp:5-27), the host read packer, and error reporting.
#include "../../include/kallisto_amd.h"
#include "kamd_host.h"

#include <cmath>
#include <cstring>
#include <random>
#include <vector>

namespace kamd {
static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace kamd

extern "C" const char* kamd_last_error(void) { return kamd::g_err.c_str(); }

extern "C" uint64_t kamd_packed_record_words(int32_t max_len) {
  return (uint64_t)((max_len + 15) / 16 + 1) + (uint64_t)((max_len + 31) / 32 + 1);
}

extern "C" int kamd_pack_reads_host(const char* seqs, const uint64_t* off, const int32_t* len, uint64_t n_reads,
                                    int32_t max_len, uint32_t* out_words, uint16_t* out_len) {
  return kamd_pack_reads_host_strided(seqs, off, len, n_reads, max_len, out_words, out_len, 1, 0);
}

extern "C" int kamd_pack_reads_host_strided(const char* seqs, const uint64_t* off, const int32_t* len, uint64_t n_reads,
                                            int32_t max_len, uint32_t* out_words, uint16_t* out_len, uint64_t rec_stride,
                                            uint64_t rec_first) {
  if (max_len <= 0 || max_len > 65535) return kamd::fail(-1, "kamd_pack_reads: max_len must be in [1, 65535]");
  if (rec_stride == 0) return kamd::fail(-1, "kamd_pack_reads: record stride must be positive");
  const uint64_t sw = (uint64_t)(max_len + 15) / 16 + 1, rec = kamd_packed_record_words(max_len);
  // byte -> 2-bit code, 4 = not ACGT (case-insensitive: KmerIterator.cpp:12 masks with 0xDF)
  static const struct Lut { uint8_t v[256]; Lut() { memset(v, 4, sizeof v); v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } } lut;
  for (uint64_t r = 0; r < n_reads; r++) {
    if (len[r] < 0 || len[r] > max_len) return kamd::fail(-1, "kamd_pack_reads: read longer than max_len");
    const uint64_t slot = rec_first + r * rec_stride;
    uint32_t* w = out_words + slot * rec;
    const unsigned char* s = (const unsigned char*)(seqs + off[r]);
    const int32_t L = len[r];
    // 32 bases per step: two sequence words and one mask word, built in registers and stored once
    int32_t i = 0;
    uint64_t wi = 0, mi = 0;
    uint32_t any_n = 0;
    for (; i + 32 <= L; i += 32) {
      uint64_t bits = 0; uint32_t m = 0;
      for (int j = 0; j < 32; j++) { const uint32_t c = lut.v[s[i + j]]; bits |= (uint64_t)(c & 3u) << (2 * j); m |= (c >> 2) << j; }
      w[wi++] = (uint32_t)bits; w[wi++] = (uint32_t)(bits >> 32); w[sw + mi++] = m; any_n |= m;
    }
    {
      uint64_t bits = 0; uint32_t m = 0;
      for (int j = 0; i + j < L; j++) { const uint32_t c = lut.v[s[i + j]]; bits |= (uint64_t)(c & 3u) << (2 * j); m |= (c >> 2) << j; }
      if (wi < sw) w[wi++] = (uint32_t)bits;
      if (wi < sw) w[wi++] = (uint32_t)(bits >> 32);
      if (sw + mi < rec) w[sw + mi++] = m;
      any_n |= m;
    }
    while (wi < sw) w[wi++] = 0;             // padding words of the record
    while (sw + mi < rec) w[sw + mi++] = 0;
    w[sw - 1] = any_n ? 1u : 0u;             // flag word (kamd_core.h REC_FLAG_HAS_N): the bases never reach the last sequence word
    out_len[slot] = (uint16_t)L;
  }
  return 0;
}

extern "C" void kamd_mean_frag_lens_trunc(const uint32_t* flens, double* out) {  // src/MinCollector.cpp:629-651
  std::vector<int> counts(KAMD_MAX_FRAG_LEN, 0);
  std::vector<double> mass(KAMD_MAX_FRAG_LEN, 0.0);
  for (int i = 0; i < KAMD_MAX_FRAG_LEN; i++) out[i] = 0.0;
  counts[0] = (int)flens[0];
  for (size_t i = 1; i < KAMD_MAX_FRAG_LEN; ++i) {
    mass[i] = static_cast<double>((size_t)flens[i] * i) + mass[i - 1];
    counts[i] = (int)flens[i] + counts[i - 1];
    if (counts[i] > 0) out[i] = mass[i] / static_cast<double>(counts[i]);
  }
}

extern "C" void kamd_trunc_gaussian_fld(int32_t start, int32_t stop, double mean, double sd, double* out) {  // src/weights.cpp:248-271
  size_t n = (size_t)(stop - start);
  double total_mass = 0.0, total_density = 0.0;
  for (size_t i = 0; i < n; ++i) {
    out[i] = 0.0;
    double x = static_cast<double>(start + (int)i);
    x = (x - mean) / sd;
    double cur_density = std::exp(-0.5 * x * x) / sd;
    total_mass += cur_density * i;
    total_density += cur_density;
    if (total_mass > 0) out[i] = total_mass / total_density;
  }
}

extern "C" void kamd_eff_lens(const int32_t* target_lens, uint64_t n, const double* t, double* eff) {  // src/weights.cpp:7-28,58-79
  const double marginal = t[KAMD_MAX_FRAG_LEN - 1];
  for (uint64_t i = 0; i < n; i++) {
    uint32_t len = (uint32_t)target_lens[i];
    double mean = len >= KAMD_MAX_FRAG_LEN ? marginal : t[len];
    double cur = static_cast<double>(len);
    double e = cur - mean + 1;
    if (e < 1.0) e = cur;
    eff[i] = e;
  }
}

extern "C" void kamd_bootstrap_seeds(uint64_t seed, int32_t n, uint64_t* seeds) {  // src/main.cpp:2746-2752
  std::mt19937_64 rand;
  rand.seed(seed);
  for (int32_t s = 0; s < n; ++s) seeds[s] = rand();
}

extern "C" void kamd_counts_to_tpm(const double* est, const double* eff, uint64_t n, double* tpm) {  // src/PlaintextWriter.cpp:5-27
  double total = 0.0;
  for (uint64_t i = 0; i < n; i++) { tpm[i] = est[i] / eff[i]; total += tpm[i]; }
  for (uint64_t i = 0; i < n; i++) tpm[i] = (tpm[i] / total) * 1e6;
}
