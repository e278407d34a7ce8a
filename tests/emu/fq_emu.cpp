// tests/emu/fq_emu.cpp -- TEST INFRASTRUCTURE: the input side of the device FASTQ parser on a box without a GPU.  The host part
// (kamd_textsource.h: TextSource, UnitCutter) is the product's own code; the device part (k_fq_count / k_fq_fill / k_fq_records of
// kamd_io.hip) is restated serially over the same host/device functions of kamd_fq_core.h.  Never linked into the product.
#include "../../kallisto_amd/csrc/kamd_fq_core.h"
#include "../../kallisto_amd/csrc/kamd_textsource.h"

namespace {
// newline positions of a text the way the kernels find them: 16 bytes per "thread" through nl_mask4, byte-wise at the tail
void emu_nlpos(const char* text, uint64_t n, std::vector<uint32_t>& pos) {
  pos.clear();
  for (uint64_t b0 = 0; b0 < n; b0 += 16) {
    uint32_t m = 0;
    if (b0 + 16 <= n) {
      uint32_t w[4];
      memcpy(w, text + b0, 16);
      m = kamd_fq::nl_mask4(w[0]) | (kamd_fq::nl_mask4(w[1]) << 4) | (kamd_fq::nl_mask4(w[2]) << 8) | (kamd_fq::nl_mask4(w[3]) << 12);
    } else {
      for (uint32_t i = 0; i < 16 && b0 + i < n; i++) if (text[b0 + i] == '\n') m |= 1u << i;
    }
    for (; m; m &= m - 1) pos.push_back((uint32_t)(b0 + (uint64_t)__builtin_ctz(m)));
  }
}
}  // namespace

extern "C" {
// one unit: status 0 ok / 1 some record not strict (first_bad) / 2 fewer lines than promised / 3 read too long; on 0 the sequences
// (mate 1, mate 2 of record j adjacent) are appended to out + *o, each followed by '\n'; -1: out too small
int fq_emu_unit(const char* const* text, const uint64_t* n_bytes, int n_files, uint64_t n_records, char* out, uint64_t cap, uint64_t* o,
                uint32_t* max_len, uint64_t* first_bad) {
  std::vector<uint32_t> pos[2];
  for (int f = 0; f < n_files; f++) { emu_nlpos(text[f], n_bytes[f], pos[f]); if (pos[f].size() < 4 * n_records) return 2; }
  uint32_t mx = 0;
  *first_bad = ~0ULL;
  std::vector<kamd_fq::Record> recs(n_records * (uint64_t)n_files);
  for (uint64_t j = 0; j < n_records; j++)
    for (int f = 0; f < n_files; f++) {
      const uint32_t* nl = pos[f].data() + 4 * j;
      const uint64_t l0 = j ? (uint64_t)nl[-1] + 1 : 0;
      const kamd_fq::Record r = kamd_fq::fq_check_record(text[f], l0, nl[0], nl[1], nl[2], nl[3]);
      if (!r.ok) { if (*first_bad == ~0ULL) *first_bad = j; }
      else mx = std::max(mx, r.seq_len);
      recs[j * n_files + f] = r;
    }
  if (*first_bad != ~0ULL) return 1;
  *max_len = std::max(*max_len, mx);
  if (mx > kamd_fq::FQ_MAX_READ) return 3;
  for (uint64_t j = 0; j < n_records; j++)
    for (int f = 0; f < n_files; f++) {
      const kamd_fq::Record& r = recs[j * n_files + f];
      if (*o + r.seq_len + 1 > cap) return -1;
      memcpy(out + *o, text[f] + r.seq_off, r.seq_len); *o += r.seq_len; out[(*o)++] = '\n';
    }
  return 0;
}

// files -> TextSource (rings from malloc) -> UnitCutter -> fq_emu_unit.  Returns the bytes written to out (sequences, '\n' after
// each; paired: mates interleaved), or a negative code: the cutter's (-1 not strict, -2 record counts differ, -3 io, -4 too long),
// -10 - status for a unit the parser declines, -100 out too small.
int64_t io_units(const char* path0, const char* path1, uint64_t ring_bytes, uint64_t target, uint64_t max_bytes, int threads, uint64_t blk,
                 char* out, uint64_t cap, uint64_t* n_records, uint64_t* n_units, uint32_t* max_len) {
  const int nf = path1 && path1[0] ? 2 : 1;
  std::vector<char> ring[2];
  std::unique_ptr<kamd_io::TextSource> src[2];
  for (int f = 0; f < nf; f++) {
    ring[f].resize(ring_bytes);
    src[f].reset(new kamd_io::TextSource(f ? path1 : path0, ring[f].data(), ring_bytes, threads, blk));
    if (src[f]->failed()) return -3;
  }
  kamd_io::UnitCutter cut(src[0].get(), nf == 2 ? src[1].get() : nullptr, target, max_bytes);
  kamd_io::UnitCut u;
  uint64_t o = 0;
  *n_records = 0; *n_units = 0; *max_len = 0;
  std::vector<char> text[2];
  for (;;) {
    const int rc = cut.next(u);
    if (rc == kamd_io::UnitCutter::DONE) break;
    if (rc < 0) return rc;
    const char* tp[2]; uint64_t tn[2];
    for (int f = 0; f < nf; f++) {
      text[f].resize((size_t)(u.end[f] - u.begin[f]));
      const char* p[2]; size_t n[2];
      const int np = src[f]->pieces(u.begin[f], u.end[f], p, n);
      size_t at = 0;
      for (int i = 0; i < np; i++) { memcpy(text[f].data() + at, p[i], n[i]); at += n[i]; }
      src[f]->release(u.end[f]);
      tp[f] = text[f].data(); tn[f] = text[f].size();
    }
    uint64_t first_bad = 0;
    const int st = fq_emu_unit(tp, tn, nf, u.n_records, out, cap, &o, max_len, &first_bad);
    if (st == -1) return -100;
    if (st) return -10 - st;
    *n_records += u.n_records; ++*n_units;
  }
  return (int64_t)o;
}
uint64_t io_count_newlines(const char* p, uint64_t n) { return kamd_io::count_newlines(p, (size_t)n); }
uint64_t io_after_kth_newline(const char* p, uint64_t n, uint64_t k) { return (uint64_t)kamd_io::after_kth_newline(p, (size_t)n, k); }
// copy + count in one pass (streaming stores where the CPU has AVX2): dst receives src, the newlines of src are returned
uint64_t io_copy_count_newlines(char* dst, const char* src, uint64_t n) { return kamd_io::copy_count_newlines(dst, src, (size_t)n); }
uint64_t io_count_newlines_sse2(const char* p, uint64_t n) { return kamd_io::count_newlines_sse2(p, (size_t)n); }
}
