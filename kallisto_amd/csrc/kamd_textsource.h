// kamd_textsource.h -- input side of the device FASTQ parser (host code only, no HIP: tests/emu drives it on a box without a GPU).
//
// FastqSequenceReader::fetchSequences (src/ProcessReads.cpp:3128-3267) parses under a lock, one record at a time.  Here the host
// threads only MOVE bytes: they pread / inflate the text of a file into a ring of pinned memory, count its newlines on the way
// (TextSource) and cut the stream into units of whole 4-line records (UnitCutter).  Finding the lines, checking the records and
// 2-bit packing happen on the GPU (kamd_fastq_unit_pack, kamd_fq_core.h).  The readers of kamd_fastq.h remain the general path
// (FASTA, multi-line records, anything the strict parser declines).
#pragma once
#include <dlfcn.h>
#include <immintrin.h>

#include <atomic>
#include <deque>
#include <memory>

#include "kamd_fastq.h"
#include "kamd_pargzip.h"

namespace kamd_io {

// newlines in [p, p + n) (SSE2: 16 bytes per compare)
inline uint64_t count_newlines_sse2(const char* p, size_t n) {
  uint64_t c = 0;
  size_t i = 0;
  const __m128i nl = _mm_set1_epi8('\n');
  for (; i + 64 <= n; i += 64) {
    const unsigned m0 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + i)), nl));
    const unsigned m1 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + i + 16)), nl));
    const unsigned m2 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + i + 32)), nl));
    const unsigned m3 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + i + 48)), nl));
    c += (uint64_t)__builtin_popcountll((uint64_t)m0 | ((uint64_t)m1 << 16) | ((uint64_t)m2 << 32) | ((uint64_t)m3 << 48));
  }
  for (; i + 16 <= n; i += 16) c += (uint64_t)__builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + i)), nl)));
  for (; i < n; i++) c += p[i] == '\n';
  return c;
}
// the same with 32-byte compares and the popcnt instruction, where the CPU has them (the readers count every byte they move:
// at a dozen GB/s per thread the count stops showing next to the copy)
__attribute__((target("avx2,popcnt"))) inline uint64_t count_newlines_avx2(const char* p, size_t n) {
  uint64_t c = 0;
  size_t i = 0;
  const __m256i nl = _mm256_set1_epi8('\n');
  for (; i + 128 <= n; i += 128) {
    const uint64_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i)), nl));
    const uint64_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 32)), nl));
    const uint64_t m2 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 64)), nl));
    const uint64_t m3 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i + 96)), nl));
    c += (uint64_t)__builtin_popcountll(m0 | (m1 << 32)) + (uint64_t)__builtin_popcountll(m2 | (m3 << 32));
  }
  return c + count_newlines_sse2(p + i, n - i);
}
inline uint64_t count_newlines(const char* p, size_t n) {
  static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
  return wide ? count_newlines_avx2(p, n) : count_newlines_sse2(p, n);
}
// copy + count in one pass over the source, the destination written with streaming stores (the ring is read next by the DMA engine, not by
// a CPU: going through the cache would cost a read of every destination line before it is overwritten)
__attribute__((target("avx2,popcnt"))) inline uint64_t copy_count_newlines_avx2(char* dst, const char* src, size_t n) {
  uint64_t c = 0;
  size_t i = 0;
  // head: up to the destination's 32-byte boundary
  const size_t head = std::min(n, (size_t)((32 - ((uintptr_t)dst & 31)) & 31));
  if (head) { memcpy(dst, src, head); c += count_newlines_sse2(src, head); i = head; }
  const __m256i nl = _mm256_set1_epi8('\n');
  for (; i + 128 <= n; i += 128) {
    const __m256i a0 = _mm256_loadu_si256((const __m256i*)(src + i)), a1 = _mm256_loadu_si256((const __m256i*)(src + i + 32));
    const __m256i a2 = _mm256_loadu_si256((const __m256i*)(src + i + 64)), a3 = _mm256_loadu_si256((const __m256i*)(src + i + 96));
    _mm256_stream_si256((__m256i*)(dst + i), a0); _mm256_stream_si256((__m256i*)(dst + i + 32), a1);
    _mm256_stream_si256((__m256i*)(dst + i + 64), a2); _mm256_stream_si256((__m256i*)(dst + i + 96), a3);
    const uint64_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a0, nl)), m1 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a1, nl));
    const uint64_t m2 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a2, nl)), m3 = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a3, nl));
    c += (uint64_t)__builtin_popcountll(m0 | (m1 << 32)) + (uint64_t)__builtin_popcountll(m2 | (m3 << 32));
  }
  _mm_sfence();
  if (i < n) { memcpy(dst + i, src + i, n - i); c += count_newlines_sse2(src + i, n - i); }
  return c;
}
inline uint64_t copy_count_newlines(char* dst, const char* src, size_t n) {
  static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
  if (wide) return copy_count_newlines_avx2(dst, src, n);
  memcpy(dst, src, n);
  return count_newlines_sse2(dst, n);
}
// offset just behind the k-th (k >= 1) newline of [p, p + n); n if there are fewer
inline size_t after_kth_newline(const char* p, size_t n, uint64_t k) {
  size_t i = 0;
  const __m128i nl = _mm_set1_epi8('\n');
  for (; i + 16 <= n; i += 16) {
    unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + i)), nl));
    const unsigned c = (unsigned)__builtin_popcount(m);
    if (c >= k) { for (;; m &= m - 1) if (--k == 0) return i + (size_t)__builtin_ctz(m) + 1; }
    k -= c;
  }
  for (; i < n; i++) if (p[i] == '\n' && --k == 0) return i + 1;
  return n;
}

// libdeflate, when the box has it (about three times zlib's inflate rate on FASTQ text), bound at run time; zlib otherwise
struct Libdeflate {
  void* (*alloc)() = nullptr; void (*free_)(void*) = nullptr;
  int (*inflate_raw)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;   // libdeflate_deflate_decompress: 0 = success
  uint32_t (*crc32_)(uint32_t, const void*, size_t) = nullptr;
  static const Libdeflate& get() {
    static const Libdeflate L = [] {
      Libdeflate l;
      if (getenv("KAMD_NO_LIBDEFLATE")) return l;
      void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
      if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
      if (!h) return l;
      l.alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
      l.free_ = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
      l.inflate_raw = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
      l.crc32_ = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
      if (!l.alloc || !l.free_ || !l.inflate_raw || !l.crc32_) l = Libdeflate();
      return l;
    }();
    return L;
  }
  bool ok() const { return alloc != nullptr; }
};

// The decompressed text of ONE file, produced in file order into a ring the caller owns (pinned memory in the front-end), with
// the newlines of every block counted as it lands.  Three producers: plain files (`threads` readers pread disjoint blocks),
// BGZF (the members are inflated by `threads` workers straight into their place in the ring) and any other gzip stream (one
// inflate thread -- a deflate stream is serial).  If the text does not end in a newline one is appended, so every line ends
// in '\n'.  The consumer (UnitCutter) asks where lines end, copies byte ranges out of the ring and releases them.
class TextSource {
 public:
  static const uint64_t NOT_FOUND = ~0ULL, TOO_BIG = ~0ULL - 1;
  enum Kind { PLAIN, GZIP, BGZF };
  TextSource(const std::string& path, char* ring, size_t ring_bytes, int threads, size_t block_bytes = 1 << 20)
      : path_(path), ring_(ring), cap_(ring_bytes), blk_(std::max<size_t>(block_bytes, 64)) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    struct stat st;
    if (fd_ < 0 || fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) { fail("could not open " + path + " as a regular file"); return; }
    fsize_ = (uint64_t)st.st_size;
    unsigned char magic[2] = {0, 0};
    const bool gz = fsize_ >= 2 && pread(fd_, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    kind_ = !gz ? PLAIN : (BgzfSource::is_bgzf(path) ? BGZF : GZIP);
    threads = std::max(1, threads);
    if (kind_ == PLAIN) {
      if (fsize_ == 0) { eof_ = true; return; }
      char last = 0;
      if (pread(fd_, &last, 1, (off_t)(fsize_ - 1)) != 1) { fail("could not read " + path); return; }
      pad_nl_ = last != '\n';
      const uint64_t text = fsize_ + (pad_nl_ ? 1 : 0);
      for (uint64_t o = 0; o < text; o += blk_) blocks_.push_back(Blk{o, std::min<uint64_t>(o + blk_, text), 0, 0, 0, false});
      // The readers copy out of a mapping of the file (one pass: copy + count, streaming stores) unless KAMD_FQ_PREAD asks for read calls
      // (the kernel's copy into the ring, then a second pass for the count)
      if (!getenv("KAMD_FQ_PREAD")) {
        void* p = mmap(nullptr, fsize_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (p != MAP_FAILED) { map_ = (const unsigned char*)p; (void)madvise(p, fsize_, MADV_SEQUENTIAL); }
      }
      int cap = 16;
      if (const char* e = getenv("KAMD_FQ_PLAIN_THREADS")) cap = std::max(1, atoi(e));
      for (int t = 0; t < std::min<int>(std::min(threads, cap), (int)blocks_.size()); t++) workers_.emplace_back([this] { plain_worker(); });
    } else if (kind_ == BGZF) {
      void* p = mmap(nullptr, fsize_, PROT_READ, MAP_PRIVATE, fd_, 0);
      if (p == MAP_FAILED) { fail("could not map " + path); return; }
      map_ = (const unsigned char*)p;
      uint64_t text = 0;
      for (uint64_t o = 0; o < fsize_;) {   // the members, hopping from header to header; ISIZE gives each its place in the text
        size_t bs = 0, doff = 0;
        if (!BgzfSource::header(map_ + o, fsize_ - o, &bs, &doff) || o + bs > fsize_) { fail(path + ": broken BGZF block at offset " + std::to_string(o)); return; }
        const unsigned char* t = map_ + o + bs - 8;
        const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (isize) blocks_.push_back(Blk{text, text + isize, 0, o + doff, bs - doff - 8, false});
        text += isize;
        o += bs;
      }
      if (blocks_.empty()) { eof_ = true; return; }
      for (auto& b : blocks_) if (b.end - b.begin + 1 > cap_) { fail(path + ": ring smaller than a BGZF block"); return; }
      for (int t = 0; t < std::min<int>(threads, (int)blocks_.size()); t++) workers_.emplace_back([this] { bgzf_worker(); });
    } else {
      // an ordinary gzip file: block-parallel inflate (kamd_pargzip.h) when there are threads for it and the file is worth it;
      // the one-thread zlib reader otherwise (KAMD_PARGZIP=0 forces it)
      size_t min_bytes = 4u << 20;
      if (const char* e = getenv("KAMD_PARGZIP_MIN_KB")) min_bytes = (size_t)std::max(0, atoi(e)) << 10;
      const char* sw = getenv("KAMD_PARGZIP");
      const bool par = threads >= 2 && fsize_ >= min_bytes && !(sw && atoi(sw) == 0);
      if (par) {
        void* p = mmap(nullptr, fsize_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (p == MAP_FAILED) { fail("could not map " + path); return; }
        map_ = (const unsigned char*)p;
        workers_.emplace_back([this, threads] { pargzip_worker(threads); });
      } else workers_.emplace_back([this] { gzip_worker(); });
    }
  }
  ~TextSource() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
    if (map_) munmap((void*)map_, fsize_);
    if (fd_ >= 0) ::close(fd_);
  }
  TextSource(const TextSource&) = delete;
  TextSource& operator=(const TextSource&) = delete;
  Kind kind() const { return kind_; }
  std::string error() { std::lock_guard<std::mutex> g(m_); return error_; }
  bool failed() { std::lock_guard<std::mutex> g(m_); return !error_.empty(); }

  // Waits until the text reaches `from + want` bytes (or ends).  Returns the number of lines that end at or before the first
  // block boundary >= from + want (at the end of the text: all lines) and that boundary; *at_end: the text ends there.
  uint64_t lines_near(uint64_t from, uint64_t want, uint64_t* boundary, bool* at_end) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return produced_ >= from + want || eof_ || !error_.empty() || stalled(from); });
    uint64_t lines = lines_, bnd = produced_;
    for (const Cum& c : cums_) if (c.end >= from + want) { lines = c.lines; bnd = c.end; break; }
    *boundary = bnd; *at_end = eof_ && bnd == produced_;
    return lines;
  }
  // Offset just behind newline number `line_no` (1-based, counted from the start of the text).  Waits for it.  NOT_FOUND: the
  // text ends first.  TOO_BIG: the ring cannot hold [unit_begin, that offset) -- the caller should ask for fewer lines.
  uint64_t locate(uint64_t line_no, uint64_t unit_begin) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return lines_ >= line_no || eof_ || !error_.empty() || stalled(unit_begin); });
    if (!error_.empty()) return NOT_FOUND;
    if (lines_ < line_no) return eof_ ? NOT_FOUND : TOO_BIG;
    size_t lo = 0, hi = cums_.size();   // the counted block that holds the line
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (cums_[mid].lines < line_no) lo = mid + 1; else hi = mid; }
    const Cum c = cums_[lo];
    lk.unlock();
    // k-th newline inside [c.begin, c.end): the block may wrap around the end of the ring
    uint64_t k = line_no - (c.lines - c.nl), pos = c.begin;
    while (pos < c.end) {
      const size_t o = (size_t)(pos % cap_), n = (size_t)std::min<uint64_t>(c.end - pos, cap_ - o);
      const uint64_t have = count_newlines(ring_ + o, n);
      if (have >= k) return pos + after_kth_newline(ring_ + o, n, k);
      k -= have; pos += n;
    }
    return NOT_FOUND;   // (cannot happen: the block was counted)
  }
  // the whole text has been produced: its size and number of lines (waits for the end)
  void totals(uint64_t* bytes, uint64_t* lines) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return eof_ || !error_.empty(); });
    *bytes = produced_; *lines = lines_;
  }
  // [begin, end) of the text as at most two pieces of the ring
  int pieces(uint64_t begin, uint64_t end, const char* p[2], size_t n[2]) const {
    if (end <= begin) return 0;
    const size_t o = (size_t)(begin % cap_), len = (size_t)(end - begin);
    p[0] = ring_ + o; n[0] = std::min(len, cap_ - o);
    if (n[0] == len) return 1;
    p[1] = ring_; n[1] = len - n[0];
    return 2;
  }
  // the bytes before `upto` are no longer needed
  void release(uint64_t upto) {
    { std::lock_guard<std::mutex> g(m_);
      released_ = std::max(released_, upto);
      while (!cums_.empty() && cums_.front().end <= released_) cums_.pop_front(); }
    cv_.notify_all();
  }
  uint64_t bytes_in() const { return fsize_; }   // size of the file on disk

 private:
  struct Blk { uint64_t begin, end, nl; uint64_t src_off, src_len; bool done; };
  struct Cum { uint64_t begin, end, nl, lines; };   // a block of the contiguous produced prefix; lines = newlines in [0, end)
  void fail(const std::string& msg) { { std::lock_guard<std::mutex> g(m_); if (error_.empty()) error_ = msg; } cv_.notify_all(); }
  // (m_ held) the producer cannot go on before bytes at or behind unit_begin are released -- which the consumer will not do
  // before it has the line it is waiting for: the ring is too small for this unit
  bool stalled(uint64_t unit_begin) const {
    if (kind_ == GZIP) return produced_ - unit_begin >= cap_;
    return next_done_ < blocks_.size() && blocks_[next_done_].end + (kind_ == BGZF ? 1 : 0) - unit_begin > cap_;
  }
  void put(uint64_t at, const char* src, size_t n) {   // copy into the ring (wraps)
    const size_t o = (size_t)(at % cap_), a = std::min(n, cap_ - o);
    memcpy(ring_ + o, src, a);
    if (a < n) memcpy(ring_, src + a, n - a);
  }
  uint64_t count_range(uint64_t begin, uint64_t end) const {
    uint64_t c = 0;
    for (uint64_t pos = begin; pos < end;) {
      const size_t o = (size_t)(pos % cap_), n = (size_t)std::min<uint64_t>(end - pos, cap_ - o);
      c += count_newlines(ring_ + o, n); pos += n;
    }
    return c;
  }
  // a block of the fixed list is complete: extend the contiguous prefix
  void finish_block(size_t i, uint64_t nl) {
    { std::lock_guard<std::mutex> g(m_);
      blocks_[i].nl = nl; blocks_[i].done = true;
      while (next_done_ < blocks_.size() && blocks_[next_done_].done) {
        const Blk& b = blocks_[next_done_];
        lines_ += b.nl; produced_ = b.end;
        cums_.push_back(Cum{b.begin, b.end, b.nl, lines_});
        ++next_done_;
      }
      if (next_done_ == blocks_.size()) eof_ = true; }
    cv_.notify_all();
  }
  // claims the next block and waits until the ring has room for it; false = stop
  bool claim(size_t* idx) {
    std::unique_lock<std::mutex> lk(m_);
    if (stop_ || !error_.empty() || next_claim_ >= blocks_.size()) return false;
    const size_t i = next_claim_++;
    const uint64_t pad = kind_ == BGZF ? 1 : 0;   // (the newline appended to a text that lacks the last one)
    cv_.wait(lk, [&] { return stop_ || !error_.empty() || blocks_[i].end + pad - released_ <= cap_; });
    if (stop_ || !error_.empty()) return false;
    *idx = i;
    return true;
  }
  void plain_worker() {
    size_t i;
    while (claim(&i)) {
      const Blk b = blocks_[i];
      uint64_t end = b.end;
      const bool padded = pad_nl_ && end == fsize_ + 1;
      if (padded) --end;
      uint64_t nl = 0;
      for (uint64_t pos = b.begin; pos < end;) {   // 256 KB at a time: the count reads what the copy just wrote while it is in L2
        const size_t o = (size_t)(pos % cap_), n = (size_t)std::min<uint64_t>(std::min<uint64_t>(end - pos, cap_ - o), 256u << 10);
        if (map_) {
#ifdef MADV_POPULATE_READ
          if (populate_ok_ && madvise((void*)(map_ + (pos & ~(uint64_t)4095)), n + (size_t)(pos & 4095), MADV_POPULATE_READ) != 0) populate_ok_ = false;   // (one call maps the piece's pages; older kernels: page faults)
#endif
          nl += copy_count_newlines(ring_ + o, (const char*)map_ + pos, n);
          pos += n;
          continue;
        }
        const ssize_t r = pread(fd_, ring_ + o, n, (off_t)pos);
        if (r <= 0) { fail("read error on " + path_); return; }
        nl += count_newlines(ring_ + o, (size_t)r);
        pos += (uint64_t)r;
      }
      if (padded) { ring_[(size_t)(end % cap_)] = '\n'; ++nl; }
      finish_block(i, nl);
    }
  }
  void bgzf_worker() {
    const Libdeflate& ld = Libdeflate::get();
    void* dec = ld.ok() ? ld.alloc() : nullptr;
    z_stream z; memset(&z, 0, sizeof z);
    if (!dec && inflateInit2(&z, -15) != Z_OK) { fail("zlib initialisation failed"); return; }
    std::vector<unsigned char> tmp(65536 + 16);
    size_t i;
    while (claim(&i)) {
      const Blk b = blocks_[i];
      const size_t n = (size_t)(b.end - b.begin), o = (size_t)(b.begin % cap_);
      const bool wraps = o + n > cap_;
      if (n > tmp.size()) tmp.resize(n);
      unsigned char* dst = wraps ? tmp.data() : (unsigned char*)ring_ + o;
      bool ok;
      if (dec) { size_t got = 0; ok = ld.inflate_raw(dec, map_ + b.src_off, (size_t)b.src_len, dst, n, &got) == 0 && got == n; }
      else {
        inflateReset(&z);
        z.next_in = (Bytef*)(map_ + b.src_off); z.avail_in = (uInt)b.src_len; z.next_out = dst; z.avail_out = (uInt)n;
        ok = inflate(&z, Z_FINISH) == Z_STREAM_END && z.total_out == n;
      }
      const unsigned char* t = map_ + b.src_off + b.src_len;
      const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
      if (ok) ok = (dec ? ld.crc32_(0, dst, n) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, (uInt)n)) == crc;
      if (!ok) { fail(path_ + ": corrupt BGZF block " + std::to_string(i)); break; }
      if (wraps) put(b.begin, (const char*)tmp.data(), n);
      uint64_t end = b.end;
      if (i + 1 == blocks_.size() && dst[n - 1] != '\n') {   // the text does not end in a newline: one is appended
        ring_[(size_t)(end % cap_)] = '\n';
        std::lock_guard<std::mutex> g(m_);
        blocks_[i].end = ++end;
      }
      finish_block(i, count_range(b.begin, end));
    }
    if (dec) ld.free_(dec); else inflateEnd(&z);
  }
  // an ordinary gzip file by all the threads this source was given: kamd_pargzip.h decodes the stream block-parallel and hands the
  // text over in order; this thread copies it into the ring (waiting for room) and counts its lines
  void pargzip_worker(int threads) {
    size_t chunk = 2u << 20;
    if (const char* e = getenv("KAMD_PARGZIP_CHUNK_KB")) chunk = (size_t)std::max(4, atoi(e)) << 10;
    const Libdeflate& ld = Libdeflate::get();
    uint64_t head = 0; char last = '\n';
    auto deliver = [&](const uint8_t* p, size_t n, uint64_t) -> bool {
      while (n) {
        size_t room;
        {
          std::unique_lock<std::mutex> lk(m_);
          cv_.wait(lk, [&] { return stop_ || head - released_ < cap_; });
          if (stop_) return false;
          room = (size_t)std::min<uint64_t>(cap_ - (head - released_), blk_);
        }
        const size_t o = (size_t)(head % cap_);
        const size_t a = std::min(std::min(room, cap_ - o), n);
        const uint64_t nl = copy_count_newlines(ring_ + o, (const char*)p, a);   // (one pass, streaming stores: this thread handles every byte of the file)
        last = (char)p[a - 1];
        { std::lock_guard<std::mutex> g(m_); lines_ += nl; cums_.push_back(Cum{head, head + a, nl, lines_}); produced_ = head + a; }
        head += a; p += a; n -= a;
        cv_.notify_all();
      }
      return true;
    };
    pgz::ParGzip P(map_, (size_t)fsize_, threads, chunk, deliver, nullptr /* the copy into the ring counts */, ld.ok() ? ld.crc32_ : nullptr);
    if (!P.run()) {
      { std::lock_guard<std::mutex> g(m_); if (stop_) return; }
      fail(path_ + ": " + (P.error().empty() ? std::string("corrupt gzip stream") : P.error()));
      return;
    }
    if (head && last != '\n') {   // the text does not end in a newline: one is appended
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return stop_ || head - released_ < cap_; });
      if (stop_) return;
      ring_[(size_t)(head % cap_)] = '\n';
      ++lines_; cums_.push_back(Cum{head, head + 1, 1, lines_}); produced_ = ++head;
    }
    { std::lock_guard<std::mutex> g(m_); eof_ = true; }
    cv_.notify_all();
  }
  // any gzip stream (concatenated members are one text, as gzread reads them): one thread, inflate straight into the ring
  void gzip_worker() {
    z_stream z; memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) { fail("zlib initialisation failed"); return; }
    std::vector<unsigned char> in(1 << 20);
    uint64_t fpos = 0, head = 0, member_out = 0;
    char last = '\n';
    bool stream_end = true, first = true;   // between members
    for (;;) {
      if (z.avail_in == 0) {
        const ssize_t r = pread(fd_, in.data(), in.size(), (off_t)fpos);
        if (r < 0) { fail("read error on " + path_); break; }
        if (r == 0) { if (!stream_end) fail(path_ + ": unexpected end of the gzip stream"); break; }
        fpos += (uint64_t)r; z.next_in = in.data(); z.avail_in = (uInt)r;
      }
      if (stream_end) {   // the next member (trailing zero padding is skipped, as gzread does)
        while (z.avail_in && *z.next_in == 0 && !first) { ++z.next_in; --z.avail_in; }
        if (!z.avail_in) continue;
        if (!first) inflateReset(&z);
        stream_end = false; first = false; member_out = 0;
      }
      size_t room;   // room in the ring: at most one block, contiguous
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || head - released_ < cap_; });
        if (stop_) break;
        room = (size_t)std::min<uint64_t>(cap_ - (head - released_), blk_);
      }
      const size_t o = (size_t)(head % cap_);
      room = std::min(room, cap_ - o);
      z.next_out = (Bytef*)ring_ + o; z.avail_out = (uInt)room;
      const int rc = inflate(&z, Z_NO_FLUSH);
      const size_t got = room - z.avail_out;
      if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
        if (head && member_out == 0 && got == 0) { stream_end = true; break; }   // bytes behind the last member that are no gzip member: ignored, as gzread does
        fail(path_ + ": corrupt gzip stream"); break;
      }
      member_out += got;
      if (got) {
        last = ring_[o + got - 1];
        const uint64_t nl = count_newlines(ring_ + o, got);
        { std::lock_guard<std::mutex> g(m_); lines_ += nl; cums_.push_back(Cum{head, head + got, nl, lines_}); produced_ = head + got; }
        head += got;
        cv_.notify_all();
      }
      if (rc == Z_STREAM_END) stream_end = true;
    }
    inflateEnd(&z);
    if (failed()) return;
    if (head && last != '\n') {   // the text does not end in a newline: one is appended
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return stop_ || head - released_ < cap_; });
      if (stop_) return;
      ring_[(size_t)(head % cap_)] = '\n';
      ++lines_; cums_.push_back(Cum{head, head + 1, 1, lines_}); produced_ = ++head;
    }
    { std::lock_guard<std::mutex> g(m_); eof_ = true; }
    cv_.notify_all();
  }

  std::string path_;
  char* ring_; size_t cap_, blk_;
  int fd_ = -1; uint64_t fsize_ = 0;
  Kind kind_ = PLAIN;
  bool pad_nl_ = false;
  const unsigned char* map_ = nullptr;
  std::atomic<bool> populate_ok_{true};
  std::vector<Blk> blocks_;          // plain / BGZF: the whole file, known at open
  size_t next_claim_ = 0, next_done_ = 0;
  std::deque<Cum> cums_;             // counted blocks of the produced prefix that are not released yet
  uint64_t produced_ = 0, lines_ = 0, released_ = 0;
  bool eof_ = false, stop_ = false;
  std::string error_;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<std::thread> workers_;
};

// Cuts the text of one file (single-end) or of the two mates' files into units of whole 4-line records: unit u is lines
// [4 R_u, 4 R_(u+1)) of every file.  File 0 decides where (the first block boundary behind `target` more bytes), the other
// file is cut behind the same line.  What a unit does NOT guarantee is that its records have the strict shape -- the device
// checks that (kamd_fq_core.h).
struct UnitCut { uint64_t begin[2] = {0, 0}, end[2] = {0, 0}, n_records = 0; };
class UnitCutter {
 public:
  enum { UNIT = 1, DONE = 0, NOT_STRICT = -1, COUNT_MISMATCH = -2, IO_ERROR = -3, TOO_LONG = -4 };
  UnitCutter(TextSource* s0, TextSource* s1, uint64_t target_bytes, uint64_t max_bytes) : n_(s1 ? 2 : 1), target_(target_bytes), max_(max_bytes) { s_[0] = s0; s_[1] = s1; }
  int next(UnitCut& u) {
    if (done_) return DONE;
    // the first units are small (a sixteenth of the target, doubling): the copy to the device and its parser start while the readers are
    // still at the head of the files
    uint64_t want = std::min<uint64_t>(target_, std::max<uint64_t>(target_ / 16, 1u << 16) << std::min<uint64_t>(units_, 8)), t_new = 0;
    bool at_end = false;
    for (;;) {
      uint64_t bnd = 0;
      const uint64_t lines = s_[0]->lines_near(pos_[0], want, &bnd, &at_end);
      if (s_[0]->failed()) return IO_ERROR;
      t_new = lines / 4 * 4;
      if (t_new > lines_ || at_end) break;
      want *= 2;                       // not one whole record in `want` bytes
      if (want > max_) return TOO_LONG;
    }
    if (t_new == lines_) return finish();
    // the same line in every file; a unit no larger than max_ bytes per file
    for (;;) {
      bool shrink = false;
      for (int f = 0; f < n_ && !shrink; f++) {
        const uint64_t e = s_[f]->locate(t_new, pos_[f]);
        if (s_[f]->failed()) return IO_ERROR;
        if (e == TextSource::NOT_FOUND) return f == 0 ? IO_ERROR : COUNT_MISMATCH;   // (file 0 counted the line itself)
        if (e == TextSource::TOO_BIG || e - pos_[f] > max_) shrink = true;
        else u.end[f] = e;
      }
      if (!shrink) break;
      const uint64_t half = (t_new - lines_) / 2 / 4 * 4;
      if (half == 0) return TOO_LONG;
      t_new = lines_ + half;
    }
    for (int f = 0; f < n_; f++) { u.begin[f] = pos_[f]; pos_[f] = u.end[f]; }
    u.n_records = (t_new - lines_) / 4;
    lines_ = t_new;
    ++units_;
    return UNIT;
  }
 private:
  // file 0 has no whole record left: what remains of every file must be blank, and the other file must end at the same record
  int finish() {
    done_ = true;
    for (int f = 0; f < n_; f++) {
      uint64_t bytes = 0, lines = 0;
      s_[f]->totals(&bytes, &lines);
      if (s_[f]->failed()) return IO_ERROR;
      if (lines / 4 * 4 != lines_) return (f > 0 && lines / 4 > lines_ / 4) ? COUNT_MISMATCH : NOT_STRICT;
      const char* p[2]; size_t n[2];
      const int np = s_[f]->pieces(pos_[f], bytes, p, n);
      for (int i = 0; i < np; i++) for (size_t j = 0; j < n[i]; j++) { const char c = p[i][j]; if (c != '\n' && c != '\r' && c != ' ' && c != '\t') return NOT_STRICT; }
    }
    return DONE;
  }
  TextSource* s_[2];
  int n_;
  uint64_t target_, max_;
  uint64_t pos_[2] = {0, 0}, lines_ = 0, units_ = 0;
  bool done_ = false;
};

}  // namespace kamd_io
