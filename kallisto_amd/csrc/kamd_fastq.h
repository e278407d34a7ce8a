// kamd_fastq.h -- host-side sequence input of the `kallisto quant` front-end (what FastqSequenceReader::fetchSequences,
// src/ProcessReads.cpp:3128-3267, does serially under a lock): plain 4-line FASTQ through mmap with one parser thread per
// slice; gzip / FASTA through one inflate-and-parse thread per file; BGZF (blocked gzip, `bgzip`) with its blocks inflated
// by several threads.  Host code only, no HIP: tests/emu/io_emu.cpp drives it on a box without a GPU.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <iostream>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace kamd_io {

// ---- BGZF: a gzip file made of independent members of at most 64 KiB whose header carries the member's size (extra
// subfield 'B','C': BSIZE = total block size - 1).  The members are found by hopping from header to header, inflated
// by `threads` workers at most `window` blocks ahead of the consumer, and handed over in file order. ----
class BgzfSource {
 public:
  static bool header(const unsigned char* p, size_t n, size_t* block_size, size_t* data_off) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (12 + xlen > n) return false;
    for (size_t q = 12; q + 4 <= 12 + xlen;) {
      const size_t slen = (size_t)p[q + 2] | ((size_t)p[q + 3] << 8);
      if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
        *block_size = ((size_t)p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
        *data_off = 12 + xlen;
        return *block_size >= *data_off + 8;
      }
      q += 4 + slen;
    }
    return false;
  }
  static bool is_bgzf(const std::string& path) {
    unsigned char m[64];
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t n = fread(m, 1, sizeof m, f); fclose(f);
    size_t bs, off;
    return header(m, n, &bs, &off);
  }
  BgzfSource(const std::string& path, int threads) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    struct stat st;
    if (fd_ < 0 || fstat(fd_, &st) != 0) { std::cerr << "Error: could not open file " << path << std::endl; exit(1); }
    size_ = (size_t)st.st_size;
    void* p = size_ ? mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0) : nullptr;
    if (size_ && p == MAP_FAILED) { std::cerr << "Error: could not map file " << path << std::endl; exit(1); }
    data_ = (const unsigned char*)p;
    for (size_t o = 0; o < size_;) {   // the members, by hopping over the headers
      size_t bs = 0, doff = 0;
      if (!header(data_ + o, size_ - o, &bs, &doff) || o + bs > size_) { std::cerr << "Error: " << path << ": broken BGZF block at offset " << o << std::endl; exit(1); }
      blocks_.push_back(Block{o + doff, bs - doff - 8, o + bs - 8});
      o += bs;
    }
    threads = std::max(1, std::min(threads, 16));
    slots_.resize((size_t)threads * 8);
    for (auto& s : slots_) s.buf.resize(65536);
    for (int t = 0; t < threads; t++) workers_.emplace_back([this] { work(); });
  }
  ~BgzfSource() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
    if (data_) munmap((void*)data_, size_);
    if (fd_ >= 0) ::close(fd_);
  }
  // like gzread: up to `cap` bytes of the decompressed stream, 0 at the end
  int read(char* dst, size_t cap) {
    size_t got = 0;
    while (got < cap) {
      if (cur_len_ == cur_pos_) {
        std::unique_lock<std::mutex> lk(m_);
        if (have_cur_) { slots_[consumed_ % slots_.size()].ready = false; ++consumed_; have_cur_ = false; lk.unlock(); cv_.notify_all(); lk.lock(); }
        if (consumed_ == blocks_.size()) break;
        Slot& s = slots_[consumed_ % slots_.size()];
        cv_.wait(lk, [&] { return s.ready && s.index == consumed_; });
        cur_ = s.buf.data(); cur_len_ = s.len; cur_pos_ = 0; have_cur_ = true;
        continue;
      }
      const size_t n = std::min(cap - got, cur_len_ - cur_pos_);
      memcpy(dst + got, cur_ + cur_pos_, n);
      got += n; cur_pos_ += n;
    }
    return (int)got;
  }

 private:
  struct Block { size_t off, clen, tail; };   // deflate data, its length, offset of CRC32 + ISIZE
  struct Slot { std::vector<unsigned char> buf; size_t len = 0, index = 0; bool ready = false; };
  void work() {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) { std::cerr << "Error: zlib initialisation failed" << std::endl; exit(1); }
    for (;;) {
      size_t i;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || next_ >= blocks_.size() || next_ < consumed_ + slots_.size(); });
        if (stop_ || next_ >= blocks_.size()) break;
        i = next_++;
      }
      Slot& s = slots_[i % slots_.size()];   // free: block i - slots has been consumed (the wait above)
      const Block& b = blocks_[i];
      const unsigned char* t = data_ + b.tail;
      const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
      const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
      if (isize > s.buf.size()) s.buf.resize(isize);
      inflateReset(&z);
      z.next_in = (Bytef*)(data_ + b.off); z.avail_in = (uInt)b.clen;
      z.next_out = s.buf.data(); z.avail_out = (uInt)s.buf.size();
      const int rc = isize ? inflate(&z, Z_FINISH) : Z_STREAM_END;
      if ((isize && rc != Z_STREAM_END) || (isize && z.total_out != isize) ||
          (isize && (uint32_t)crc32(crc32(0L, Z_NULL, 0), s.buf.data(), isize) != crc)) {
        std::cerr << "Error: corrupt BGZF block " << i << std::endl; exit(1);
      }
      {
        std::lock_guard<std::mutex> g(m_);
        s.len = isize; s.index = i; s.ready = true;
      }
      cv_.notify_all();
    }
    inflateEnd(&z);
  }
  int fd_ = -1;
  const unsigned char* data_ = nullptr;
  size_t size_ = 0;
  std::vector<Block> blocks_;
  std::vector<Slot> slots_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_;
  size_t next_ = 0, consumed_ = 0;
  bool stop_ = false, have_cur_ = false;
  const unsigned char* cur_ = nullptr;
  size_t cur_len_ = 0, cur_pos_ = 0;
};

// ---- FASTA/FASTQ reader over gzFile (plain or gzip) or a BgzfSource, one record at a time ----
class SeqReader {
 public:
  // inflate_threads > 1: a BGZF file is inflated block-parallel (any other gzip file is one serial deflate stream)
  explicit SeqReader(const std::string& path, int inflate_threads = 1) : buf_(1 << 22) {
    if (inflate_threads > 1 && BgzfSource::is_bgzf(path)) { bgzf_ = new BgzfSource(path, inflate_threads); return; }
    f_ = gzopen(path.c_str(), "r");
    if (!f_) { std::cerr << "Error: could not open file " << path << std::endl; exit(1); }
    gzbuffer(f_, 1 << 20);
  }
  ~SeqReader() { if (f_) gzclose(f_); delete bgzf_; }
  SeqReader(const SeqReader&) = delete;
  SeqReader& operator=(const SeqReader&) = delete;
  // Fast path for the common shape -- a 4-line FASTQ record that lies completely inside the read buffer: four memchr calls,
  // the sequence is appended to `dst` straight from the buffer (no per-line strings).  Returns 1 = appended (*len set),
  // 0 = not applicable here (record crosses the buffer end, multi-line record, FASTA, pending header ...): use next().
  int next_fast(std::string& dst, int32_t* len) {
    if (pending_header_ || pos_ >= len_) return 0;
    const char* b = buf_.data() + pos_;
    const char* end = buf_.data() + len_;
    if (*b != '@') return 0;
    const char* e1 = (const char*)memchr(b, '\n', end - b);
    if (!e1) return 0;
    const char* e2 = (const char*)memchr(e1 + 1, '\n', end - (e1 + 1));
    if (!e2 || e2 + 1 >= end || e2[1] != '+') return 0;
    const char* e3 = (const char*)memchr(e2 + 1, '\n', end - (e2 + 1));
    if (!e3) return 0;
    const char* e4 = (const char*)memchr(e3 + 1, '\n', end - (e3 + 1));
    if (!e4) return 0;
    size_t sl = (size_t)(e2 - (e1 + 1)), ql = (size_t)(e4 - (e3 + 1));
    if (sl && e2[-1] == '\r') --sl;
    if (ql && e4[-1] == '\r') --ql;
    if (sl != ql || sl == 0) return 0;     // multi-line or empty records go through the general reader
    dst.append(e1 + 1, sl);
    *len = (int32_t)sl;
    pos_ = (size_t)(e4 + 1 - buf_.data());
    return 1;
  }
  // next sequence into `out`; returns false at end of file.  Character-level semantics of the reference's reader (kseq.h
  // kseq_read, ext/bifrost/src/kseq.h as used by src/ProcessReads.cpp:3128-3267): a record starts at the next '>' or '@'
  // anywhere in the stream (leading junk is skipped -- the reference's own functional tests write "\@>t1" headers), the rest
  // of that line is the name; sequence lines follow until a line that starts with '>', '@' or '+' (empty lines skipped,
  // trailing '\r' dropped); after '+' as many quality characters as there were bases are consumed.
  bool next(std::string& out) {
    int c;
    if (!pending_header_) {
      while ((c = getch()) >= 0 && c != '>' && c != '@') {}
      if (c < 0) return false;
    }
    pending_header_ = false;
    while ((c = getch()) >= 0 && c != '\n') {}            // name and comment
    if (c < 0) return false;
    out.clear();
    while ((c = getch()) >= 0 && c != '>' && c != '+' && c != '@') {
      if (c == '\n') continue;
      out.push_back((char)c);
      append_line(out);
    }
    if (c == '>' || c == '@') pending_header_ = true;      // the first character of the next header has been read
    if (c != '+') return true;                             // FASTA record, or the file ends here
    while ((c = getch()) >= 0 && c != '\n') {}            // rest of the '+' line
    size_t q = 0;
    std::string ql;
    while (q < out.size()) { ql.clear(); if (!append_line(ql)) break; q += ql.size(); }
    return true;
  }

 private:
  int getch() {
    if (pos_ == len_) {
      const int n = bgzf_ ? bgzf_->read(buf_.data(), buf_.size()) : gzread(f_, buf_.data(), (unsigned)buf_.size());
      if (n <= 0) return -1;
      len_ = (size_t)n; pos_ = 0;
    }
    return (unsigned char)buf_[pos_++];
  }
  // the rest of the current line appended to s (without the line end); false if the stream ended before any byte was read
  bool append_line(std::string& s) {
    bool any = false;
    for (;;) {
      if (pos_ == len_) {
        const int n = bgzf_ ? bgzf_->read(buf_.data(), buf_.size()) : gzread(f_, buf_.data(), (unsigned)buf_.size());
        if (n <= 0) break;
        len_ = (size_t)n; pos_ = 0;
      }
      any = true;
      char* b = buf_.data() + pos_;
      char* e = (char*)memchr(b, '\n', len_ - pos_);
      if (e) { s.append(b, e - b); pos_ = (size_t)(e - buf_.data()) + 1; break; }
      s.append(b, len_ - pos_); pos_ = len_;
    }
    if (!s.empty() && s.back() == '\r') s.pop_back();
    return any;
  }
  bool getline(std::string& s) {
    s.clear();
    for (;;) {
      if (pos_ == len_) {
        const int n = bgzf_ ? bgzf_->read(buf_.data(), buf_.size()) : gzread(f_, buf_.data(), (unsigned)buf_.size());
        if (n <= 0) return !s.empty();
        len_ = (size_t)n; pos_ = 0;
      }
      char* b = buf_.data() + pos_;
      char* e = (char*)memchr(b, '\n', len_ - pos_);
      if (e) { s.append(b, e - b); pos_ = (size_t)(e - buf_.data()) + 1; if (!s.empty() && s.back() == '\r') s.pop_back(); return true; }
      s.append(b, len_ - pos_); pos_ = len_;
    }
  }
  gzFile f_ = nullptr;
  BgzfSource* bgzf_ = nullptr;
  std::vector<char> buf_;
  size_t pos_ = 0, len_ = 0;
  bool pending_header_ = false;   // the '>' / '@' of the next record has already been consumed
};

// ---- gzip / FASTA input: one decompress-and-parse thread per file hands over chunks of `n` sequences, so the two mates'
// files inflate concurrently and the main thread only pairs chunks and packs them with all host threads ----
struct SeqChunk { std::string seqs; std::vector<uint64_t> off; std::vector<int32_t> len; };
class ChunkReader {
 public:
  ChunkReader(const std::string& path, uint64_t n, int inflate_threads = 1) : r_(path, inflate_threads), n_(n), th_([this] { loop(); }) {}
  ~ChunkReader() { { std::lock_guard<std::mutex> g(m_); stop_ = true; } cv_.notify_all(); if (th_.joinable()) th_.join(); }
  // next chunk (fewer than n sequences only at the end of the file); false when the file is exhausted
  bool next(SeqChunk& out) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !q_.empty() || eof_; });
    if (q_.empty()) return false;
    out = std::move(q_.front()); q_.erase(q_.begin());
    lk.unlock();
    cv_.notify_all();
    return true;
  }
 private:
  void loop() {
    std::string s;
    for (;;) {
      SeqChunk c;
      while (c.off.size() < n_) {
        int32_t fl = 0;
        const size_t at = c.seqs.size();
        if (r_.next_fast(c.seqs, &fl)) {
          if (fl > 65535) { std::cerr << "Error: reads longer than 65535 bp are outside the short-read GPU path" << std::endl; exit(1); }
          c.off.push_back(at); c.len.push_back(fl);
          continue;
        }
        if (!r_.next(s)) break;
        if (s.size() > 65535) { std::cerr << "Error: reads longer than 65535 bp are outside the short-read GPU path" << std::endl; exit(1); }
        c.off.push_back(c.seqs.size()); c.len.push_back((int32_t)s.size()); c.seqs += s;
      }
      const bool last = c.off.size() < n_;
      std::unique_lock<std::mutex> lk(m_);
      if (!c.off.empty()) {
        cv_.wait(lk, [&] { return q_.size() < 3 || stop_; });   // at most 3 chunks ahead
        if (stop_) return;
        q_.push_back(std::move(c));
      }
      if (last) eof_ = true;
      lk.unlock();
      cv_.notify_all();
      if (last) return;
    }
  }
  SeqReader r_;
  uint64_t n_;
  std::vector<SeqChunk> q_;
  std::mutex m_;
  std::condition_variable cv_;
  bool eof_ = false, stop_ = false;
  std::thread th_;
};

// ---- fast path for plain (uncompressed) 4-line FASTQ: mmap + one parser thread per chunk --------------------------------
// FastqSequenceReader::fetchSequences (src/ProcessReads.cpp:3128-3267) parses serially under a lock; here every thread scans
// its slice of the file for record starts and the records are paired by index afterwards.
struct MappedFastq {
  const char* data = nullptr; size_t size = 0; int fd = -1;
  std::vector<uint64_t> off; std::vector<int32_t> len;   // sequence line of every record
  bool open(const std::string& path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) return false;
    size = (size_t)st.st_size;
    void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return false;
    madvise(p, size, MADV_SEQUENTIAL);
    data = (const char*)p;
    return true;
  }
  void close() { if (data) munmap((void*)data, size); if (fd >= 0) ::close(fd); data = nullptr; fd = -1; }
  static bool is_gzip(const std::string& path) {
    unsigned char m[2] = {0, 0};
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    size_t n = fread(m, 1, 2, f); fclose(f);
    return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
  }
  const char* next_line(const char* p) const { const char* e = (const char*)memchr(p, '\n', data + size - p); return e ? e + 1 : data + size; }
  // first record start at or after p: a line starting with '@' whose third line starts with '+' and whose second and
  // fourth lines have equal length (a quality line may itself start with '@')
  const char* find_record(const char* p) const {
    const char* end = data + size;
    if (p != data) { const char* q = (const char*)memchr(p - 1, '\n', end - (p - 1)); p = q ? q + 1 : end; }
    while (p < end) {
      if (*p == '@') {
        const char* l1 = next_line(p); const char* l2 = next_line(l1); const char* l3 = next_line(l2); const char* l4 = next_line(l3);
        if (l2 < end && *l2 == '+') {
          const int64_t seq_len = (l2 - l1) - ((l2 > l1 && l2[-1] == '\n') ? 1 : 0);
          const int64_t qual_len = (l4 - l3) - ((l4 > l3 && l4[-1] == '\n') ? 1 : 0);
          if (seq_len == qual_len) return p;
        }
      }
      p = next_line(p);
    }
    return end;
  }
  // returns false if the file is not plain 4-line FASTQ (the caller then uses the serial reader)
  bool index_records(int threads) {
    if (size == 0 || data[0] != '@') return false;
    size_t min_chunk = 1 << 20;
    if (const char* e = getenv("KAMD_FASTQ_CHUNK")) min_chunk = std::max<size_t>(64, strtoull(e, nullptr, 10));
    threads = std::max(1, std::min(threads, (int)(size / min_chunk) + 1));
    std::vector<const char*> starts(threads + 1);
    starts[0] = data; starts[threads] = data + size;
    for (int t = 1; t < threads; t++) starts[t] = find_record(data + size / threads * t);
    std::vector<std::vector<uint64_t>> offs(threads); std::vector<std::vector<int32_t>> lens(threads);
    std::vector<char> ok(threads, 1);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back([&, t] {
      const char* p = starts[t]; const char* end = starts[t + 1];
      while (p < end) {
        if (*p != '@') { ok[t] = 0; return; }
        const char* l1 = next_line(p); const char* l2 = next_line(l1);
        if (l1 >= data + size || l2 > data + size || (l2 < data + size && *l2 != '+')) { ok[t] = 0; return; }
        int64_t n = (l2 - l1) - ((l2 > l1 && l2[-1] == '\n') ? 1 : 0);
        if (n > 0 && l1[n - 1] == '\r') --n;
        offs[t].push_back((uint64_t)(l1 - data)); lens[t].push_back((int32_t)n);
        p = next_line(next_line(l2));
      }
    });
    for (auto& x : th) x.join();
    size_t total = 0;
    for (int t = 0; t < threads; t++) { if (!ok[t]) return false; total += offs[t].size(); }
    off.reserve(total); len.reserve(total);
    for (int t = 0; t < threads; t++) { off.insert(off.end(), offs[t].begin(), offs[t].end()); len.insert(len.end(), lens[t].begin(), lens[t].end()); }
    return true;
  }
};

}  // namespace kamd_io
