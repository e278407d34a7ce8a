// kamd_pargzip.h -- block-parallel inflate of an ordinary gzip file (one deflate stream, as `gzip` writes it).
//
// A deflate stream is serial: a block can only be decoded when the 32 KiB of text before it are known, and nothing in a
// gzip file says where blocks start.  The front-end's gzip reader was therefore one zlib thread per file -- 0.6 GB/s of text,
// two orders of magnitude below what the device parser takes.  Here the stream is decoded by all host threads, the way pugz /
// rapidgzip do it:
//   1. the compressed bytes are cut into chunks; for every chunk a worker FINDS the first deflate block that starts inside
//      it (a dynamic-Huffman block header is self-validating: code-length code, literal/length code and distance code must
//      all be complete prefix codes -- random bit positions practically never are) ...
//   2. ... and decodes from there to the start of the next chunk's block WITHOUT knowing the window: the output is 16-bit
//      symbols, a byte or a MARKER "byte i of the unknown 32 KiB window"; copies of markers copy markers;
//   3. one thread walks the chunks in order: it checks that chunk k starts exactly where chunk k - 1 ended (otherwise the found
//      "block" was not one: the piece is decoded serially with the known window instead), resolves the markers of the chunk's
//      LAST 32 KiB -- the next chunk's window -- and hands the chunk back to the workers, which replace the remaining markers,
//      count newlines and checksum in parallel; the pieces are delivered in order, the member's CRC-32 and size are checked
//      against its trailer (crc32_combine over the chunks).
// Members that follow the first one, stored / fixed-Huffman blocks at a chunk's start, text that is not ASCII: all decoded
// correctly, through the serial path of step 3 where the parallel one does not apply.  The decoder is this file's own
// (table-driven, 64-bit bit buffer): zlib cannot produce markers, and libdeflate cannot stream.
//
// Reference: the reader this replaces is kseq's gzread loop, src/kseq.h:78-107 under FastqSequenceReader::fetchSequences,
// src/ProcessReads.cpp:3128-3267 (one thread, zlib).
#pragma once
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <cstdlib>
#include <memory>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifndef KAMD_PGZ_TIME
#define KAMD_PGZ_TIME(i)   // (a measuring harness defines this as a scoped timer of phase i: 0 decode, 1 resolve, 2 newline count, 3 CRC, 4 block search)
#endif
namespace kamd_io {
namespace pgz {

static const uint32_t WIN = 32768;          // deflate window
static const uint16_t MARK = 0x8000;        // symbol = MARK | index into the unknown window

// ---- bit reader: LSB-first, 64-bit buffer ------------------------------------------------------------------------------------
struct Bits {
  const uint8_t* base; size_t n; size_t pos = 0;   // pos: next byte to load
  uint64_t buf = 0; int cnt = 0;
  int over = 0;   // zero bytes invented behind the end of the data
  Bits(const uint8_t* b, size_t bytes, uint64_t bit) : base(b), n(bytes) { seek(bit); }
  void seek(uint64_t bit) { pos = (size_t)(bit >> 3); buf = 0; cnt = 0; over = 0; refill(); const int s = (int)(bit & 7); buf >>= s; cnt -= s; }
  uint64_t bitpos() const { return ((uint64_t)pos + (uint64_t)over) * 8 - (uint64_t)cnt; }
  // at least 56 valid bits afterwards.  (Fast path: the bits of `buf` above cnt are the stream's own next bits -- loading the same
  // bytes again later ORs in the same values.)  Behind the end of the data zero bytes are invented; eof() tells when they were used.
  inline void refill() {
    if (pos + 8 <= n) {
      uint64_t w; memcpy(&w, base + pos, 8);
      buf |= w << cnt;
      const int take = (63 - cnt) >> 3;
      pos += (size_t)take; cnt += take * 8;
    } else {
      buf &= cnt >= 64 ? ~0ULL : ((1ULL << cnt) - 1);
      while (cnt <= 56 && pos < n) { buf |= (uint64_t)base[pos++] << cnt; cnt += 8; }
      if (cnt <= 56) { const int z = (64 - cnt) >> 3; over += z; cnt += z * 8; }
    }
  }
  bool eof() const { return bitpos() > (uint64_t)n * 8; }
  inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ULL << k) - 1)); }
  inline void drop(int k) { buf >>= k; cnt -= k; }
  inline uint32_t take(int k) { const uint32_t v = peek(k); drop(k); return v; }
};

// ---- Huffman decoding tables (two levels) ----------------------------------------------------------------------------------
struct Ent { uint16_t val; uint8_t bits; uint8_t op; };   // op: 0 literal, 1 base + (op >> 4) extra bits, 2 end of block, 3 sub-table (val = offset, op >> 4 = its bits),
                                                          // 4 two literals (val = first | second << 8, bits = both codes), 15 invalid.  (op & 11) == 0: literal(s)
static const int LIT_PB = 11, DIST_PB = 8;
struct Table { std::vector<Ent> e; int pb = 0; };

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t rev_bits(uint32_t c, int len) { uint32_t r = 0; for (int i = 0; i < len; i++) { r = (r << 1) | (c & 1); c >>= 1; } return r; }

// kind: 0 = literal/length alphabet, 1 = distance alphabet.  Returns false for an over-subscribed code, or an incomplete one
// (allowed, as in zlib, only for a code with a single symbol -- or, for distances, none at all).
inline bool build_table(const uint8_t* lens, int n, int kind, Table* T) {
  int count[16] = {0};
  for (int i = 0; i < n; i++) count[lens[i]]++;
  count[0] = 0;
  int left = 1, maxlen = 0, used = 0;
  for (int l = 1; l <= 15; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; if (count[l]) maxlen = l; used += count[l]; }
  if (left > 0 && maxlen > 1) return false;   // incomplete: only a code of one-bit codes may be (zlib's inflate_table)
  if (kind == 0 && used == 0) return false;
  const int pb = kind == 0 ? LIT_PB : DIST_PB;
  T->pb = pb;
  const Ent bad{0, 0, 15};
  T->e.assign((size_t)1 << pb, bad);
  uint32_t next[16]; { uint32_t c = 0; for (int l = 1; l <= 15; l++) { c = (c + (uint32_t)count[l - 1]) << 1; next[l] = c; } next[0] = 0; }
  // codes of every symbol (canonical order: by length, then by symbol)
  std::vector<uint32_t> code((size_t)n, 0);
  for (int i = 0; i < n; i++) if (lens[i]) code[(size_t)i] = next[lens[i]]++;
  auto entry_of = [&](int sym, int len) -> Ent {
    if (kind == 0) {
      if (sym < 256) return Ent{(uint16_t)sym, (uint8_t)len, 0};
      if (sym == 256) return Ent{0, (uint8_t)len, 2};
      if (sym > 285) return Ent{0, (uint8_t)len, 15};
      return Ent{LEN_BASE[sym - 257], (uint8_t)len, (uint8_t)(1 | (LEN_EXTRA[sym - 257] << 4))};
    }
    if (sym > 29) return Ent{0, (uint8_t)len, 15};
    return Ent{DIST_BASE[sym], (uint8_t)len, (uint8_t)(1 | (DIST_EXTRA[sym] << 4))};
  };
  // sub-tables: the longest code behind each primary prefix
  std::vector<uint8_t> sub_bits;
  if (maxlen > pb) {
    sub_bits.assign((size_t)1 << pb, 0);
    for (int i = 0; i < n; i++) if (lens[i] > pb) { const uint32_t r = rev_bits(code[(size_t)i], lens[i]) & ((1u << pb) - 1); sub_bits[r] = std::max<uint8_t>(sub_bits[r], (uint8_t)(lens[i] - pb)); }
    for (uint32_t p = 0; p < (1u << pb); p++) if (sub_bits[p]) {
      const size_t off = T->e.size();
      if (off > 0xFFFF) return false;
      T->e[p] = Ent{(uint16_t)off, (uint8_t)pb, (uint8_t)(3 | (sub_bits[p] << 4))};
      T->e.resize(off + ((size_t)1 << sub_bits[p]), bad);
    }
  }
  for (int i = 0; i < n; i++) {
    const int len = lens[i];
    if (!len) continue;
    const uint32_t r = rev_bits(code[(size_t)i], len);
    const Ent en = entry_of(i, len);
    if (len <= pb) { for (uint32_t k = r; k < (1u << pb); k += 1u << len) T->e[k] = en; }
    else {
      const uint32_t p = r & ((1u << pb) - 1), hi = r >> pb;
      const Ent ptr = T->e[p];
      const int sb = ptr.op >> 4;
      for (uint32_t k = hi; k < (1u << sb); k += 1u << (len - pb)) T->e[(size_t)ptr.val + k] = en;
    }
  }
  // Literal/length table: where the index bits behind a literal's code hold the whole code of another literal, the entry yields both.
  // Decoding a literal is a chain of dependent steps (look up, shift, look up ...): text whose frequent symbols have codes of two or three
  // bits -- the bases of a FASTQ file -- takes half as many steps.  (The bits behind the first code index the table with zeros above them: an
  // entry whose code is no longer than the bits that are known is the entry every completion of the index would give.)
  if (kind == 0) {
    Ent single[1u << LIT_PB];
    memcpy(single, T->e.data(), sizeof single);
    for (uint32_t k = 0; k < (1u << pb); k++) {
      const Ent a = single[k];
      if (a.op != 0 || a.bits >= pb) continue;
      const Ent b2 = single[k >> a.bits];
      if (b2.op != 0 || a.bits + b2.bits > pb) continue;
      T->e[k] = Ent{(uint16_t)(a.val | (b2.val << 8)), (uint8_t)(a.bits + b2.bits), 4};
    }
  }
  return true;
}
inline Ent lookup(const Table& T, const Bits& b) {
  Ent e = T.e[b.peek(T.pb)];
  if ((e.op & 15) == 3) e = T.e[(size_t)e.val + ((b.buf >> T.pb) & ((1u << (e.op >> 4)) - 1))];
  return e;
}

struct Codes { Table lit, dist; bool has_dist = true; };
inline const Codes& fixed_codes() {
  static const Codes F = [] {
    Codes c; uint8_t l[288];
    for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    build_table(l, 288, 0, &c.lit);
    uint8_t d[32]; for (int i = 0; i < 32; i++) d[i] = 5;
    build_table(d, 32, 1, &c.dist);
    return c;
  }();
  return F;
}
// the header of a dynamic block (behind BFINAL / BTYPE); false = not a valid header
inline bool read_dynamic_header(Bits& b, Codes* C) {
  b.refill();
  const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
  if (hlit > 286 || hdist > 30) return false;
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint8_t cl[19] = {0};
  for (int i = 0; i < hclen; i++) { if (b.cnt < 3) b.refill(); cl[order[i]] = (uint8_t)b.take(3); }
  // the code-length code: complete, 7 bits at most -> one flat table
  { int left = 1, used = 0, cnt[8] = {0}; for (int i = 0; i < 19; i++) cnt[cl[i]]++;
    for (int l = 1; l <= 7; l++) { left <<= 1; left -= cnt[l]; if (left < 0) return false; used += cnt[l]; }
    if (left > 0 && used != 1) return false;
    if (used == 0) return false; }
  uint8_t clt_sym[128], clt_len[128]; memset(clt_len, 0, sizeof clt_len);
  { uint32_t next[8], c = 0; int cnt[8] = {0}; for (int i = 0; i < 19; i++) cnt[cl[i]]++; cnt[0] = 0;
    for (int l = 1; l <= 7; l++) { c = (c + (uint32_t)cnt[l - 1]) << 1; next[l] = c; }
    for (int i = 0; i < 19; i++) if (cl[i]) { const uint32_t r = rev_bits(next[cl[i]]++, cl[i]); for (uint32_t k = r; k < 128; k += 1u << cl[i]) { clt_sym[k] = (uint8_t)i; clt_len[k] = cl[i]; } } }
  uint8_t lens[320]; int i = 0;
  const int total = hlit + hdist;
  while (i < total) {
    b.refill();
    const uint32_t k = b.peek(7);
    if (!clt_len[k]) return false;
    const int sym = clt_sym[k]; b.drop(clt_len[k]);
    if (sym < 16) lens[i++] = (uint8_t)sym;
    else {
      int rep; uint8_t v = 0;
      if (sym == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)b.take(2); }
      else if (sym == 17) rep = 3 + (int)b.take(3);
      else rep = 11 + (int)b.take(7);
      if (i + rep > total) return false;
      while (rep--) lens[i++] = v;
    }
  }
  if (b.eof()) return false;
  if (lens[256] == 0) return false;   // no end-of-block code
  if (!build_table(lens, hlit, 0, &C->lit)) return false;
  bool any = false; for (int j = 0; j < hdist; j++) any = any || lens[hlit + j];
  C->has_dist = any;
  if (any && !build_table(lens + hlit, hdist, 1, &C->dist)) return false;
  return true;
}

// ---- decoding: T = uint8_t (the window is known: the WIN bytes in front of out[0] are real text) or uint16_t (markers) ----------
enum { D_STOP = 1, D_FINAL = 2, D_ERROR = -1 };
// a growable array that is neither zeroed nor copied element-wise (the output of a chunk is tens of megabytes, written once)
template <class T>
struct RawBuf {
  T* p = nullptr; size_t n = 0;
  RawBuf() = default;
  explicit RawBuf(size_t elems) { grow(elems); }
  RawBuf(const RawBuf&) = delete; RawBuf& operator=(const RawBuf&) = delete;
  RawBuf(RawBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  RawBuf& operator=(RawBuf&& o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~RawBuf() { free(p); }
  void grow(size_t elems) { if (elems <= n) return; const size_t nn = std::max(elems, n + n / 2); T* q = (T*)realloc(p, nn * sizeof(T)); if (!q) throw std::bad_alloc(); p = q; n = nn; }
  size_t size() const { return n; }
  T* data() { return p; }
};
template <class T>
struct Out {   // v[0 .. WIN): the window in front of the output; the output grows from v[WIN]
  RawBuf<T>* v; size_t o;
  T* ensure(size_t more) { v->grow(o + more); return v->data(); }
};
inline bool is_text(uint32_t c) { return c == 9 || c == 10 || c == 13 || (c >= 32 && c < 127); }
// Decodes whole blocks from the reader's position until a block ends at or behind stop_bit (D_STOP), the final block ends (D_FINAL)
// or the data is invalid (D_ERROR).  *end_bit: the bit position behind the last complete block.  ascii_only: a literal that is
// not text makes the data invalid (used while a found block start is still unverified).
template <class T>
inline int inflate_blocks(Bits& b, Out<T>& O, uint64_t stop_bit, uint64_t* end_bit, bool ascii_only) {
  Codes dyn;
  for (;;) {
    *end_bit = b.bitpos();
    if (*end_bit >= stop_bit) return D_STOP;
    b.refill();
    if (b.eof()) return D_ERROR;
    const uint32_t bfinal = b.take(1), btype = b.take(2);
    if (btype == 3) return D_ERROR;
    if (btype == 0) {
      b.drop(b.cnt & 7);   // to the byte boundary
      b.refill();
      const uint32_t len = b.take(16), nlen = b.take(16);
      if ((len ^ nlen) != 0xFFFFu) return D_ERROR;
      T* out = O.ensure(len + 8);
      for (uint32_t left = len; left; --left) {
        if (b.cnt < 8) { b.refill(); if (b.eof()) return D_ERROR; }
        const uint32_t c = b.take(8);
        if (ascii_only && !is_text(c)) return D_ERROR;
        out[O.o++] = (T)c;
      }
    } else {
      const Codes* C = &fixed_codes();
      if (btype == 2) { if (!read_dynamic_header(b, &dyn)) return D_ERROR; C = &dyn; }
      // the hot loop works on copies of the reader's state and of the table pointers (the output stores may alias neither)
      const Ent* const lt = C->lit.e.data(); const Ent* const dt = C->dist.e.data();
      const bool has_dist = C->has_dist;
      Bits r = b;
      size_t o = O.o;
      T* out = O.ensure(1u << 16);
      size_t lim = O.v->size() - 320;   // room for six literals (one more is written behind them) and one match (plus what the word-wise copy writes too far) is checked once per iteration
      int rc_in = 0;
      // The look-up of the next symbol is issued as early as its bits are known: behind the last literal of a run, and in front of a
      // match's copy (the table load and the copy's loads and stores then run side by side).  A length / distance code and its extra
      // bits leave the bit buffer in one shift; the extra bits are cut out of a saved copy, off the chain that leads to the next look-up.
#define KAMD_PGZ_LOOKUP(E)                                                                                                           \
      {                                                                                                                              \
        E = lt[r.buf & ((1u << LIT_PB) - 1)];                                                                                        \
        if (__builtin_expect(((E).op & 15) == 3, 0)) E = lt[(size_t)(E).val + ((r.buf >> LIT_PB) & ((1u << ((E).op >> 4)) - 1))];   \
      }
      // both bytes of a literal entry are stored, the output advances by one or two (what lies behind is overwritten by what comes next)
#define KAMD_PGZ_LITS(E)                                                                                                             \
      {                                                                                                                              \
        if (ascii_only && (!is_text((E).val & 0xFF) || ((E).op == 4 && !is_text((E).val >> 8)))) { rc_in = D_ERROR; break; }        \
        r.drop((E).bits);                                                                                                            \
        out[o] = (T)((E).val & 0xFF); out[o + 1] = (T)((E).val >> 8);                                                               \
        o += 1 + ((E).op >> 2);                                                                                                      \
      }
      r.refill();
      Ent e;
      KAMD_PGZ_LOOKUP(e)
      for (;;) {   // e: the entry of the next symbol; at least 56 - 30 bits behind it in the buffer
        if (__builtin_expect(o > lim, 0)) { O.o = o; out = O.ensure(1u << 16); lim = O.v->size() - 320; }
        if ((e.op & 11) == 0) {   // literals: up to three look-ups per refill (3 x 15 bits), each one literal or two
          KAMD_PGZ_LITS(e)
          KAMD_PGZ_LOOKUP(e)
          if ((e.op & 11) == 0) {
            KAMD_PGZ_LITS(e)
            KAMD_PGZ_LOOKUP(e)
            if ((e.op & 11) == 0) {
              KAMD_PGZ_LITS(e)
              if (__builtin_expect(r.over != 0, 0) && r.eof()) { rc_in = D_ERROR; break; }   // (zeros invented behind the end of the data decode to literals for ever)
              r.refill();
              KAMD_PGZ_LOOKUP(e)
              continue;
            }
          }
          r.refill();   // (e was looked up in bits that were valid already: refilling only adds bits above them)
        }
        const int op = e.op & 15;
        if (__builtin_expect(op == 1, 1)) {
          const uint64_t saved_l = r.buf;
          const int xl = e.op >> 4;
          r.drop(e.bits + xl);
          const uint32_t len = (uint32_t)e.val + (uint32_t)((saved_l >> e.bits) & ((1u << xl) - 1));
          if (!has_dist) { rc_in = D_ERROR; break; }
          Ent d = dt[r.buf & ((1u << DIST_PB) - 1)];
          if (__builtin_expect((d.op & 15) == 3, 0)) d = dt[(size_t)d.val + ((r.buf >> DIST_PB) & ((1u << (d.op >> 4)) - 1))];
          if ((d.op & 15) != 1) { rc_in = D_ERROR; break; }
          const uint64_t saved_d = r.buf;
          const int xd = d.op >> 4;
          r.drop(d.bits + xd);
          const uint32_t dist = (uint32_t)d.val + (uint32_t)((saved_d >> d.bits) & ((1u << xd) - 1));
          if (dist > WIN) { rc_in = D_ERROR; break; }
          r.refill();
          KAMD_PGZ_LOOKUP(e)
          const T* s = out + o - dist; T* t = out + o;
          constexpr uint32_t W = 8 / sizeof(T);   // elements per 64-bit word
          if (dist >= 2 * W) {   // 16 bytes at a time, up to 4 W - 1 elements too far (the room is there)
            // zlib turns the bases of a FASTQ file into matches of ten symbols on average: the first 32 bytes go without a look at the
            // length (no loop whose trip count the branch predictor would have to guess), only longer matches loop
            { uint64_t w0, w1; memcpy(&w0, s, 8); memcpy(&w1, s + W, 8); memcpy(t, &w0, 8); memcpy(t + W, &w1, 8); }
            { uint64_t w0, w1; memcpy(&w0, s + 2 * W, 8); memcpy(&w1, s + 3 * W, 8); memcpy(t + 2 * W, &w0, 8); memcpy(t + 3 * W, &w1, 8); }
            if (__builtin_expect(len > 4 * W, 0))
              for (uint32_t i = 4 * W; i < len; i += 2 * W) { uint64_t w0, w1; memcpy(&w0, s + i, 8); memcpy(&w1, s + i + W, 8); memcpy(t + i, &w0, 8); memcpy(t + i + W, &w1, 8); }
          } else if (dist >= W) {
            for (uint32_t i = 0; i < len; i += W) { uint64_t w; memcpy(&w, s + i, 8); memcpy(t + i, &w, 8); }
          } else for (uint32_t i = 0; i < len; i++) t[i] = s[i];
          o += len;
        } else if (op == 2) { r.drop(e.bits); break; }
        else { rc_in = D_ERROR; break; }
        if (__builtin_expect(r.over != 0, 0) && r.eof()) { rc_in = D_ERROR; break; }
      }
#undef KAMD_PGZ_LITS
#undef KAMD_PGZ_LOOKUP
      b = r;
      if (rc_in) return rc_in;
      O.o = o;
    }
    if (b.eof()) return D_ERROR;
    if (bfinal) { *end_bit = b.bitpos(); return D_FINAL; }
  }
}

// ---- finding a block start ---------------------------------------------------------------------------------------------------
// First bit position in [from_bit, until_bit) where a non-final dynamic-Huffman block with a valid header starts whose first
// symbols decode to text; ~0 if none.  (Stored and fixed-Huffman blocks carry nothing to recognise them by: a chunk that begins
// with one is decoded by the serial path.)
inline uint64_t find_block(const uint8_t* data, size_t n, uint64_t from_bit, uint64_t until_bit) {
  RawBuf<uint16_t> scratch(WIN + (1u << 17));
  for (uint64_t p = from_bit; p < until_bit; p++) {
    if ((p >> 3) + 8 >= n) break;
    // cheap filter on the first 13 bits: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29
    uint64_t w; memcpy(&w, data + (p >> 3), 8); w >>= (p & 7);
    if ((w & 7) != 4) continue;
    if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
    Bits b(data, n, p + 3);
    Codes C;
    if (!read_dynamic_header(b, &C)) continue;   // (rejects at the code-length code's Kraft sum nearly always)
    // trial: the whole block must decode, to text
    for (uint32_t i = 0; i < WIN; i++) scratch.p[i] = (uint16_t)(MARK | i);
    Bits t(data, n, p);
    Out<uint16_t> O{&scratch, WIN};
    uint64_t eb = 0;
    if (inflate_blocks<uint16_t>(t, O, p + 1, &eb, true) == D_ERROR) continue;   // (stop_bit = p + 1: exactly the block at p)
    return p;
  }
  return ~0ULL;
}

// ---- gzip member header / trailer ----------------------------------------------------------------------------------------------
// returns the offset of the deflate data, 0 = no gzip header at `at`
inline size_t gzip_header(const uint8_t* d, size_t n, size_t at) {
  if (at + 18 > n || d[at] != 0x1f || d[at + 1] != 0x8b || d[at + 2] != 8) return 0;
  const int flg = d[at + 3];
  size_t p = at + 10;
  if (flg & 4) { if (p + 2 > n) return 0; p += 2 + ((size_t)d[p] | ((size_t)d[p + 1] << 8)); }
  if (flg & 8) { while (p < n && d[p]) ++p; ++p; }
  if (flg & 16) { while (p < n && d[p]) ++p; ++p; }
  if (flg & 2) p += 2;
  return p < n ? p : 0;
}

// ---- the parallel reader ----------------------------------------------------------------------------------------------------------
// deliver(bytes, n, newlines): the text, in order, from ONE thread (the one that calls run()); returns false to stop early.
class ParGzip {
 public:
  struct Stats { uint64_t chunks = 0, parallel = 0, serial_bytes = 0, rejected = 0, members = 0; };
  ParGzip(const uint8_t* data, size_t n, int threads, size_t chunk_bytes, std::function<bool(const uint8_t*, size_t, uint64_t)> deliver,
          uint64_t (*count_nl)(const char*, size_t), uint32_t (*crc_fn)(uint32_t, const void*, size_t) = nullptr)
      : d_(data), n_(n), T_(std::max(1, threads)), C_(std::max<size_t>(chunk_bytes, 1 << 12)), deliver_(std::move(deliver)), count_nl_(count_nl), crc_fn_(crc_fn) {}
  ~ParGzip() { stop_workers(); }
  const std::string& error() const { return error_; }
  const Stats& stats() const { return st_; }

  // decodes the whole file; false = error (error() says what) or stopped by deliver
  bool run() {
    const size_t ds = gzip_header(d_, n_, 0);
    if (!ds) { error_ = "not a gzip stream"; return false; }
    data_start_ = ds;
    // the first chunks are an eighth of the size (two rounds of the workers): the text starts to flow after milliseconds instead of after
    // the time a whole chunk takes -- what a small file consists of
    small_ = C_ >= (1u << 16) ? C_ / 8 : C_;
    n_small_ = small_ < C_ ? (size_t)(2 * T_) : 0;
    {
      const size_t bytes = n_ - ds;
      if (bytes <= n_small_ * small_) n_chunks_ = (bytes + small_ - 1) / small_;
      else n_chunks_ = n_small_ + (bytes - n_small_ * small_ + C_ - 1) / C_;
    }
    chunks_.resize(n_chunks_);
    for (auto& c : chunks_) c.reset(new Chunk);
    for (int t = 0; t < T_; t++) workers_.emplace_back([this] { worker(); });
    const bool ok = coordinate();
    stop_workers();
    return ok;
  }

 private:
  enum State { IDLE = 0, DECODED, CONVERTED };
  struct Chunk {
    std::mutex m; std::condition_variable cv;
    int state = IDLE;
    bool start_known = false; uint64_t start_bit = ~0ULL;   // first block found at or behind the chunk's nominal start
    uint64_t end_bit = 0; int rc = D_ERROR;
    RawBuf<uint16_t> sym;         // WIN markers in front, then the symbols (from the pool)
    size_t n_out = 0;
    std::vector<uint8_t> window;  // the WIN bytes in front of the chunk (set by the coordinator before conversion)
    RawBuf<uint8_t> bytes; size_t n_bytes = 0; uint64_t nl = 0; uint32_t crc = 0;
    bool computing = false;       // some thread is looking for the chunk's block start
  };
  uint64_t nominal_bit(size_t k) const {
    if (k >= n_chunks_) return (uint64_t)n_ * 8;
    const uint64_t off = k <= n_small_ ? (uint64_t)k * small_ : (uint64_t)n_small_ * small_ + (uint64_t)(k - n_small_) * C_;
    return ((uint64_t)data_start_ + off) * 8;
  }
  // start of chunk k's first block (memoised; any thread)
  uint64_t start_of(size_t k) {
    if (k >= n_chunks_) return ~0ULL;
    Chunk& c = *chunks_[k];
    {
      std::unique_lock<std::mutex> lk(c.m);
      if (c.computing) c.cv.wait(lk, [&] { return c.start_known; });
      if (c.start_known) return c.start_bit;
      c.computing = true;
    }
    uint64_t s = ~0ULL;
    try { KAMD_PGZ_TIME(4); s = k == 0 ? (uint64_t)data_start_ * 8 : find_block(d_, n_, nominal_bit(k), nominal_bit(k + 1)); }
    catch (...) { fatal_ = true; }   // (out of memory: "no block found" lets everyone waiting for this answer go on; the coordinator gives up)
    std::lock_guard<std::mutex> lk(c.m);
    c.start_bit = s; c.start_known = true;
    c.cv.notify_all();
    return s;
  }
  // the first chunk behind k that has a block start: where chunk k's decoder stops
  uint64_t stop_for(size_t k) {
    for (size_t j = k + 1; j < n_chunks_; j++) { const uint64_t s = start_of(j); if (s != ~0ULL) return s; }
    return ~0ULL;
  }
  void decode_chunk(size_t k) {
    Chunk& c = *chunks_[k];
    const uint64_t s = start_of(k);
    int rc = D_ERROR; uint64_t eb = 0;
    if (s != ~0ULL) {
      const uint64_t stop = stop_for(k);
      c.sym = take_sym();
      c.sym.grow(WIN + 5 * (k < n_small_ ? small_ : C_) + (1u << 17));
      for (uint32_t i = 0; i < WIN; i++) c.sym.p[i] = (uint16_t)(MARK | i);
      Bits b(d_, n_, s);
      Out<uint16_t> O{&c.sym, WIN};
      rc = D_STOP;
      KAMD_PGZ_TIME(0);
      if (k != 0) rc = inflate_blocks<uint16_t>(b, O, s + 1, &eb, true);   // the found block itself: must be text (the finder tried it already)
      if (rc == D_STOP) rc = inflate_blocks<uint16_t>(b, O, stop, &eb, false);
      c.n_out = O.o - WIN;
    }
    std::lock_guard<std::mutex> lk(c.m);
    c.rc = rc; c.end_bit = eb; c.state = DECODED;
    c.cv.notify_all();
  }
  void convert_chunk(size_t k) {
    Chunk& c = *chunks_[k];
    c.bytes = take_bytes();
    c.bytes.grow(c.n_out + 16);
    c.n_bytes = c.n_out;
    const uint16_t* s = c.sym.data() + WIN; const uint8_t* w = c.window.data(); uint8_t* o = c.bytes.data();
    { KAMD_PGZ_TIME(1); resolve(s, c.n_out, w, o); }
    { KAMD_PGZ_TIME(2); c.nl = count_nl_ ? count_nl_((const char*)o, c.n_out) : 0; }
    { KAMD_PGZ_TIME(3); c.crc = crc_of(o, c.n_out); }
    give_sym(std::move(c.sym));
    std::lock_guard<std::mutex> lk(c.m);
    c.state = CONVERTED;
    c.cv.notify_all();
  }
  // buffers go round: a decoded chunk's symbols and a converted chunk's bytes are tens of megabytes that would otherwise be mapped,
  // faulted in and unmapped for every chunk
  RawBuf<uint16_t> take_sym() { std::lock_guard<std::mutex> lk(pm_); if (sym_pool_.empty()) return RawBuf<uint16_t>(); RawBuf<uint16_t> b = std::move(sym_pool_.back()); sym_pool_.pop_back(); return b; }
  void give_sym(RawBuf<uint16_t>&& b) { if (!b.p) return; std::lock_guard<std::mutex> lk(pm_); sym_pool_.push_back(std::move(b)); }
  RawBuf<uint8_t> take_bytes() { std::lock_guard<std::mutex> lk(pm_); if (byte_pool_.empty()) return RawBuf<uint8_t>(); RawBuf<uint8_t> b = std::move(byte_pool_.back()); byte_pool_.pop_back(); return b; }
  void give_bytes(RawBuf<uint8_t>&& b) { if (!b.p) return; std::lock_guard<std::mutex> lk(pm_); byte_pool_.push_back(std::move(b)); }
  // symbols -> bytes.  FASTQ text keeps its markers for ever (the constant parts of a record are copied from the record before, and so on back
  // into the unknown window: half of all symbols in a file with constant quality lines), so the general case is a table look-up without a
  // branch: tab[v] = v for a literal byte, tab[MARK | i] = window[i] -- 64 KiB per worker thread, the upper half rewritten per chunk.  Runs of
  // sixteen symbols without a marker are packed in one go.
  static void resolve(const uint16_t* s, size_t n, const uint8_t* w, uint8_t* o) {
    static thread_local std::unique_ptr<uint8_t[]> tab_holder;
    if (!tab_holder) { tab_holder.reset(new uint8_t[1u << 16]); for (uint32_t v = 0; v < 0x8000u; v++) tab_holder[v] = (uint8_t)v; }
    uint8_t* const tab = tab_holder.get();
    memcpy(tab + MARK, w, WIN);
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i mk = _mm_set1_epi16((short)0x8000);
    for (; i + 16 <= n; i += 16) {
      const __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 8));
      if (_mm_movemask_epi8(_mm_and_si128(_mm_or_si128(a, b), mk)) == 0) _mm_storeu_si128((__m128i*)(o + i), _mm_packus_epi16(a, b));
      else {
        const uint16_t* q = s + i; uint8_t* t = o + i;
        t[0] = tab[q[0]]; t[1] = tab[q[1]]; t[2] = tab[q[2]]; t[3] = tab[q[3]]; t[4] = tab[q[4]]; t[5] = tab[q[5]]; t[6] = tab[q[6]]; t[7] = tab[q[7]];
        t[8] = tab[q[8]]; t[9] = tab[q[9]]; t[10] = tab[q[10]]; t[11] = tab[q[11]]; t[12] = tab[q[12]]; t[13] = tab[q[13]]; t[14] = tab[q[14]]; t[15] = tab[q[15]];
      }
    }
#endif
    for (; i < n; i++) o[i] = tab[s[i]];
  }
  uint32_t crc_of(const uint8_t* p, size_t n) const {
    if (crc_fn_) return crc_fn_(0, p, n);
    uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
    while (n) { const size_t a = std::min<size_t>(n, 1u << 30); c = (uint32_t)crc32(c, p, (uInt)a); p += a; n -= a; }
    return c;
  }
  void worker() {
    for (;;) {
      size_t k; bool conv;
      {
        std::unique_lock<std::mutex> lk(qm_);
        qcv_.wait(lk, [&] { return stop_ || !conv_q_.empty() || (next_decode_ < n_chunks_ && next_decode_ < window_hi_); });
        if (stop_) return;
        if (!conv_q_.empty()) { k = conv_q_.front(); conv_q_.pop_front(); conv = true; }
        else { k = next_decode_++; conv = false; }
      }
      try {
        if (conv) convert_chunk(k); else decode_chunk(k);
      } catch (...) {   // (out of memory for a chunk's buffers): the coordinator gives up instead of waiting for this chunk for ever
        Chunk& c = *chunks_[k];
        fatal_ = true;
        std::lock_guard<std::mutex> lk(c.m);
        c.rc = D_ERROR; c.state = conv ? CONVERTED : DECODED; c.n_bytes = 0;
        c.cv.notify_all();
      }
    }
  }
  void stop_workers() {
    { std::lock_guard<std::mutex> lk(qm_); stop_ = true; }
    qcv_.notify_all();
    for (auto& t : workers_) if (t.joinable()) t.join();
    workers_.clear();
  }
  // chunks below `hi` may be decoded (bounds the memory of decoded chunks that wait for their turn)
  void allow_up_to(size_t hi) {
    { std::lock_guard<std::mutex> lk(qm_); if (hi > window_hi_) window_hi_ = hi; }
    qcv_.notify_all();
  }
  void fold(const uint8_t* p, size_t n, uint32_t crc) {   // the member's checksum and size; the window = the last WIN bytes of the text
    crc_ = (uint32_t)crc32_combine(crc_, crc, (z_off_t)n);
    member_out_ += n;
    (void)p;
  }
  void slide_window(const uint8_t* p, size_t n) {
    if (n >= WIN) memcpy(win_.data(), p + n - WIN, WIN);
    else if (n) { memmove(win_.data(), win_.data() + n, WIN - n); memcpy(win_.data() + WIN - n, p, n); }
  }
  // text the coordinator decoded itself
  bool emit(const uint8_t* p, size_t n) {
    if (!n) return true;
    fold(p, n, crc_of(p, n));
    slide_window(p, n);
    return deliver_(p, n, count_nl_ ? count_nl_((const char*)p, n) : 0);
  }
  // serial decode from cur_bit_ (window known) until a block ends at or behind stop_bit / the member ends; the text is emitted
  int serial_until(uint64_t stop_bit, uint64_t* end_bit) {
    RawBuf<uint8_t> buf(WIN + (1u << 20));
    Bits b(d_, n_, cur_bit_);
    uint64_t at = cur_bit_;
    for (;;) {
      // a few blocks at a time, so that the text leaves in pieces
      memcpy(buf.data(), win_.data(), WIN);
      Out<uint8_t> O{&buf, WIN};
      const uint64_t piece_stop = std::min<uint64_t>(stop_bit, at + (8u << 20));
      const int rc = inflate_blocks<uint8_t>(b, O, piece_stop, end_bit, false);
      st_.serial_bytes += O.o - WIN;
      if (!emit(buf.data() + WIN, O.o - WIN)) { stopped_ = true; return D_ERROR; }
      if (rc != D_STOP || *end_bit >= stop_bit) return rc;
      at = *end_bit;
    }
  }
  bool coordinate() {
    win_.assign(WIN, 0);
    cur_bit_ = (uint64_t)data_start_ * 8;
    crc_ = 0; member_out_ = 0; st_.members = 1;
    std::deque<size_t> pending;   // accepted chunks, in order, not delivered yet
    const size_t ahead = (size_t)(2 * T_ + 2);
    allow_up_to(ahead);
    auto flush_one = [&]() -> bool {
      const size_t j = pending.front(); pending.pop_front();
      Chunk& c = *chunks_[j];
      { std::unique_lock<std::mutex> lk(c.m); c.cv.wait(lk, [&] { return c.state == CONVERTED; }); }
      if (fatal_) { if (error_.empty()) error_ = "out of memory while inflating"; return false; }
      fold(c.bytes.data(), c.n_bytes, c.crc);
      const bool ok = deliver_(c.bytes.data(), c.n_bytes, c.nl);
      give_bytes(std::move(c.bytes)); std::vector<uint8_t>().swap(c.window);
      if (!ok) stopped_ = true;
      return ok;
    };
    auto fail = [&](const char* msg) { if (error_.empty() && !stopped_) error_ = msg; return false; };
    size_t k = 0;
    for (;;) {
      if (k >= n_chunks_) {   // nothing left to take from the workers: the rest of the stream serially
        while (!pending.empty()) if (!flush_one()) return false;
        uint64_t eb = 0;
        const int rc = serial_until(~0ULL, &eb);
        if (rc != D_FINAL) return fail("corrupt gzip stream");
        cur_bit_ = eb;
        const int m = next_member();
        if (m <= 0) return m == 0;
        continue;
      }
      Chunk& c = *chunks_[k];
      // the workers may decode up to `ahead` chunks beyond the oldest one that is not done with (delivered or skipped): chunk k must be among them
      while (!pending.empty() && k >= pending.front() + ahead) if (!flush_one()) return false;
      allow_up_to((pending.empty() ? k : pending.front()) + ahead);
      { std::unique_lock<std::mutex> lk(c.m); c.cv.wait(lk, [&] { return c.state >= DECODED; }); }
      if (fatal_) return fail("out of memory while inflating");
      const uint64_t s = c.start_bit;
      if (s == ~0ULL || s < cur_bit_ || c.rc == D_ERROR) {   // no block found in the chunk / already covered / the found one was none
        if (s != ~0ULL) ++st_.rejected;
        give_sym(std::move(c.sym));
        ++st_.chunks; ++k;
        continue;
      }
      if (s > cur_bit_) {
        // the piece in front of the chunk's block: serially, with the window (everything before it is delivered first)
        while (!pending.empty()) if (!flush_one()) return false;
        uint64_t eb = 0;
        const int rc = serial_until(s, &eb);
        if (rc == D_ERROR) return fail("corrupt gzip stream");
        cur_bit_ = eb;
        if (rc == D_FINAL) { const int m = next_member(); if (m <= 0) return m == 0; }
        continue;   // chunk k again: cur_bit_ == s (accept), > s (reject) or still < s (a new member began in between)
      }
      // s == cur_bit_: the chunk continues the verified stream.  Its window is the last WIN bytes so far; the next window is its own
      // last WIN symbols with their markers resolved.
      ++st_.parallel; ++st_.chunks;
      c.window = win_;
      {
        const uint16_t* sy = c.sym.data() + WIN;
        const size_t n = c.n_out;
        const uint8_t* w = c.window.data();
        if (n >= WIN) { for (uint32_t i = 0; i < WIN; i++) { const uint16_t v = sy[n - WIN + i]; win_[i] = v & MARK ? w[v & 0x7FFF] : (uint8_t)v; } }
        else { memmove(win_.data(), win_.data() + n, WIN - n); for (size_t i = 0; i < n; i++) { const uint16_t v = sy[i]; win_[WIN - n + i] = v & MARK ? w[v & 0x7FFF] : (uint8_t)v; } }
      }
      { std::lock_guard<std::mutex> lk(qm_); conv_q_.push_back(k); }
      qcv_.notify_all();
      pending.push_back(k);
      cur_bit_ = c.end_bit;
      const int rc = c.rc;
      ++k;
      while (pending.size() > (size_t)(T_ + 1)) if (!flush_one()) return false;
      if (rc == D_FINAL) {
        while (!pending.empty()) if (!flush_one()) return false;
        const int m = next_member(); if (m <= 0) return m == 0;
      }
    }
  }
  // behind a final block: trailer check, then the next member if there is one.  1 = another member begins, 0 = end of the text, -1 = error
  int next_member() {
    size_t p = (size_t)((cur_bit_ + 7) >> 3);
    if (p + 8 > n_) { error_ = "unexpected end of the gzip stream"; return -1; }
    const uint32_t crc = (uint32_t)d_[p] | ((uint32_t)d_[p + 1] << 8) | ((uint32_t)d_[p + 2] << 16) | ((uint32_t)d_[p + 3] << 24);
    const uint32_t isz = (uint32_t)d_[p + 4] | ((uint32_t)d_[p + 5] << 8) | ((uint32_t)d_[p + 6] << 16) | ((uint32_t)d_[p + 7] << 24);
    if (crc != crc_ || isz != (uint32_t)member_out_) { error_ = "corrupt gzip stream (CRC or length of a member)"; return -1; }
    p += 8;
    while (p < n_ && d_[p] == 0) ++p;   // zero padding between / behind members is skipped, as gzread does
    if (p >= n_) return 0;
    const size_t ds = gzip_header(d_, n_, p);
    if (!ds) return 0;                  // bytes behind the last member that are no gzip member: ignored, as gzread does
    cur_bit_ = (uint64_t)ds * 8; crc_ = 0; member_out_ = 0; ++st_.members;
    return 1;
  }

  const uint8_t* d_; size_t n_; int T_; size_t C_;
  std::function<bool(const uint8_t*, size_t, uint64_t)> deliver_;
  uint64_t (*count_nl_)(const char*, size_t);
  uint32_t (*crc_fn_)(uint32_t, const void*, size_t);
  size_t data_start_ = 0, n_chunks_ = 0, small_ = 0, n_small_ = 0;
  std::vector<std::unique_ptr<Chunk>> chunks_;
  std::vector<std::thread> workers_;
  std::mutex qm_; std::condition_variable qcv_;
  std::deque<size_t> conv_q_; size_t next_decode_ = 0, window_hi_ = 0; bool stop_ = false;
  std::vector<uint8_t> win_;
  std::mutex pm_; std::vector<RawBuf<uint16_t>> sym_pool_; std::vector<RawBuf<uint8_t>> byte_pool_;
  uint64_t cur_bit_ = 0; uint32_t crc_ = 0; uint64_t member_out_ = 0;
  bool stopped_ = false;
  std::atomic<bool> fatal_{false};
  std::string error_;
  Stats st_;
};

}  // namespace pgz
}  // namespace kamd_io
