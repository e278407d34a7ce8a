// kamd_index.cpp -- host side of seam S1: read a kallisto index (format v13) and flatten it into the tables the
// gfx950 kernels consume.  Replaces KmerIndex::load (src/KmerIndex.cpp:1330-1559) for the quant path.
//
// What is kept from the file: the graph section's 2-bit unitigs (ext/bifrost/src/IO.tcc:1635-1738), the per-unitig
// mosaic blocks with their transcript sets and position words (src/Node.hpp:63-72, src/BlockArray.hpp:441-471,
// src/SparseVector.tcc:424-515), target lengths/names and the on-list.  Bifrost's minimizer index and the BooPHF blob
// are skipped: k-mer lookup is served by our own bucketed Robin-Hood table (layout in kamd_core.h) built here by
// enumerating every k-mer of every unitig, which reproduces CompactedDBG::find exactly because each canonical k-mer
// occurs exactly once in a compacted de Bruijn graph.
#include "../../include/kallisto_amd.h"
#include "kamd_core.h"
#include "kamd_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

namespace {

struct Cursor {
  const uint8_t* p; size_t n; size_t pos = 0; bool bad = false;
  template <class T> T get() {
    T v{};
    if (pos + sizeof(T) > n) { bad = true; return v; }
    memcpy(&v, p + pos, sizeof(T)); pos += sizeof(T);
    return v;
  }
  const uint8_t* take(size_t len) {
    if (pos > n || len > n - pos) { bad = true; return nullptr; }   // (len comes from the file: no wrap-around)
    const uint8_t* r = p + pos; pos += len; return r;
  }
};

// --- Roaring readers (formats: ext/bifrost/src/roaring.c:8554-8629 "native", :10555-10700 portable) -------------------
bool roaring_portable(const uint8_t* b, size_t n, std::vector<uint32_t>& out) {
  Cursor c{b, n};
  uint32_t cookie = c.get<uint32_t>();
  int32_t size; bool hasrun = false;
  if ((cookie & 0xFFFF) == 12347) { size = (int32_t)(cookie >> 16) + 1; hasrun = true; }
  else if (cookie == 12346) size = c.get<int32_t>();
  else return false;
  if (size < 0 || size > 65536) return false;
  const uint8_t* runbm = hasrun ? c.take((size_t)(size + 7) / 8) : nullptr;
  const uint8_t* kc = c.take((size_t)size * 4);
  if (!hasrun || size >= 4) c.take((size_t)size * 4);
  if (c.bad) return false;
  for (int32_t i = 0; i < size; i++) {
    uint16_t key, cm1; memcpy(&key, kc + 4 * i, 2); memcpy(&cm1, kc + 4 * i + 2, 2);
    uint32_t card = (uint32_t)cm1 + 1, hi = (uint32_t)key << 16;
    bool isrun = hasrun && (runbm[i / 8] & (1 << (i % 8)));
    if (isrun) {
      uint16_t nr = c.get<uint16_t>();
      for (uint16_t r = 0; r < nr; r++) {
        uint16_t st = c.get<uint16_t>(), ln = c.get<uint16_t>();
        for (uint32_t x = st; x <= (uint32_t)st + ln; x++) out.push_back(hi | x);
      }
    } else if (card > 4096) {
      const uint8_t* w = c.take(8192);
      if (!w) return false;
      for (uint32_t j = 0; j < 1024; j++) {
        uint64_t word; memcpy(&word, w + 8 * j, 8);
        while (word) { out.push_back(hi | (j * 64 + (uint32_t)__builtin_ctzll(word))); word &= word - 1; }
      }
    } else {
      const uint8_t* a = c.take((size_t)card * 2);
      if (!a) return false;
      for (uint32_t j = 0; j < card; j++) { uint16_t x; memcpy(&x, a + 2 * j, 2); out.push_back(hi | x); }
    }
    if (c.bad) return false;
  }
  return true;
}
bool roaring_native(const uint8_t* b, size_t n, std::vector<uint32_t>& out) {
  out.clear();
  if (n < 1) return false;
  if (b[0] == 1) {
    if (n < 5) return false;
    uint32_t card; memcpy(&card, b + 1, 4);
    if (5 + (size_t)card * 4 > n) return false;
    out.resize(card);
    memcpy(out.data(), b + 5, (size_t)card * 4);
    return true;
  }
  if (b[0] == 2) return roaring_portable(b + 1, n - 1, out);
  return false;
}

// smallest and largest member of a set without materialising it (position sets: only the two are used, and most hold one element)
bool roaring_minmax(const uint8_t* b, size_t n, std::vector<uint32_t>& scratch, uint32_t* mn, uint32_t* mx) {
  if (n >= 5 && b[0] == 1) {   // the native array form: cardinality, then the members in increasing order
    uint32_t card; memcpy(&card, b + 1, 4);
    if (card == 0 || 5 + (size_t)card * 4 > n) return false;
    memcpy(mn, b + 5, 4); memcpy(mx, b + 5 + (size_t)(card - 1) * 4, 4);
    return true;
  }
  if (!roaring_native(b, n, scratch) || scratch.empty()) return false;
  *mn = scratch.front(); *mx = scratch.back();
  return true;
}

struct VecHash {
  size_t operator()(const std::vector<uint32_t>& v) const {
    uint64_t h = 0x9e3779b97f4a7c15ULL ^ v.size();
    for (uint32_t x : v) h = kamd::mix64(h ^ x);
    return (size_t)h;
  }
};

struct RawBlock { uint32_t lb, ub, ec; uint64_t pos_off; };

}  // namespace

// allocator whose resize() leaves new elements uninitialised: the three big tables of a flattened index file are filled by
// parallel reads straight away (value-initialising 3 GB first costs a third of the load)
// Large blocks (the k-mer table and its two aux arrays: gigabytes that are then written at random) come straight from mmap and are
// advised to use transparent huge pages: with 4 KB pages the first touch of 3.4 GB is 0.85 M page faults and the placement pass misses
// the TLB on nearly every k-mer.  Measured on the pool's boxes (human-sized index, 16 CPUs, scratch/round4_calls/r4_call15.sh): "layout +
// allocation" 0.36 -> 0.09 s, the whole load 1.05-1.19 -> 0.86-0.89 s, a flattened file 0.22-0.30 -> 0.07 s; in the 8-CPU build container
// the placement pass goes from 1.85 to 1.17 s while the sandbox's huge-page faults cost what they save.  Where the kernel's THP mode is
// `never` the advice is ignored and nothing changes; KAMD_NO_THP=1 leaves it out.
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { typedef NoInitAlloc<U> other; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  static constexpr size_t HUGE_FROM = 8u << 20, HUGE_PAGE = 2u << 20;
  static size_t mapped_bytes(size_t n) { return (n * sizeof(T) + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1); }
  T* allocate(size_t n) {
    if (n * sizeof(T) >= HUGE_FROM) {
      void* p = mmap(nullptr, mapped_bytes(n), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (p == MAP_FAILED) throw std::bad_alloc();
      static const bool thp = getenv("KAMD_NO_THP") == nullptr;
      if (thp) (void)madvise(p, mapped_bytes(n), MADV_HUGEPAGE);
      return static_cast<T*>(p);
    }
    return std::allocator<T>::allocate(n);
  }
  void deallocate(T* p, size_t n) {
    if (n * sizeof(T) >= HUGE_FROM) { (void)munmap(p, mapped_bytes(n)); return; }
    std::allocator<T>::deallocate(p, n);
  }
  template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using BigVec = std::vector<T, NoInitAlloc<T>>;

struct kamd_index {
  int32_t k = 0;
  uint64_t n_kmers = 0, n_unitigs = 0, n_long = 0, n_short = 0, n_abund = 0, dlist_size = 0;
  std::vector<uint32_t> unitig_len;
  std::vector<uint64_t> unitig_blk_off;
  std::vector<uint32_t> blk_unitig, blk_lb, blk_ub, blk_ec, blk_uec;
  std::vector<uint64_t> blk_pos_off;
  std::vector<uint32_t> blk_posw;
  std::vector<uint8_t> blk_sense;
  std::vector<uint32_t> uec_ec;
  std::vector<uint64_t> ec_off;
  std::vector<uint32_t> ec_ids;
  std::vector<int32_t> target_lens;
  std::vector<std::string> target_names;
  std::vector<uint32_t> onlist_bits;
  uint64_t n_targets = 0;           // real targets (what abundance.tsv lists); target_lens also covers the D-list pseudo-targets
  uint64_t n_buckets = 0, pad_buckets = 0;
  // layout of `table` (kamd_core.h): LAYOUT_WIDE with BUCKET_SLOTS slots per bucket, or LAYOUT_COMPACT with COMPACT_SLOTS and its shifts
  uint32_t layout = kamd::LAYOUT_WIDE, slots = kamd::BUCKET_SLOTS, tag_q = 0, tag_dsh = 0, tag_w = 0;
  BigVec<uint64_t> table;
  BigVec<uint32_t> slot_block, slot_dist;
  std::vector<uint32_t> utext;          // 2-bit text of all unitigs, end to end (read packing), + 2 words of padding
  std::vector<uint64_t> unitig_gpos;    // first base of every unitig in it
  uint64_t text_bases = 0;
  // D-list
  std::vector<uint64_t> dlist_keys;   // canonical, right-aligned; [0] = the dummy (the one that is in the graph)
  uint64_t n_dbuckets = 0, dpad_buckets = 0;
  std::vector<uint64_t> dtable;
  uint64_t dummy_slot = 0; uint32_t dummy_uec = 0, dummy_strand = 0;
  // identity of the kallisto index these tables were built from (size in bytes, hash of its first and last 64 KiB): written into a
  // flattened file so that one picked up beside an index can be tied to that index (kamd_flat_index_matches); 0 / 0 = unknown
  uint64_t src_size = 0, src_hash = 0;
};

namespace {

// k-mers of one unitig, MSB-first right-aligned, forward orientation; calls f(dist, fwd_value)
template <class F>
inline void for_each_kmer(const uint8_t* packed, uint64_t len, int k, F&& f) {
  const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t v = 0;
  for (uint64_t j = 0; j < len; j++) {
    uint64_t b = (packed[j >> 2] >> ((j & 3) << 1)) & 3;  // CompressedSequence::getChar (CompressedSequence.cpp:311-314)
    v = ((v << 2) | b) & mask;
    if (j + 1 >= (uint64_t)k) f((uint32_t)(j + 1 - k), v);
  }
}
// the same with the reverse complement rolled along (what kamd::revcomp_msb(v, k) gives): calls f(dist, fwd_value, rc_value)
template <class F>
inline void for_each_kmer_rc(const uint8_t* packed, uint64_t len, int k, F&& f) {
  const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const int top = 2 * (k - 1);
  uint64_t v = 0, rc = 0;
  for (uint64_t j = 0; j < len; j++) {
    const uint64_t b = (packed[j >> 2] >> ((j & 3) << 1)) & 3;
    v = ((v << 2) | b) & mask;
    rc = (rc >> 2) | ((3 - b) << top);
    if (j + 1 >= (uint64_t)k) f((uint32_t)(j + 1 - k), v, rc);
  }
}

}  // namespace

// ---- the flattened index as a file (kamd_index_save / kamd_index_load on such a file) ------------------------------------------
// Building the device tables from a kallisto index takes seconds (enumerate the k-mers of every unitig twice, hash, place); a
// front-end that runs sample after sample against one index can write them once and read them back with plain reads.  Layout:
// magic, format version, the scalars, then every array as {u64 count, bytes}; native endianness, for this machine's eyes only.
namespace {
const char FLAT_MAGIC[8] = {'K', 'A', 'M', 'D', 'F', 'L', 'T', '4'};   // 4: the source index's identity behind the stamp
// what the layout of the tables depends on, written behind the magic and compared on load
const uint32_t FLAT_STAMP[4] = {(uint32_t)kamd::BUCKET_SLOTS, (uint32_t)sizeof(uint64_t) * 8u /* bytes per bucket */, 30u /* bits of a class id in the payload */, 13u /* kallisto index version */};
// identity of a kallisto index file: FNV-1a over its size and its first and last 64 KiB (header, k, start of the graph; transcript names and
// lengths at the end) -- cheap enough for every start of the front-end, and what a replaced index changes even when its mtime is kept
const size_t SRC_HASH_SPAN = 64u << 10;
uint64_t fnv1a(uint64_t h, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; } return h; }
uint64_t source_hash(const uint8_t* data, uint64_t size) {
  uint64_t h = fnv1a(0xcbf29ce484222325ULL, &size, sizeof size);
  const size_t head = (size_t)std::min<uint64_t>(size, SRC_HASH_SPAN);
  h = fnv1a(h, data, head);
  if (size > head) { const size_t tail = (size_t)std::min<uint64_t>(size - head, SRC_HASH_SPAN); h = fnv1a(h, data + size - tail, tail); }
  return h ? h : 1;
}
bool source_identity_of_file(const char* path, uint64_t* size, uint64_t* hash) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  bool ok = fseek(f, 0, SEEK_END) == 0;
  const long sz = ok ? ftell(f) : -1;
  ok = ok && sz >= 0;
  std::vector<uint8_t> b;
  if (ok) {
    const uint64_t n = (uint64_t)sz;
    const size_t head = (size_t)std::min<uint64_t>(n, SRC_HASH_SPAN), tail = n > head ? (size_t)std::min<uint64_t>(n - head, SRC_HASH_SPAN) : 0;
    b.resize(head + tail);
    ok = fseek(f, 0, SEEK_SET) == 0 && fread(b.data(), 1, head, f) == head && (tail == 0 || (fseek(f, (long)(n - tail), SEEK_SET) == 0 && fread(b.data() + head, 1, tail, f) == tail));
    if (ok) {   // the same bytes source_hash() sees when it is given the whole file
      uint64_t h = fnv1a(0xcbf29ce484222325ULL, &n, sizeof n);
      h = fnv1a(h, b.data(), head + tail);
      *size = n; *hash = h ? h : 1;
    }
  }
  fclose(f);
  return ok;
}
struct FlatOut {
  FILE* f; bool ok = true;
  void raw(const void* p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
  template <class T> void scalar(const T& x) { raw(&x, sizeof x); }
  template <class V> void vec(const V& v) { const uint64_t n = v.size(); scalar(n); raw(v.data(), n * sizeof(typename V::value_type)); }
};
struct FlatIn {
  FILE* f; bool ok = true;
  struct Job { char* dst; uint64_t off, bytes; };
  std::vector<Job> jobs;   // arrays of 16 MB and more: skipped in the stream, read afterwards by several threads (run_jobs)
  void raw(void* p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
  template <class T> void scalar(T& x) { raw(&x, sizeof x); }
  template <class V> void vec(V& v) {
    typedef typename V::value_type T;
    uint64_t n = 0; scalar(n);
    if (!ok || n > (1ULL << 40) / sizeof(T)) { ok = false; return; }
    v.resize(n);
    const uint64_t bytes = n * sizeof(T);
    if (bytes >= (16u << 20)) {
      const long at = ftell(f);
      if (at < 0 || fseek(f, (long)bytes, SEEK_CUR) != 0) { ok = false; return; }
      jobs.push_back(Job{(char*)v.data(), (uint64_t)at, bytes});
    } else raw(v.data(), bytes);
  }
  void run_jobs(const char* path, int threads) {
    if (!ok || jobs.empty()) return;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { ok = false; return; }
    struct Piece { char* dst; uint64_t off, bytes; };
    std::vector<Piece> pieces;
    for (const Job& j : jobs) for (uint64_t a = 0; a < j.bytes; a += (8u << 20)) pieces.push_back(Piece{j.dst + a, j.off + a, std::min<uint64_t>(8u << 20, j.bytes - a)});
    std::atomic<size_t> next{0}; std::atomic<bool> good{true};
    std::vector<std::thread> th;
    for (int t = 0; t < std::max(1, std::min(threads, 32)); t++) th.emplace_back([&] {
      for (size_t i; (i = next++) < pieces.size();) {
        uint64_t done = 0;
        while (done < pieces[i].bytes) {
          const ssize_t r = pread(fd, pieces[i].dst + done, pieces[i].bytes - done, (off_t)(pieces[i].off + done));
          if (r <= 0) { good = false; return; }
          done += (uint64_t)r;
        }
      }
    });
    for (auto& x : th) x.join();
    ::close(fd);
    if (!good) ok = false;
  }
};
template <class IO> void flat_fields(IO& io, kamd_index& x) {
  io.scalar(x.k); io.scalar(x.n_kmers); io.scalar(x.n_unitigs); io.scalar(x.n_long); io.scalar(x.n_short); io.scalar(x.n_abund); io.scalar(x.dlist_size);
  io.scalar(x.n_targets); io.scalar(x.n_buckets); io.scalar(x.pad_buckets); io.scalar(x.text_bases);
  io.scalar(x.n_dbuckets); io.scalar(x.dpad_buckets); io.scalar(x.dummy_slot); io.scalar(x.dummy_uec); io.scalar(x.dummy_strand);
  io.scalar(x.layout); io.scalar(x.slots); io.scalar(x.tag_q); io.scalar(x.tag_dsh); io.scalar(x.tag_w);
  io.vec(x.unitig_len); io.vec(x.unitig_blk_off); io.vec(x.blk_unitig); io.vec(x.blk_lb); io.vec(x.blk_ub); io.vec(x.blk_ec); io.vec(x.blk_uec);
  io.vec(x.blk_pos_off); io.vec(x.blk_posw); io.vec(x.blk_sense); io.vec(x.uec_ec); io.vec(x.ec_off); io.vec(x.ec_ids); io.vec(x.target_lens);
  io.vec(x.onlist_bits); io.vec(x.table); io.vec(x.slot_block); io.vec(x.slot_dist); io.vec(x.utext); io.vec(x.unitig_gpos); io.vec(x.dlist_keys); io.vec(x.dtable);
}
uint32_t bits_of(uint64_t n) { uint32_t b = 0; while (b < 64 && (n >> b)) ++b; return b; }   // bits that hold the values 0..n
// the shifts of the compact layout for a table of nb home buckets; false: a field does not fit (kamd_core.h)
bool compact_shifts(int k, uint64_t nb, uint64_t n_uec, uint64_t text_bases, uint32_t* q, uint32_t* dsh, uint32_t* w) {
  *q = kamd::compact_q_of(nb);
  *dsh = *q + (uint32_t)std::max(0, 2 * k - 32);
  *w = *dsh + (*dsh + 4 + bits_of(n_uec) <= 64 ? 4 : 3);   // the displacement: four bits when the class ids leave room for them
  return *w + bits_of(n_uec) <= 64 && text_bases <= kamd::COMPACT_GPOS_MASK;
}
bool layout_is_consistent(const kamd_index& x) {
  if (x.layout == kamd::LAYOUT_WIDE) return x.slots == (uint32_t)kamd::BUCKET_SLOTS;
  if (x.layout != kamd::LAYOUT_COMPACT || x.slots != (uint32_t)kamd::COMPACT_SLOTS || x.n_buckets < 16) return false;
  uint32_t q, dsh, w;
  return compact_shifts(x.k, x.n_buckets, x.uec_ec.size(), x.text_bases, &q, &dsh, &w) && q == x.tag_q && dsh == x.tag_dsh && w == x.tag_w;
}
int load_flat(const char* path, int threads, kamd_index** out) {
  FILE* f = fopen(path, "rb");
  if (!f) return kamd::fail(-2, std::string("index input file could not be opened: ") + path);
  char magic[8];
  FlatIn in{f};
  in.raw(magic, 8);
  uint32_t stamp[4] = {0, 0, 0, 0};
  in.raw(stamp, sizeof stamp);
  if (!in.ok || memcmp(magic, FLAT_MAGIC, 8) != 0 || memcmp(stamp, FLAT_STAMP, sizeof stamp) != 0) {
    fclose(f);
    return kamd::fail(-3, std::string("flattened index file is damaged or of another format version: ") + path);
  }
  std::unique_ptr<kamd_index> ix(new kamd_index);
  in.scalar(ix->src_size); in.scalar(ix->src_hash);
  flat_fields(in, *ix);
  uint64_t n_names = 0; in.scalar(n_names);
  if (in.ok && n_names < (1ULL << 32)) {
    ix->target_names.resize(n_names);
    for (uint64_t i = 0; i < n_names && in.ok; i++) { uint32_t l = 0; in.scalar(l); if (l > (1u << 20)) { in.ok = false; break; } ix->target_names[i].resize(l); in.raw(&ix->target_names[i][0], l); }
  } else in.ok = false;
  fclose(f);
  in.run_jobs(path, threads);
  // the arrays must be consistent with the scalars -- and with each other -- the kernels trust: every length, and the largest value
  // of every array that is used as an index (a truncated or stale file must not turn into out-of-bounds reads on the device)
  const uint64_t S = ix->slots, nb = ix->n_buckets + ix->pad_buckets;
  bool good = in.ok && ix->k >= 3 && ix->k <= 31 && layout_is_consistent(*ix) && ix->table.size() == nb * 8 && ix->slot_block.size() == nb * S && ix->slot_dist.size() == nb * S &&
              ix->unitig_len.size() == ix->n_unitigs && ix->unitig_blk_off.size() == ix->n_unitigs + 1 && ix->unitig_gpos.size() == ix->n_unitigs + 1 &&
              !ix->ec_off.empty() && ix->ec_off.back() == ix->ec_ids.size() && ix->target_lens.size() >= ix->n_targets + ix->dlist_size &&
              ix->target_names.size() >= ix->n_targets && ix->utext.size() >= (ix->text_bases + 15) / 16 + 2 &&
              ix->dtable.size() == (ix->n_dbuckets ? (ix->n_dbuckets + ix->dpad_buckets) * 8 : 0) && ix->onlist_bits.size() >= (ix->n_targets + 31) / 32;
  if (good) {
    const uint64_t n_blocks = ix->unitig_blk_off.back(), n_ecs = ix->ec_off.size() - 1, n_uec = ix->uec_ec.size(), n_tr = ix->n_targets + ix->dlist_size;
    good = ix->blk_unitig.size() == n_blocks && ix->blk_lb.size() == n_blocks && ix->blk_ub.size() == n_blocks && ix->blk_ec.size() == n_blocks &&
           ix->blk_uec.size() == n_blocks && ix->blk_pos_off.size() == n_blocks + 1 && ix->blk_posw.size() == ix->blk_pos_off.back() &&
           ix->blk_sense.size() == ix->blk_pos_off.back() && n_uec <= kamd::UEC_MASK && n_ecs <= kamd::EC_ID_MASK && ix->dummy_uec < std::max<uint64_t>(n_uec, 1) &&
           ix->dummy_slot < std::max<uint64_t>(nb * S, 1) && ix->unitig_gpos.back() <= ix->text_bases &&
           (ix->dlist_size == 0 || ix->dlist_keys.size() == ix->dlist_size);
    for (size_t i = 0; good && i + 1 < ix->ec_off.size(); i++) good = ix->ec_off[i] <= ix->ec_off[i + 1];
    for (size_t i = 0; good && i + 1 < ix->unitig_blk_off.size(); i++) good = ix->unitig_blk_off[i] <= ix->unitig_blk_off[i + 1];
    for (size_t i = 0; good && i + 1 < ix->blk_pos_off.size(); i++) good = ix->blk_pos_off[i] <= ix->blk_pos_off[i + 1];
    for (size_t i = 0; good && i < ix->ec_ids.size(); i++) good = ix->ec_ids[i] < n_tr;
    for (size_t i = 0; good && i < n_uec; i++) good = ix->uec_ec[i] < n_ecs;
    for (size_t i = 0; good && i < n_blocks; i++) good = ix->blk_ec[i] < n_ecs && ix->blk_uec[i] < n_uec && ix->blk_unitig[i] < ix->n_unitigs && ix->blk_lb[i] <= ix->blk_ub[i];
    if (good) {   // the per-slot block ids (0xFFFFFFFF = empty slot), in parallel: 4 bytes per slot of the big table
      std::atomic<bool> ok2{true};
      std::vector<std::thread> th;
      const int nt = std::max(1, std::min(threads, 16));
      for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
        const size_t a = ix->slot_block.size() * t / nt, e = ix->slot_block.size() * (t + 1) / nt;
        bool g = true;
        for (size_t i = a; i < e; i++) g = g && (ix->slot_block[i] == 0xFFFFFFFFu || ix->slot_block[i] < n_blocks);
        if (!g) ok2 = false;
      });
      for (auto& x : th) x.join();
      good = ok2;
    }
  }
  if (!good) return kamd::fail(-3, std::string("flattened index file is damaged or of another format version: ") + path);
  *out = ix.release();
  return 0;
}
}  // namespace

extern "C" int kamd_index_save(const kamd_index* ix, const char* path) {
  if (!ix || !path) return kamd::fail(-1, "kamd_index_save: null argument");
  FILE* f = fopen(path, "wb");
  if (!f) return kamd::fail(-2, std::string("kamd_index_save: could not open ") + path);
  FlatOut o{f};
  o.raw(FLAT_MAGIC, 8);
  o.raw(FLAT_STAMP, sizeof FLAT_STAMP);
  o.scalar(ix->src_size); o.scalar(ix->src_hash);
  flat_fields(o, const_cast<kamd_index&>(*ix));
  const uint64_t n_names = ix->target_names.size(); o.scalar(n_names);
  for (const std::string& nm : ix->target_names) { const uint32_t l = (uint32_t)nm.size(); o.scalar(l); o.raw(nm.data(), l); }
  const bool ok = o.ok && fclose(f) == 0;
  if (!ok) return kamd::fail(-2, std::string("kamd_index_save: write failed: ") + path);
  return 0;
}

extern "C" int kamd_flat_index_matches(const char* flat_path, const char* index_path) {
  if (!flat_path || !index_path) return kamd::fail(-1, "kamd_flat_index_matches: null argument");
  FILE* f = fopen(flat_path, "rb");
  if (!f) return 0;
  char magic[8]; uint32_t stamp[4]; uint64_t id[2] = {0, 0};
  const bool head_ok = fread(magic, 1, 8, f) == 8 && fread(stamp, 1, sizeof stamp, f) == sizeof stamp && fread(id, 1, sizeof id, f) == sizeof id;
  fclose(f);
  if (!head_ok || memcmp(magic, FLAT_MAGIC, 8) != 0 || memcmp(stamp, FLAT_STAMP, sizeof stamp) != 0 || id[1] == 0) return 0;
  uint64_t size = 0, hash = 0;
  if (!source_identity_of_file(index_path, &size, &hash)) return 0;
  return size == id[0] && hash == id[1] ? 1 : 0;
}

namespace {
int load_index_impl(const char* path, int threads, int want_compact, double compact_load, kamd_index** out, bool layout_requested);
}
extern "C" int kamd_index_load(const char* path, int threads, kamd_index** out) {
  // the layout of the k-mer table from the environment: KAMD_TABLE_LAYOUT = wide (default) | compact | auto, KAMD_TABLE_LOAD
  // (default since round 4: auto.  Measured on MI355X, profiles/r04_gencode_size_table_layouts.json: at GENCODE size -- 130.6 M k-mers --
  // kernel A takes 13.9 ms on the wide table (5.6 GB) and 11.7 ms on the compact one at a load of 0.6 (3.5 GB); at config #3's 56.8 M
  // k-mers 10.43 wide, 10.29 compact at 0.5, 10.55 at 0.6)
  int want_compact = KAMD_TABLE_AUTO;
  if (const char* e = getenv("KAMD_TABLE_LAYOUT")) {
    if (!strcmp(e, "compact")) want_compact = KAMD_TABLE_COMPACT;
    else if (!strcmp(e, "auto")) want_compact = KAMD_TABLE_AUTO;
    else if (!strcmp(e, "wide")) want_compact = KAMD_TABLE_WIDE;
    else if (*e) return kamd::fail(-1, std::string("KAMD_TABLE_LAYOUT: wide, compact or auto expected, not ") + e);
  }
  double compact_load = 0.0;
  if (const char* e = getenv("KAMD_TABLE_LOAD")) compact_load = atof(e);
  return load_index_impl(path, threads, want_compact, compact_load, out, getenv("KAMD_TABLE_LAYOUT") != nullptr && *getenv("KAMD_TABLE_LAYOUT"));
}
extern "C" int kamd_index_load_layout(const char* path, int threads, int layout, double load, kamd_index** out) {
  if (layout != KAMD_TABLE_WIDE && layout != KAMD_TABLE_COMPACT && layout != KAMD_TABLE_AUTO) return kamd::fail(-1, "kamd_index_load_layout: layout must be KAMD_TABLE_WIDE, _COMPACT or _AUTO");
  return load_index_impl(path, threads, layout, load, out, true);
}
namespace {
int load_index_impl(const char* path, int threads, int want_compact, double compact_load_arg, kamd_index** out, bool layout_requested) {
  if (!out) return kamd::fail(-1, "kamd_index_load: null output pointer");
  *out = nullptr;
  if (threads <= 0) {
    // the CPUs this process may keep busy, not the host's processors: a container routinely shows all of them while its cgroup grants a
    // fraction (cpu.max), and four times as many builder threads as CPUs made the load slower, not faster
    threads = (int)std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0}; long long period = 0;
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        const long long quota = atoll(q);
        if (quota > 0) threads = (int)std::max<long long>(1, std::min<long long>(threads, (quota + period / 2) / period));
      }
      fclose(f);
    }
  }
  threads = std::min(threads, 64);
  {   // a file written by kamd_index_save?
    FILE* f = path ? fopen(path, "rb") : nullptr;
    char magic[8] = {0};
    const bool flat = f && fread(magic, 1, 8, f) == 8 && memcmp(magic, FLAT_MAGIC, 7) == 0;   // (any version: load_flat refuses the others by name)
    if (f) fclose(f);
    if (flat) {
      if (int rc = load_flat(path, threads, out)) return rc;
      // a flattened file carries its layout: a caller that named one (kamd_index_load_layout, KAMD_TABLE_LAYOUT) and is handed the
      // other must hear about it -- the compact table is asked for to fit a footprint, the wide one to reproduce a measurement
      const uint32_t have = (*out)->layout;
      if (layout_requested && ((want_compact == KAMD_TABLE_COMPACT && have != kamd::LAYOUT_COMPACT) || (want_compact == KAMD_TABLE_WIDE && have != kamd::LAYOUT_WIDE))) {
        delete *out; *out = nullptr;
        return kamd::fail(-3, std::string("flattened index file holds the ") + (have == kamd::LAYOUT_COMPACT ? "compact" : "wide") + " k-mer table but the " +
                          (want_compact == KAMD_TABLE_COMPACT ? "compact" : "wide") + " one was asked for (flatten the kallisto index with that layout, or load the index itself): " + path);
      }
      return 0;
    }
  }
  // the file as a read-only mapping (a copy into a zero-filled vector was 0.1-0.15 s of the 0.8 s for a human-sized index); a file that cannot
  // be mapped (a pipe, a file system without mmap) is read
  struct FileBytes {
    const uint8_t* p = nullptr; size_t n = 0; bool mapped = false; std::vector<uint8_t> copy;
    ~FileBytes() { if (mapped && p) munmap(const_cast<uint8_t*>(p), n); }
  } buf;
  {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return kamd::fail(-2, std::string("index input file could not be opened: ") + path);
    const off_t sz = lseek(fd, 0, SEEK_END);
    void* m = sz > 0 ? mmap(nullptr, (size_t)sz, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0) : MAP_FAILED;
    if (m != MAP_FAILED) { buf.p = (const uint8_t*)m; buf.n = (size_t)sz; buf.mapped = true; (void)madvise(m, (size_t)sz, MADV_SEQUENTIAL); }
    close(fd);
    if (!buf.mapped) {
      std::ifstream in(path, std::ios::binary | std::ios::ate);
      if (!in) return kamd::fail(-2, std::string("index input file could not be opened: ") + path);
      std::streamsize rsz = in.tellg();
      in.seekg(0);
      buf.copy.resize((size_t)rsz);
      if (rsz > 0 && !in.read((char*)buf.copy.data(), rsz)) return kamd::fail(-2, "index: short read");
      buf.p = buf.copy.data(); buf.n = buf.copy.size();
    }
  }
  Cursor c{buf.p, buf.n};
  std::unique_ptr<kamd_index> ix(new kamd_index);
  ix->src_size = buf.n; ix->src_hash = source_hash(buf.p, buf.n);
  // KAMD_INDEX_TIMING=1: seconds per phase on stderr
  const bool timing = getenv("KAMD_INDEX_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto tick = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[index] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  tick("read file");

  // 1. version (KmerIndex.cpp:1351-1360)
  if (c.get<uint64_t>() != 13) return kamd::fail(-3, "incompatible index: expected version 13");
  // 2. graph section
  uint64_t dbg_bytes = c.get<uint64_t>() & (~0ULL >> 1);
  size_t pos1 = c.pos;
  if ((c.get<uint64_t>() >> 32) != 0x7e215f3fULL) return kamd::fail(-3, "index: bad graph header");
  int32_t k = c.get<int32_t>();
  (void)c.get<int32_t>();  // g
  if (k < 3 || k > 31 || !(k & 1)) return kamd::fail(-3, "index: unsupported k (need odd k <= 31)");
  ix->k = k;
  ix->n_long = c.get<uint64_t>();
  // (every count below comes from the file and sizes an allocation: bounded by what the rest of the file could hold, so that a damaged index
  // ends in an error code and not in std::bad_alloc / std::length_error crossing the C ABI)
  auto fits = [&](uint64_t count, uint64_t min_bytes_each) { return !c.bad && c.pos <= c.n && count <= (c.n - c.pos) / min_bytes_each; };
  if (!fits(ix->n_long, 8)) return kamd::fail(-3, "index: bad unitig count");
  struct U { const uint8_t* data; uint64_t len; };
  std::vector<U> units;
  units.reserve(ix->n_long);
  uint64_t n_kmers = 0;
  for (uint64_t i = 0; i < ix->n_long; i++) {
    uint64_t len = c.get<uint64_t>();
    if (c.bad || len < (uint64_t)k || len > 4 * (uint64_t)c.n) return kamd::fail(-3, "index: bad unitig record");
    const uint8_t* d = c.take((len + 3) / 4);
    if (c.bad) return kamd::fail(-3, "index: bad unitig record");
    units.push_back({d, len});
    n_kmers += len - k + 1;
  }
  ix->n_short = c.get<uint64_t>();
  if (!fits(ix->n_short, 8)) return kamd::fail(-3, "index: bad short/abundant unitig section");
  const uint8_t* short_p = c.take(ix->n_short * 8);
  ix->n_abund = c.get<uint64_t>();
  if (!fits(ix->n_abund, 8)) return kamd::fail(-3, "index: bad short/abundant unitig section");
  const uint8_t* abund_p = c.take(ix->n_abund * 8);
  if (c.bad) return kamd::fail(-3, "index: bad short/abundant unitig section");
  ix->n_unitigs = ix->n_long + ix->n_short + ix->n_abund;
  ix->n_kmers = n_kmers + ix->n_short + ix->n_abund;
  ix->unitig_len.resize(ix->n_unitigs);
  for (uint64_t i = 0; i < ix->n_long; i++) ix->unitig_len[i] = (uint32_t)units[i].len;
  for (uint64_t i = ix->n_long; i < ix->n_unitigs; i++) ix->unitig_len[i] = (uint32_t)k;
  // forward k-mer of a short/abundant unitig: stored left-aligned MSB-first (Kmer.cpp:92-107) -> right-align
  auto single_kmer = [&](uint64_t i) -> uint64_t {
    uint64_t raw;
    if (i < ix->n_long + ix->n_short) memcpy(&raw, short_p + 8 * (i - ix->n_long), 8);
    else memcpy(&raw, abund_p + 8 * (i - ix->n_long - ix->n_short), 8);
    return raw >> (64 - 2 * k);
  };
  // head k-mer (forward text) -> unitig id, to attach node records (KmerIndex.cpp:1420-1428 uses dbg.find(head))
  // (a flat open-addressing table at a load of at most a half: a std::unordered_map of a million nodes cost 0.1 s to fill and as much again to free)
  struct HeadMap {
    std::vector<uint64_t> key; std::vector<uint32_t> val; uint64_t mask = 0;
    enum : uint32_t { NONE = 0xFFFFFFFFu };
    void init(uint64_t n) { uint64_t cap = 16; while (cap < 2 * n) cap <<= 1; key.assign(cap, 0); val.assign(cap, NONE); mask = cap - 1; }
    void emplace(uint64_t k, uint32_t v) {   // (the first value of a key stays, as with std::unordered_map::emplace)
      uint64_t s = kamd::mix64(k) & mask;
      while (val[s] != NONE) { if (key[s] == k) return; s = (s + 1) & mask; }
      key[s] = k; val[s] = v;
    }
    const uint32_t* find(uint64_t k) const {
      uint64_t s = kamd::mix64(k) & mask;
      while (val[s] != NONE) { if (key[s] == k) return &val[s]; s = (s + 1) & mask; }
      return nullptr;
    }
  } head_of;
  head_of.init(ix->n_unitigs * 2);
  for (uint64_t i = 0; i < ix->n_unitigs; i++) {
    uint64_t v = 0;
    if (i < ix->n_long) {
      for (int j = 0; j < k; j++) v = (v << 2) | ((units[i].data[j >> 2] >> ((j & 3) << 1)) & 3);
    } else v = single_kmer(i);
    head_of.emplace(v, (uint32_t)i);
    head_of.emplace(kamd::revcomp_msb(v, k), (uint32_t)i);
  }
  tick("unitigs + head map");
  // ---- k-mer table: two passes over all k-mers (count per home bucket, then place), no transient copy ----
  // The layout (kamd_core.h): wide = 3 slots of 20 bytes per line at a load of 0.5; compact = 4 slots of 16 bytes at a load of 0.6
  // (or the caller's): KAMD_TABLE_COMPACT is an error when a field does not fit, KAMD_TABLE_AUTO builds the wide one then.
  // load factor when the caller names none: the sparsest of 0.4 / 0.5 whose table stays under the 2.4 GB up to which dependent random
  // reads run at full rate on MI355X (fewer continue flags: 9.57 / 9.90 / 10.53 bucket lines per pair of config #3 at 0.4 / 0.5 / 0.6,
  // kernel A 10.13 / 10.26 / 10.55 ms; profiles/README.md), 0.6 beyond (fewer bytes beat fewer lines per probe there)
  // The footprint is a property of the device (MI355X: 2.4 GB, measured with kamd_debug_random_lines_span -- 54 G lines/s up to there, 28 G/s at
  // 3.6 GB; bench.py reports the live figure as random_line_ceiling), not of the library: KAMD_TABLE_KNEE_GB overrides it for another part.
  double knee = 2.4e9;
  if (const char* e = getenv("KAMD_TABLE_KNEE_GB")) { const double v = atof(e); if (v > 0.05 && v < 1024.0) knee = v * 1e9; }
  const double bytes_at_1 = (double)ix->n_kmers * (64.0 / kamd::COMPACT_SLOTS);
  const double default_load = bytes_at_1 / 0.4 <= knee ? 0.4 : bytes_at_1 / 0.5 <= knee ? 0.5 : 0.6;
  const double compact_load = (compact_load_arg >= 0.2 && compact_load_arg <= 0.9) ? compact_load_arg : default_load;
  bool compact = want_compact != KAMD_TABLE_WIDE;
  const uint64_t nb_wide = std::max<uint64_t>(16, (ix->n_kmers * 2 + kamd::BUCKET_SLOTS - 1) / kamd::BUCKET_SLOTS);  // load factor 0.5 over 3-slot buckets
  uint64_t S = compact ? kamd::COMPACT_SLOTS : kamd::BUCKET_SLOTS;
  uint64_t nb = compact ? std::max<uint64_t>(16, (uint64_t)((double)ix->n_kmers / compact_load / (double)S) + 1) : nb_wide;
  if (std::max(nb, nb_wide) >= 0xF0000000ULL) return kamd::fail(-3, "index: too many k-mers for 32-bit bucket numbers");
  // [0, n) in contiguous pieces, one per thread: the big arrays are first touched (and later scanned) by all threads
  auto parallel_range = [&](uint64_t n, auto&& body) {
    std::vector<std::thread> th;
    const uint64_t per = (n + (uint64_t)threads - 1) / (uint64_t)threads;
    for (int t = 0; t < threads; t++) {
      const uint64_t a = std::min(n, per * (uint64_t)t), b = std::min(n, a + per);
      if (a < b) th.emplace_back([&body, a, b] { body(a, b); });
    }
    for (auto& t : th) t.join();
  };
  BigVec<uint32_t> fill; fill.resize(nb + 1);
  auto fill_atomic = reinterpret_cast<std::atomic<uint32_t>*>(fill.data());
  // all unitigs, handed out in runs of 256; every thread has a state of its own (make()), flushed when it runs out of work (drain())
  auto run_parallel = [&](auto&& make, auto&& body, auto&& drain) {
    std::vector<std::thread> th;
    std::atomic<uint64_t> next{0};
    const uint64_t chunk = 256;
    for (int t = 0; t < threads; t++) th.emplace_back([&] {
      auto st = make();
      for (;;) {
        uint64_t s = next.fetch_add(chunk);
        if (s >= ix->n_unitigs) break;
        uint64_t e = std::min(ix->n_unitigs, s + chunk);
        for (uint64_t u = s; u < e; u++) body(u, st);
      }
      drain(st);
    });
    for (auto& t : th) t.join();
  };
  // Both passes touch a random cache line per k-mer (the bucket's counter; then the bucket's line of the 2.4 GB table and its two aux
  // words): the k-mers go through a small ring, a line is prefetched when a k-mer enters it and used when it leaves, RING k-mers later
  constexpr int RING = 16;
  struct CountState { uint64_t hb[RING]; int n = 0, head = 0; };
  auto count_one = [&](CountState& st, uint64_t hb) {
    __builtin_prefetch(&fill[hb], 1, 0);
    if (st.n == RING) { fill_atomic[st.hb[st.head]].fetch_add(1, std::memory_order_relaxed); st.hb[st.head] = hb; st.head = (st.head + 1) % RING; }
    else st.hb[(st.head + st.n++) % RING] = hb;
  };
  // The count pass only needs the unitigs: it runs on the worker threads from here on, underneath the (serial) parsing of the node
  // records, and is joined in front of the placement.
  auto count_pass = [&] {   // (for the table size `nb` of the moment)
    parallel_range(nb + 1, [&](uint64_t a, uint64_t b) { memset(fill.data() + a, 0, (b - a) * sizeof(uint32_t)); });
    run_parallel([] { return CountState(); },
                 [&](uint64_t u, CountState& st) {
                   auto count = [&](uint32_t, uint64_t v, uint64_t rc) { count_one(st, kamd::home_bucket(v < rc ? v : rc, nb)); };
                   if (u < ix->n_long) for_each_kmer_rc(units[u].data, units[u].len, k, count);
                   else { const uint64_t v = single_kmer(u); count(0, v, kamd::revcomp_msb(v, k)); }
                 },
                 [&](CountState& st) { for (int i = 0; i < st.n; i++) fill_atomic[st.hb[(st.head + i) % RING]].fetch_add(1, std::memory_order_relaxed); });
  };
  std::thread count_bg(count_pass);
  struct BgJoin { std::thread& t; ~BgJoin() { if (t.joinable()) t.join(); } } count_join{count_bg};
  // skip minimizer index + BooPHF (KmerIndex.cpp:1368-1376)
  c.pos = pos1 + dbg_bytes;
  { uint64_t mphf = c.get<uint64_t>(); c.take(mphf); }
  if (c.bad) return kamd::fail(-3, "index: bad mphf section");
  // 2.2 D-list (KmerIndex.cpp:1386-1403)
  ix->dlist_size = c.get<uint64_t>();
  (void)c.get<uint64_t>();  // overhang: only used when the index is built
  if (!fits(ix->dlist_size, 8)) return kamd::fail(-3, "index: truncated D-list");
  {
    const uint8_t* dl = c.take(ix->dlist_size * 8);
    if (ix->dlist_size && !dl) return kamd::fail(-3, "index: truncated D-list");
    ix->dlist_keys.resize(ix->dlist_size);
    for (uint64_t i = 0; i < ix->dlist_size; i++) {
      uint64_t raw; memcpy(&raw, dl + 8 * i, 8);          // a Kmer object: base j at bits 62 - 2j (Kmer.cpp:92-114)
      const uint64_t v = raw >> (64 - 2 * k), rc = kamd::revcomp_msb(v, k);
      ix->dlist_keys[i] = v < rc ? v : rc;                 // stored as rep() already; canonicalised again for safety
    }
  }

  // 3. nodes
  uint64_t n_nodes = c.get<uint64_t>();
  std::vector<std::vector<RawBlock>> ublocks(ix->n_unitigs);
  // transcript set -> id in order of first appearance: an open-addressing table of ids whose keys are the sets already stored in
  // ec_ids / ec_off (no copy of a set per entry, no allocation per insertion)
  std::vector<uint32_t> ec_tab(1u << 16, 0xFFFFFFFFu);
  uint64_t ec_tab_used = 0;
  auto ec_hash = [](const std::vector<uint32_t>& v) { uint64_t h = 0x9e3779b97f4a7c15ULL ^ v.size(); for (uint32_t x : v) h = kamd::mix64(h ^ x); return h; };
  std::vector<uint32_t> posw_all; std::vector<uint8_t> sense_all;
  ix->ec_off.push_back(0);
  // The node table in three steps.  (a) One serial walk finds where every node's record starts and ends (a head k-mer, a size, the body).
  // (b) All threads decode the bodies -- the roaring sets are most of the work -- into arenas of their own: per block its bounds, its
  // transcript set with the set's hash, and the position word / sense of every (block, transcript) pair.  (c) One serial walk in node
  // order interns the sets by their pre-computed hashes (ids stay in order of first appearance, as a serial parse gives them) and strings
  // the position arrays together.  The serial parse took 0.45 of the load's 0.87 s on the pool's boxes; parsing in parallel with the
  // interning left serial inside the loop had been tried in round 3 and was slower.
  struct NodeExt { size_t head, body, end; };
  // (the count comes from the file: every node record holds at least a head k-mer, its size word, an id and a flag -- a truncated or corrupt
  // index must fail here with -3, not in a multi-gigabyte allocation)
  if (c.bad || n_nodes > (c.n - std::min(c.pos, c.n)) / ((uint64_t)k + 4 + 5)) return kamd::fail(-3, "index: truncated node table");
  std::vector<NodeExt> ext(n_nodes);
  for (uint64_t i = 0; i < n_nodes; i++) {
    const uint8_t* hs = c.take((size_t)k);
    if (!hs) return kamd::fail(-3, "index: truncated node table");
    const uint32_t node_size = c.get<uint32_t>();
    if (c.bad || c.pos + node_size > c.n) return kamd::fail(-3, "index: truncated node table");
    ext[i] = NodeExt{(size_t)(hs - c.p), c.pos, c.pos + node_size};
    c.pos += node_size;
  }
  struct BlockTmp { uint32_t lb, ub; uint64_t set_off; uint32_t set_len; uint64_t hash; uint64_t pos_off; };   // offsets into the arena
  struct NodeTmp { uint32_t gid; uint32_t n_blocks; uint64_t first_block; };
  struct Arena { std::vector<BlockTmp> blocks; std::vector<uint32_t> sets, posw; std::vector<uint8_t> sense; std::string err; };
  // (KAMD_INDEX_DECODE_THREADS: experiments -- the count pass of the k-mer table runs on `threads` workers in the background meanwhile)
  int nth = std::max(1, std::min(threads, 32));
  if (const char* e = getenv("KAMD_INDEX_DECODE_THREADS")) nth = std::max(1, std::min(atoi(e), 64));
  std::vector<Arena> arenas((size_t)nth);
  std::vector<NodeTmp> nodes(n_nodes);
  {
    std::vector<std::thread> th;
    const uint64_t per = (n_nodes + (uint64_t)nth - 1) / (uint64_t)nth;
    for (int t = 0; t < nth; t++) th.emplace_back([&, t] {
      Arena& A = arenas[(size_t)t];
      std::vector<uint32_t> set, tmp;
      const uint64_t lo = std::min<uint64_t>(n_nodes, per * (uint64_t)t), hi = std::min<uint64_t>(n_nodes, lo + per);
      for (uint64_t i = lo; i < hi && A.err.empty(); i++) {
        const uint8_t* hs = c.p + ext[i].head;
        uint64_t head = 0;
        for (int j = 0; j < k; j++) { uint64_t ch = hs[j]; uint64_t x = (ch & 4) >> 1; head = (head << 2) | (x + ((x ^ (ch & 2)) >> 1)); }
        const uint32_t* it = head_of.find(head);
        if (!it) { A.err = "Corrupted index; unitig not found"; break; }
        Cursor nc{c.p, ext[i].end, ext[i].body};
        (void)nc.get<uint32_t>();  // Node::id: only a sort key in the reference
        const uint8_t flag = nc.get<uint8_t>();
        const uint64_t nb = flag == 0 ? 0 : (flag == 1 ? 1 : nc.get<uint64_t>());
        nodes[i] = NodeTmp{*it, 0, A.blocks.size()};
        for (uint64_t b = 0; b < nb; b++) {
          BlockTmp bt; bt.lb = nc.get<uint32_t>(); bt.ub = nc.get<uint32_t>(); bt.pos_off = A.posw.size();
          const uint64_t sz = nc.get<uint64_t>();
          const uint8_t* rp = nc.take(sz);
          if (!rp || !roaring_native(rp, sz, set)) { A.err = "index: bad transcript set"; break; }
          const uint64_t vsz = nc.get<uint64_t>();
          if (vsz != set.size()) { A.err = "index: SparseVector size mismatch"; break; }
          for (uint64_t q = 0; q < vsz; q++) {
            const uint64_t s2 = nc.get<uint64_t>();
            const uint8_t* pp = nc.take(s2);
            uint32_t mn = 0, mx = 0;
            if (!pp || !roaring_minmax(pp, s2, tmp, &mn, &mx)) { A.err = "index: bad position set"; break; }
            A.posw.push_back(mn);
            const bool smin = (mn & 0x7FFFFFFFu) == mn, smax = (mx & 0x7FFFFFFFu) == mx;
            A.sense.push_back(smin != smax ? 2 : (uint8_t)smin);
          }
          if (!A.err.empty()) break;
          bt.set_off = A.sets.size(); bt.set_len = (uint32_t)set.size(); bt.hash = ec_hash(set);
          A.sets.insert(A.sets.end(), set.begin(), set.end());
          A.blocks.push_back(bt);
          ++nodes[i].n_blocks;
        }
        if (A.err.empty() && (nc.bad || nc.pos != ext[i].end)) A.err = "index: node size mismatch";
      }
    });
    for (auto& x : th) x.join();
    for (const Arena& A : arenas) if (!A.err.empty()) return kamd::fail(-3, A.err);
  }
  {
    // (c) interning and stringing together, in node order.  The table holds ids; a probe compares the candidate set with the stored one.
    auto intern_hashed = [&](const uint32_t* v, uint32_t n, uint64_t h) -> uint32_t {
      if ((ec_tab_used + 1) * 2 > ec_tab.size()) {   // grow: re-insert the ids by the hash of their stored sets
        std::vector<uint32_t> nt(ec_tab.size() * 4, 0xFFFFFFFFu);
        std::vector<uint32_t> t;
        for (uint32_t id : ec_tab) if (id != 0xFFFFFFFFu) {
          t.assign(ix->ec_ids.begin() + ix->ec_off[id], ix->ec_ids.begin() + ix->ec_off[id + 1]);
          uint64_t p = ec_hash(t) & (nt.size() - 1);
          while (nt[p] != 0xFFFFFFFFu) p = (p + 1) & (nt.size() - 1);
          nt[p] = id;
        }
        ec_tab.swap(nt);
      }
      uint64_t p = h & (ec_tab.size() - 1);
      for (;; p = (p + 1) & (ec_tab.size() - 1)) {
        const uint32_t id = ec_tab[p];
        if (id == 0xFFFFFFFFu) break;
        const uint64_t a0 = ix->ec_off[id], len = ix->ec_off[id + 1] - a0;
        if (len == n && std::equal(v, v + n, ix->ec_ids.begin() + a0)) return id;
      }
      const uint32_t id = (uint32_t)(ix->ec_off.size() - 1);
      ix->ec_ids.insert(ix->ec_ids.end(), v, v + n); ix->ec_off.push_back(ix->ec_ids.size());
      ec_tab[p] = id; ++ec_tab_used;
      return id;
    };
    const uint64_t per = (n_nodes + (uint64_t)nth - 1) / (uint64_t)nth;
    for (uint64_t i = 0; i < n_nodes; i++) {
      const Arena& A = arenas[(size_t)std::min<uint64_t>((uint64_t)nth - 1, per ? i / per : 0)];
      const NodeTmp& nd = nodes[i];
      auto& bl = ublocks[nd.gid];
      bl.clear();
      for (uint32_t b = 0; b < nd.n_blocks; b++) {
        const BlockTmp& bt = A.blocks[nd.first_block + b];
        RawBlock rb; rb.lb = bt.lb; rb.ub = bt.ub; rb.pos_off = posw_all.size();
        posw_all.insert(posw_all.end(), A.posw.begin() + bt.pos_off, A.posw.begin() + bt.pos_off + bt.set_len);
        sense_all.insert(sense_all.end(), A.sense.begin() + bt.pos_off, A.sense.begin() + bt.pos_off + bt.set_len);
        rb.ec = intern_hashed(A.sets.data() + bt.set_off, bt.set_len, bt.hash);
        bl.push_back(rb);
      }
      std::stable_sort(bl.begin(), bl.end(), [](const RawBlock& a, const RawBlock& b) { return a.lb < b.lb; });
    }
  }
  // (the decode arenas hold every block's set, positions and senses once more: freed here, not at the end of the load -- ADVICE r4)
  { std::vector<Arena>().swap(arenas); std::vector<NodeTmp>().swap(nodes); std::vector<NodeExt>().swap(ext); }
  tick("node records");
  // flatten blocks; assign (unitig, set) classes
  ix->unitig_blk_off.assign(ix->n_unitigs + 1, 0);
  for (uint64_t u = 0; u < ix->n_unitigs; u++) {
    ix->unitig_blk_off[u] = ix->blk_lb.size();
    auto& bl = ublocks[u];
    if (bl.empty()) return kamd::fail(-3, "index: unitig without a node record");
    size_t first = ix->blk_lb.size();
    for (auto& rb : bl) {
      uint32_t uec = 0xFFFFFFFFu;
      for (size_t j = first; j < ix->blk_lb.size(); j++) if (ix->blk_ec[j] == rb.ec) { uec = ix->blk_uec[j]; break; }
      if (uec == 0xFFFFFFFFu) { uec = (uint32_t)ix->uec_ec.size(); ix->uec_ec.push_back(rb.ec); }
      ix->blk_unitig.push_back((uint32_t)u); ix->blk_lb.push_back(rb.lb); ix->blk_ub.push_back(rb.ub);
      ix->blk_ec.push_back(rb.ec); ix->blk_uec.push_back(uec);
      ix->blk_pos_off.push_back(ix->blk_posw.size());
      uint64_t n = ix->ec_off[rb.ec + 1] - ix->ec_off[rb.ec];
      ix->blk_posw.insert(ix->blk_posw.end(), posw_all.begin() + rb.pos_off, posw_all.begin() + rb.pos_off + n);
      ix->blk_sense.insert(ix->blk_sense.end(), sense_all.begin() + rb.pos_off, sense_all.begin() + rb.pos_off + n);
    }
    std::vector<RawBlock>().swap(bl);
  }
  ix->unitig_blk_off[ix->n_unitigs] = ix->blk_lb.size();
  ix->blk_pos_off.push_back(ix->blk_posw.size());
  if (ix->uec_ec.size() >= kamd::UEC_MASK) return kamd::fail(-3, "index: too many (unitig, set) classes");   // class lists keep 30 bits
  if (ix->ec_off.size() - 1 >= kamd::EC_ID_MASK) return kamd::fail(-3, "index: too many transcript sets");        // set lists keep two mate flags

  tick("flatten blocks");
  // 4-6. targets (KmerIndex.cpp:1470-1519)
  int32_t nt = c.get<int32_t>();
  if (c.bad || nt < 0) return kamd::fail(-3, "index: bad target count");
  // the count includes one pseudo-target per D-list k-mer ("d_list.N", KmerIndex.cpp:940-960); lengths and names are only
  // stored for the real ones (num_trans -= d_list.size(), :1297).  The pseudo-targets keep ids [n_real, nt) in the
  // transcript sets (off-list); their length entries are 0 so that every id can index target_lens.
  const int64_t n_real = (int64_t)nt - (int64_t)ix->dlist_size;
  if (n_real < 0 || !fits((uint64_t)n_real, 4 + 8)) return kamd::fail(-3, "index: bad target count");   // (a length and a name record per real target)
  ix->n_targets = (uint64_t)n_real;
  ix->target_lens.assign((size_t)nt, 0);
  for (int64_t i = 0; i < n_real; i++) ix->target_lens[i] = c.get<int32_t>();
  for (int64_t i = 0; i < n_real; i++) {
    uint64_t n = c.get<uint64_t>();
    const uint8_t* s = c.take(n);
    if (!s) return kamd::fail(-3, "index: bad target name");
    ix->target_names.emplace_back((const char*)s, n);
  }
  // 7. on-list (KmerIndex.cpp:1522-1526)
  {
    uint64_t n = c.get<uint64_t>();
    const uint8_t* s = c.take(n);
    std::vector<uint32_t> ol;
    if (!s || !roaring_portable(s, n, ol)) return kamd::fail(-3, "index: bad on-list");
    ix->onlist_bits.assign(((size_t)nt + 31) / 32 + 1, 0);
    for (uint32_t t : ol) if (t < (uint32_t)nt) ix->onlist_bits[t >> 5] |= 1u << (t & 31);
  }
  if (c.bad) return kamd::fail(-3, "index: truncated file");

  tick("targets + on-list");
  // ---- unitig text: every unitig's bases end to end, 16 per word (the reads' packing) ----
  {
    ix->unitig_gpos.assign(ix->n_unitigs + 1, 0);
    uint64_t g = 0;
    for (uint64_t u = 0; u < ix->n_unitigs; u++) { ix->unitig_gpos[u] = g; g += ix->unitig_len[u]; }
    ix->unitig_gpos[ix->n_unitigs] = g;
    ix->text_bases = g;
    if (g >= 0xFFFFFF00ULL) return kamd::fail(-3, "index: unitig text exceeds 2^32 bases");
    ix->utext.assign((g + 15) / 16 + 2, 0);
  }
  tick("unitig text: layout");
  // (the count pass of the k-mer table has been running since the unitigs were read)
  count_bg.join();
  tick("table: count pass (rest)");
  // placement: keys grouped by home bucket are laid down sequentially, never before their home (Robin Hood order)
  BigVec<uint64_t> base;
  uint64_t cursor = 0;
  uint32_t tag_q = 0, tag_dsh = 0, tag_w = 0;
  for (;;) {
    // the compact layout must hold the class ids and the text positions beside the tag, and a key no farther from its home than the
    // displacement field can say (14 or 6 buckets): otherwise the wide layout (auto), or a larger table (a sixteenth more buckets) and the
    // count pass again
    auto recount = [&] { fill.resize(nb + 1); fill_atomic = reinterpret_cast<std::atomic<uint32_t>*>(fill.data()); count_pass(); };
    if (compact && !compact_shifts(k, nb, ix->uec_ec.size(), ix->text_bases, &tag_q, &tag_dsh, &tag_w)) {
      if (want_compact == KAMD_TABLE_COMPACT) return kamd::fail(-3, "index: the compact k-mer table cannot hold this index (class ids / text positions too wide); use KAMD_TABLE_LAYOUT=wide or auto");
      compact = false; S = kamd::BUCKET_SLOTS; nb = nb_wide;
      recount();
    }
    base.resize(nb);
    cursor = 0;
    uint64_t max_disp = 0;
    for (uint64_t b = 0; b < nb; b++) {   // (a running maximum: serial, but only 12 bytes per bucket go through it)
      cursor = std::max(cursor, b * S);
      base[b] = cursor;
      cursor += fill[b];
      if (fill[b]) max_disp = std::max(max_disp, (cursor - 1) / S - b);
    }
    if (!compact || max_disp <= (1u << (tag_w - tag_dsh)) - 2u) break;
    nb += nb / 16 + 1;
    recount();
  }
  ix->n_buckets = nb;
  ix->layout = compact ? kamd::LAYOUT_COMPACT : kamd::LAYOUT_WIDE; ix->slots = (uint32_t)S;
  ix->tag_q = compact ? tag_q : 0; ix->tag_dsh = compact ? tag_dsh : 0; ix->tag_w = compact ? tag_w : 0;
  kamd::Table ktab{nullptr, nb};
  ktab.layout = (uint8_t)ix->layout; ktab.q = (uint8_t)ix->tag_q; ktab.dsh = (uint8_t)ix->tag_dsh; ktab.tagw = (uint8_t)ix->tag_w;
  const uint64_t end_cursor = cursor;
  uint64_t total_buckets = std::max(nb, (cursor + S - 1) / S) + 1;
  ix->pad_buckets = total_buckets - nb;
  // a bucket continues into the next one when keys homed at or before it spill past its slots: then all its slots are taken, and the
  // key that lands in slot 0 carries the flag (set by the placement below)
  auto bucket_continues = [&](uint64_t b) { const uint64_t reach = b + 1 < nb ? base[b + 1] : end_cursor; return reach > (b + 1) * S; };
  // the tables are allocated untouched and initialised by all threads (value-initialising 3.3 GB on one thread was a third of the load)
  ix->table.resize(total_buckets * 8);
  ix->slot_block.resize(total_buckets * S);
  ix->slot_dist.resize(total_buckets * S);
  parallel_range(total_buckets, [&](uint64_t a, uint64_t b) {
    for (uint64_t i = a; i < b; i++) {
      uint64_t* w = ix->table.data() + 8 * i;
      if (compact) { for (uint64_t j = 0; j < 4; j++) { w[2 * j] = ~0ULL; w[2 * j + 1] = 0; } }   // displacement all ones = empty
      else {
        for (uint64_t j = 0; j < S; j++) w[j] = kamd::KEY_EMPTY;
        for (uint64_t j = S; j < 8; j++) w[j] = 0;
      }
    }
    std::fill(ix->slot_block.begin() + a * S, ix->slot_block.begin() + b * S, 0xFFFFFFFFu);
    memset(ix->slot_dist.data() + a * S, 0, (b - a) * S * sizeof(uint32_t));
  });
  parallel_range(nb + 1, [&](uint64_t a, uint64_t b) { memset(fill.data() + a, 0, (b - a) * sizeof(uint32_t)); });
  tick("table: layout + allocation");
  auto utext_atomic = reinterpret_cast<std::atomic<uint32_t>*>(ix->utext.data());
  // stage 1: the k-mer's home bucket (counter + base prefetched); stage 2, RING k-mers later: its slot (table line and aux words
  // prefetched); stage 3, RING k-mers later again: the stores
  struct Pend1 { uint64_t cn, hb, payload; uint32_t gpos, block, dist; };
  struct Pend2 { uint64_t cn, slot, payload; uint32_t gpos, block, dist; };
  struct PlaceState { Pend1 a[RING]; Pend2 b[RING]; int na = 0, ha = 0, nb2 = 0, hb2 = 0; };
  auto stage3 = [&](const Pend2& q) {
    const uint64_t bk = q.slot / S, j = q.slot % S;
    if (compact) {
      const uint32_t h = kamd::kmer_hash32(q.cn);
      const uint64_t uec = (q.payload >> 32) & 0x7FFFFFFFULL;
      ix->table[8 * bk + 2 * j] = kamd::compact_tag(ktab, q.cn, h, (uint32_t)(bk - kamd::bucket_of_hash(h, nb))) | (uec << ix->tag_w);
      ix->table[8 * bk + 2 * j + 1] = (q.payload & 0xFFFFFFFFULL) | ((uint64_t)q.gpos << 32) | ((q.payload >> 63) ? kamd::COMPACT_FWD : 0ULL) |
                                      ((j == 0 && bucket_continues(bk)) ? kamd::COMPACT_CONT : 0ULL);
    } else {
      ix->table[8 * bk + j] = (j == 0 && bucket_continues(bk)) ? (q.cn | kamd::KEY_CONT) : q.cn;
      ix->table[8 * bk + S + j] = q.payload;
      reinterpret_cast<uint32_t*>(&ix->table[8 * bk + 2 * S])[j] = q.gpos;
    }
    ix->slot_block[q.slot] = q.block;
    ix->slot_dist[q.slot] = q.dist;
  };
  auto stage2 = [&](PlaceState& st, const Pend1& p) {
    const uint64_t slot = base[p.hb] + fill_atomic[p.hb].fetch_add(1, std::memory_order_relaxed);
    __builtin_prefetch(&ix->table[8 * (slot / S)], 1, 0);
    __builtin_prefetch(&ix->slot_block[slot], 1, 0);
    __builtin_prefetch(&ix->slot_dist[slot], 1, 0);
    const Pend2 q{p.cn, slot, p.payload, p.gpos, p.block, p.dist};
    if (st.nb2 == RING) { stage3(st.b[st.hb2]); st.b[st.hb2] = q; st.hb2 = (st.hb2 + 1) % RING; }
    else st.b[(st.hb2 + st.nb2++) % RING] = q;
  };
  auto stage1 = [&](PlaceState& st, const Pend1& p) {
    __builtin_prefetch(&fill[p.hb], 1, 0);
    __builtin_prefetch(&base[p.hb], 0, 0);
    if (st.na == RING) { stage2(st, st.a[st.ha]); st.a[st.ha] = p; st.ha = (st.ha + 1) % RING; }
    else st.a[(st.ha + st.na++) % RING] = p;
  };
  run_parallel([] { return PlaceState(); },
               [&](uint64_t u, PlaceState& st) {
    uint64_t b0 = ix->unitig_blk_off[u], b1 = ix->unitig_blk_off[u + 1];
    uint64_t cur = b0;
    const uint64_t g0 = ix->unitig_gpos[u];
    auto place = [&](uint32_t dist, uint64_t v, uint64_t rc) {
      // block containing dist: BlockArray::get_block_at = last block with lb <= dist (BlockArray.hpp:306-322)
      if (b1 - b0 > 1) { while (cur + 1 < b1 && ix->blk_lb[cur + 1] <= dist) ++cur; }
      const bool f = v < rc; const uint64_t cn = f ? v : rc;
      const uint32_t lb = ix->blk_lb[cur], ub = ix->blk_ub[cur];
      const uint32_t rem_f = ub - 1 - dist, rem_b = dist - lb;   // KmerIndex.cpp:1780-1789
      stage1(st, Pend1{cn, kamd::home_bucket(cn, nb), kamd::make_payload(rem_f, rem_b, ix->blk_uec[cur], f), (uint32_t)(g0 + dist), (uint32_t)cur, dist});
    };
    if (u < ix->n_long) for_each_kmer_rc(units[u].data, units[u].len, k, place);
    else { const uint64_t v = single_kmer(u); place(0, v, kamd::revcomp_msb(v, k)); }
    // the unitig's bases into the text (neighbouring unitigs share words: atomic OR)
    const uint64_t len = ix->unitig_len[u];
    const uint64_t sk = u < ix->n_long ? 0 : single_kmer(u);
    uint32_t acc = 0; uint64_t wi = g0 >> 4;
    for (uint64_t j = 0; j < len; j++) {
      const uint64_t g = g0 + j;
      if ((g >> 4) != wi) { if (acc) utext_atomic[wi].fetch_or(acc, std::memory_order_relaxed); acc = 0; wi = g >> 4; }
      const uint32_t b = u < ix->n_long ? (uint32_t)((units[u].data[j >> 2] >> ((j & 3) << 1)) & 3)
                                        : (uint32_t)((sk >> (2 * (k - 1 - (int)j))) & 3);
      acc |= b << (2 * (g & 15));
    }
    if (acc) utext_atomic[wi].fetch_or(acc, std::memory_order_relaxed);
  },
               [&](PlaceState& st) {
    for (int i = 0; i < st.na; i++) stage2(st, st.a[(st.ha + i) % RING]);
    st.na = 0;
    for (int i = 0; i < st.nb2; i++) stage3(st.b[(st.hb2 + i) % RING]);
    st.nb2 = 0;
  });
  tick("table: place pass");
  // ---- D-list table (same bucket layout, built serially: it is small) and the dummy hit ----
  if (ix->dlist_size) {
    constexpr uint64_t S = kamd::BUCKET_SLOTS;   // (the D-list table is always of the wide layout)
    const uint64_t nd = ix->dlist_size, ndb = std::max<uint64_t>(16, (nd * 2 + S - 1) / S);
    std::vector<std::pair<uint64_t, uint64_t>> byhome(nd);   // (home bucket, key)
    for (uint64_t i = 0; i < nd; i++) byhome[i] = {kamd::home_bucket(ix->dlist_keys[i], ndb), ix->dlist_keys[i]};
    std::sort(byhome.begin(), byhome.end());
    byhome.erase(std::unique(byhome.begin(), byhome.end()), byhome.end());
    // Robin Hood order: keys of a home bucket are laid down sequentially, never before their home
    std::vector<uint64_t> slot_of(byhome.size());
    uint64_t cur = 0;
    for (size_t i = 0; i < byhome.size(); i++) { cur = std::max(cur, byhome[i].first * S); slot_of[i] = cur++; }
    const uint64_t tb = std::max(ndb, (cur + S - 1) / S) + 1;
    ix->n_dbuckets = ndb; ix->dpad_buckets = tb - ndb;
    ix->dtable.assign(tb * 8, 0);
    for (uint64_t b = 0; b < tb; b++) for (uint64_t j = 0; j < S; j++) ix->dtable[8 * b + j] = kamd::KEY_EMPTY;
    for (size_t i = 0; i < byhome.size(); i++) ix->dtable[8 * (slot_of[i] / S) + slot_of[i] % S] = byhome[i].second;
    // a bucket continues into the next one when keys homed at or before it spill past its slots
    {
      uint64_t reach = 0;  // one past the last slot used by keys homed in buckets <= b
      size_t i = 0;
      for (uint64_t b = 0; b < tb; b++) {
        while (i < byhome.size() && byhome[i].first == b) { reach = slot_of[i] + 1; ++i; }
        if (reach > (b + 1) * S) ix->dtable[8 * b] |= kamd::KEY_CONT;
      }
    }
    kamd::Table mt = ktab; mt.slots = ix->table.data();
    const kamd::Probe pd = kamd::probe_table(mt, ix->dlist_keys[0], true, nullptr);
    if (!pd.found) return kamd::fail(-3, "index: Dummy k-mer not found in graph");   // KmerIndex.cpp:1398-1401
    ix->dummy_slot = pd.slot; ix->dummy_uec = pd.uec; ix->dummy_strand = pd.strand ? 1 : 0;
  }
  tick("D-list table + dummy hit");
  *out = ix.release();
  return 0;
}
}  // namespace

extern "C" void kamd_index_free(kamd_index* ix) { delete ix; }

extern "C" int kamd_index_get_view(const kamd_index* ix, kamd_index_view* v) {
  if (!ix || !v) return kamd::fail(-1, "kamd_index_get_view: null argument");
  memset(v, 0, sizeof *v);
  v->k = ix->k; v->n_kmers = ix->n_kmers; v->n_unitigs = ix->n_unitigs; v->n_blocks = ix->blk_lb.size();
  v->n_uec = ix->uec_ec.size(); v->n_ecs = ix->ec_off.size() - 1; v->ec_nnz = ix->ec_ids.size();
  v->n_targets = ix->n_targets; v->dlist_size = ix->dlist_size;
  v->n_buckets = ix->n_buckets; v->pad_buckets = ix->pad_buckets;
  v->table_layout = ix->layout; v->slots_per_bucket = ix->slots; v->tag_q = ix->tag_q; v->tag_dsh = ix->tag_dsh; v->tag_w = ix->tag_w;
  v->table = ix->table.data(); v->slot_block = ix->slot_block.data(); v->slot_dist = ix->slot_dist.data();
  v->uec_ec = ix->uec_ec.data(); v->ec_off = ix->ec_off.data(); v->ec_ids = ix->ec_ids.data();
  v->unitig_blk_off = ix->unitig_blk_off.data(); v->unitig_len = ix->unitig_len.data();
  v->blk_unitig = ix->blk_unitig.data(); v->blk_lb = ix->blk_lb.data(); v->blk_ub = ix->blk_ub.data(); v->blk_ec = ix->blk_ec.data();
  v->blk_pos_off = ix->blk_pos_off.data(); v->blk_posw = ix->blk_posw.data(); v->blk_sense = ix->blk_sense.data();
  v->target_lens = ix->target_lens.data(); v->onlist_bits = ix->onlist_bits.data(); v->onlist_words = ix->onlist_bits.size();
  v->dtable = ix->dtable.empty() ? nullptr : ix->dtable.data(); v->n_dbuckets = ix->n_dbuckets; v->dpad_buckets = ix->dpad_buckets;
  v->dummy_slot = ix->dummy_slot; v->dummy_uec = ix->dummy_uec; v->dummy_strand = ix->dummy_strand;
  v->utext = ix->utext.data(); v->utext_words = ix->utext.size(); v->text_bases = ix->text_bases;
  v->unitig_gpos = ix->unitig_gpos.data();
  return 0;
}

extern "C" const char* kamd_index_target_name(const kamd_index* ix, uint64_t i) {
  return (ix && i < ix->target_names.size()) ? ix->target_names[i].c_str() : nullptr;
}
