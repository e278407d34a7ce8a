// kamd_core.h -- per-item logic of the pseudoalignment kernels, written once as host/device inline functions.
//
// The __global__ kernels in kamd_match.hip / kamd_ec.hip are thin wrappers (LDS staging, wave-aggregated output, atomics) around
// the functions below.  They are also compilable for the host so that tests/emu can drive exactly the same logic on a
// CPU-only box (hipcc cross-compiles here but there is no GPU); that emulation build is TEST infrastructure, the
// product library never calls it.
//
// Reference semantics restated here (file:line in the reference tree):
//   KmerIterator                 ext/bifrost/src/KmerIterator.cpp:6-63
//   CompactedDBG::find           ext/bifrost/src/CompactedDBG.tcc:999-1119  (as an exact k-mer dictionary)
//   KmerIndex::match             src/KmerIndex.cpp:1698-1940  (default flags; quirks Q1-Q4 of SURVEY.md section 8a)
//   MinCollector::intersectKmers src/MinCollector.cpp:160-218,425-496
//   KmerIndex::mapPair           src/KmerIndex.cpp:1622-1693
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KAMD_HD __host__ __device__ __forceinline__
#else
#define KAMD_HD inline
#endif

namespace kamd {

// ---------------------------------------------------------------------------------------------------------------
// k-mer table layout (built by kamd_index.cpp, probed by the kernels)
//
//   bucket  = one 64-byte line = 3 slots, stored as 8 u64 words {key0, key1, key2, pay0, pay1, pay2, gpos0 | gpos1 << 32,
//             gpos2}: a slot is {key, payload, gpos} = 20 bytes, the last 4 bytes of the line are spare.
//   key     = canonical k-mer, MSB-first 2-bit, right-aligned (<= 62 bits).  Empty slot = KEY_EMPTY (all ones in the low
//             62 bits, never canonical because its reverse complement is 0).  Bit 63 of slot 0's key is the bucket's
//             "continue" flag: some key whose home is <= this bucket was placed beyond it.
//   gpos    = position of the k-mer's first base in `utext`, the 2-bit text of all unitigs laid end to end in unitig-forward
//             orientation (same packing as the reads: base i at bits 2*(i&15) of word i>>4).  A k-mer that match() expects on
//             the SAME unitig at a known offset from an earlier hit (the jump target, the middle k-mer, the back-off k-mer) is
//             first compared with the text at that offset: equality is exactly the answer CompactedDBG::find would give (a
//             k-mer occurs once in a compacted de Bruijn graph), and it costs a read of a small MALL-resident array instead
//             of a random 64-byte line of the table; only a mismatch is looked up in the table.
//   payload = rem_f[15:0] | rem_b[31:16] | uec[62:32] | fwd_is_canon[63]
//             rem_f = (ub-1-dist), rem_b = (dist-lb): k-mers left to the end of the mosaic block when walking the unitig
//             forward / backward, saturated at 65535 (exact for reads shorter than 65535+k);
//             uec   = id of the (unitig, transcript-set) class of the block: two hits satisfy
//                     "isSameReferenceUnitig && ec == ec" (KmerIndex.cpp:1810) iff their uec are equal.
//   Keys are laid out in home-bucket order (Robin-Hood linear probing at bucket granularity), so a lookup reads the home
//   bucket and follows continue flags; at load <= 0.5 that is 1.0x bucket reads.
// ---------------------------------------------------------------------------------------------------------------
//
// The COMPACT layout (LAYOUT_COMPACT; chosen when the index is loaded, kamd_index.cpp) holds the same information in 16-byte
// slots, FOUR per 64-byte line, so that a GENCODE-sized table shrinks from 43 to 21-29 bytes per k-mer.  It is exact -- no
// fingerprint that a second read would have to verify -- by quotienting: kmer_hash32 below is a bijection of the low 32 bits
// of the key for every value of the high bits (xor with a function of the high bits, odd multiplications, xor-shifts), and
// fastrange is monotone in the hash, so the keys of one home bucket have consecutive hashes and (high bits of the key, low
// q bits of the hash), q = ceil(log2(2^32 / n_buckets)), identify a key within its home bucket.
//   slot    = {w0, w1}; bucket = slots 0..3 = words {w0, w1} x 4
//   w0      = tag[tagw-1:0] | uec << tagw,   tag = hash & (2^q - 1) | (key >> 32) << q | displacement << (q + max(0, 2k - 32))
//             displacement (4 bits when the class ids leave room for them, else 3) = buckets between the slot's bucket and the key's
//             home, 0..14 (0..6); all ones marks an empty slot (w0 = all ones).  A lookup in bucket home + d compares with the tag of
//             displacement d: one 64-bit compare per slot.
//   w1      = rem_f[15:0] | rem_b[31:16] | gpos[61:32] | fwd_is_canon[62] | continue flag of the bucket[63] (slot 0 only)
//   The builder refuses the layout (and the loader falls back to the wide one unless it was asked for by name) when a field
//   does not fit: tagw + bits(uec) <= 64, text positions < 2^30.
// ---------------------------------------------------------------------------------------------------------------
static const int LAYOUT_WIDE = 0, LAYOUT_COMPACT = 1;
static const int COMPACT_SLOTS = 4;
static const uint64_t COMPACT_CONT = 1ULL << 63, COMPACT_FWD = 1ULL << 62;
static const uint32_t COMPACT_GPOS_MASK = 0x3FFFFFFFu;
static const uint64_t KEY_MASK = (1ULL << 62) - 1;
static const uint64_t KEY_EMPTY = KEY_MASK;
static const uint64_t KEY_CONT = 1ULL << 63;
static const uint32_t REM_CAP = 65535u;
static const uint32_t NO_UEC = 0x7FFFFFFFu;
static const int BUCKET_SLOTS = 3;
static const uint32_t UEC_MASK = 0x3FFFFFFFu;   // class lists keep two mate flags above 30 bits of uec

KAMD_HD uint64_t mix64(uint64_t x) {
  x ^= x >> 31; x *= 0x7fb5d329728ea185ULL;
  x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL;
  x ^= x >> 33;
  return x;
}
// home bucket of a canonical k-mer: a 32-bit mix (ten 32-bit VALU operations on the device; mix64 + a 64-bit fastrange
// were ~40) followed by fastrange, monotone in the hash so that sorting by hash == sorting by home.  n_buckets < 2^32.
KAMD_HD uint32_t kmer_hash32(uint64_t canon) {
  const uint32_t lo = (uint32_t)canon, hi = (uint32_t)(canon >> 32);
  uint32_t h = hi * 0x9E3779B1u; h ^= h >> 15;
  uint32_t x = (lo ^ h) * 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
KAMD_HD uint64_t bucket_of_hash(uint32_t h, uint64_t n_buckets) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__umulhi(h, (uint32_t)n_buckets);
#else
  return ((uint64_t)h * (uint32_t)n_buckets) >> 32;
#endif
}
KAMD_HD uint64_t home_bucket(uint64_t canon, uint64_t n_buckets) { return bucket_of_hash(kmer_hash32(canon), n_buckets); }
// compact layout: q for a table of n_buckets home buckets -- the hashes of one bucket are at most ceil(2^32 / n) consecutive values
KAMD_HD uint32_t compact_q_of(uint64_t n_buckets) {
  const uint64_t span = ((1ULL << 32) + n_buckets - 1) / n_buckets;
  uint32_t q = 0;
  while ((1ULL << q) < span) ++q;
  return q;
}
KAMD_HD uint64_t make_payload(uint32_t rem_f, uint32_t rem_b, uint32_t uec, bool fwd_is_canon) {
  if (rem_f > REM_CAP) rem_f = REM_CAP;
  if (rem_b > REM_CAP) rem_b = REM_CAP;
  return (uint64_t)rem_f | ((uint64_t)rem_b << 16) | ((uint64_t)(uec & 0x7FFFFFFFu) << 32) | ((uint64_t)fwd_is_canon << 63);
}

// reverse the order of the 2-bit bases of a 64-bit word
KAMD_HD uint64_t rev_bases64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  x = __brevll(x);
#else
  x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
#endif
  return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);  // un-swap the two bits of each base
}

// MSB-first value of the reverse complement of an MSB-first right-aligned k-mer
KAMD_HD uint64_t revcomp_msb(uint64_t v, int k) { return rev_bases64(~v) >> (64 - 2 * k) ; }

// ---------------------------------------------------------------------------------------------------------------
// packed reads: base i at bits 2*(i&15) of word i>>4 (LSB-first), N mask bit i&31 of mask word i>>5
// ---------------------------------------------------------------------------------------------------------------
struct ReadView {
  const uint32_t* seq;   // 2-bit words (LDS on the device)
  const uint32_t* mask;  // non-ACGT mask words
  int len;               // bases
  int stride = 1;        // distance between consecutive sequence words (64 for the lane-transposed LDS layout of kernel A)
  int mask_stride = 1;   // ... and between mask words (kernel A v3 leaves the mask plane in global memory: 1)
  bool has_n = true;     // false: the packer saw only ACGT (flag word of the record), the mask plane is never consulted
};
// the last sequence word of a packed record is padding (the bases fill at most seq_words - 1 words); the packers store the
// "read has a non-ACGT base" flag there
static const uint32_t REC_FLAG_HAS_N = 1u;

// bits [2w, 2w+2k) of the read, LSB-first: x = sum base[w+i] << 2i
KAMD_HD uint64_t window_lsb(const ReadView& r, int w, int k) {
  int bit = 2 * w;
  int wi = bit >> 5, sh = bit & 31;
  uint64_t lo = (uint64_t)r.seq[wi * r.stride] | ((uint64_t)r.seq[(wi + 1) * r.stride] << 32);
  uint64_t x = lo >> sh;
  if (sh + 2 * k > 64) x |= (uint64_t)r.seq[(wi + 2) * r.stride] << (64 - sh);  // third word only when the window reaches it
  return x & ((k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1));
}
// mask bits of bases [w, w+k) (k <= 32)
KAMD_HD uint32_t window_mask(const ReadView& r, int w, int k) {
  int wi = w >> 5, sh = w & 31;
  uint64_t lo = (uint64_t)r.mask[wi * r.mask_stride] | ((uint64_t)r.mask[(wi + 1) * r.mask_stride] << 32);  // one pad word per plane
  return (uint32_t)(lo >> sh) & (uint32_t)((1ULL << k) - 1);
}
// first window start >= t whose k bases are all ACGT and that fits in the read; -1 if none   (KmerIterator::operator++)
KAMD_HD int next_valid_window(const ReadView& r, int t, int k) {
  if (!r.has_n) return t + k <= r.len ? t : -1;
  while (t + k <= r.len) {
    uint32_t m = window_mask(r, t, k);
    if (m == 0) return t;
    t += 32 - __builtin_clz(m);  // jump past the last non-ACGT base of the window
  }
  return -1;
}
// KmerIterator::operator+=(n) applied to an iterator standing on window w (n >= 0)
KAMD_HD int advance_window(const ReadView& r, int w, int n, int k) {
  if (n == 0) return w;
  if (n == 1) return next_valid_window(r, w + 1, k);
  if (w + n + k > r.len) return -1;  // str[pos_e + n - 1] == '\0'
  return next_valid_window(r, w + n, k);
}

// canonical MSB-first key of the k-mer at window w; *is_fwd_canon = read-forward k-mer is the canonical one
KAMD_HD uint64_t window_canon(const ReadView& r, int w, int k, bool* is_fwd_canon) {
  uint64_t x = window_lsb(r, w, k);
  uint64_t fwd = rev_bases64(x) >> (64 - 2 * k);                    // MSB-first value of the read k-mer
  uint64_t rc = (~x) & ((1ULL << (2 * k)) - 1);                     // MSB-first value of its reverse complement
  *is_fwd_canon = fwd < rc;
  return fwd < rc ? fwd : rc;
}

// ---------------------------------------------------------------------------------------------------------------
// table probe
// ---------------------------------------------------------------------------------------------------------------
struct Table {
  const uint64_t* slots;  // 8 words per bucket
  uint64_t n_buckets;
  // D-list (kb-python style indices): the distinguishing flanking k-mers as a second table of the same layout (payload
  // unused), and the hit that is pushed when a read contains one of them -- um_dummy = dbg.find(first D-list k-mer),
  // src/KmerIndex.cpp:1386-1403.  n_dbuckets == 0: the index carries no D-list.
  const uint64_t* dslots = nullptr;
  uint64_t n_dbuckets = 0;
  uint32_t dummy_uec = 0;
  uint64_t dummy_slot = 0;
  bool dummy_strand = false;
  bool partial = false;   // match(..., partial): single-end reads (src/ProcessReads.cpp:1058)
  bool no_jump = false;   // --no-jump: every k-mer of the read is looked up (KmerIndex.cpp:1776)
  // layout of `slots` (the D-list table is always wide): LAYOUT_COMPACT with its shifts -- q, shift of the displacement, width of the tag
  uint8_t layout = LAYOUT_WIDE, q = 0, dsh = 0, tagw = 0;
};
// compact layout: the farthest a key may lie from its home bucket (the displacement field's all-ones value marks an empty slot)
KAMD_HD uint32_t compact_max_disp(const Table& t) { return (1u << (t.tagw - t.dsh)) - 2u; }
struct Probe {
  bool found;
  bool strand;      // read k-mer equals the unitig's forward text (const_UnitigMap::strand)
  uint32_t uec;
  uint32_t dist;    // "dist" of KmerIndex.cpp:1789: k-mers to the end of the block in read direction
  uint64_t slot;    // slot index (for the aux tables)
  uint32_t gpos;    // position of the k-mer in the unitig text
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long __attribute__((ext_vector_type(2))) kamd_u64x2;
#endif

// one 64-byte bucket, and what a probe finds in it
struct BucketLine { uint64_t k0, k1, k2, p0, p1, p2, g01, g2; };
KAMD_HD BucketLine load_bucket(const uint64_t* slots, uint64_t b) {
  const uint64_t* bp = slots + b * 8;
  BucketLine L;
#if defined(__HIP_DEVICE_COMPILE__)
  // four 16-byte loads of one 64-byte line, issued back to back
  // (non-temporal loads of the line were measured: 14.8 against 9.0-9.5 ms for kernel A on config #3 -- the four 16-byte loads of a line no longer
  // merge; round 6)
  const kamd_u64x2 s0 = ((const kamd_u64x2*)bp)[0];
  const kamd_u64x2 s1 = ((const kamd_u64x2*)bp)[1];
  const kamd_u64x2 s2 = ((const kamd_u64x2*)bp)[2];
  const kamd_u64x2 s3 = ((const kamd_u64x2*)bp)[3];
  L.k0 = s0.x; L.k1 = s0.y; L.k2 = s1.x; L.p0 = s1.y; L.p1 = s2.x; L.p2 = s2.y; L.g01 = s3.x; L.g2 = s3.y;
#else
  L.k0 = bp[0]; L.k1 = bp[1]; L.k2 = bp[2]; L.p0 = bp[3]; L.p1 = bp[4]; L.p2 = bp[5]; L.g01 = bp[6]; L.g2 = bp[7];
#endif
  return L;
}
enum { BUCKET_ABSENT = 0, BUCKET_FOUND = 1, BUCKET_CONTINUE = 2 };   // CONTINUE: the key may sit in the next bucket
KAMD_HD int match_bucket(const BucketLine& L, uint64_t canon, bool is_fwd_canon, uint64_t b, Probe& p) {
  p.found = false; p.strand = false; p.uec = NO_UEC; p.dist = 0; p.slot = 0; p.gpos = 0;
  uint64_t pay = 0; uint32_t gp = 0; int hit = -1;
  if ((L.k0 & KEY_MASK) == canon) { pay = L.p0; gp = (uint32_t)L.g01; hit = 0; }
  else if (L.k1 == canon) { pay = L.p1; gp = (uint32_t)(L.g01 >> 32); hit = 1; }
  else if (L.k2 == canon) { pay = L.p2; gp = (uint32_t)L.g2; hit = 2; }
  if (hit < 0) return (L.k0 & KEY_CONT) ? BUCKET_CONTINUE : BUCKET_ABSENT;
  const bool fwd_is_canon = (pay >> 63) != 0;
  p.found = true;
  p.strand = (is_fwd_canon == fwd_is_canon);
  p.uec = (uint32_t)(pay >> 32) & 0x7FFFFFFFu;
  p.dist = p.strand ? (uint32_t)(pay & 0xFFFF) : (uint32_t)((pay >> 16) & 0xFFFF);
  p.slot = b * BUCKET_SLOTS + (uint64_t)hit;
  p.gpos = gp;
  return BUCKET_FOUND;
}
// compact layout: the tag a key carries in a slot `disp` buckets beyond its home
KAMD_HD uint64_t compact_tag(const Table& t, uint64_t canon, uint32_t hash, uint32_t disp) {
  return (uint64_t)(hash & ((1u << t.q) - 1u)) | ((canon >> 32) << t.q) | ((uint64_t)disp << t.dsh);
}
KAMD_HD int match_bucket_compact(const BucketLine& L, const Table& t, uint64_t tag, bool is_fwd_canon, uint64_t b, Probe& p) {
  p.found = false; p.strand = false; p.uec = NO_UEC; p.dist = 0; p.slot = 0; p.gpos = 0;
  const uint64_t tm = (1ULL << t.tagw) - 1ULL;
  uint64_t w0 = 0, w1 = 0; int hit = -1;
  if ((L.k0 & tm) == tag) { w0 = L.k0; w1 = L.k1; hit = 0; }
  else if ((L.k2 & tm) == tag) { w0 = L.k2; w1 = L.p0; hit = 1; }
  else if ((L.p1 & tm) == tag) { w0 = L.p1; w1 = L.p2; hit = 2; }
  else if ((L.g01 & tm) == tag) { w0 = L.g01; w1 = L.g2; hit = 3; }
  // Not here.  The key may sit further on when keys homed at or before this bucket spill past it (the continue flag: the bucket is
  // full then) AND the key in the last slot is not homed beyond the key looked for: keys lie in the order of their homes, so a last
  // slot whose displacement is smaller than the lookup's closes the run of the lookup's home bucket.
  if (hit < 0) return ((L.k1 & COMPACT_CONT) && ((L.g01 & tm) >> t.dsh) >= (tag >> t.dsh)) ? BUCKET_CONTINUE : BUCKET_ABSENT;
  const bool fwd_is_canon = (w1 & COMPACT_FWD) != 0;
  p.found = true;
  p.strand = (is_fwd_canon == fwd_is_canon);
  p.uec = (uint32_t)(w0 >> t.tagw);
  p.dist = p.strand ? (uint32_t)(w1 & 0xFFFF) : (uint32_t)((w1 >> 16) & 0xFFFF);
  p.slot = b * COMPACT_SLOTS + (uint64_t)hit;
  p.gpos = (uint32_t)(w1 >> 32) & COMPACT_GPOS_MASK;
  return BUCKET_FOUND;
}
KAMD_HD Probe probe_table_compact(const Table& t, uint64_t canon, bool is_fwd_canon, uint32_t* bucket_reads) {
  Probe p;
  const uint32_t h = kmer_hash32(canon);
  const uint64_t home = bucket_of_hash(h, t.n_buckets);
  const uint32_t max_disp = compact_max_disp(t);
  for (uint32_t d = 0;; d++) {
    const BucketLine L = load_bucket(t.slots, home + d);
    if (bucket_reads) ++*bucket_reads;
    if (match_bucket_compact(L, t, compact_tag(t, canon, h, d), is_fwd_canon, home + d, p) != BUCKET_CONTINUE || d == max_disp) return p;
  }
}
KAMD_HD Probe probe_table(const Table& t, uint64_t canon, bool is_fwd_canon, uint32_t* bucket_reads) {
  if (t.layout == LAYOUT_COMPACT) return probe_table_compact(t, canon, is_fwd_canon, bucket_reads);
  Probe p;
  uint64_t b = home_bucket(canon, t.n_buckets);
  for (;;) {
    const BucketLine L = load_bucket(t.slots, b);
    if (bucket_reads) ++*bucket_reads;
    if (match_bucket(L, canon, is_fwd_canon, b, p) != BUCKET_CONTINUE) return p;
    ++b;  // the table carries pad buckets at the end, so this never runs off
  }
}

// The k-mer of the unitig text at base position g (unitig-forward): canonical MSB-first key, as window_canon gives for a read.
// `utext` carries two words of padding at the end.
struct TextWords { uint32_t a, b, c; };
KAMD_HD TextWords load_text(const uint32_t* utext, uint32_t g) { const uint32_t wi = g >> 4; return TextWords{utext[wi], utext[wi + 1], utext[wi + 2]}; }
KAMD_HD uint64_t text_canon_of(const TextWords& w, uint32_t g, int k) {
  const int sh = (int)(g & 15u) * 2;
  const uint32_t a = w.a, b = w.b, c = w.c;
  uint64_t x = ((uint64_t)a | ((uint64_t)b << 32)) >> sh;
  if (sh + 2 * k > 64) x |= (uint64_t)c << (64 - sh);
  x &= (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const uint64_t fwd = rev_bases64(x) >> (64 - 2 * k);
  const uint64_t rc = (~x) & ((1ULL << (2 * k)) - 1);
  return fwd < rc ? fwd : rc;
}
KAMD_HD uint64_t text_canon(const uint32_t* utext, uint32_t g, int k) { return text_canon_of(load_text(utext, g), g, k); }

// ---------------------------------------------------------------------------------------------------------------
// per-item accumulation of distinct transcript-set ids (sorted, unique, bounded)
// ---------------------------------------------------------------------------------------------------------------
// Entries are transcript-set ids; with --union they also carry which mate's hits had the set (EC_MATE1 / EC_MATE2: the union
// is taken per mate, MinCollector.cpp:163-169), otherwise the flag bits are zero.
static const uint32_t EC_MATE1 = 0x40000000u, EC_MATE2 = 0x80000000u, EC_ID_MASK = 0x3FFFFFFFu;
struct EcList {
  uint32_t* e;   // `cap` entries (LDS in the main kernel, global scratch in the overflow kernel)
  int cap;
  int n;
  bool overflow; // more than `cap` distinct sets: the item is re-run by the overflow kernel with a larger list
};
KAMD_HD void eclist_add(EcList& l, uint32_t ec_flags) {   // sorted by id; flags of equal ids are merged
  const uint32_t ec = ec_flags & EC_ID_MASK;
  int i = 0;
  while (i < l.n && (l.e[i] & EC_ID_MASK) < ec) ++i;
  if (i < l.n && (l.e[i] & EC_ID_MASK) == ec) { l.e[i] |= ec_flags; return; }
  if (l.n == l.cap) { l.overflow = true; return; }
  for (int j = l.n; j > i; --j) l.e[j] = l.e[j - 1];
  l.e[i] = ec_flags;
  ++l.n;
}

struct MateInfo {
  int n_hits;        // v.size()
  int n_nonempty;    // != 0 when some hit carries a non-empty transcript set
  // first pushed hit == findFirstMappingKmer (smallest recorded position, first wins) == mapPair's first present k-mer
  uint64_t first_slot;
  int first_pos;
  bool first_strand;
  uint32_t probes, bucket_reads;
};

// distinct (slot_block, strand) pairs of a mate's hits -- what the per-hit (`comprehensive`) strand filter needs
// (src/ProcessReads.cpp:62-82); entry = block << 1 | strand
struct HitBlocks {
  const uint32_t* slot_block; uint32_t* e; int cap; int n; bool overflow;
};
KAMD_HD void hitblocks_add(HitBlocks& h, uint64_t slot, bool strand) {
  const uint32_t x = (h.slot_block[slot] << 1) | (strand ? 1u : 0u);
  for (int i = 0; i < h.n; i++) if (h.e[i] == x) return;
  if (h.n == h.cap) { h.overflow = true; return; }
  h.e[h.n++] = x;
}
// KmerIndex::match for one mate.  `ecs` receives every pushed hit's transcript-set id (uec -> ec through uec_ec) once, with
// `mflag` (0, EC_MATE1 or EC_MATE2) or-ed in; `hb` (optional) every hit's block and strand.
KAMD_HD void match_mate(const Table& t, const uint32_t* uec_ec, const uint8_t* ec_nonempty, const ReadView& r, int k,
                        EcList& ecs, MateInfo& mi, uint32_t mflag = 0, HitBlocks* hb = nullptr) {
  mi.n_hits = 0; mi.n_nonempty = 0; mi.first_slot = 0; mi.first_pos = -1; mi.first_strand = false;
  mi.probes = 0; mi.bucket_reads = 0;
  const int l = r.len;
  uint32_t last_uec = NO_UEC;
  // push = v.push_back({um, pos}) : only the set id matters downstream, plus the very first hit
#define KAMD_PUSH(P, POS)                                                          \
  do {                                                                             \
    if (mi.n_hits == 0) { mi.first_slot = (P).slot; mi.first_pos = (POS); mi.first_strand = (P).strand; } \
    ++mi.n_hits;                                                                   \
    if (hb) hitblocks_add(*hb, (P).slot, (P).strand);                              \
    if ((P).uec != last_uec) {                                                     \
      last_uec = (P).uec;                                                          \
      uint32_t ec_ = uec_ec[(P).uec];                                              \
      if (ec_nonempty == nullptr || ec_nonempty[ec_]) { eclist_add(ecs, ec_ | mflag); mi.n_nonempty = 1; } \
    }                                                                              \
  } while (0)

  int w = next_valid_window(r, 0, k);
  while (w >= 0) {
    bool fc; uint64_t canon = window_canon(r, w, k, &fc);
    ++mi.probes;
    Probe um = probe_table(t, canon, fc, &mi.bucket_reads);
    if (um.found) {
      const int pos = w;
      KAMD_PUSH(um, pos);                                                          // KmerIndex.cpp:1774
      const int dist = (int)um.dist;                                               // :1789
      if (!t.no_jump && dist >= 2) {                                               // :1776, :1792
        int nextPos = pos + dist;
        if (pos + dist >= l - k) nextPos = l - k;                                  // :1796-1799
        int w2 = advance_window(r, w, nextPos - pos, k);                           // :1802-1803
        if (w2 < 0) break;                                                         // :1882-1886 (Q4)
        bool fc2; uint64_t c2 = window_canon(r, w2, k, &fc2);
        ++mi.probes;
        Probe um2 = probe_table(t, c2, fc2, &mi.bucket_reads);
        bool found2 = false; int found2pos = pos + dist;
        if (!um2.found) { found2 = true; found2pos = pos; }                        // :1807-1809 (Q2)
        else if (um2.uec == um.uec) { found2 = true; }                             // :1810-1815
        if (found2) {
          if (found2pos >= l - k) { KAMD_PUSH(um, l - k); break; }                 // :1819-1822
          KAMD_PUSH(um, found2pos); w = w2;                                        // :1823-1826
        } else {
          bool foundMiddle = false;
          if (dist > 4) {                                                          // :1831
            int middlePos = (pos + nextPos) / 2;
            int w3 = advance_window(r, w, middlePos - pos, k);
            if (w3 >= 0) {
              bool fc3; uint64_t c3 = window_canon(r, w3, k, &fc3);
              ++mi.probes;
              Probe um3 = probe_table(t, c3, fc3, &mi.bucket_reads);
              if (um3.found && (um3.uec == um.uec || um3.uec == um2.uec)) {        // :1842-1850
                foundMiddle = true;
                KAMD_PUSH(um3, 0);                                                 // :1866 (position irrelevant: not first)
                if (nextPos >= l - k) break;                                       // :1867-1868
                w = w2;                                                            // :1870 (Q3)
              }
            }
          }
          if (!foundMiddle) {                                                      // :1876-1925 with Q1: one-step back-off
            w = next_valid_window(r, w + 1, k);
            if (w < 0) break;
            bool fc4; uint64_t c4 = window_canon(r, w, k, &fc4);
            ++mi.probes;
            Probe um4 = probe_table(t, c4, fc4, &mi.bucket_reads);
            if (um4.found) KAMD_PUSH(um4, w);
          }
        }
      }
    }
    w = next_valid_window(r, w + 1, k);
  }
  // D-list (:1928-1939): the first k-mer of the read that is a distinguishing flanking k-mer pushes the dummy hit.  (The
  // early exit of :1818-1826 -- same test on the jump target in `partial` mode -- only ends the search sooner: the dummy
  // hit is pushed either way, and it is the set of pushed classes that decides the outcome.)
  if (t.n_dbuckets && (mi.n_hits > 0 || !t.partial)) {
    const Table dt{t.dslots, t.n_dbuckets};
    for (int wd = next_valid_window(r, 0, k); wd >= 0; wd = next_valid_window(r, wd + 1, k)) {
      bool fcd; const uint64_t cd = window_canon(r, wd, k, &fcd);
      if (probe_table(dt, cd, fcd, &mi.bucket_reads).found) {
        Probe dm; dm.found = true; dm.strand = t.dummy_strand; dm.uec = t.dummy_uec; dm.dist = 0; dm.slot = t.dummy_slot; dm.gpos = 0;
        KAMD_PUSH(dm, wd);
        break;
      }
    }
  }
#undef KAMD_PUSH
}

// ---------------------------------------------------------------------------------------------------------------
// The same match() as a resumable state machine that asks for exactly ONE table probe per step.  Kernel A (v2) keeps one
// of these per lane so that every loop iteration issues one probe for every lane (no serialised probe sites), and a lane
// that finishes a read immediately starts the next one.  States follow the probe sites of KmerIndex::match:
//   SCAN (:1753)  JUMP (:1804)  MIDDLE (:1839)  BACKOFF (:1900)
// ---------------------------------------------------------------------------------------------------------------
enum { PH_SCAN = 0, PH_JUMP = 1, PH_MIDDLE = 2, PH_BACKOFF = 3, PH_DONE = 4, PH_DLIST = 5 };  // PH_DLIST: the D-list scan of :1928-1939
struct MatchState {
  int phase;
  int w;        // window the next probe must look up
  int w0;       // window of the hit under examination (the iterator position `kit`)
  int w2;       // jump target window (`kit2`)
  int dist;     // :1789
  int nextPos;  // :1794-1799
  uint32_t um_uec, um2_uec;
  uint32_t um_gpos;   // text position of the hit under examination ...
  bool um_strand;     // ... and its orientation: window w0 + n lies at um_gpos + n (strand) or um_gpos - n of the unitig text
  bool text_tried;    // the pending window was compared with the text and differs: it goes to the table
  uint32_t disp;      // buckets past the home bucket that have been read for the pending window (continue flags)
};
// May the pending probe of a JUMP / MIDDLE / BACK-OFF step be answered from the unitig text?  Only while the window is
// within `dist` k-mers of the hit (then it lies in the same block of the same unitig, :1780-1799) -- an iterator that had to
// skip non-ACGT windows can land beyond it.
KAMD_HD bool text_applies(const MatchState& st) {
  return (st.phase == PH_JUMP || st.phase == PH_MIDDLE || st.phase == PH_BACKOFF) && !st.text_tried && st.w - st.w0 <= st.dist && st.w > st.w0;
}
KAMD_HD uint32_t text_pos_of(const MatchState& st) {
  const uint32_t n = (uint32_t)(st.w - st.w0);
  return st.um_strand ? st.um_gpos + n : st.um_gpos - n;
}
// distinct (unitig, set) classes seen by one item: entry = uec | mate flags (bit 30: mate 1, bit 31: mate 2)
struct UecList {
  uint32_t* e; int cap; int n; bool overflow;
  int stride = 1;   // distance between consecutive entries (kernel A v3 keeps the lists thread-transposed in LDS)
  // append mode (kernel A's pass over the items whose eight-entry list overflowed): the list lies in global memory and is never read back --
  // a class equal to the one appended last (the common case: consecutive hits on one unitig) only merges its mate flag, anything else is
  // appended, O(1) per hit; the duplicates that are not neighbours are removed when the classes are mapped to sets (k_classify_long)
  bool append = false;
  uint32_t last = NO_UEC, last_flags = 0;
};
KAMD_HD void ueclist_add(UecList& l, uint32_t uec, int mate) {
  const uint32_t flag = mate ? 0x80000000u : 0x40000000u;
  if (l.append) {
    if (l.n > 0 && l.last == uec) {
      if (!(l.last_flags & flag)) { l.last_flags |= flag; l.e[(l.n - 1) * l.stride] = uec | l.last_flags; }
      return;
    }
    if (l.n == l.cap) { l.overflow = true; return; }
    l.e[l.n * l.stride] = uec | flag;
    l.last = uec; l.last_flags = flag;
    ++l.n;
    return;
  }
  for (int i = l.n - 1; i >= 0; --i)
    if ((l.e[i * l.stride] & 0x3FFFFFFFu) == uec) { l.e[i * l.stride] |= flag; return; }
  if (l.n == l.cap) { l.overflow = true; return; }
  l.e[l.n * l.stride] = uec | flag;
  ++l.n;
}
struct MateFirst { int n_hits; uint64_t slot; int pos; bool strand; };

// the table the pending probe of a phase goes to
KAMD_HD Table phase_table(const Table& t, int phase) { return phase == PH_DLIST ? Table{t.dslots, t.n_dbuckets} : t; }
// when match()'s loop is over: start the D-list scan if the index has one (:1928-1930)
// (DL: the index has a D-list -- a compile-time switch so that kernels for ordinary indices carry none of this)
template <bool DL>
KAMD_HD void match_finish(MatchState& st, const ReadView& r, int k, const Table& t, const MateFirst& mf) {
  st.phase = PH_DONE;
  if (DL && t.n_dbuckets && (mf.n_hits > 0 || !t.partial)) {
    st.w = next_valid_window(r, 0, k);
    if (st.w >= 0) st.phase = PH_DLIST;
  }
}
KAMD_HD void match_init(MatchState& st, const ReadView& r, int k) {
  st.w = next_valid_window(r, 0, k);
  st.phase = st.w >= 0 ? PH_SCAN : PH_DONE;
  st.w0 = st.w2 = st.dist = st.nextPos = 0; st.um_uec = st.um2_uec = NO_UEC;
  st.um_gpos = 0; st.um_strand = false; st.text_tried = false; st.disp = 0;
}
// consume the probe result of window st.w.  Written data-flow style: every phase only decides (a) whether the hit is
// recorded, (b) where the next window search starts and (c) the phase that follows; the list insertion and the single
// next_valid_window call are shared by all phases, so a wavefront whose lanes are in different phases executes them once.
template <bool DL>
KAMD_HD void match_feed(MatchState& st, const ReadView& r, int k, const Probe& p, UecList& list, int mate, MateFirst& mf, const Table& t) {
  const int lk = r.len - k;
  const int ph = st.phase;
  st.text_tried = false; st.disp = 0;
  if (DL && ph == PH_DLIST) {   // p = probe of the D-list table
    if (p.found) {
      if (mf.n_hits == 0) { mf.slot = t.dummy_slot; mf.pos = st.w; mf.strand = t.dummy_strand; }
      ++mf.n_hits;
      ueclist_add(list, t.dummy_uec, mate);
      st.phase = PH_DONE;
      return;
    }
    st.w = next_valid_window(r, st.w + 1, k);
    if (st.w < 0) st.phase = PH_DONE;
    return;
  }
  bool hit = false, add = false, done = false;
  int start = -1;              // next_valid_window(start) decides the next window ...
  bool guard_len = false;      // ... after operator+='s end-of-string test (start + k > len -> end), KmerIterator.cpp:47-60
  bool stay = false;           // advance by 0: the iterator stays on st.w0 (Q4)
  int ph_ok = PH_SCAN;
  bool fallback_backoff = false;  // the middle window does not exist: take the back-off path instead

  if (ph == PH_SCAN) {
    start = st.w + 1;
    if (p.found) {
      hit = add = true;
      if (mf.n_hits == 0) { mf.slot = p.slot; mf.pos = st.w; mf.strand = p.strand; }
      const int dist = (int)p.dist;                                                // :1789
      if (!t.no_jump && dist >= 2) {                                               // :1776, :1792
        const int pos = st.w;
        const int nextPos = (pos + dist >= lk) ? lk : pos + dist;                  // :1794-1799
        const int n = nextPos - pos;                                               // kit2 += nextPos - pos (:1803)
        st.w0 = pos; st.dist = dist; st.nextPos = nextPos; st.um_uec = p.uec;
        st.um_gpos = p.gpos; st.um_strand = p.strand;
        ph_ok = PH_JUMP;
        if (n == 0) stay = true; else { start = pos + n; guard_len = n >= 2; }
      }
    }
  } else if (ph == PH_JUMP) {
    const bool found2 = !p.found || p.uec == st.um_uec;                            // :1807-1815
    if (found2) {
      hit = true;                                                                  // push {um, found2pos}: um's class is in the list
      const int found2pos = p.found ? st.w0 + st.dist : st.w0;                     // (Q2)
      if (found2pos >= lk) done = true;                                            // :1819-1822
      else start = st.w2 + 1;                                                      // kit = kit2; ++kit
    } else {
      st.um2_uec = p.uec;
      start = st.w0 + 1; ph_ok = PH_BACKOFF;                                       // :1876-1925 with Q1: one-step back-off
      if (st.dist > 4) {                                                           // :1831: try the middle k-mer first
        const int n = (st.w0 + st.nextPos) / 2 - st.w0;                            // middlePos - pos
        ph_ok = PH_MIDDLE; fallback_backoff = true;
        if (n == 0) stay = true; else { start = st.w0 + n; guard_len = n >= 2; }
      }
    }
  } else if (ph == PH_MIDDLE) {
    if (p.found && (p.uec == st.um_uec || p.uec == st.um2_uec)) {                  // :1842-1850
      hit = add = true;                                                            // :1866 (Q3)
      if (st.nextPos >= lk) done = true;                                           // :1867-1868
      else start = st.w2 + 1;
    } else { start = st.w0 + 1; ph_ok = PH_BACKOFF; }
  } else {  // PH_BACKOFF
    if (p.found) hit = add = true;                                                 // :1900-1917
    start = st.w + 1;
  }
  if (hit) ++mf.n_hits;
  if (add) ueclist_add(list, p.uec, mate);
  if (done) { match_finish<DL>(st, r, k, t, mf); return; }
  int w;
  if (stay) w = st.w0;
  else w = (guard_len && start + k > r.len) ? -1 : next_valid_window(r, start, k);
  if (w < 0 && fallback_backoff) { w = next_valid_window(r, st.w0 + 1, k); ph_ok = PH_BACKOFF; }
  if (ph_ok == PH_JUMP) { if (w < 0) { match_finish<DL>(st, r, k, t, mf); return; } st.w2 = w; }  // :1882-1886 (Q4)
  if (w < 0) { match_finish<DL>(st, r, k, t, mf); return; }
  st.w = w;
  st.phase = ph_ok;
}

// map the item's (unitig, set) classes to sorted distinct non-empty transcript-set ids; reports per mate whether any of
// its hits carried a non-empty set (MinCollector::intersectECs skips empty sets, MinCollector.cpp:463-471)
KAMD_HD void uecs_to_ecs(const uint32_t* uecs, int n, const uint32_t* uec_ec, const uint8_t* ec_nonempty, EcList& out,
                         bool* nonempty0, bool* nonempty1, bool keep_flags = false) {
  *nonempty0 = *nonempty1 = false;
  for (int i = 0; i < n; i++) {
    const uint32_t u = uecs[i];
    const uint32_t ec = uec_ec[u & 0x3FFFFFFFu];
    if (ec_nonempty != nullptr && !ec_nonempty[ec]) continue;
    if (u & 0x40000000u) *nonempty0 = true;
    if (u & 0x80000000u) *nonempty1 = true;
    eclist_add(out, keep_flags ? (ec | (u & 0xC0000000u)) : ec);   // (the class lists use the same two flag bits)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The transcript set of an item before the on-list mask and the positional filters (MinCollector::intersectKmers,
// src/MinCollector.cpp:160-218), enumerated in increasing order:
//   default   the intersection of the collected sets (intersectECs per mate, then u1 & u2 -- one intersection of all)
//   --union   (union of mate 1's sets) & (union of mate 2's sets); a mate without non-empty sets imposes nothing
//             (unionECs :498-546; the emptiness rules of :172-202 are pair_is_mapped's)
// `cur`: ecs.n words of scratch (cursors of the k-way merge), only used with --union.
// ---------------------------------------------------------------------------------------------------------------
struct SetTables { const uint64_t* ec_off; const uint32_t* ec_ids; };
template <class F>
KAMD_HD void for_each_in_set(const SetTables& st, const EcList& ecs, bool union_mode, uint32_t* cur, F&& f) {
  if (ecs.n == 0) return;
  if (!union_mode) {
    int best = 0; uint64_t best_sz = ~0ULL;
    for (int j = 0; j < ecs.n; j++) { const uint32_t e = ecs.e[j] & EC_ID_MASK; const uint64_t sz = st.ec_off[e + 1] - st.ec_off[e]; if (sz < best_sz) { best_sz = sz; best = j; } }
    const uint32_t* base = st.ec_ids + st.ec_off[ecs.e[best] & EC_ID_MASK];
    for (uint64_t c = 0; c < best_sz; c++) {
      const uint32_t x = base[c];
      bool ok = true;
      for (int j = 0; ok && j < ecs.n; j++) {
        if (j == best) continue;
        const uint32_t e = ecs.e[j] & EC_ID_MASK;
        const uint32_t* ids = st.ec_ids + st.ec_off[e];
        uint64_t lo = 0, hi = st.ec_off[e + 1] - st.ec_off[e];
        const uint64_t n = hi;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (ids[mid] < x) lo = mid + 1; else hi = mid; }
        ok = lo < n && ids[lo] == x;
      }
      if (ok) f(x);
    }
    return;
  }
  bool has1 = false, has2 = false;
  for (int j = 0; j < ecs.n; j++) { cur[j] = 0; has1 = has1 || (ecs.e[j] & EC_MATE1); has2 = has2 || (ecs.e[j] & EC_MATE2); }
  for (;;) {
    uint32_t x = 0xFFFFFFFFu;
    for (int j = 0; j < ecs.n; j++) {
      const uint32_t e = ecs.e[j] & EC_ID_MASK;
      const uint64_t p = st.ec_off[e] + cur[j];
      if (p < st.ec_off[e + 1]) { const uint32_t h = st.ec_ids[p]; if (h < x) x = h; }
    }
    if (x == 0xFFFFFFFFu) return;
    bool in1 = false, in2 = false;
    for (int j = 0; j < ecs.n; j++) {
      const uint32_t e = ecs.e[j] & EC_ID_MASK;
      const uint64_t p = st.ec_off[e] + cur[j];
      if (p < st.ec_off[e + 1] && st.ec_ids[p] == x) { ++cur[j]; in1 = in1 || (ecs.e[j] & EC_MATE1); in2 = in2 || (ecs.e[j] & EC_MATE2); }
    }
    if ((in1 || !has1) && (in2 || !has2)) f(x);
  }
}
// upper bound of the set's size (what a record of it may need): smallest list / sum of the lists
KAMD_HD uint64_t set_size_bound(const SetTables& st, const uint32_t* e, int n, bool union_mode) {
  uint64_t mn = ~0ULL, sum = 0;
  for (int j = 0; j < n; j++) { const uint32_t x = e[j] & EC_ID_MASK; const uint64_t sz = st.ec_off[x + 1] - st.ec_off[x]; sum += sz; if (sz < mn) mn = sz; }
  return union_mode ? sum : (n ? mn : 0);
}

// Outcome of intersectKmers' emptiness rules (MinCollector.cpp:172-202) given per-mate facts.
// Returns true when the item is pseudoaligned to the intersection of the collected (non-empty) sets.
KAMD_HD bool pair_is_mapped(const MateInfo& a, const MateInfo& b) {
  bool u1_empty = (a.n_nonempty == 0), u2_empty = (b.n_nonempty == 0);
  if (u1_empty && u2_empty) return false;
  if (u1_empty) return a.n_hits == 0;
  if (u2_empty) return b.n_hits == 0;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// positional filters of ReadProcessor::processBuffer (src/ProcessReads.cpp:1095-1145)
//   - fragment-length compatibility of single-end reads / orphan mates via KmerIndex::findPosition
//     (src/KmerIndex.cpp:2188-2292)
//   - doStrandSpecificity (src/ProcessReads.cpp:61-124, non-comprehensive branch)
// Both replace u by {tr in u : keep(tr)} where keep() only depends on the first mapping k-mer of a mate
// (findFirstMappingKmer) and on tr, so the final set is u filtered by the conjunction of the predicates.
// ---------------------------------------------------------------------------------------------------------------
struct PosTables {
  const uint64_t* unitig_blk_off;  // [n_unitigs+1]
  const uint32_t* unitig_len;      // bp
  const uint32_t* blk_unitig;
  const uint32_t* blk_lb;
  const uint32_t* blk_ub;
  const uint32_t* blk_ec;
  const uint64_t* blk_pos_off;     // into blk_posw / blk_sense
  const uint32_t* blk_posw;
  const uint8_t* blk_sense;
  const uint64_t* ec_off;
  const uint32_t* ec_ids;
  const int32_t* target_lens;
  int k;
};
struct FirstHit {      // findFirstMappingKmer(v): um + p
  bool valid;
  uint32_t block;      // global block id of the hit (slot_block)
  uint32_t dist;       // um.dist (slot_dist)
  bool strand;         // um.strand
  int pos;             // p
};

// rank of tr in the transcript set of block b, or -1
KAMD_HD int64_t blk_rank(const PosTables& pt, uint64_t b, uint32_t tr) {
  const uint32_t ec = pt.blk_ec[b];
  const uint32_t* ids = pt.ec_ids + pt.ec_off[ec];
  uint64_t n = pt.ec_off[ec + 1] - pt.ec_off[ec], lo = 0, hi = n;
  while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (ids[m] < tr) lo = m + 1; else hi = m; }
  return (lo < n && ids[lo] == tr) ? (int64_t)lo : -1;
}
// BlockArray::get_block_at(idx) with the reference's wrap-around for idx = (uint32)-1 (BlockArray.hpp:306-322)
KAMD_HD uint64_t block_at(const PosTables& pt, uint64_t b0, uint64_t b1, int64_t idx) {
  if (b1 - b0 == 1) return b0;
  if (idx < 0) return b1 - 1;
  uint64_t a = b0, b = b1;
  while (a < b) { uint64_t m = (a + b) >> 1; if ((int64_t)pt.blk_lb[m] <= idx) a = m + 1; else b = m; }
  return a == b0 ? b0 : a - 1;
}

// KmerIndex::findPosition(tr, km, um, p) -> (pos, sense)
KAMD_HD int find_position(const PosTables& pt, uint32_t tr, const FirstHit& um, bool* sense_out) {
  const int k = pt.k;
  const uint32_t gid = pt.blk_unitig[um.block];
  const uint64_t b0 = pt.unitig_blk_off[gid], b1 = pt.unitig_blk_off[gid + 1];
  const uint64_t vb = um.block;                 // ecs.back(): the block containing um.dist
  const int64_t nlead = (int64_t)(vb - b0) + 1; // get_leading_vals(um.dist).size()
  const int p = um.pos;
  const bool csense = um.strand;
  int64_t rk = blk_rank(pt, vb, tr);
  if (rk < 0) { *sense_out = true; return -1; }
  const uint32_t rawpos = pt.blk_posw[pt.blk_pos_off[vb] + (uint64_t)rk];
  const int trpos = (int)(rawpos & 0x7FFFFFFFu);
  const bool trsense = ((uint32_t)trpos == rawpos);
  uint32_t mc_first = pt.blk_lb[vb], mc_second = pt.blk_ub[vb];
  const int64_t um_size = pt.unitig_len[gid], um_dist = um.dist;
  int64_t ret; bool rsense;
  if (trsense) {
    if (csense) {                                                                    // Case I   :2213-2225
      int64_t padding = 0;
      if (trpos == 0) {
        for (int64_t i = nlead - 2; i >= 0; i--) {
          if (blk_rank(pt, b0 + (uint64_t)i, tr) < 0) { padding = mc_first; break; }
          uint64_t nb = block_at(pt, b0, b1, (int64_t)mc_first - 1);
          mc_first = pt.blk_lb[nb]; mc_second = pt.blk_ub[nb];
        }
      }
      ret = (int64_t)trpos - p + um_dist + 1 - padding; rsense = csense;
    } else {                                                                         // Case III :2226-2240
      const int64_t initial = mc_second;
      int right_one = 0, left_one = 0;
      for (int64_t i = nlead - 1; i >= 0; i--) {
        if (i == nlead - 1) right_one = (int)mc_second;
        if (blk_rank(pt, b0 + (uint64_t)i, tr) < 0) { left_one = (int)mc_second; break; }
        else if (i == 0) left_one = 0;
        uint64_t nb = block_at(pt, b0, b1, (int64_t)mc_first - 1);
        mc_first = pt.blk_lb[nb]; mc_second = pt.blk_ub[nb];
      }
      const int64_t padding = -((int64_t)left_one + right_one - um_size + k - 1);
      ret = (int64_t)trpos + p + k - (um_size - k - um_dist) + initial - 1 + padding; rsense = csense;
    }
  } else {                                                                           // Cases IV / II :2243-2288
    int64_t curr_mc = 0; int left_one = 0, right_one = 0, unmapped_len = 0; bool found_first_mapped = false;
    for (uint64_t i = 0; i < b1 - b0; i++) {
      const uint64_t mb = block_at(pt, b0, b1, curr_mc);
      const bool has = blk_rank(pt, b0 + i, tr) >= 0;
      if (!has && found_first_mapped) {
        if (unmapped_len == 0) left_one = (int)pt.blk_lb[mb];
        right_one = (int)pt.blk_ub[mb];
        unmapped_len += (int)(pt.blk_ub[mb] - pt.blk_lb[mb]);
      }
      if (has) found_first_mapped = true;
      curr_mc = pt.blk_ub[mb];
    }
    if (csense) {
      int64_t start = 0;
      start -= right_one - left_one;
      start += um_size - k;
      ret = (int64_t)trpos + (-(um_dist - start)) + k + p; rsense = !csense;
    } else {
      unmapped_len = right_one - left_one;
      const int64_t padding = um_size - um_dist - unmapped_len - k + 1;
      ret = (int64_t)trpos + padding - p; rsense = !csense;
    }
  }
  *sense_out = rsense;
  return (int)ret;
}

struct FilterCfg {
  bool fraglen;   // !single_overhang && has_mean_fl && (single-end || one mate without hits)
  int fl;         // (int) tc.get_mean_frag_len()
  int strand;     // 0 none, 1 FR, 2 RF
  // --union / --no-jump switch doStrandSpecificity to its per-hit mode (src/ProcessReads.cpp:62-82, 1139-1140): the filter
  // runs once per hit of mate 1 (that hit alone, then mate 2's first mapping k-mer) and once per hit of mate 2 (as written
  // there: an empty first list, then mate 2's first mapping k-mer), and the results are united.  Every run is a predicate on
  // the transcript, so: mate 2 has hits -> its first mapping k-mer alone decides (the runs of mate 1's hits give subsets);
  // mate 2 has none -> a transcript is kept if ANY hit of mate 1 passes it (`hits1`, the distinct block / strand pairs).
  bool comprehensive = false;
  const uint32_t* hits1 = nullptr; int n_hits1 = 0;
};
// the test of doStrandSpecificity for one hit (:87-99): u &= ec, then sense against the wanted strand
KAMD_HD bool strand_ok(const PosTables& pt, uint32_t block, bool um_strand, bool want, uint32_t tr) {
  const int64_t rk = blk_rank(pt, block, tr);
  if (rk < 0) return false;
  const int sense = pt.blk_sense[pt.blk_pos_off[block] + (uint64_t)rk];
  return (((um_strand == (sense != 0)) == want) || sense == 2);
}
// keep(tr) for one item: h1 / h2 = first mapping k-mers of mate 1 / 2 (valid = the mate has hits)
KAMD_HD bool keep_transcript(const PosTables& pt, const FilterCfg& cfg, const FirstHit& h1, const FirstHit& h2, uint32_t tr) {
  if (cfg.fraglen) {
    const FirstHit& um = h2.valid ? h2 : h1;  // ProcessReads.cpp:1104-1115: mate 2's k-mer wins when both are set
    bool sense; int x = find_position(pt, tr, um, &sense);
    bool keep = false;
    if (sense && x + cfg.fl <= (int)pt.target_lens[tr]) keep = true;    // :1122-1126
    if (!sense && x - cfg.fl >= 0) keep = true;                          // :1127-1131
    if (!keep) return false;
  }
  if (cfg.strand && cfg.comprehensive) {
    if (h2.valid) return strand_ok(pt, h2.block, h2.strand, cfg.strand == 2, tr);
    for (int i = 0; i < cfg.n_hits1; i++)
      if (strand_ok(pt, cfg.hits1[i] >> 1, (cfg.hits1[i] & 1u) != 0, cfg.strand == 1, tr)) return true;
    return false;
  }
  if (cfg.strand) {
    for (int mate = 0; mate < 2; mate++) {
      const FirstHit& um = mate ? h2 : h1;
      if (!um.valid) continue;
      const bool want = mate ? (cfg.strand == 2) : (cfg.strand == 1);   // :87 firstStrand = FR, :106 secondStrand = RF
      if (!strand_ok(pt, um.block, um.strand, want, tr)) return false;  // u &= ec; :98
    }
  }
  return true;
}

}  // namespace kamd
