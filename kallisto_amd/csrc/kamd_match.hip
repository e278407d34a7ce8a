// kamd_match.hip -- kernel A and the straight-line matchers: kamd_pseudoalign, kamd_fld_* (KmerIndex::match, MinCollector::intersectKmers, KmerIndex::mapPair)
#include "kamd_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Kernel A
// ------------------------------------------------------------------------------------------------------------------
struct AlignOut {
  u32* dense_counts;   // [n_ecs]
  u64* dense_first;    // [n_ecs] record-stream offset of the first single-set item that hit the set (first-occurrence order)
  u32* stream;         // records [cnt, m, e0..e(m-1)]
  u64* rec_off;        // word offset of each record
  u64* overflow_items; // item indices for the overflow kernel
  u64* explicit_items; // items whose set a positional filter changed (re-run by k_explicit_write)
  u64* explicit_items_big;
  DevState* st;
};

// Common tail of kernel A: classify one item per lane and emit it (must be called by all 64 lanes of the wavefront).
//   single non-empty set  -> dense count vector;  several sets -> tuple record, ONE allocation per wavefront (inclusive
//   scan of the record sizes across the lanes);  list overflow -> overflow kernel;  set changed by a positional filter
//   -> explicit-set pass.
template <bool PAIRED, bool FILTER>
__device__ __forceinline__ void emit_item(const DevIndex& ix, const FilterDev& fd, const AlignOut& out, const kamd::EcList& ecs,
                                          const kamd::MateInfo& m0, const kamd::MateInfo& m1, u64 item, bool active) {
  // classify: 0 unmapped, 1 single set, 2 tuple, 3 overflow, 4 explicit set (positional filter changed it)
  int kind = 0;
  if (active) {
    if (ecs.overflow) kind = 3;
    else if (kamd::pair_is_mapped(m0, m1)) kind = ecs.n == 1 ? 1 : 2;
  }
  if (FILTER && (kind == 1 || kind == 2)) {
    u32 kept = 0;
    u32 cur[TUPLE_CAP];
    const int oc = filter_outcome(ix, fd, PAIRED, m0, m1, ecs, &kept, cur);
    if (oc == 1) kind = 0;
    else if (oc == 2) {
      kind = 4;
      const u64 k = atomicAdd(&out.st->n_explicit, 1ULL);
      out.explicit_items[k] = item;
      atomicAdd(&out.st->exp_words, (u64)kept + 2);
    }
  }
  if (kind == 1) atomicAdd(&out.dense_counts[ecs.e[0]], 1u);
  const u32 need = kind == 2 ? (u32)ecs.n + 2u : 0u;
  const u32 incl = wave_incl_scan(need);
  const u32 wave_total = __shfl(incl, 63, 64);
  const u64 multi_mask = __ballot(kind == 2);
  u64 base_words = 0, base_recs = 0;
  if (wave_total) {
    if (lane_id() == 0) {
      base_words = atomicAdd(&out.st->stream_words, (u64)wave_total);
      base_recs = atomicAdd(&out.st->n_recs, (u64)__popcll(multi_mask));
    }
    base_words = __shfl(base_words, 0, 64);
    base_recs = __shfl(base_recs, 0, 64);
    if (kind == 2) {
      const u64 off = base_words + (incl - need);
      const u64 ridx = base_recs + (u64)__popcll(multi_mask & ((1ULL << lane_id()) - 1));
      u32* w = out.stream + off;
      w[0] = 1u; w[1] = (u32)ecs.n;
      for (int i = 0; i < ecs.n; i++) w[2 + i] = ecs.e[i];
      out.rec_off[ridx] = off;
    }
  }
  if (kind == 3) { u64 i = atomicAdd(&out.st->n_overflow, 1ULL); out.overflow_items[i] = item; }
  const u64 s_single = (u64)__popcll(__ballot(kind == 1));
  const u64 s_multi = (u64)__popcll(multi_mask);
  const u64 s_proc = (u64)__popcll(__ballot(active));
  if (lane_id() == 0) {
    if (s_single) atomicAdd(&out.st->st_single, s_single);
    if (s_multi) atomicAdd(&out.st->st_multi, s_multi);
    atomicAdd(&out.st->st_processed, s_proc);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Kernel A: per-lane match() state machines (kamd_core.h MatchState).
//   k_match_v3   every lane owns a resumable match(); every iteration all busy lanes issue one memory access together, and a
//                lane that finishes its item takes the next one of the wavefront's chunk, its packed reads being fetched while
//                the others probe.  Reads live in a lane-transposed LDS layout (word j of lane i at [j*64+i]: conflict-free).
//                Output: one raw record per item {header, distinct (unitig,set) classes}.
//   k_classify   one thread per item: classes -> sorted distinct transcript-set ids, then the common emit_item tail.
// (The two earlier versions -- block-staged reads with a straight-line match per lane, 42 ms; one table probe per iteration with
// 12-entry lists, 14 ms -- were removed once version 3 had replaced them; the straight-line matcher lives on in the overflow,
// explicit-set and fragment-length kernels.)
// ------------------------------------------------------------------------------------------------------------------
constexpr u32 RAW_OVERFLOW = 1u << 8, RAW_HIT0 = 1u << 9, RAW_HIT1 = 1u << 10;

// ------------------------------------------------------------------------------------------------------------------
// Kernel A, version 3: the state machines of version 2, with
//   * the unitig text in front of the table: a JUMP / MIDDLE / BACK-OFF window that lies within the block of the hit under
//     examination is compared with the text at its expected position (12 bytes of an 18 MB array that lives in L2 / MALL);
//     a match is what the table would have answered, a mismatch sends the window to the table in the next iteration.  On
//     config #3 a quarter of all probes (every successful jump) never touch the 2.4 GB table;
//   * 24 words of LDS per lane instead of 38 (sequence words of both mates + 8 classes; the non-ACGT plane stays in global
//     memory and is only read for items whose record carries the has-N flag): 24 wavefronts per CU instead of 16;
//   * items handed out from a wavefront-uniform cursor (ballot + prefix count, no LDS atomic), class lists thread-transposed
//     in LDS (conflict-free), and the 32-bit k-mer hash.
// More than V3_LIST_CAP distinct classes (0.5 % of config #3's pairs): the item goes to the overflow kernel, as before.
// ------------------------------------------------------------------------------------------------------------------
constexpr int V3_LIST_CAP = 8;
constexpr int V3_LIST_CAP_LONG = 192;   // the second pass over the items whose list overflowed (k_match_v3<..., V3_LIST_CAP_LONG, true> on an item list)
// LCAP: class entries per item (in LDS; APPEND: in the item's raw record in global memory, appended, never scanned -- kamd_core.h UecList).
// item_idx != null: the launch works on the items item_idx[0 .. n_items) of the batch (their raw records go to slots 0 .. n_items of `raw`):
// the second pass over the items whose list overflowed in the first -- a pair inside a repeat family or a poly-A stretch has dozens of
// distinct classes --, with the same data-flow matcher at the same occupancy instead of 64 divergent straight-line ones per wavefront.
template <bool PAIRED, bool FILTER, bool DL, bool TEXT, int LAYOUT, int LCAP = V3_LIST_CAP, bool APPEND = false>
__global__ __launch_bounds__(BLOCK) void k_match_v3(DevIndex ix, const u32* __restrict__ words, const uint16_t* __restrict__ lens,
                                                    u64 n_items, int seq_words, int rec_words, int items_per_wave, int refill_min,
                                                    u32* raw, int raw_stride, DevStatsA* st, const u64* __restrict__ item_idx = nullptr) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  constexpr int WAVES = BLOCK / 64;
  constexpr int NM = PAIRED ? 2 : 1;
  const int item_words = rec_words * NM;
  const int lane_words = seq_words * NM;                      // words of an item kept in LDS: the sequence planes only
  const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32* wave_words = lds + (size_t)wv * 64 * lane_words;       // the wavefront's items, lane-transposed
  u32* my_words = wave_words + lane;                          // word j at my_words[j * 64]
  u32* my_list = APPEND ? nullptr : lds + (size_t)WAVES * 64 * lane_words + threadIdx.x;   // entry j at my_list[j * BLOCK]
  const u64 wave_global = (u64)blockIdx.x * WAVES + wv;
  const u64 chunk0 = wave_global * (u64)items_per_wave;
  const u32 chunk_n = chunk0 < n_items ? (u32)min((u64)items_per_wave, n_items - chunk0) : 0u;
  u32 cursor = 0;                                             // next item of the chunk (wavefront-uniform)
  const kamd::Table t = make_table(ix, !PAIRED);
  const int k = ix.k;

  kamd::MatchState ms; ms.phase = kamd::PH_DONE; ms.w = 0; ms.w0 = 0; ms.w2 = 0; ms.dist = 0; ms.nextPos = 0;
  ms.um_uec = ms.um2_uec = kamd::NO_UEC; ms.um_gpos = 0; ms.um_strand = false; ms.text_tried = false; ms.disp = 0;
  kamd::UecList ul{my_list, LCAP, 0, false, APPEND ? 1 : BLOCK};
  ul.append = APPEND;
  kamd::MateFirst mf0{0, 0, -1, false}, mf1{0, 0, -1, false};
  u64 my_item = 0;        // the lane's item in the batch (chunk0 + my_idx, or item_idx[chunk0 + my_idx])
  int mate = 0, len0 = 0, len1 = 0;
  bool n0 = false, n1 = false;   // has-N flags of the two mates
  u32 my_idx = 0;
  bool have = false;      // the lane owns an item whose raw record is not written yet
  bool busy = false;      // ... and its state machine still wants probes
  bool exhausted = false;
  u32 probes = 0, breads = 0, raw_words = 0, text_hits = 0;
  u32 wave_iters = 0, lane_iters = 0;   // loop trips of this wavefront / lanes that probed in them (lane utilisation of the kernel)

  for (;;) {
    // 1. lanes without an item take the next ones of the chunk; their sequence words come by LDS-DMA loads, in flight during
    // the probe below.  Refills are batched (at least refill_min free lanes, or nothing left to probe).
    bool loading = false;
    const u64 idle_mask = __ballot(!have && !exhausted);
    const bool do_refill = idle_mask != 0ULL && (__popcll(idle_mask) >= refill_min || __ballot(have) == 0ULL);
    if (do_refill) {
      if (!have && !exhausted) {
        my_idx = cursor + (u32)__popcll(idle_mask & ((1ULL << lane) - 1ULL));
        if (my_idx >= chunk_n) exhausted = true;
        else {
          loading = true;
          const u64 item = item_idx ? item_idx[chunk0 + my_idx] : chunk0 + my_idx;
          my_item = item;
          const u32* src = words + item * item_words;
#pragma unroll 4
          for (int j = 0; j < seq_words; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j),
                                             (__attribute__((address_space(3))) void*)(wave_words + (size_t)j * 64), 4, 0, 0);
          if (PAIRED) {
#pragma unroll 4
            for (int j = 0; j < seq_words; j++)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + rec_words + j),
                                               (__attribute__((address_space(3))) void*)(wave_words + (size_t)(seq_words + j) * 64), 4, 0, 0);
          }
          len0 = PAIRED ? (int)lens[2 * item] : (int)lens[item];
          len1 = PAIRED ? (int)lens[2 * item + 1] : 0;
        }
      }
      cursor += (u32)__popcll(idle_mask);
    }
    if (__ballot(have || loading) == 0ULL) break;
    ++wave_iters; lane_iters += (u32)__popcll(__ballot(have && busy));
    // 2. every busy lane: ONE memory request -- the unitig text where the window's place on the unitig is known, else a bucket
    // of the table -- all of them issued before any is waited for; a text mismatch or a bucket whose continue flag sends the
    // key to the next bucket costs the lane another iteration, never the wavefront a second round trip
    if (have && busy) {
      const u32* base = my_words + (size_t)(mate ? seq_words : 0) * 64;
      const u32* mplane = words + my_item * (u64)item_words + (size_t)(mate ? rec_words : 0) + seq_words;
      kamd::ReadView rv{base, mplane, mate ? len1 : len0, 64, 1, mate ? n1 : n0};
      bool fc;
      const uint64_t canon = kamd::window_canon(rv, ms.w, k, &fc);
      const bool use_text = TEXT && kamd::text_applies(ms);
      const kamd::Table pt = DL ? kamd::phase_table(t, ms.phase) : t;
      const u32 tpos = use_text ? kamd::text_pos_of(ms) : 0u;
      // (compact layout: the D-list table keeps the wide one, so the phase decides)
      const bool compact = LAYOUT == kamd::LAYOUT_COMPACT && (!DL || ms.phase != kamd::PH_DLIST);
      const u32 khash = kamd::kmer_hash32(canon);
      const uint64_t bucket = kamd::bucket_of_hash(khash, pt.n_buckets) + ms.disp;
      kamd::TextWords tw{0u, 0u, 0u};
      kamd::BucketLine bl{0, 0, 0, 0, 0, 0, 0, 0};
      if (use_text) tw = kamd::load_text(ix.utext, tpos);
      else { bl = kamd::load_bucket(pt.slots, bucket); ++breads; }
      kamd::Probe p; p.found = false; p.strand = false; p.uec = kamd::NO_UEC; p.dist = 0; p.slot = 0; p.gpos = 0;
      bool feed = true;
      if (use_text) {
        if (kamd::text_canon_of(tw, tpos, k) == canon) { p.found = true; p.strand = ms.um_strand; p.uec = ms.um_uec; ++text_hits; }   // (only uec is looked at in these phases)
        else { ms.text_tried = true; feed = false; }
      } else if (compact) {
        if (kamd::match_bucket_compact(bl, pt, kamd::compact_tag(pt, canon, khash, ms.disp), fc, bucket, p) == kamd::BUCKET_CONTINUE &&
            ms.disp < kamd::compact_max_disp(pt)) { ++ms.disp; feed = false; }
      } else if (kamd::match_bucket(bl, canon, fc, bucket, p) == kamd::BUCKET_CONTINUE) { ++ms.disp; feed = false; }
      if (feed) {
        if (!DL || ms.phase != kamd::PH_DLIST) ++probes;   // dbg.find calls of match(); the D-list scan is counted as bucket reads only
        kamd::match_feed<DL>(ms, rv, k, p, ul, mate, mate ? mf1 : mf0, t);
        if (ms.phase == kamd::PH_DONE && PAIRED && mate == 0) {
          mate = 1;
          const u32* b1 = my_words + (size_t)seq_words * 64;
          kamd::ReadView r1{b1, mplane + rec_words, len1, 64, 1, n1};
          kamd::match_init(ms, r1, k);
        }
        busy = ms.phase != kamd::PH_DONE;
      }
    }
    // 3. lanes that fetched an item: its words are in LDS once the DMA loads have landed; start mate 1
    if (loading) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ul.n = 0; ul.overflow = false; ul.last = kamd::NO_UEC; ul.last_flags = 0;
      if (APPEND) ul.e = raw + (chunk0 + my_idx) * (u64)raw_stride + 1;   // the item's classes go straight into its raw record
      mf0 = kamd::MateFirst{0, 0, -1, false}; mf1 = kamd::MateFirst{0, 0, -1, false};
      mate = 0;
      n0 = (my_words[(size_t)(seq_words - 1) * 64] & kamd::REC_FLAG_HAS_N) != 0;
      n1 = PAIRED ? (my_words[(size_t)(2 * seq_words - 1) * 64] & kamd::REC_FLAG_HAS_N) != 0 : false;
      const u32* mplane = words + my_item * (u64)item_words + seq_words;
      kamd::ReadView r0{my_words, mplane, len0, 64, 1, n0};
      kamd::match_init(ms, r0, k);
      if (ms.phase == kamd::PH_DONE && PAIRED) {
        mate = 1;
        const u32* b1 = my_words + (size_t)seq_words * 64;
        kamd::ReadView r1{b1, mplane + rec_words, len1, 64, 1, n1};
        kamd::match_init(ms, r1, k);
      }
      have = true;
      busy = ms.phase != kamd::PH_DONE;
    }
    // 4. finished items: write the raw record (plain stores, nothing waits for them) and free the lane
    if (have && !busy) {
      u32* o = raw + (chunk0 + my_idx) * (u64)raw_stride;
      if constexpr (!APPEND) {
        // (words 0 .. n of the slot in 8-byte stores -- a slot starts at a multiple of 40 bytes --: on average two requests per item instead of three
        // and a half, in a kernel that lives on the memory system's request rate; the word behind the last class is whatever the list held)
        static_assert(LCAP % 2 == 0, "pairs of words");
        uint2* o2 = reinterpret_cast<uint2*>(o);
#pragma unroll
        for (int p = 1; p <= LCAP / 2; p++)
          if (2 * p <= ul.n) o2[p] = make_uint2(my_list[(size_t)(2 * p - 1) * BLOCK], 2 * p < LCAP ? my_list[(size_t)(2 * p) * BLOCK] : 0u);
        o2[0] = make_uint2((u32)ul.n | (ul.overflow ? RAW_OVERFLOW : 0u) | (mf0.n_hits > 0 ? RAW_HIT0 : 0u) | (mf1.n_hits > 0 ? RAW_HIT1 : 0u), my_list[0]);
      } else o[0] = (u32)ul.n | (ul.overflow ? RAW_OVERFLOW : 0u) | (mf0.n_hits > 0 ? RAW_HIT0 : 0u) | (mf1.n_hits > 0 ? RAW_HIT1 : 0u);
      raw_words += 1u + (u32)ul.n;
      if (FILTER) {
        uint2* f2 = reinterpret_cast<uint2*>(o + 2 + LCAP);
        f2[0] = make_uint2((u32)mf0.slot, (u32)(mf0.pos & 0xFFFF) | (mf0.strand ? 0x10000u : 0u));
        f2[1] = make_uint2((u32)mf1.slot, (u32)(mf1.pos & 0xFFFF) | (mf1.strand ? 0x10000u : 0u));
      }
      have = false;
    }
  }
  const u64 s_probes = wave_sum64((u64)probes), s_reads = wave_sum64((u64)breads), s_raw = wave_sum64((u64)raw_words), s_text = wave_sum64((u64)text_hits);
  if (lane == 0) {
    atomicAdd(&st->probes, s_probes); atomicAdd(&st->bucket_reads, s_reads); atomicAdd(&st->raw_words, s_raw);
    if (s_text) atomicAdd(&st->text_hits, s_text);
    atomicAdd(&st->wave_iters, (u64)wave_iters); atomicAdd(&st->lane_iters, (u64)lane_iters);
  }
}

// Persistent blocks (grid-stride over 256-item tiles) so that the per-launch bookkeeping costs a handful of same-address
// atomics per BLOCK instead of per wavefront (a device-scope atomic on one address retires every ~12 ns: half a million
// wavefronts x 5 counters was most of kernel A v1's time).  Each item's slot of the stream is rewritten IN PLACE as a
// tuple record [1, m, e0..] (or [0, ..] when the item is not a tuple), so no stream allocation is needed at all; counts
// of single-set items go through an LDS cache that absorbs the hot sets before touching the dense vector.
__device__ __forceinline__ void wave_lds_fence() {   // the wavefront's LDS writes are visible to its own later reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr int DENSE_CACHE = 2048;
constexpr int CLS_MAX_STRIDE = 2 + V3_LIST_CAP + 4;   // words of a raw record with the positional filters' four
// The slots travel through LDS (end of round 6): a wavefront's 64 slots are one contiguous run of 64 x stride words, read and written back with
// 16-byte accesses (40 lines per wavefront each way).  Before, every thread read and rewrote its own slot word by word -- up to ten 4-byte
// accesses each way at a stride of 40 bytes, every one of them 40 line requests per wavefront -- and the kernel ran at the memory system's request
// rate, not its bandwidth.  The item's set list is built in the slot's own words 2 .. 2 + CAP (the raw classes are in registers by then), so the
// kernel needs no second LDS array; a slot stride of 10 / 14 words is a 2-way bank conflict.  A wavefront synchronises with itself only.
template <bool PAIRED, bool FILTER, int CAP>   // CAP: class entries of a raw record (12: kernel A v2, 8: v3)
__global__ __launch_bounds__(BLOCK) void k_classify(DevIndex ix, u32* __restrict__ slots, int stride, u64 n_items, u64 slot_base,
                                                    u64 rec_base, u64 key_base, u64 item_base, FilterDev fd, AlignOut out) {
  static_assert(2 + CAP + 4 <= CLS_MAX_STRIDE, "slot stride");
  __shared__ __attribute__((aligned(16))) u32 lds_slots[BLOCK * CLS_MAX_STRIDE];
  __shared__ u32 cache_key[DENSE_CACHE];
  __shared__ u32 cache_cnt[DENSE_CACHE];
  __shared__ u32 cache_min[DENSE_CACHE];  // smallest item index (within this launch) that hit the cached set
  __shared__ u32 blk_stats[3];
  for (int i = threadIdx.x; i < DENSE_CACHE; i += BLOCK) { cache_key[i] = 0xFFFFFFFFu; cache_cnt[i] = 0u; cache_min[i] = 0xFFFFFFFFu; }
  if (threadIdx.x < 3) blk_stats[threadIdx.x] = 0u;
  __syncthreads();
  u32 s_single = 0, s_multi = 0, s_proc = 0;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
  u32* ws = lds_slots + (size_t)wv * 64 * stride;
  const bool al16 = ((uintptr_t)slots & 15) == 0;   // (a wavefront's run starts at a multiple of 256 x stride bytes)
  for (u64 tile = blockIdx.x; tile * BLOCK < n_items; tile += gridDim.x) {
    const u64 wave_item0 = tile * BLOCK + (u64)wv * 64;
    if (wave_item0 >= n_items) continue;   // (wavefront-uniform)
    const u32 nwords = (u32)std::min<u64>(64, n_items - wave_item0) * (u32)stride;
    u32* gsl = slots + wave_item0 * (u64)stride;
    if (al16) {
      for (u32 k = (u32)lane * 4; k < nwords; k += 256) {
        if (k + 4 <= nwords) *reinterpret_cast<uint4*>(ws + k) = *reinterpret_cast<const uint4*>(gsl + k);
        else for (u32 j = k; j < nwords; j++) ws[j] = gsl[j];
      }
    } else for (u32 k = (u32)lane; k < nwords; k += 64) ws[k] = gsl[k];
    wave_lds_fence();
    const u64 item = wave_item0 + (u64)lane;
    if (item < n_items) {
    u32* r = ws + lane * stride;
    const u32 h = r[0];
    const int n = (int)(h & 0xFFu);
    u32 uecs[CAP];
#pragma unroll
    for (int j = 0; j < CAP; j++) uecs[j] = j < n ? r[1 + j] : 0u;
    u32 ec[CAP];
#pragma unroll
    for (int j = 0; j < CAP; j++) ec[j] = j < n ? ix.uec_ecn[uecs[j] & 0x3FFFFFFFu] : 0u;   // independent loads, issued together
    kamd::EcList ecs; ecs.e = r + 2; ecs.cap = CAP; ecs.n = 0; ecs.overflow = false;   // (the record's own place: [1, m, e0 ..])
    kamd::MateInfo m0, m1;
    m0.first_slot = m1.first_slot = 0; m0.first_pos = m1.first_pos = -1; m0.first_strand = m1.first_strand = false;
    if (FILTER) {   // (words 2 + CAP ..: behind the list)
      m0.first_slot = r[2 + CAP]; m0.first_pos = (int)(r[3 + CAP] & 0xFFFF); m0.first_strand = (r[3 + CAP] >> 16) & 1u;
      m1.first_slot = r[4 + CAP]; m1.first_pos = (int)(r[5 + CAP] & 0xFFFF); m1.first_strand = (r[5 + CAP] >> 16) & 1u;
    }
    bool ne0 = false, ne1 = false;
#pragma unroll
    for (int j = 0; j < CAP; j++) {
      if (j < n && (ec[j] & 0x80000000u)) {   // the set is non-empty
        const u32 id = ec[j] & kamd::EC_ID_MASK;
        if (uecs[j] & 0x40000000u) ne0 = true;
        if (uecs[j] & 0x80000000u) ne1 = true;
        kamd::eclist_add(ecs, ix.union_mode ? (id | (uecs[j] & 0xC0000000u)) : id);   // --union keeps the mates apart
      }
    }
    ecs.overflow = (h & RAW_OVERFLOW) != 0;
    m0.n_hits = (h & RAW_HIT0) ? 1 : 0; m1.n_hits = (h & RAW_HIT1) ? 1 : 0;
    m0.n_nonempty = ne0; m1.n_nonempty = ne1;
    // classify: 0 unmapped, 1 single set, 2 tuple, 3 overflow, 4 explicit set (positional filter changed it)
    int kind = 0;
    if (ecs.overflow) kind = 3;
    else if (kamd::pair_is_mapped(m0, m1)) kind = ecs.n == 1 ? 1 : 2;
    if (FILTER && (kind == 1 || kind == 2)) {
      u32 kept = 0;
      u32 cur[CAP];
      const int oc = filter_outcome(ix, fd, PAIRED, m0, m1, ecs, &kept, cur);
      if (oc == 1) kind = 0;
      else if (oc == 2) {
        kind = 4;
        const u64 k = atomicAdd(&out.st->n_explicit, 1ULL);
        out.explicit_items[k] = item_base + item;
        atomicAdd(&out.st->exp_words, (u64)kept + 2);
      }
    }
    ++s_proc;
    if (kind == 1) {
      ++s_single;
      const u32 e = ecs.e[0] & kamd::EC_ID_MASK;
      const u32 hh = (e * 2654435761u) >> (32 - 11);
      const u32 old = atomicCAS(&cache_key[hh], 0xFFFFFFFFu, e);
      if (old == 0xFFFFFFFFu || old == e) { atomicAdd(&cache_cnt[hh], 1u); atomicMin(&cache_min[hh], (u32)item); }
      else { atomicAdd(&out.dense_counts[e], 1u); if (out.dense_first) atomicMin(&out.dense_first[e], key_base + item); }
    }
    if (kind == 2) { ++s_multi; r[1] = (u32)ecs.n; }   // (the sets are in place)
    r[0] = kind == 2 ? 1u : 0u;  // record count: 0 = not a tuple record (skipped by the de-duplication)
    // (the slot's place is only looked up for the overflow items -- through rec_off until the second pass redirects it to a long record --: every
    // other record of the batch is found by its item number, absorb_tuples' fixed stride)
    if (kind == 3) { out.rec_off[rec_base + item] = slot_base + item * (u64)stride; u64 i = atomicAdd(&out.st->n_overflow, 1ULL); out.overflow_items[i] = item_base + item; }
    }
    wave_lds_fence();
    if (al16) {
      for (u32 k = (u32)lane * 4; k < nwords; k += 256) {
        if (k + 4 <= nwords) *reinterpret_cast<uint4*>(gsl + k) = *reinterpret_cast<const uint4*>(ws + k);
        else for (u32 j = k; j < nwords; j++) gsl[j] = ws[j];
      }
    } else for (u32 k = (u32)lane; k < nwords; k += 64) gsl[k] = ws[k];
    wave_lds_fence();   // (the next tile's loads overwrite the run)
  }
  // flush: block-level statistics and the cached single-set counts
  const u64 w_single = wave_sum64((u64)s_single), w_multi = wave_sum64((u64)s_multi), w_proc = wave_sum64((u64)s_proc);
  if (lane_id() == 0) { atomicAdd(&blk_stats[0], (u32)w_single); atomicAdd(&blk_stats[1], (u32)w_multi); atomicAdd(&blk_stats[2], (u32)w_proc); }
  __syncthreads();
  for (int i = threadIdx.x; i < DENSE_CACHE; i += BLOCK)
    if (cache_cnt[i]) {
      atomicAdd(&out.dense_counts[cache_key[i]], cache_cnt[i]);
      if (out.dense_first) atomicMin(&out.dense_first[cache_key[i]], key_base + (u64)cache_min[i]);
    }
  if (threadIdx.x == 0) {
    if (blk_stats[0]) atomicAdd(&out.st->st_single, (u64)blk_stats[0]);
    if (blk_stats[1]) atomicAdd(&out.st->st_multi, (u64)blk_stats[1]);
    atomicAdd(&out.st->st_processed, (u64)blk_stats[2]);
  }
}

// items whose hits carried more than TUPLE_CAP distinct sets: same logic, lists in global scratch, reads from HBM
template <bool PAIRED, bool FILTER>
__global__ __launch_bounds__(64) void k_pseudoalign_overflow(DevIndex ix, const u32* __restrict__ words,
                                                             const uint16_t* __restrict__ lens, const u64* items, u64 n,
                                                             int seq_words, int rec_words, u32* scratch, FilterDev fd,
                                                             u64 rec_base, AlignOut out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 item = items[i];
  const int item_words = rec_words * (PAIRED ? 2 : 1);
  // scratch of an item: the list (a sorted array kept by insertion, eclist_add), then as many words for the cursors of the --union merge
  kamd::EcList ecs; ecs.n = 0; ecs.overflow = false;
  ecs.e = scratch + i * (2 * TUPLE_CAP_BIG); ecs.cap = TUPLE_CAP_BIG;
  u32* cur = scratch + i * (2 * TUPLE_CAP_BIG) + TUPLE_CAP_BIG;
  kamd::MateInfo m0, m1; m1.n_hits = 0; m1.n_nonempty = 0;
  const kamd::Table t = make_table(ix, !PAIRED);
  const u32* rec = words + item * item_words;
  kamd::ReadView r0{rec, rec + seq_words, PAIRED ? (int)lens[2 * item] : (int)lens[item]};
  kamd::match_mate(t, ix.uec_ec, ix.ec_nonempty, r0, ix.k, ecs, m0, ix.union_mode ? kamd::EC_MATE1 : 0u);
  if (PAIRED) {
    kamd::ReadView r1{rec + rec_words, rec + rec_words + seq_words, (int)lens[2 * item + 1]};
    kamd::match_mate(t, ix.uec_ec, ix.ec_nonempty, r1, ix.k, ecs, m1, ix.union_mode ? kamd::EC_MATE2 : 0u);
  }
  if (ecs.overflow || !kamd::pair_is_mapped(m0, m1)) return;  // > TUPLE_CAP_BIG distinct sets cannot occur for 16-bit read lengths
  if (FILTER) {
    u32 kept = 0;
    const int oc = filter_outcome(ix, fd, PAIRED, m0, m1, ecs, &kept, cur);
    if (oc == 1) return;
    if (oc == 2) {
      const u64 k = atomicAdd(&out.st->n_explicit_big, 1ULL);
      out.explicit_items_big[k] = item;
      atomicAdd(&out.st->exp_words, (u64)kept + 2);
      return;
    }
  }
  u64 off = atomicAdd(&out.st->stream_words, (u64)ecs.n + 2);
  // v2: the item's own record (index rec_base + item, so record indices stay in input order) is redirected to the big
  // record; v1 (rec_base == ~0): records are appended
  const u64 ridx = rec_base == ~0ULL ? atomicAdd(&out.st->n_recs, 1ULL) : rec_base + item;
  u32* w = out.stream + off;
  w[0] = 1u; w[1] = (u32)ecs.n;
  for (int j = 0; j < ecs.n; j++) w[2 + j] = ecs.e[j];
  out.rec_off[ridx] = off;
  atomicAdd(&out.st->st_multi, 1ULL);
}

// The second pass's raw records (k_match_v3<..., V3_LIST_CAP_LONG, true> over the items whose list overflowed in the first: appended classes,
// duplicates among them unless they were neighbours) -> tuple records, with k_pseudoalign_overflow's tail: the item's own record is
// redirected to a long record appended to the stream.  The de-duplication happens here, in the sorted insertion into the item's set list
// (LDS).  Items whose list overflowed again (more than CAP appended classes) are listed for the straight-line kernel.
constexpr int CL_WAVES = 4, CL_ITEMS = 8;   // k_classify_long: wavefronts per block, items a wavefront takes one after the other
template <bool PAIRED, bool FILTER, int CAP>
__global__ __launch_bounds__(64 * CL_WAVES) void k_classify_long(DevIndex ix, const u32* __restrict__ raw, int stride, const u64* __restrict__ items, u64 n, u32* scratch,
                                                                 FilterDev fd, u64 rec_base, AlignOut out, u64* items_left) {
  // One WAVEFRONT per item (round 6; a thread per item kept a sorted list by insertion -- 64 divergent insertion sorts per wavefront -- and paid two
  // same-address atomics per item: 8.9 ms for the 440 k such items of 8 M stress pairs).  The item's appended classes (<= CAP = 3 per lane, with
  // duplicates) are mapped to set ids, compacted, and ranked by counting: an entry is kept if no earlier entry has its id (the mate flags of all
  // entries with that id are merged into it), and its place in the sorted record is the number of kept entries with a smaller id -- every lane
  // reads the list out of LDS by broadcast, no sort network.  A wavefront takes CL_ITEMS items and allocates their records with ONE atomic.
  static_assert(CAP <= 192, "three entries per lane");
  __shared__ u32 s_in_all[CL_WAVES][CAP];
  __shared__ u32 s_out_all[CL_WAVES][CL_ITEMS][CAP];
  __shared__ u32 s_n_all[CL_WAVES][CL_ITEMS];   // sets of the item's record; 0 = no record
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
  u32* s_in = s_in_all[w];
  const u64 i0 = ((u64)blockIdx.x * CL_WAVES + w) * CL_ITEMS;
  constexpr u32 DUP = 0xFFFFFFFFu;
  u32 total_words = 0, n_records = 0;
  // every load of the wavefront's items first (header, appended classes, their sets: three dependent rounds for all CL_ITEMS items together instead of
  // one after the other per item -- the kernel is a few thousand wavefronts waiting on exactly these)
  u32 hq[CL_ITEMS], uq[CL_ITEMS][3], eq[CL_ITEMS][3];
#pragma unroll
  for (int q = 0; q < CL_ITEMS; q++) hq[q] = i0 + q < n ? raw[(i0 + q) * (u64)stride] : RAW_OVERFLOW;
#pragma unroll
  for (int q = 0; q < CL_ITEMS; q++) {
    const u32 nc = (hq[q] & RAW_OVERFLOW) ? 0u : (hq[q] & 0xFFu);
#pragma unroll
    for (int t = 0; t < 3; t++) { const u32 j = (u32)lane + 64u * t; uq[q][t] = j < nc ? raw[(i0 + q) * (u64)stride + 1 + j] : 0u; }
  }
#pragma unroll
  for (int q = 0; q < CL_ITEMS; q++) {
    const u32 nc = (hq[q] & RAW_OVERFLOW) ? 0u : (hq[q] & 0xFFu);
#pragma unroll
    for (int t = 0; t < 3; t++) { const u32 j = (u32)lane + 64u * t; eq[q][t] = j < nc ? ix.uec_ecn[uq[q][t] & 0x3FFFFFFFu] : 0u; }
  }
#pragma unroll
  for (int q = 0; q < CL_ITEMS; q++) {
    const u64 i = i0 + q;
    if (lane == 0) s_n_all[w][q] = 0;
    if (i >= n) continue;
    const u64 item = items[i];
    const u32* r = raw + i * (u64)stride;
    const u32 h = hq[q];
    if (h & RAW_OVERFLOW) { if (lane == 0) { const u64 k = atomicAdd(&out.st->n_overflow, 1ULL); items_left[k] = item; } continue; }
    const u32 nc = h & 0xFFu;
    // classes -> non-empty sets, compacted into s_in (id | mate flags with --union)
    u32 nv = 0;
    bool ne0 = false, ne1 = false;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const u32 j = (u32)lane + 64u * t;
      u32 v = 0; bool valid = false;
      if (j < nc) {
        const u32 uec = uq[q][t];
        const u32 ec = eq[q][t];
        valid = (ec & 0x80000000u) != 0;   // the set is non-empty
        v = (ec & kamd::EC_ID_MASK) | (ix.union_mode ? (uec & 0xC0000000u) : 0u);
        ne0 = ne0 || (valid && (uec & 0x40000000u)); ne1 = ne1 || (valid && (uec & 0x80000000u));
      }
      const u64 bm = __ballot(valid);
      if (valid) s_in[nv + __popcll(bm & ((1ULL << lane) - 1ULL))] = v;
      nv += (u32)__popcll(bm);
    }
    ne0 = __ballot(ne0) != 0ULL; ne1 = __ballot(ne1) != 0ULL;
    wave_lds_fence();
    // pass 1: first occurrences and merged flags
    u32 x[3], fl[3]; bool first[3];
#pragma unroll
    for (int t = 0; t < 3; t++) { const u32 j = (u32)lane + 64u * t; x[t] = j < nv ? s_in[j] : DUP; fl[t] = 0; first[t] = j < nv; }
    for (u32 k = 0; k < nv; k++) {
      const u32 y = s_in[k], yid = y & kamd::EC_ID_MASK;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        if (yid == (x[t] & kamd::EC_ID_MASK) && x[t] != DUP) { fl[t] |= y & 0xC0000000u; if (k < (u32)lane + 64u * t) first[t] = false; }
      }
    }
    wave_lds_fence();
#pragma unroll
    for (int t = 0; t < 3; t++) { const u32 j = (u32)lane + 64u * t; if (j < nv) s_in[j] = first[t] ? ((x[t] & kamd::EC_ID_MASK) | fl[t]) : DUP; }
    wave_lds_fence();
    // pass 2: the place of a kept entry = kept entries with a smaller id
    u32 rank[3] = {0, 0, 0};
    for (u32 k = 0; k < nv; k++) {
      const u32 y = s_in[k];
      if (y == DUP) continue;   // (wavefront-uniform: a broadcast read)
      const u32 yid = y & kamd::EC_ID_MASK;
#pragma unroll
      for (int t = 0; t < 3; t++) rank[t] += yid < (x[t] & kamd::EC_ID_MASK) ? 1u : 0u;
    }
    u32 nd = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const bool keep = first[t] && x[t] != DUP;
      if (keep) s_out_all[w][q][rank[t]] = (x[t] & kamd::EC_ID_MASK) | fl[t];
      nd += (u32)__popcll(__ballot(keep));
    }
    wave_lds_fence();
    kamd::MateInfo m0, m1;
    m0.first_slot = m1.first_slot = 0; m0.first_pos = m1.first_pos = -1; m0.first_strand = m1.first_strand = false;
    m0.n_hits = (h & RAW_HIT0) ? 1 : 0; m1.n_hits = (h & RAW_HIT1) ? 1 : 0;
    m0.n_nonempty = ne0; m1.n_nonempty = ne1;
    if (FILTER) {
      m0.first_slot = r[2 + CAP]; m0.first_pos = (int)(r[3 + CAP] & 0xFFFF); m0.first_strand = (r[3 + CAP] >> 16) & 1u;
      m1.first_slot = r[4 + CAP]; m1.first_pos = (int)(r[5 + CAP] & 0xFFFF); m1.first_strand = (r[5 + CAP] >> 16) & 1u;
    }
    if (!kamd::pair_is_mapped(m0, m1)) continue;
    if (FILTER) {   // (the positional filters walk the set thread-serially: lane 0, as in the straight-line kernel)
      int oc = 0;
      if (lane == 0) {
        kamd::EcList ecs; ecs.e = s_out_all[w][q]; ecs.cap = CAP; ecs.n = (int)nd; ecs.overflow = false;
        u32 kept = 0;
        u32* cur = scratch + i * (u64)(2 * TUPLE_CAP_BIG) + TUPLE_CAP_BIG;
        oc = filter_outcome(ix, fd, PAIRED, m0, m1, ecs, &kept, cur);
        if (oc == 2) {
          const u64 k = atomicAdd(&out.st->n_explicit_big, 1ULL);
          out.explicit_items_big[k] = item;
          atomicAdd(&out.st->exp_words, (u64)kept + 2);
        }
      }
      oc = __shfl(oc, 0, 64);
      if (oc != 0) continue;
    }
    if (lane == 0) s_n_all[w][q] = nd;
    total_words += nd + 2; ++n_records;
  }
  if (n_records == 0) return;
  u64 off = 0;
  if (lane == 0) { off = atomicAdd(&out.st->stream_words, (u64)total_words); atomicAdd(&out.st->st_multi, (u64)n_records); }
  off = shfl_u64(off, 0);
  wave_lds_fence();
  for (int q = 0; q < CL_ITEMS; q++) {
    const u32 nd = s_n_all[w][q];
    if (nd == 0) continue;
    u32* wr = out.stream + off;
    if (lane == 0) { wr[0] = 1u; wr[1] = nd; out.rec_off[rec_base + items[i0 + q]] = off; }
    for (u32 j = lane; j < nd; j += 64) wr[2 + j] = s_out_all[w][q][j];
    off += nd + 2;
  }
}

// items whose transcript set was changed by a positional filter: write the filtered set as an explicit record
// [1, n, t0..t(n-1)] (same format as the candidate stream of the finalize step)
template <bool PAIRED>
__global__ __launch_bounds__(64) void k_explicit_write(DevIndex ix, const u32* __restrict__ words, const uint16_t* __restrict__ lens,
                                                       const u64* items, u64 n, int seq_words, int rec_words, u32* scratch,
                                                       int cap, FilterDev fd, u32* exp_stream, u64* exp_off, u64* exp_key, u64 key_base,
                                                       u64 key_stride, DevState* st) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 item = items[i];
  const int item_words = rec_words * (PAIRED ? 2 : 1);
  // scratch of an item: the list (cap words), the cursors of the --union merge (cap), the distinct block / strand pairs of
  // mate 1's hits for the per-hit strand filter (EXPLICIT_HITS)
  u32* sbase = scratch + i * (u64)(2 * cap + EXPLICIT_HITS);
  kamd::EcList ecs; ecs.e = sbase; ecs.cap = cap; ecs.n = 0; ecs.overflow = false;
  u32* cur = sbase + cap;
  kamd::HitBlocks hb{ix.slot_block, sbase + 2 * cap, EXPLICIT_HITS, 0, false};
  kamd::MateInfo m0, m1; m1.n_hits = 0; m1.n_nonempty = 0; m1.first_slot = 0; m1.first_pos = -1; m1.first_strand = false;
  const kamd::Table t = make_table(ix, !PAIRED);
  const u32* rec = words + item * item_words;
  kamd::ReadView r0{rec, rec + seq_words, PAIRED ? (int)lens[2 * item] : (int)lens[item]};
  kamd::match_mate(t, ix.uec_ec, ix.ec_nonempty, r0, ix.k, ecs, m0, ix.union_mode ? kamd::EC_MATE1 : 0u,
                   (fd.comprehensive && fd.strand) ? &hb : nullptr);   // (only looked at when mate 2 has no hits)
  if (PAIRED) {
    kamd::ReadView r1{rec + rec_words, rec + rec_words + seq_words, (int)lens[2 * item + 1]};
    kamd::match_mate(t, ix.uec_ec, ix.ec_nonempty, r1, ix.k, ecs, m1, ix.union_mode ? kamd::EC_MATE2 : 0u);
  }
  kamd::FilterCfg cfg = item_filter_cfg(fd, PAIRED, m0, m1);
  cfg.hits1 = hb.e; cfg.n_hits1 = hb.n;
  const kamd::PosTables pt = pos_tables(ix);
  const kamd::FirstHit h0 = first_hit(ix, m0), h1 = first_hit(ix, m1);
  u32 keep = 0;
  if (!hb.overflow) for_each_member(ix, ecs, cur, [&](u32 tr) { keep += kamd::keep_transcript(pt, cfg, h0, h1, tr) ? 1u : 0u; });
  else atomicAdd(&st->n_hit_overflow, 1ULL);   // (reported as an error by the caller: EXPLICIT_HITS distinct blocks per read is far beyond real data)
  const u64 off = atomicAdd(&st->cand_words, (u64)keep + 2);   // cursor of this pass (exp_words holds the bound)
  const u64 r = atomicAdd(&st->exp_recs, 1ULL);
  u32* w = exp_stream + off;
  w[0] = keep ? 1u : 0u; w[1] = keep;   // count 0: the filters left nothing (the record is skipped downstream)
  u32 o = 0;
  for_each_member(ix, ecs, cur, [&](u32 tr) { if (keep && kamd::keep_transcript(pt, cfg, h0, h1, tr)) w[2 + o++] = tr; });
  exp_off[r] = off;
  if (exp_key) exp_key[r] = key_base + item * key_stride;  // position of the item in the input (first-occurrence order)
}

// ------------------------------------------------------------------------------------------------------------------
// FLD probe kernel: per item the fragment length KmerIndex::mapPair would return and |u| (first items only)
// ------------------------------------------------------------------------------------------------------------------
constexpr u32 FLD_OVERFLOW = 0xFFFFFFFFu;
// Phase 1 of the sample: KmerIndex::mapPair's own test and nothing else -- the first present k-mer of either mate by a linear
// scan (KmerIndex.cpp:1636-1668; it is also match()'s first hit), same block, opposite strands, 0 < tl < MAX_FRAG_LEN.  One or
// two probes per mate, no class lists.  Only the pairs that pass (a few per cent) go on to k_fld, which adds |u| == 1.
__global__ __launch_bounds__(BLOCK) void k_fld_first(DevIndex ix, const u32* __restrict__ words, const uint16_t* __restrict__ lens, u64 n_items,
                                                     int seq_words, int rec_words, int32_t* tl_out, u32* card_out, u64* cand, u32* n_cand) {
  const u64 item = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  bool pass = false;
  if (item < n_items) {
    const kamd::Table t = make_table(ix, false);
    const u32* rec = words + item * (u64)(rec_words * 2);
    u64 slot[2] = {0, 0}; int pos[2] = {0, 0}; bool strand[2] = {false, false}, found[2] = {false, false};
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const u32* rm = rec + m * rec_words;
      kamd::ReadView r{rm, rm + seq_words, (int)lens[2 * item + m]};
      r.has_n = (rm[seq_words - 1] & kamd::REC_FLAG_HAS_N) != 0;
      int w = kamd::next_valid_window(r, 0, ix.k);
      while (w >= 0) {
        bool fc; const u64 canon = kamd::window_canon(r, w, ix.k, &fc);
        const kamd::Probe p = kamd::probe_table(t, canon, fc, nullptr);
        if (p.found) { slot[m] = p.slot; pos[m] = w; strand[m] = p.strand; found[m] = true; break; }
        w = kamd::next_valid_window(r, w + 1, ix.k);
      }
      if (!found[m]) break;
    }
    int32_t tl = -1;
    if (found[0] && found[1] && strand[0] != strand[1] && ix.slot_block[slot[0]] == ix.slot_block[slot[1]]) {
      const int d0 = (int)ix.slot_dist[slot[0]], d1 = (int)ix.slot_dist[slot[1]];
      const int p1 = strand[0] ? d0 - pos[0] : d0 + ix.k + pos[0];
      const int p2 = strand[1] ? d1 - pos[1] : d1 + ix.k + pos[1];
      tl = p1 > p2 ? p1 - p2 : p2 - p1;
    }
    pass = tl > 0 && tl < KAMD_MAX_FRAG_LEN;
    tl_out[item] = pass ? tl : -1; card_out[item] = 0;
  }
  const u64 bal = __ballot(pass);
  if (bal) {
    u32 base = 0;
    if (lane_id() == 0) base = atomicAdd(n_cand, (u32)__popcll(bal));
    base = __shfl(base, 0, 64);
    if (pass) cand[base + __popcll(bal & ((1ULL << lane_id()) - 1))] = item;
  }
}
// (scratch == nullptr: the per-item list of distinct transcript sets lives in LDS, TUPLE_CAP entries; an item that needs more
// is reported as FLD_OVERFLOW and re-run with a TUPLE_CAP_BIG list in global memory)
__global__ __launch_bounds__(BLOCK) void k_fld(DevIndex ix, const u32* __restrict__ words, const uint16_t* __restrict__ lens,
                                               const u64* __restrict__ items, u64 n_items, int seq_words, int rec_words,
                                               u32* scratch, int cap, FilterDev fd, int32_t* tl_out, u32* card_out,
                                               const u32* __restrict__ n_items_dev = nullptr) {
  __shared__ u32 lds_list[BLOCK * TUPLE_CAP];
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_items_dev) n_items = *n_items_dev;   // the candidate list of k_fld_first: its length never visits the host
  if (i >= n_items) return;
  const u64 item = items ? items[i] : i;
  const int item_words = rec_words * 2;
  kamd::EcList ecs; ecs.n = 0; ecs.overflow = false;
  u32 cur_small[TUPLE_CAP];
  u32* cur = cur_small;   // cursors of the --union merge: behind the list in the item's scratch (2 x cap words per item)
  if (scratch) { ecs.e = scratch + i * (u64)(2 * cap); ecs.cap = cap; cur = ecs.e + cap; } else { ecs.e = lds_list + threadIdx.x * TUPLE_CAP; ecs.cap = TUPLE_CAP; }
  kamd::MateInfo m0, m1;
  const kamd::Table t = make_table(ix, false);
  const u32* rec = words + item * item_words;
  kamd::ReadView r0{rec, rec + seq_words, (int)lens[2 * item]};
  kamd::ReadView r1{rec + rec_words, rec + rec_words + seq_words, (int)lens[2 * item + 1]};
  kamd::match_mate(t, ix.uec_ec, ix.ec_nonempty, r0, ix.k, ecs, m0, ix.union_mode ? kamd::EC_MATE1 : 0u);
  kamd::match_mate(t, ix.uec_ec, ix.ec_nonempty, r1, ix.k, ecs, m1, ix.union_mode ? kamd::EC_MATE2 : 0u);
  int32_t tl = -1; u32 card = 0;
  if (!ecs.overflow && kamd::pair_is_mapped(m0, m1)) {
    // |u| after the strand filter (the fragment-length filter cannot be active while the FLD is being estimated)
    {
      const kamd::FilterCfg cfg = item_filter_cfg(fd, true, m0, m1);
      const kamd::PosTables pt = pos_tables(ix);
      const kamd::FirstHit h0 = first_hit(ix, m0), h1 = first_hit(ix, m1);
      // (with the per-hit strand filter only mate 2's first mapping k-mer matters whenever it has hits -- and a pair
      // without hits on mate 2 has no fragment length)
      for_each_member(ix, ecs, cur, [&](u32 tr) {
        card += (!(cfg.fraglen || cfg.strand) || kamd::keep_transcript(pt, cfg, h0, h1, tr)) ? 1u : 0u;
      });
    }
    if (m0.n_hits > 0 && m1.n_hits > 0) {  // KmerIndex::mapPair (KmerIndex.cpp:1622-1693) on the first present k-mers
      const u32 b0 = ix.slot_block[m0.first_slot], b1 = ix.slot_block[m1.first_slot];
      if (b0 == b1 && (m0.first_strand != m1.first_strand)) {  // same unitig + same set + same block end <=> same block
        const int d0 = (int)ix.slot_dist[m0.first_slot], d1 = (int)ix.slot_dist[m1.first_slot];
        const int p1 = m0.first_strand ? d0 - m0.first_pos : d0 + ix.k + m0.first_pos;
        const int p2 = m1.first_strand ? d1 - m1.first_pos : d1 + ix.k + m1.first_pos;
        tl = p1 > p2 ? p1 - p2 : p2 - p1;
      }
    }
  }
  if (ecs.overflow) {   // re-run by the host with a larger list (counted next to the candidate count, so the host need not search)
    card = FLD_OVERFLOW;
    if (n_items_dev) atomicAdd(const_cast<u32*>(n_items_dev) + 1, 1u);
  }
  tl_out[item] = tl; card_out[item] = card;
}

// The sample in input order, on the device: out[r] = fragment length of the r-th qualifying pair (|u| == 1, 0 < tl < MAX_FRAG_LEN)
// of the prefix, r < want; head[2] = qualifying pairs of the prefix.  The host then reads 40 KB instead of searching two 4 MB
// vectors.  Three small launches (count per block of 8192 items, scan of the block counts, emit): a single block walking the
// prefix was starved by kernel A, which runs at the same time (5.6 ms for 65 trips).
constexpr int FLD_RANK_BLOCK = 1024, FLD_RANK_PER = 8;   // 8192 items per block
__device__ __forceinline__ u32 fld_rank_load(const int32_t* __restrict__ tl, const u32* __restrict__ card, u64 n, u64 i0, int32_t* t) {
  u32 mine = 0;
#pragma unroll
  for (int j = 0; j < FLD_RANK_PER; j++) {
    const u64 i = i0 + j;
    t[j] = (i < n && card[i] == 1u) ? tl[i] : -1;
    if (!(t[j] > 0 && t[j] < KAMD_MAX_FRAG_LEN)) t[j] = -1;
    mine += t[j] > 0;
  }
  return mine;
}
__global__ __launch_bounds__(FLD_RANK_BLOCK) void k_fld_count(const int32_t* __restrict__ tl, const u32* __restrict__ card, u64 n, u32* blk) {
  __shared__ u32 tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  int32_t t[FLD_RANK_PER];
  const u32 mine = fld_rank_load(tl, card, n, ((u64)blockIdx.x * FLD_RANK_BLOCK + threadIdx.x) * FLD_RANK_PER, t);
  const u32 w = (u32)wave_sum64(mine);
  if (lane_id() == 0 && w) atomicAdd(&tot, w);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}
__global__ __launch_bounds__(FLD_RANK_BLOCK) void k_fld_scan(u32* blk, u32 n_blk, u32* head) {   // one block; n_blk <= a few hundred
  __shared__ u32 wsum[FLD_RANK_BLOCK / 64];
  __shared__ u32 carry_s;
  const int tid = (int)threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (u32 b0 = 0; b0 < n_blk; b0 += FLD_RANK_BLOCK) {
    const u32 i = b0 + (u32)tid;
    const u32 v = i < n_blk ? blk[i] : 0;
    const u32 incl = wave_incl_scan(v);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u32 before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < FLD_RANK_BLOCK / 64; k++) { const u32 x = wsum[k]; total += x; if (k < wv) before += x; }
    const u32 carry = carry_s;
    if (i < n_blk) blk[i] = carry + before + incl - v;
    __syncthreads();
    if (tid == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (tid == 0) head[2] = carry_s;
}
__global__ __launch_bounds__(FLD_RANK_BLOCK) void k_fld_emit(const int32_t* __restrict__ tl, const u32* __restrict__ card, u64 n, u32 want,
                                                            const u32* __restrict__ blk, int32_t* out) {
  __shared__ u32 wsum[FLD_RANK_BLOCK / 64];
  const u32 base = blk[blockIdx.x];
  if (base >= want) return;
  const int tid = (int)threadIdx.x, wv = tid >> 6, lane = tid & 63;
  int32_t t[FLD_RANK_PER];
  const u32 mine = fld_rank_load(tl, card, n, ((u64)blockIdx.x * FLD_RANK_BLOCK + threadIdx.x) * FLD_RANK_PER, t);
  const u32 incl = wave_incl_scan(mine);
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  u32 before = 0;
#pragma unroll
  for (int k = 0; k < FLD_RANK_BLOCK / 64; k++) if (k < wv) before += wsum[k];
  u32 r = base + before + incl - mine;
#pragma unroll
  for (int j = 0; j < FLD_RANK_PER; j++) if (t[j] > 0) { if (r < want) out[r] = t[j]; ++r; }
}

}  // namespace

namespace {
// items whose reads do not fit the LDS-resident kernel: every item is flagged for the overflow kernel, which reads from HBM
__global__ void k_mark_overflow(u32* raw, int raw_stride, u64 n_items) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_items) raw[i * (u64)raw_stride] = RAW_OVERFLOW;
}
// Kernel A of items [first, first + n) of the batch, on stream s: raw records into the items' slots
template <bool PAIRED, bool FILTER>
int launch_match_chunk(kamd_ctx* c, hipStream_t s, const u32* d_words, const uint16_t* d_len, u64 first, u64 n, int seq_words, int rec_words, u32* slots,
                       int stride) {
  constexpr int WAVES = BLOCK / 64;
  constexpr int NM = PAIRED ? 2 : 1;
  const int lane_words = seq_words * NM;
  size_t lds_bytes = ((size_t)WAVES * 64 * lane_words + (size_t)BLOCK * V3_LIST_CAP) * sizeof(u32);
  // diagnostic: unused LDS per block lowers the number of resident wavefronts (occupancy sensitivity; room for another stream's kernels)
  if (c->tune.lds_pad > 0 && lds_bytes <= 64 * 1024) lds_bytes = std::min<size_t>(64 * 1024, lds_bytes + (size_t)c->tune.lds_pad);
  const u64 n_waves = (n + c->items_per_wave - 1) / c->items_per_wave;
  const u32* w = d_words + first * (u64)(rec_words * NM);
  const uint16_t* l = d_len + first * NM;
  u32* raw = slots + first * (u64)stride;
  if (lds_bytes > 64 * 1024) {
    // reads of more than ~480 bases (pairs) / ~980 (single): the kernel would hold too few wavefronts per CU (or none: the
    // CU has 160 KB) -- the reference has no length limit, so such batches take the HBM-resident path item by item
    hipLaunchKernelGGL(k_mark_overflow, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, raw, stride, n);
  } else {
#define KAMD_LAUNCH_V3L(DLV, TXT, LAY)                                                                                                   \
  do {                                                                                                                                  \
    HIPC(hipFuncSetAttribute((const void*)k_match_v3<PAIRED, FILTER, DLV, TXT, LAY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
    hipLaunchKernelGGL((k_match_v3<PAIRED, FILTER, DLV, TXT, LAY>), dim3(grid_for(n_waves, WAVES)), dim3(BLOCK), lds_bytes, s, c->ix, w, l, n, seq_words, \
                       rec_words, c->items_per_wave, c->refill_min, raw, stride, c->stats_a.as<DevStatsA>());                           \
  } while (0)
#define KAMD_LAUNCH_V3(DLV, TXT)                                                                                                         \
  do { if (c->ix.table_layout == kamd::LAYOUT_COMPACT) KAMD_LAUNCH_V3L(DLV, TXT, kamd::LAYOUT_COMPACT); else KAMD_LAUNCH_V3L(DLV, TXT, kamd::LAYOUT_WIDE); } while (0)
    const bool dl = c->ix.n_dbuckets != 0, txt = c->tune.text_verify == 1;
    if (dl) { if (txt) KAMD_LAUNCH_V3(true, true); else KAMD_LAUNCH_V3(true, false); }
    else { if (txt) KAMD_LAUNCH_V3(false, true); else KAMD_LAUNCH_V3(false, false); }
#undef KAMD_LAUNCH_V3
#undef KAMD_LAUNCH_V3L
  }
  HIPC(hipGetLastError());
  return 0;
}
// While a batch is processed the context's work stream may be a side stream: kernel A's chunks run on the caller's stream, everything
// that follows a chunk (classification, de-duplication, and the host synchronisations between them) on the side stream, so that
// chunk k + 1 is matched while chunk k is classified and absorbed -- kernel A is bound by memory requests, the rest by atomics and
// dependent gathers, and the two overlap well.  Leaving the scope joins the side stream back into the caller's.
struct WorkStream {
  kamd_ctx* c; hipStream_t user; bool swapped = false;
  explicit WorkStream(kamd_ctx* ctx) : c(ctx), user(ctx->stream) {}
  int fork() {
    if (!c->al_stream) {
      HIPC(hipStreamCreateWithFlags(&c->al_stream, hipStreamNonBlocking));
      HIPC(hipEventCreateWithFlags(&c->al_ev_in, hipEventDisableTiming));
      HIPC(hipEventCreateWithFlags(&c->al_ev_out, hipEventDisableTiming));
    }
    HIPC(hipEventRecord(c->al_ev_in, user));
    HIPC(hipStreamWaitEvent(c->al_stream, c->al_ev_in, 0));
    c->stream = c->al_stream; swapped = true;
    return 0;
  }
  ~WorkStream() {
    if (!swapped) return;
    (void)hipEventRecord(c->al_ev_out, c->al_stream);
    (void)hipStreamWaitEvent(user, c->al_ev_out, 0);
    c->stream = user;
  }
};
// One batch: kernel A in `chunks` launches on the caller's stream, every chunk classified and its tuple records absorbed as soon as
// it is matched.  On return every item that is neither an overflow item nor one a positional filter changed is accounted for.
int fld_launch(kamd_ctx* c, const FilterDev& fd, const u32* w, const uint16_t* l, u64 n, int seq_words, int rec_words, hipStream_t s);
template <bool PAIRED, bool FILTER>
int overflow_side_launch(kamd_ctx* c, const u32* d_words, const uint16_t* d_len, u64 nov, int seq_words, int rec_words, const FilterDev& fd, AlignOut& out);
template <bool PAIRED, bool FILTER>
int align_batch(kamd_ctx* c, WorkStream& ws, const u32* d_words, const uint16_t* d_len, u64 n_items, int seq_words, int rec_words, const FilterDev& fd,
                AlignOut& out, u64 key_base) {
  const int stride = 2 + V3_LIST_CAP + (FILTER ? 4 : 0);
  // every item owns a fixed slot of the batch's record stream: raw record from k_match_v3, rewritten in place by k_classify.  The
  // stream belongs to this batch only (absorb_tuples moves what is new into the tuple store)
  if (int rc = c->stream_buf.ensure(n_items * (u64)stride * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->rec_off.ensure(n_items * sizeof(u64), 0, c->stream)) return rc;
  out.stream = c->stream_buf.as<u32>(); out.rec_off = c->rec_off.as<u64>();
  c->host_state.stream_words = n_items * (u64)stride;
  c->host_state.n_recs = n_items;
  if (int rc = push_state(c)) return rc;
  u32* slots = c->stream_buf.as<u32>();
  // (one launch by default: measured on config #3, 4 / 8 chunks overlap perfectly and gain nothing -- kernel A, k_classify and
  // k_tup_absorb all live on the memory system's rate of random requests, so side by side they only share it; profiles/README.md)
  int chunks = c->tune.align_chunks > 0 ? c->tune.align_chunks : 1;
  chunks = (int)std::min<u64>((u64)std::min(chunks, 64), std::max<u64>(1, n_items / 65536));
  const u64 per = (n_items + chunks - 1) / chunks;
  if (chunks > 1) {
    if (c->al_ev_chunk.size() < (size_t)chunks) { const size_t o = c->al_ev_chunk.size(); c->al_ev_chunk.resize((size_t)chunks, nullptr); for (size_t i = o; i < c->al_ev_chunk.size(); i++) HIPC(hipEventCreateWithFlags(&c->al_ev_chunk[i], hipEventDisableTiming)); }
    if (int rc = ws.fork()) return rc;   // (behind push_state and whatever produced the reads on the caller's stream)
  }
  HIPC(hipEventRecord(c->ev0, ws.user));
  for (int k = 0; k < chunks; k++) {
    const u64 first = (u64)k * per, n = std::min(per, n_items - first);
    if (int rc = launch_match_chunk<PAIRED, FILTER>(c, ws.user, d_words, d_len, first, n, seq_words, rec_words, slots, stride)) return rc;
    if (chunks > 1) HIPC(hipEventRecord(c->al_ev_chunk[k], ws.user));
  }
  HIPC(hipEventRecord(c->ev1, ws.user));
  // (another batch, or the same buffers with other options -- ring buffers are reused --: stale.  ADVICE r4)
  if (c->fld_deferred.valid && !(c->fld_deferred.w == d_words && c->fld_deferred.l == d_len && c->fld_deferred.n <= n_items &&
                                 (c->fld_deferred.max_len + 15) / 16 + 1 == seq_words && c->fld_deferred.strand == fd.strand &&
                                 c->fld_deferred.so == fd.single_overhang && c->fld_deferred.comp == c->ix.comprehensive)) c->fld_deferred.valid = false;
  if (c->fld_deferred.valid) {
    // the deferred fragment-length prefetch of this batch: its kernels start when kernel A has finished and run beside what follows
    c->fld_deferred.valid = false;
    const auto& q = c->fld_deferred;
    const FilterDev ffd{q.so, 0, 0, q.strand, q.comp};
    HIPC(hipStreamWaitEvent(c->fld_stream, c->ev1, 0));
    if (int rc = fld_launch(c, ffd, q.w, q.l, q.n, (q.max_len + 15) / 16 + 1, (int)kamd_packed_record_words(q.max_len), c->fld_stream)) return rc;
    HIPC(hipEventRecord(c->fld_ev, c->fld_stream));
    c->fld_pending.w = q.w; c->fld_pending.l = q.l; c->fld_pending.n = q.n; c->fld_pending.max_len = q.max_len;
    c->fld_pending.strand = q.strand; c->fld_pending.so = q.so; c->fld_pending.valid = true;
  }
  c->last_classify_ms = 0.f;
  for (int k = 0; k < chunks; k++) {
    const u64 first = (u64)k * per, n = std::min(per, n_items - first);
    if (chunks > 1) HIPC(hipStreamWaitEvent(c->stream, c->al_ev_chunk[k], 0));
    HIPC(hipEventRecord(c->ev2, c->stream));
    // (persistent blocks: as many as are resident at once -- 38 KB of LDS each, four per CU -- so that none starts when the others are done)
    const unsigned grid = (unsigned)std::min<u64>(grid_for(n, BLOCK), (u64)std::max(1, c->n_cus) * 4);
    hipLaunchKernelGGL((k_classify<PAIRED, FILTER, V3_LIST_CAP>), dim3(grid), dim3(BLOCK), 0, c->stream, c->ix, slots + first * (u64)stride, stride, n,
                       first * (u64)stride, first, key_base + first, first, fd, out);
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(c->ev3, c->stream));
    if (int rc = sync_state(c)) return rc;
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, c->ev2, c->ev3));
    c->last_classify_ms += ms;
    // (one chunk: every overflow item of the batch is known -- their second pass starts here, beside the absorption; kamd_pseudoalign joins it)
    if (chunks == 1 && c->host_state.n_overflow && c->tune.overflow_second_pass == 1 && !FILTER && !c->ix.union_mode) {
      const int sr = overflow_side_launch<PAIRED, FILTER>(c, d_words, d_len, c->host_state.n_overflow, seq_words, rec_words, fd, out);
      if (sr < 0) return sr;
    }
    // the chunk's tuple records join the distinct tuples of the run (overflow items have no tuple record yet: see kamd_pseudoalign)
    if (int rc = absorb_tuples(c, c->stream_buf.as<u32>(), c->rec_off.as<u64>() + first, n, c->host_state.stream_words, key_base + first,
                               c->host_state.st_multi - c->multi_before, nullptr, (u32)stride, first)) return rc;
    c->multi_before = c->host_state.st_multi;
  }
  HIPC(hipEventSynchronize(c->ev1));
  HIPC(hipEventElapsedTime(&c->last_align_ms, c->ev0, c->ev1));
  return 0;
}
template <bool PAIRED, bool FILTER>
void launch_overflow(kamd_ctx* c, const u32* d_words, const uint16_t* d_len, u64 nov, int seq_words, int rec_words, const FilterDev& fd,
                     u64 rec_base, const AlignOut& out, const u64* items = nullptr) {
  hipLaunchKernelGGL((k_pseudoalign_overflow<PAIRED, FILTER>), dim3(grid_for(nov, 64)), dim3(64), 0, c->stream, c->ix, d_words, d_len,
                     items ? items : (const u64*)c->overflow_items.as<u64>(), nov, seq_words, rec_words, c->overflow_scratch.as<u32>(), fd, rec_base, out);
}
// The second pass of kernel A over the items whose class list overflowed: the data-flow matcher with an append-only list of up to
// V3_LIST_CAP_LONG classes per item in global memory (no list in LDS: kernel A's occupancy), reading the items through overflow_items[];
// their records through k_classify_long.  Returns 0 = done (items whose list overflowed again
// are in c->overflow_left, their number in host_state.n_overflow), 1 = not applicable (reads too long for the LDS layout: the caller takes the
// straight-line kernel for all items), < 0 = error.
// `s`: the stream the two kernels are launched on; `side`: they count into out.st, a DevState of their own that the caller has initialised and will
// read back itself (overflow_side_launch) -- no host synchronisation here.
template <bool PAIRED, bool FILTER>
int overflow_second_pass(kamd_ctx* c, hipStream_t s, bool side, const u32* d_words, const uint16_t* d_len, u64 nov, int seq_words, int rec_words, const FilterDev& fd,
                         const AlignOut& out) {
  constexpr int WAVES = BLOCK / 64;
  constexpr int NM = PAIRED ? 2 : 1;
  constexpr int LC = V3_LIST_CAP_LONG;
  const int lane_words = seq_words * NM;
  const size_t lds_bytes = (size_t)WAVES * 64 * lane_words * sizeof(u32);
  if (lds_bytes > 96 * 1024) return 1;
  const int stride2 = 2 + LC + (FILTER ? 4 : 0);
  if (int rc = c->raw2.ensure(nov * (u64)stride2 * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->overflow_left.ensure(nov * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->stats_b.ensure(sizeof(DevStatsA), 0, c->stream)) return rc;
  // (the probes of this pass are not kernel A's of the roofline: counters of their own)
  HIPC(hipMemsetAsync(c->stats_b.p, 0, sizeof(DevStatsA), s));
  // items per wavefront: enough wavefronts for every CU, at least a wavefront's worth of items each
  // (8 / 16 / 32 wavefronts per CU measured alike on config #3 -- 0.15 M items -- and on 1.65 M stress items, 32 ahead by 1 %)
  const int ipw = (int)std::min<u64>(1024, std::max<u64>(64, nov / ((u64)std::max(1, c->n_cus) * 32)));
  const u64 n_waves = (nov + (u64)ipw - 1) / (u64)ipw;
  u32* raw2 = c->raw2.as<u32>();
#define KAMD_LAUNCH_V3L2(DLV, TXT, LAY)                                                                                                  \
  do {                                                                                                                                  \
    HIPC(hipFuncSetAttribute((const void*)k_match_v3<PAIRED, FILTER, DLV, TXT, LAY, LC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
    hipLaunchKernelGGL((k_match_v3<PAIRED, FILTER, DLV, TXT, LAY, LC, true>), dim3(grid_for(n_waves, WAVES)), dim3(BLOCK), lds_bytes, s, c->ix, d_words, d_len, nov, \
                       seq_words, rec_words, ipw, std::min(c->refill_min, 8), raw2, stride2, c->stats_b.as<DevStatsA>(), (const u64*)c->overflow_items.as<u64>()); \
  } while (0)
#define KAMD_LAUNCH_V32(DLV, TXT)                                                                                                        \
  do { if (c->ix.table_layout == kamd::LAYOUT_COMPACT) KAMD_LAUNCH_V3L2(DLV, TXT, kamd::LAYOUT_COMPACT); else KAMD_LAUNCH_V3L2(DLV, TXT, kamd::LAYOUT_WIDE); } while (0)
  const bool dl = c->ix.n_dbuckets != 0, txt = c->tune.text_verify == 1;
  if (dl) { if (txt) KAMD_LAUNCH_V32(true, true); else KAMD_LAUNCH_V32(true, false); }
  else { if (txt) KAMD_LAUNCH_V32(false, true); else KAMD_LAUNCH_V32(false, false); }
#undef KAMD_LAUNCH_V32
#undef KAMD_LAUNCH_V3L2
  if (!side) {
    c->host_state.n_overflow = 0;   // k_classify_long counts the items whose list overflowed again
    if (int rc = push_state(c)) return rc;
  }
  hipLaunchKernelGGL((k_classify_long<PAIRED, FILTER, LC>), dim3(grid_for(nov, CL_WAVES * CL_ITEMS)), dim3(64 * CL_WAVES), 0, s, c->ix, (const u32*)raw2, stride2,
                     (const u64*)c->overflow_items.as<u64>(), nov, c->overflow_scratch.as<u32>(), fd, 0ULL, out, c->overflow_left.as<u64>());
  HIPC(hipGetLastError());
  return side ? 0 : sync_state(c);
}
// The second pass BESIDE the absorption of the batch's other tuple records (called from align_batch when k_classify has counted the overflow items, before
// absorb_tuples): the pass is a few wavefronts per CU waiting on long chains of dependent probes (0.8 + 0.5 ms for the 0.15 M such pairs of config #3's
// 30 M), the absorption is memory-side atomics -- side by side they cost the longer of the two.  Its kernels run on c->ov_stream and count into
// c->ov_state (stream_words continues the batch's, the rest starts at zero); kamd_pseudoalign joins (overflow_side_join) before it looks at the counters.
// Not with the positional filters / --union (their cursors and counters stay on the one-stream path).  Returns 1 = not applicable, < 0 = error.
template <bool PAIRED, bool FILTER>
int overflow_side_launch(kamd_ctx* c, const u32* d_words, const uint16_t* d_len, u64 nov, int seq_words, int rec_words, const FilterDev& fd, AlignOut& out) {
  if ((size_t)(BLOCK / 64) * 64 * seq_words * (PAIRED ? 2 : 1) * sizeof(u32) > 96 * 1024) return 1;   // (overflow_second_pass's own test: reads too long)
  if (!c->ov_stream) {
    HIPC(hipStreamCreateWithFlags(&c->ov_stream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&c->ov_ev_in, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&c->ov_ev_done, hipEventDisableTiming));
    HIPC(hipEventCreate(&c->ov_ev_t0)); HIPC(hipEventCreate(&c->ov_ev_t1));
    HIPC(hipHostMalloc((void**)&c->ov_pin, 2 * sizeof(DevState), hipHostMallocDefault));
  }
  // every buffer the pass appends to, before the absorption takes its pointers (growing one moves it)
  const u64 w = c->host_state.stream_words, r = c->host_state.n_recs;
  if (int rc = c->stream_buf.ensure((w + nov * (TUPLE_CAP_BIG + 2)) * sizeof(u32), w * sizeof(u32), c->stream)) return rc;
  if (int rc = c->rec_off.ensure((r + nov) * sizeof(u64), r * sizeof(u64), c->stream)) return rc;
  if (int rc = c->ov_state.ensure(sizeof(DevState), 0, c->stream)) return rc;
  out.stream = c->stream_buf.as<u32>(); out.rec_off = c->rec_off.as<u64>();
  memset(&c->ov_pin[0], 0, sizeof(DevState));
  c->ov_pin[0].stream_words = w;
  HIPC(hipMemcpyAsync(c->ov_state.p, &c->ov_pin[0], sizeof(DevState), hipMemcpyHostToDevice, c->stream));
  HIPC(hipEventRecord(c->ov_ev_in, c->stream));
  HIPC(hipStreamWaitEvent(c->ov_stream, c->ov_ev_in, 0));
  AlignOut out2 = out;
  out2.st = (DevState*)c->ov_state.p;
  HIPC(hipEventRecord(c->ov_ev_t0, c->ov_stream));
  const int sp = overflow_second_pass<PAIRED, FILTER>(c, c->ov_stream, true, d_words, d_len, nov, seq_words, rec_words, fd, out2);
  if (sp != 0) return sp;   // (1 cannot happen: tested above)
  HIPC(hipEventRecord(c->ov_ev_t1, c->ov_stream));
  HIPC(hipMemcpyAsync(&c->ov_pin[1], c->ov_state.p, sizeof(DevState), hipMemcpyDeviceToHost, c->ov_stream));
  HIPC(hipEventRecord(c->ov_ev_done, c->ov_stream));
  c->ov_side_pending = true;
  return 0;
}
// ... and its end: the pass's counters join the context's (host and device); returns the number of items whose list overflowed again
int overflow_side_join(kamd_ctx* c, u64 nov, u64* n_again) {
  HIPC(hipEventSynchronize(c->ov_ev_done));
  c->ov_side_pending = false;
  const DevState& e = c->ov_pin[1];
  c->host_state.stream_words = e.stream_words;
  c->host_state.st_multi += e.st_multi;
  c->host_state.n_overflow = 0;
  *n_again = e.n_overflow;
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, c->ov_ev_t0, c->ov_ev_t1));
  c->overflow_ms += ms; c->overflow_total += nov;
  return push_state(c);
}
}  // namespace

extern "C" int kamd_pseudoalign(kamd_ctx* c, const kamd_quant_opts* o, const uint32_t* d_words, const uint16_t* d_len,
                                uint64_t n_items, int32_t max_len) {
  if (!c || !o) return kamd::fail(-1, "kamd_pseudoalign: null argument");
  if (!c->has_index) return kamd::fail(-1, "kamd_pseudoalign: no index uploaded");
  if (!o->paired && !(o->fld > 0.0 && o->sd > 0.0))
    return kamd::fail(-1, "kamd_pseudoalign: fragment length mean and sd must be supplied for single-end reads (-l, -s)");
  if (o->strand < 0 || o->strand > 2) return kamd::fail(-1, "kamd_pseudoalign: bad strand option");
  if (!o->paired && o->do_union && !o->single_overhang)
    // (the reference itself aborts there: findPosition looks the union's transcripts up in the first mapping k-mer's set,
    // "Index not present in SparseVector")
    return kamd::fail(-5, "kamd_pseudoalign: --single with --union needs --single-overhang");
  apply_quant_opts(c, o);
  if (max_len <= 0 || max_len > 65535) return kamd::fail(-1, "kamd_pseudoalign: max_len must be in [1, 65535]");
  if (n_items == 0) return 0;
  HIPC(hipSetDevice(c->device));
  if (c->ov_side_pending) { HIPC(hipStreamSynchronize(c->ov_stream)); c->ov_side_pending = false; }   // (left by a call that failed before its join)
  const int seq_words = (max_len + 15) / 16 + 1;
  const int rec_words = (int)kamd_packed_record_words(max_len);
  // positional filters (ProcessReads.cpp:1095-1145): has_mean_fl is set by -l only, and while the reads are processed
  // mean_fl is the -l value itself (MinCollector constructor, MinCollector.h:38-41; the truncated-Gaussian mean of
  // init_mean_fl_trunc -- 199.99999999999994 for -l 200 -s 25 -- replaces it only after ProcessReads, main.cpp:2668-2671);
  // with an estimated FLD has_mean_fl stays false while reads are processed
  FilterDev fd{o->single_overhang, o->fld != 0.0 ? 1 : 0, 0, o->strand, c->ix.comprehensive};
  if (fd.has_mean_fl) fd.fl = (int)o->fld;  // (int) tc.get_mean_frag_len() (ProcessReads.cpp:1098)
  const bool filter = fd.strand != 0 || (!fd.single_overhang && fd.has_mean_fl);
  // capacity for the worst case of this batch (its record stream is recycled from batch to batch)
  const u64 key_base = c->recs_total;   // position of the batch's first item in the run's input (first-occurrence keys)
  if (int rc = c->stream_buf.ensure(n_items * (TUPLE_CAP + 2) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->rec_off.ensure(n_items * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->overflow_items.ensure(n_items * sizeof(u64), 0, c->stream)) return rc;
  if (filter) {
    if (int rc = c->explicit_items.ensure(n_items * sizeof(u64), 0, c->stream)) return rc;
    if (int rc = c->explicit_items_big.ensure(n_items * sizeof(u64), 0, c->stream)) return rc;
  }
  AlignOut out{c->dense.as<u32>(), c->track_order ? c->dense_first.as<u64>() : nullptr, c->stream_buf.as<u32>(), c->rec_off.as<u64>(), c->overflow_items.as<u64>(),
               c->explicit_items.as<u64>(), c->explicit_items_big.as<u64>(), (DevState*)c->state.p};
  WorkStream ws(c);   // (from here to the end of the call the context's work stream may be the side stream)
  int rc = 0;
  if (o->paired) rc = filter ? align_batch<true, true>(c, ws, d_words, d_len, n_items, seq_words, rec_words, fd, out, key_base)
                             : align_batch<true, false>(c, ws, d_words, d_len, n_items, seq_words, rec_words, fd, out, key_base);
  else rc = filter ? align_batch<false, true>(c, ws, d_words, d_len, n_items, seq_words, rec_words, fd, out, key_base)
                   : align_batch<false, false>(c, ws, d_words, d_len, n_items, seq_words, rec_words, fd, out, key_base);
  if (rc) return rc;
  // (every chunk of kernel A has completed: the record stream may be reallocated from here on)
  if (c->host_state.n_overflow) {
    const u64 nov = c->host_state.n_overflow;
    c->had_overflow_items = true;
    // The items go through kernel A's loop once more with an append-only class list in global memory (overflow_second_pass); what overflows
    // again -- or every item when the tuning says so, or when the reads are too long for the LDS layout -- takes the straight-line kernel.  The
    // cursor scratch (2 x 1024 words per item) is only needed by the filters / --union in the second pass and by the straight-line kernel.
    // (Round 5's second pass kept a SCANNED 64-entry list per lane in LDS: 8 wavefronts per CU instead of 24 and a scan per hit, 41.5 ms against
    // the straight-line kernel's 33-37 for the 1.65 M such pairs of 30 M stress pairs; it was opt-in and is gone.)
    const bool try_second = c->tune.overflow_second_pass != 2;
    const bool need_cursors = filter || c->ix.union_mode;
    const bool beside = c->ov_side_pending;   // align_batch started the second pass beside the absorption of the batch's other records
    if (!try_second || need_cursors) if (int rc2 = c->overflow_scratch.ensure(nov * 2 * TUPLE_CAP_BIG * sizeof(u32), 0, c->stream)) return rc2;
    if (!beside) {
      const u64 w = c->host_state.stream_words, r = c->host_state.n_recs;
      if (int rc2 = c->stream_buf.ensure((w + nov * (TUPLE_CAP_BIG + 2)) * sizeof(u32), w * sizeof(u32), c->stream)) return rc2;
      if (int rc2 = c->rec_off.ensure((r + nov) * sizeof(u64), r * sizeof(u64), c->stream)) return rc2;
      out.stream = c->stream_buf.as<u32>(); out.rec_off = c->rec_off.as<u64>();
    }
    const u64 ov_base = 0;   // (record indices of the batch)
    if (!c->ev_ov0) { HIPC(hipEventCreate(&c->ev_ov0)); HIPC(hipEventCreate(&c->ev_ov1)); }
    u64 n_straight = nov;
    const u64* straight_items = nullptr;   // (null: all of overflow_items)
    if (beside) {
      if (int rc2 = overflow_side_join(c, nov, &n_straight)) return rc2;
      straight_items = c->overflow_left.as<u64>();
    }
    HIPC(hipEventRecord(c->ev_ov0, c->stream));
    if (try_second && !beside) {
      int sp = 1;
      if (o->paired) sp = filter ? overflow_second_pass<true, true>(c, c->stream, false, d_words, d_len, nov, seq_words, rec_words, fd, out)
                                 : overflow_second_pass<true, false>(c, c->stream, false, d_words, d_len, nov, seq_words, rec_words, fd, out);
      else sp = filter ? overflow_second_pass<false, true>(c, c->stream, false, d_words, d_len, nov, seq_words, rec_words, fd, out)
                       : overflow_second_pass<false, false>(c, c->stream, false, d_words, d_len, nov, seq_words, rec_words, fd, out);
      if (sp < 0) return sp;
      if (sp == 0) { n_straight = c->host_state.n_overflow; straight_items = c->overflow_left.as<u64>(); }
    }
    c->overflow_second_total += nov - n_straight;
    if (n_straight) {
      if (int rc2 = c->overflow_scratch.ensure(n_straight * 2 * TUPLE_CAP_BIG * sizeof(u32), 0, c->stream)) return rc2;
      if (o->paired) { if (filter) launch_overflow<true, true>(c, d_words, d_len, n_straight, seq_words, rec_words, fd, ov_base, out, straight_items);
                       else launch_overflow<true, false>(c, d_words, d_len, n_straight, seq_words, rec_words, fd, ov_base, out, straight_items); }
      else { if (filter) launch_overflow<false, true>(c, d_words, d_len, n_straight, seq_words, rec_words, fd, ov_base, out, straight_items);
             else launch_overflow<false, false>(c, d_words, d_len, n_straight, seq_words, rec_words, fd, ov_base, out, straight_items); }
    }
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(c->ev_ov1, c->stream));
    if (int rc2 = sync_state(c)) return rc2;
    { float ms = 0.f; HIPC(hipEventElapsedTime(&ms, c->ev_ov0, c->ev_ov1)); c->overflow_ms += ms; if (!beside) c->overflow_total += nov; }
    c->host_state.n_overflow = 0;
    if (int rc2 = push_state(c)) return rc2;
    // their records (rec_off of an overflow item now points at its long record) join the distinct tuples
    if (int rc2 = absorb_tuples(c, c->stream_buf.as<u32>(), c->rec_off.as<u64>(), nov, c->host_state.stream_words, key_base,
                                c->host_state.st_multi - c->multi_before, c->overflow_items.as<u64>())) return rc2;
    c->multi_before = c->host_state.st_multi;
  }
  if (filter && (c->host_state.n_explicit || c->host_state.n_explicit_big)) {
    // second pass over the items whose set was changed: write the filtered sets as explicit records
    const u64 ne = c->host_state.n_explicit, nb = c->host_state.n_explicit_big;
    const u64 have_w = c->exp_words_done, have_r = c->host_state.exp_recs;
    if (int rc2 = c->exp_stream.ensure((c->host_state.exp_words + 2) * sizeof(u32), have_w * sizeof(u32), c->stream)) return rc2;
    if (int rc2 = c->exp_off.ensure((have_r + ne + nb + 1) * sizeof(u64), have_r * sizeof(u64), c->stream)) return rc2;
    if (c->track_order) if (int rc2 = c->exp_key.ensure((have_r + ne + nb + 1) * sizeof(u64), have_r * sizeof(u64), c->stream)) return rc2;
    u64* exp_key = c->track_order ? c->exp_key.as<u64>() : nullptr;
    c->host_state.cand_words = have_w;  // write cursor of this pass
    if (int rc2 = push_state(c)) return rc2;
    for (int big = 0; big < 2; big++) {
      const u64 n = big ? nb : ne;
      if (!n) continue;
      const int cap = big ? TUPLE_CAP_BIG : TUPLE_CAP;
      if (int rc2 = c->exp_scratch.ensure(n * (u64)(2 * cap + EXPLICIT_HITS) * sizeof(u32), 0, c->stream)) return rc2;
      const u64* items = big ? c->explicit_items_big.as<u64>() : c->explicit_items.as<u64>();
      if (o->paired) hipLaunchKernelGGL(k_explicit_write<true>, dim3(grid_for(n, 64)), dim3(64), 0, c->stream, c->ix, d_words, d_len, items, n,
                                        seq_words, rec_words, c->exp_scratch.as<u32>(), cap, fd, c->exp_stream.as<u32>(),
                                        c->exp_off.as<u64>(), exp_key, key_base, 1ULL, (DevState*)c->state.p);
      else hipLaunchKernelGGL(k_explicit_write<false>, dim3(grid_for(n, 64)), dim3(64), 0, c->stream, c->ix, d_words, d_len, items, n,
                              seq_words, rec_words, c->exp_scratch.as<u32>(), cap, fd, c->exp_stream.as<u32>(), c->exp_off.as<u64>(),
                              exp_key, key_base, 1ULL, (DevState*)c->state.p);
      HIPC(hipGetLastError());
      HIPC(hipStreamSynchronize(c->stream));  // exp_scratch is reused by the second launch
    }
    if (int rc2 = sync_state(c)) return rc2;
    if (c->host_state.n_hit_overflow) return kamd::fail(-4, "kamd_pseudoalign: a read's hits touch more blocks than the per-hit strand filter keeps");
    c->exp_words_done = c->host_state.cand_words;
    c->host_state.n_explicit = 0; c->host_state.n_explicit_big = 0;
    if (int rc2 = push_state(c)) return rc2;
  }
  // (the batch's tuple records have joined the distinct tuples of the run; its record stream is free again)
  c->recs_total += n_items;
  c->finalized = false;
  return 0;
}

extern "C" int kamd_align_stats_get(kamd_ctx* c, kamd_align_stats* s) {
  if (!c || !s) return kamd::fail(-1, "kamd_align_stats_get: null argument");
  HIPC(hipSetDevice(c->device));
  if (int rc = sync_state(c)) return rc;
  s->n_processed = c->host_state.st_processed; s->n_single = c->host_state.st_single; s->n_multi = c->host_state.st_multi;
  DevStatsA sa{};
  HIPC(hipMemcpyAsync(&sa, c->stats_a.p, sizeof sa, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  s->n_probes = sa.probes; s->n_bucket_reads = sa.bucket_reads;
  s->n_distinct_tuples = c->n_distinct_tuples; s->n_stream_words = c->host_state.stream_words;
  s->n_raw_words = sa.raw_words;
  s->n_text_hits = sa.text_hits;
  s->n_wave_iters = sa.wave_iters; s->n_lane_iters = sa.lane_iters;
  return 0;
}

namespace {
// ~20 000 qualifying pairs at config #3's rate: one pass, with a margin for sparser data
constexpr u64 FLD_FIRST_CHUNK = 1048576;
constexpr int FLD_CAP_SMALL = 64;   // list entries per item in global scratch (an LDS list of TUPLE_CAP entries sends too many items to
                                    // the re-run, which costs ~1 ms per launch however few they are)
// buffers for a prefix of n items + k_fld_first / k_fld / k_fld_rank + the copy of the ranked sample, all on stream s (no synchronisation)
constexpr u32 FLD_WANT = 10000;   // pairs in the sample (ProcessReads.cpp:981-985)
int32_t* fld_host_sample(kamd_ctx* c) { return (int32_t*)((u32*)c->fld_host + 2 * c->fld_host_cap + 4); }   // [FLD_WANT], behind the two vectors + head
u32* fld_host_head(kamd_ctx* c) { return (u32*)c->fld_host + 2 * c->fld_host_cap; }                          // {candidates, list overflows, qualifying}
int fld_launch(kamd_ctx* c, const FilterDev& fd, const u32* w, const uint16_t* l, u64 n, int seq_words, int rec_words, hipStream_t s) {
  if (int rc = c->fld_tl.ensure(n * 4, 0, c->stream)) return rc;
  if (int rc = c->fld_card.ensure(n * 4, 0, c->stream)) return rc;
  if (int rc = c->fld_scratch.ensure(n * 2 * FLD_CAP_SMALL * 4, 0, c->stream)) return rc;
  if (n > c->fld_host_cap) {   // pinned staging: the two result vectors (only copied when a class list overflowed), head, ranked sample
    if (c->fld_host) (void)hipHostFree(c->fld_host);
    c->fld_host = nullptr; c->fld_host_cap = 0;
    if (hipHostMalloc(&c->fld_host, n * 8 + 16 + FLD_WANT * 4, hipHostMallocDefault) != hipSuccess) return kamd::fail(-100, "kamd_fld_from_batch: pinned allocation failed");
    c->fld_host_cap = n;
  }
  // k_fld_first leaves the pairs mapPair gives a usable length for; k_fld (the whole match + |u|) runs on those only
  if (int rc = c->fld_cand.ensure((n + 2) * 8 + FLD_WANT * 4 + (n / (FLD_RANK_BLOCK * FLD_RANK_PER) + 2) * 4, 0, c->stream)) return rc;
  u32* head = (u32*)c->fld_cand.p;                  // {candidates, list overflows, qualifying, -}
  u64* cand = c->fld_cand.as<u64>() + 2;
  int32_t* sample = (int32_t*)(cand + n);
  HIPC(hipMemsetAsync(head, 0, 16, s));
  hipLaunchKernelGGL(k_fld_first, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, c->ix, w, l, n, seq_words, rec_words, c->fld_tl.as<int32_t>(),
                     c->fld_card.as<u32>(), cand, head);
  hipLaunchKernelGGL(k_fld, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, s, c->ix, w, l, (const u64*)cand, n, seq_words, rec_words,
                     c->fld_scratch.as<u32>(), FLD_CAP_SMALL, fd, c->fld_tl.as<int32_t>(), c->fld_card.as<u32>(), (const u32*)head);
  const u32 n_blk = (u32)grid_for(n, FLD_RANK_BLOCK * FLD_RANK_PER);
  u32* blk = (u32*)(sample + FLD_WANT);
  hipLaunchKernelGGL(k_fld_count, dim3(n_blk), dim3(FLD_RANK_BLOCK), 0, s, c->fld_tl.as<int32_t>(), c->fld_card.as<u32>(), n, blk);
  hipLaunchKernelGGL(k_fld_scan, dim3(1), dim3(FLD_RANK_BLOCK), 0, s, blk, n_blk, head);
  hipLaunchKernelGGL(k_fld_emit, dim3(n_blk), dim3(FLD_RANK_BLOCK), 0, s, c->fld_tl.as<int32_t>(), c->fld_card.as<u32>(), n, FLD_WANT, blk, sample);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(fld_host_head(c), head, 16, hipMemcpyDeviceToHost, s));
  HIPC(hipMemcpyAsync(fld_host_sample(c), sample, FLD_WANT * 4, hipMemcpyDeviceToHost, s));
  return 0;
}
}  // namespace

// The first prefix of kamd_fld_from_batch, launched on a side stream: call it BEFORE kamd_pseudoalign on the same batch and
// the fragment-length kernel (latency-bound, few wavefronts) runs underneath kernel A instead of after it.
extern "C" int kamd_fld_prefetch(kamd_ctx* c, const kamd_quant_opts* o, const uint32_t* d_words, const uint16_t* d_len,
                                 uint64_t n_items, int32_t max_len) {
  if (!c || !o) return kamd::fail(-1, "kamd_fld_prefetch: null argument");
  if (!o->paired || o->fld != 0.0) return kamd::fail(-1, "kamd_fld_prefetch: the FLD is only estimated for paired reads without -l");
  if (!c->has_index) return kamd::fail(-1, "kamd_fld_prefetch: no index uploaded");
  apply_quant_opts(c, o);
  HIPC(hipSetDevice(c->device));
  if (n_items == 0) return 0;
  if (!c->fld_stream) {
    HIPC(hipStreamCreateWithFlags(&c->fld_stream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&c->fld_ev, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&c->fld_ev_in, hipEventDisableTiming));
  }
  if (c->fld_pending.valid) { HIPC(hipStreamSynchronize(c->fld_stream)); c->fld_pending.valid = false; }
  const u64 n = std::min<u64>(FLD_FIRST_CHUNK, n_items);
  // launched by the kamd_pseudoalign call on the same batch, behind its kernel A (align_batch): the fragment-length kernels then run beside
  // k_classify / the tuple de-duplication instead of underneath kernel A, which lives on the memory system's request rate
  c->fld_deferred.w = d_words; c->fld_deferred.l = d_len; c->fld_deferred.n = n; c->fld_deferred.max_len = max_len;
  c->fld_deferred.strand = o->strand; c->fld_deferred.so = o->single_overhang; c->fld_deferred.comp = c->ix.comprehensive; c->fld_deferred.valid = true;
  return 0;
}

extern "C" int kamd_fld_from_batch(kamd_ctx* c, const kamd_quant_opts* o, const uint32_t* d_words, const uint16_t* d_len,
                                   uint64_t n_items, int32_t max_len, uint32_t* flens, uint64_t* n_used) {
  if (!c || !flens || !o) return kamd::fail(-1, "kamd_fld_from_batch: null argument");
  if (!o->paired || o->fld != 0.0) return kamd::fail(-1, "kamd_fld_from_batch: the FLD is only estimated for paired reads without -l");
  if (!c->has_index) return kamd::fail(-1, "kamd_fld_from_batch: no index uploaded");
  apply_quant_opts(c, o);
  const FilterDev fd{o->single_overhang, 0, 0, o->strand, c->ix.comprehensive};
  HIPC(hipSetDevice(c->device));
  const int seq_words = (max_len + 15) / 16 + 1;
  const int rec_words = (int)kamd_packed_record_words(max_len);
  c->fld_deferred.valid = false;   // (a prefetch that no kamd_pseudoalign call picked up is dropped: the sample is computed here)
  u64 found = n_used ? *n_used : 0, done = 0;  // continues a sample started on earlier batches
  const u64 found0 = found;
  // the sample is the first 10000 qualifying pairs: start with a prefix that suffices when a few per cent of the pairs
  // qualify (config #3: 3.6 % -- one transcript after the filters AND both mates on one block), then size the next prefix
  // from the rate seen so far.  (Matching the prefix with kernel A's FILTER variant + a kernel over its raw records was
  // tried: 0.5 + 1.0 ms per 262 k pairs plus 1.1 ms for the few items whose class list overflows -- not better than k_fld.)
  u64 chunk = FLD_FIRST_CHUNK;
  DBuf &tl = c->fld_tl, &card = c->fld_card, &scratch = c->fld_scratch, &items = c->fld_items;
  std::vector<u64> h_items;
  int rc = 0;
  while (done < n_items && found < 10000 && rc == 0) {
    const u64 n = std::min(chunk, n_items - done);
    const u32* w = d_words + done * (u64)rec_words * 2;
    const uint16_t* l = d_len + 2 * done;
    const bool prefetched = done == 0 && c->fld_pending.valid && c->fld_pending.w == w && c->fld_pending.l == l && c->fld_pending.n == n &&
                            c->fld_pending.max_len == max_len && c->fld_pending.strand == o->strand && c->fld_pending.so == o->single_overhang;
    if (c->fld_pending.valid) {   // either consumed now or stale
      if (hipEventSynchronize(c->fld_ev) != hipSuccess) { rc = kamd::fail(-100, "k_fld (prefetched) failed"); break; }
      c->fld_pending.valid = false;
    }
    if (!prefetched) {
      if ((rc = fld_launch(c, fd, w, l, n, seq_words, rec_words, c->stream))) break;
      if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = kamd::fail(-100, "k_fld copy failed"); break; }
    }
    // items with more than FLD_CAP_SMALL distinct transcript sets (the head counts them): same kernel again with the large list,
    // then the sample is taken on the host from the two full vectors
    const u32* head = fld_host_head(c);
    if (head[1]) {
      int32_t* h_tl = (int32_t*)c->fld_host; u32* h_card = (u32*)c->fld_host + n;
      if (hipMemcpyAsync(h_card, card.p, n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = kamd::fail(-100, "k_fld copy failed"); break; }
      h_items.clear();
      for (u64 i = 0; i < n; i++) if (h_card[i] == FLD_OVERFLOW) h_items.push_back(i);
      const u64 no = h_items.size();
      if ((rc = items.ensure(no * 8, 0, c->stream))) break;
      if ((rc = scratch.ensure(std::max<u64>(no * 2 * TUPLE_CAP_BIG * 4, n * 2 * FLD_CAP_SMALL * 4), 0, c->stream))) break;
      if (hipMemcpyAsync(items.p, h_items.data(), no * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = kamd::fail(-100, "k_fld copy failed"); break; }
      hipLaunchKernelGGL(k_fld, dim3(grid_for(no, BLOCK)), dim3(BLOCK), 0, c->stream, c->ix, w, l, items.as<u64>(), no, seq_words,
                         rec_words, scratch.as<u32>(), TUPLE_CAP_BIG, fd, tl.as<int32_t>(), card.as<u32>());
      if (hipMemcpyAsync(h_tl, tl.p, n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
          hipMemcpyAsync(h_card, card.p, n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
          hipStreamSynchronize(c->stream) != hipSuccess) { rc = kamd::fail(-100, "k_fld copy failed"); break; }
      // first 10000 qualifying pairs in input order (ProcessReads.cpp:981-1017,1174-1181 at -t 1)
      for (u64 i = 0; i < n && found < 10000; i++)
        if (h_card[i] == 1 && h_tl[i] > 0 && h_tl[i] < KAMD_MAX_FRAG_LEN) { flens[h_tl[i]]++; found++; }
    } else {
      // k_fld_rank's list: the qualifying pairs of the prefix in input order
      const int32_t* smp = fld_host_sample(c);
      const u32 have = std::min<u32>(head[2], FLD_WANT);
      for (u32 r = 0; r < have && found < 10000; r++) { flens[smp[r]]++; found++; }
    }
    done += n;
    const double rate = std::max((double)(found - found0) / (double)done, 1e-4);
    chunk = std::min<u64>(std::max<u64>((u64)((double)(10000 - std::min<u64>(found, 10000)) / rate * 1.5), 65536), 2097152);
  }
  if (n_used) *n_used = found;
  return rc;
}

// ---- exchange helpers -------------------------------------------------------------------------------------------------
