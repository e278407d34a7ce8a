// kamd_ec.hip -- exact de-duplication of records and tuples, EC resolution, kamd_ec_* (MasterProcessor::update, MinCollector::intersectECs / increaseCount)
#include "kamd_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// exact de-duplication of records [cnt, n, w0..w(n-1)]
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_table_init(TSlot* t, u64 cap) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) { t[i].tag = 0; t[i].owner = ~0ULL; t[i].count = 0; t[i].first = ~0ULL; }
}
// idx == nullptr: records r0..r0+n-1 ; else records idx[0..n-1]
__global__ void k_rec_insert(const u32* __restrict__ stream, const u64* __restrict__ rec_off, const u64* __restrict__ idx,
                             u64 r0, u64 n, TSlot* table, u64 mask, u64 seed, u64* rec_slot) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 r = idx ? idx[i] : r0 + i;
  const u64 off = rec_off[r];
  if (off == ~0ULL || stream[off] == 0u) return;  // no record (empty intersection) / slot of an item that is not a tuple record
  const u32 m = stream[off + 1];
  const u64 tag = rec_hash(stream + off + 1, m + 1, seed);
  u64 s = (tag >> 1) & mask;
  for (;;) {
    u64 old = atomicCAS(&table[s].tag, 0ULL, tag);
    if (old == 0ULL || old == tag) break;
    s = (s + 1) & mask;
  }
  atomicMin(&table[s].owner, off);
  rec_slot[r] = s;
}
// `list` receives the table slot of every distinct record exactly once (appended by the record that owns the slot;
// one global atomic per block)
constexpr int VERIFY_BLOCK = BLOCK;
__global__ __launch_bounds__(VERIFY_BLOCK) void k_rec_verify(const u32* __restrict__ stream, const u64* __restrict__ rec_off,
                                                      const u64* __restrict__ idx, u64 r0, u64 n, TSlot* table,
                                                      const u64* __restrict__ rec_slot, u64* retry, u64* list,
                                                      const u64* __restrict__ keys, int track, DevState* st) {
  __shared__ u32 blk_n; __shared__ u64 blk_base;
  if (threadIdx.x == 0) blk_n = 0;
  __syncthreads();
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  bool is_owner = false; u64 s = 0; u32 my = 0;
  if (i < n) {
    const u64 r = idx ? idx[i] : r0 + i;
    const u64 off = rec_off[r];
    if (off != ~0ULL && stream[off] != 0u) {
      s = rec_slot[r];
      const u64 own = table[s].owner;
      const u32 m = stream[off + 1];
      bool same = stream[own + 1] == m;
      for (u32 j = 0; same && j < m; j++) same = stream[own + 2 + j] == stream[off + 2 + j];
      if (same) {
        atomicAdd(&table[s].count, (u64)stream[off]); is_owner = (own == off);
        if (track) atomicMin(&table[s].first, keys ? keys[r] : r);  // first occurrence: record indices follow the input order
      }
      else { u64 k = atomicAdd(&st->n_retry, 1ULL); retry[k] = r; }
    }
  }
  if (is_owner) my = atomicAdd(&blk_n, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && blk_n) blk_base = atomicAdd(&st->n_list, (u64)blk_n);
  __syncthreads();
  if (is_owner) list[blk_base + my] = s;
}

// The same de-duplication in ONE launch: the slot's tag word holds 32 bits of the hash AND the stream offset (+1) of the record that
// installed it, so a single compare-and-swap elects the owner and publishes where its contents are; a later record with the
// same 32 hash bits compares itself with that record at once (equal: its count is added; different contents -- a collision
// of the 32 bits -- : next slot).  Exact, no second pass, no per-record slot array, no retries.
// max_probe: give up after that many slots (counted in st->n_retry; the host then repeats the run with a larger table) -- lets the
// table be sized for the EXPECTED number of distinct records instead of the number of records.
__global__ __launch_bounds__(BLOCK) void k_rec_dedup(const u32* __restrict__ stream, const u64* __restrict__ rec_off, u64 r0, u64 n, TSlot* table,
                                                     u64 mask, u64* list, const u64* __restrict__ keys, int track, u32 max_probe, DevState* st) {
  __shared__ u32 blk_n; __shared__ u64 blk_base;
  if (threadIdx.x == 0) blk_n = 0;
  __syncthreads();
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  bool is_owner = false; u64 s = 0; u32 my = 0;
  if (i < n) {
    const u64 r = r0 + i;
    const u64 off = rec_off[r];
    if (off != ~0ULL && stream[off] != 0u) {
      const u32 m = stream[off + 1];
      const u64 h = rec_hash(stream + off + 1, m + 1, 1);
      const u64 mine = (h & 0xFFFFFFFF00000000ULL) | (off + 1);   // off + 1 < 2^32 (checked by the host), so the word is never 0
      s = (h >> 1) & mask;
      bool placed = false;
      for (u32 probes = 0; probes < max_probe; probes++) {
        u64 old = table[s].tag;   // (plain load first, as in k_tup_absorb)
        if (old == 0ULL) old = atomicCAS(&table[s].tag, 0ULL, mine);
        if (old == 0ULL) { is_owner = true; table[s].owner = off; placed = true; break; }   // (owner: read by the kernels after this one)
        if ((old >> 32) == (mine >> 32)) {
          const u64 ooff = (old & 0xFFFFFFFFULL) - 1;
          bool same = stream[ooff + 1] == m;
          for (u32 j = 0; same && j < m; j++) same = stream[ooff + 2 + j] == stream[off + 2 + j];
          if (same) { placed = true; break; }
        }
        s = (s + 1) & mask;
      }
      if (placed) {
        atomicAdd(&table[s].count, (u64)stream[off]);
        if (track) atomicMin(&table[s].first, keys ? keys[r] : r);   // first occurrence: record indices follow the input order
      } else atomicAdd(&st->n_retry, 1ULL);
    }
  }
  if (is_owner) my = atomicAdd(&blk_n, 1u);
  __syncthreads();
  if (threadIdx.x == 0 && blk_n) blk_base = atomicAdd(&st->n_list, (u64)blk_n);
  __syncthreads();
  if (is_owner) list[blk_base + my] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// The distinct tuples of a run, kept ACROSS batches (MinCollector keeps O(#ECs), src/MinCollector.cpp:251-269: so do we).
// After every batch its tuple records are absorbed into a persistent table -- k_rec_dedup's scheme: tag = 32 hash bits + where the
// owner's record lies, one compare-and-swap elects the owner, equal contents add their count -- whose owners live in the tuple
// STORE, a compact stream of distinct records only; a record that opens a new slot is compared against in the batch's own stream
// until k_tup_store has moved it into the store (bit 31 of the tag's offset tells which of the two streams it refers to).  The
// batch's record stream is then recycled: device memory no longer grows with the number of reads, only with the number of
// distinct tuples.
// ------------------------------------------------------------------------------------------------------------------
constexpr u64 TAG_LOCAL = 0x80000000ULL;   // the tag's offset refers to the batch's stream (the record is not in the store yet)
__global__ __launch_bounds__(BLOCK) void k_tup_absorb(const u32* __restrict__ batch, const u32* __restrict__ store, const u64* __restrict__ rec_off,
                                                      const u64* __restrict__ idx, u64 n, TSlot* table, u64 mask, u64* list, u64 key_base, int track,
                                                      u32 max_probe, u64* fail, DevState* st, u32 fixed_stride = 0, u64 item0 = 0) {
  __shared__ u32 blk_n, blk_words; __shared__ u64 blk_base;
  if (threadIdx.x == 0) { blk_n = 0; blk_words = 0; }
  __syncthreads();
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  bool is_owner = false; u64 s = 0; u32 my = 0, m = 0;
  if (i < n) {
    const u64 r = idx ? idx[i] : i;
    // (kernel A's records sit in fixed slots -- record r of the chunk at (item0 + r) x stride, what k_classify wrote into rec_off --: the
    // main pass computes the place instead of loading it, one dependent load less in a chain of three)
    const u64 off = fixed_stride ? (item0 + r) * (u64)fixed_stride : rec_off[r];
    if (off != ~0ULL && batch[off] != 0u) {
      m = batch[off + 1];
      const u64 h = rec_hash(batch + off + 1, m + 1, 1);
      const u64 mine = (h & 0xFFFFFFFF00000000ULL) | TAG_LOCAL | (off + 1);   // off + 1 < 2^31 (checked by the host)
      s = (h >> 1) & mask;
      bool placed = false;
      for (u32 probes = 0; probes < max_probe; probes++) {
        // nine records in ten find their tuple already in the table: a plain load sees the tag without the trip to the memory side
        // that a device-scope atomic costs (a tag goes from 0 to its value once; a stale 0 from this XCD's L2 only means the
        // compare-and-swap below is taken after all)
        u64 old = table[s].tag;
        if (old == 0ULL) old = atomicCAS(&table[s].tag, 0ULL, mine);
        if (old == 0ULL) { is_owner = true; table[s].owner = off; placed = true; break; }
        if ((old >> 32) == (mine >> 32)) {
          const u32* o = ((old & TAG_LOCAL) ? batch : store) + ((old & 0x7FFFFFFFULL) - 1);
          bool same = o[1] == m;
          for (u32 j = 0; same && j < m; j++) same = o[2 + j] == batch[off + 2 + j];
          if (same) { placed = true; break; }
        }
        s = (s + 1) & mask;
      }
      if (placed) {
        atomicAdd(&table[s].count, (u64)batch[off]);
        if (track & 1) atomicMin(&table[s].first, key_base + r);   // first occurrence: record indices follow the input order
      } else fail[atomicAdd(&st->tl_fail, 1ULL)] = r;
    }
  }
  if (is_owner) { my = atomicAdd(&blk_n, 1u); atomicAdd(&blk_words, m + 2u); }
  __syncthreads();
  if (threadIdx.x == 0 && blk_n) { blk_base = atomicAdd(&st->tl_n, (u64)blk_n); atomicAdd(&st->bound_words, (u64)blk_words); }
  __syncthreads();
  if (is_owner) list[blk_base + my] = s;
}
// The main pass of a batch (kernel A's records in their fixed slots), FOUR records per thread: the chain record -> hash -> slot -> owner's
// record is a sequence of dependent random reads, and with one record per lane the kernel waited 89 % of its cycles on it (VERDICT r4:
// 11 G dependent requests/s on a chip that serves 54 G independent ones).  Nine records in ten find their tuple in the table at the
// first probe: that case runs here for four records at once in straight-line code -- the four record reads, the four tag reads and
// the four owner reads are each in flight together --, everything else (an empty slot to claim, a collision, a tuple of more than four
// sets) takes k_tup_absorb's loop, one record at a time, afterwards.  Thread t takes records t, t + T, t + 2T, t + 3T (T = threads of
// the launch): every one of the four rounds reads consecutive slots.
constexpr int ABS_Q = 4;
__global__ __launch_bounds__(BLOCK) void k_tup_absorb4(const u32* __restrict__ batch, const u32* __restrict__ store, u64 n, TSlot* table, u64 mask, u64* list,
                                                       u64 key_base, int track, u32 max_probe, u64* fail, DevState* st, u32 stride, u64 item0) {
  __shared__ u32 blk_n, blk_words; __shared__ u64 blk_base;
  if (threadIdx.x == 0) { blk_n = 0; blk_words = 0; }
  __syncthreads();
  const u64 T = (u64)gridDim.x * blockDim.x;
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 off[ABS_Q]; u32 w[ABS_Q][6]; bool live[ABS_Q], slow[ABS_Q];
#pragma unroll
  for (int q = 0; q < ABS_Q; q++) {
    const u64 r = i + (u64)q * T;
    live[q] = r < n;
    off[q] = (item0 + (live[q] ? r : 0)) * (u64)stride;
    const uint2* p = reinterpret_cast<const uint2*>(batch + off[q]);   // (slots of 14 words: 8-byte aligned)
    const uint2 a = live[q] ? p[0] : make_uint2(0u, 0u);
    const uint2 b = live[q] ? p[1] : make_uint2(0u, 0u), cc = live[q] ? p[2] : make_uint2(0u, 0u);
    w[q][0] = a.x; w[q][1] = a.y; w[q][2] = b.x; w[q][3] = b.y; w[q][4] = cc.x; w[q][5] = cc.y;
    live[q] = live[q] && a.x != 0u;
    slow[q] = live[q] && a.y > 4u;
  }
  u64 h[ABS_Q], sl[ABS_Q], old[ABS_Q];
#pragma unroll
  for (int q = 0; q < ABS_Q; q++) {
    const u32 m = w[q][1];
    u64 x = kamd::mix64(1ULL ^ (0x9e3779b97f4a7c15ULL * ((u64)(m + 1) + 1)));   // rec_hash(words 1 .. m + 1, seed 1)
    x = kamd::mix64(x ^ m);
#pragma unroll
    for (int j = 0; j < 4; j++) if ((u32)j < m) x = kamd::mix64(x ^ w[q][2 + j]);
    h[q] = x | 1ULL;
    sl[q] = (h[q] >> 1) & mask;
    old[q] = (live[q] && !slow[q]) ? table[sl[q]].tag : 0ULL;
  }
  u32 ow[ABS_Q][5]; bool cand[ABS_Q];
#pragma unroll
  for (int q = 0; q < ABS_Q; q++) {
    cand[q] = live[q] && !slow[q] && old[q] != 0ULL && (old[q] >> 32) == (h[q] >> 32);
    const u32* o = ((old[q] & TAG_LOCAL) ? batch : store) + ((old[q] & 0x7FFFFFFFULL) - 1);
    const u32 m = w[q][1];
#pragma unroll
    for (int j = 0; j < 5; j++) ow[q][j] = (cand[q] && (u32)j <= m) ? o[1 + j] : 0u;   // the owner's m and its first four sets
  }
#pragma unroll
  for (int q = 0; q < ABS_Q; q++) {
    if (!live[q] || slow[q]) continue;
    bool same = cand[q] && ow[q][0] == w[q][1];
#pragma unroll
    for (int j = 0; j < 4; j++) if ((u32)j < w[q][1]) same = same && ow[q][1 + j] == w[q][2 + j];
    if (same) {
      atomicAdd(&table[sl[q]].count, (u64)w[q][0]);
      if (track & 1) atomicMin(&table[sl[q]].first, key_base + i + (u64)q * T);
    } else slow[q] = true;
  }
  // the rest: k_tup_absorb's loop
  u32 my[ABS_Q]; u64 own_s[ABS_Q]; bool is_owner[ABS_Q];
#pragma unroll
  for (int q = 0; q < ABS_Q; q++) {
    is_owner[q] = false; my[q] = 0; own_s[q] = 0;
    if (!slow[q]) continue;
    const u64 r = i + (u64)q * T, o = off[q];
    const u32 m = batch[o + 1];
    const u64 hh = rec_hash(batch + o + 1, m + 1, 1);
    const u64 mine = (hh & 0xFFFFFFFF00000000ULL) | TAG_LOCAL | (o + 1);   // o + 1 < 2^31 (checked by the host)
    u64 s = (hh >> 1) & mask;
    bool placed = false;
    for (u32 probes = 0; probes < max_probe; probes++) {
      u64 od = table[s].tag;
      if (od == 0ULL) od = atomicCAS(&table[s].tag, 0ULL, mine);
      if (od == 0ULL) { is_owner[q] = true; table[s].owner = o; placed = true; break; }
      if ((od >> 32) == (mine >> 32)) {
        const u32* ob = ((od & TAG_LOCAL) ? batch : store) + ((od & 0x7FFFFFFFULL) - 1);
        bool same = ob[1] == m;
        for (u32 j = 0; same && j < m; j++) same = ob[2 + j] == batch[o + 2 + j];
        if (same) { placed = true; break; }
      }
      s = (s + 1) & mask;
    }
    if (placed) {
      atomicAdd(&table[s].count, (u64)batch[o]);
      if (track & 1) atomicMin(&table[s].first, key_base + r);
    } else fail[atomicAdd(&st->tl_fail, 1ULL)] = r;
    if (is_owner[q]) { own_s[q] = s; my[q] = atomicAdd(&blk_n, 1u); atomicAdd(&blk_words, m + 2u); }
  }
  __syncthreads();
  if (threadIdx.x == 0 && blk_n) { blk_base = atomicAdd(&st->tl_n, (u64)blk_n); atomicAdd(&st->bound_words, (u64)blk_words); }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < ABS_Q; q++) if (is_owner[q]) list[blk_base + my[q]] = own_s[q];
}
// the records that opened a slot in the last k_tup_absorb move from the batch's stream into the store
__global__ __launch_bounds__(BLOCK) void k_tup_store(const u32* __restrict__ batch, u32* store, TSlot* table, u64* list, u64 first_new,
                                                     u64 n_new, DevState* st) {
  __shared__ u32 wsum[BLOCK / 64]; __shared__ u64 blk_base;
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s = 0, off = 0; u32 need = 0;
  if (i < n_new) { s = list[first_new + i] & 0xFFFFFFFFULL; off = table[s].owner; need = batch[off + 1] + 2u; }
  const u32 incl = wave_incl_scan(need);
  if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int j = 0; j < BLOCK / 64; j++) t += wsum[j]; blk_base = t ? atomicAdd(&st->ts_words, (u64)t) : 0ULL; }
  __syncthreads();
  if (i >= n_new) return;
  u32 before = 0;
  for (u32 j = 0; j < (threadIdx.x >> 6); j++) before += wsum[j];
  const u64 dst = blk_base + before + incl - need;
  for (u32 j = 0; j < need; j++) store[dst + j] = batch[off + j];
  table[s].owner = dst;
  table[s].tag = (table[s].tag & 0xFFFFFFFF00000000ULL) | (dst + 1);   // (dst + 1 < 2^31: checked by the host)
  // the list entry carries the record's place in the store beside the slot: the kernels that walk the distinct tuples (k_bound_tuples,
  // k_resolve) start on the record at once instead of going list -> slot -> record, one dependent random read less per tuple
  list[first_new + i] = s | (dst << 32);
}
// a larger table: every distinct tuple (all of them in the store by now) takes a slot of the new one, the list follows
__global__ void k_tup_rehash(const u32* __restrict__ store, const TSlot* __restrict__ old, TSlot* nu, u64 mask, u64* list, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const TSlot o = old[list[i] & 0xFFFFFFFFULL];
  const u32 m = store[o.owner + 1];
  const u64 h = rec_hash(store + o.owner + 1, m + 1, 1);
  const u64 tag = (h & 0xFFFFFFFF00000000ULL) | (o.owner + 1);
  u64 s = (h >> 1) & mask;
  while (atomicCAS(&nu[s].tag, 0ULL, tag) != 0ULL) s = (s + 1) & mask;
  nu[s].owner = o.owner; nu[s].count = o.count; nu[s].first = o.first;
  list[i] = s | (o.owner << 32);
}

// ------------------------------------------------------------------------------------------------------------------
// resolve: candidates = transcript sets of (a) index sets with a non-zero dense count, (b) distinct tuples
// ------------------------------------------------------------------------------------------------------------------
// upper bound of the candidate stream size: sum over candidates of (smallest list + 2)
constexpr u32 RES_BIG_MIN = 16;   // (= RES_LANES) a tuple whose smallest set has more members goes to k_resolve_big
__global__ void k_bound_tuples(DevIndex ix, const u32* __restrict__ stream, const TSlot* table, const u64* list, u64 n,
                               u32* per_tuple, u32* big_idx, DevState* st) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 b = 0;
  bool big = false, huge = false;
  if (i < n) {
    const u64 off = list[i] >> 32;   // (the record's place in the store rides in the list entry: k_tup_store)
    const u32 m = stream[off + 1];
    const kamd::SetTables stt{(const uint64_t*)ix.ec_off, ix.ec_ids};
    b = kamd::set_size_bound(stt, stream + off + 2, (int)m, ix.union_mode != 0) + 2;   // smallest set / sum of the sets (--union)
    per_tuple[i] = (u32)b;   // the tuple's slot in the candidate stream (k_resolve writes there: no allocation at run time)
    big = big_idx && !ix.union_mode && b - 2 > RES_BIG_MIN;
    huge = big && b - 2 > 1024;   // (= RB_CAND_BIG)
  }
  b = wave_sum64(b);
  if (lane_id() == 0 && b) atomicAdd(&st->bound_words, b);
  // the work lists of k_resolve_big (the second one grows from the end of the same array): one atomic per wavefront and list
  const u64 bm = __ballot(big && !huge), hm = __ballot(huge);
  if (bm) {
    u64 base = 0;
    if (lane_id() == 0) base = atomicAdd(&st->n_big, (u64)__popcll(bm));
    base = shfl_u64(base, 0);
    if (big && !huge) big_idx[base + __popcll(bm & ((1ULL << lane_id()) - 1ULL))] = (u32)i;
  }
  if (hm) {
    u64 base = 0;
    if (lane_id() == 0) base = atomicAdd(&st->n_huge, (u64)__popcll(hm));
    base = shfl_u64(base, 0);
    if (huge) big_idx[n - 1 - (base + __popcll(hm & ((1ULL << lane_id()) - 1ULL)))] = (u32)i;
  }
}
__global__ void k_bound_singles(DevIndex ix, const u32* __restrict__ dense, DevState* st) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 b = 0;
  if (e < ix.n_ecs && dense[e]) b = ix.ec_off[e + 1] - ix.ec_off[e] + 2;
  b = wave_sum64(b);
  if (lane_id() == 0 && b) atomicAdd(&st->bound_words, b);
}
// (a) one thread per index set with a count: copy its on-listed members as a candidate record
__global__ void k_cand_singles(DevIndex ix, const u32* __restrict__ dense, const u64* __restrict__ dense_first, u32* cand,
                               u64* cand_off, u64* cand_key, DevState* st) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = e < ix.n_ecs && dense[e] != 0;
  const u32* ids = nullptr; u32 n = 0;
  if (have) { ids = ix.ec_ids + ix.ec_off[e]; n = (u32)(ix.ec_off[e + 1] - ix.ec_off[e]); }
  const bool longset = n > 64;
  if (have && !longset) {
    u32 keep = 0;
    for (u32 j = 0; j < n; j++) keep += onlisted(ix.onlist_bits, ids[j]);
    if (keep) {
      u64 off = atomicAdd(&st->cand_words, (u64)keep + 2);
      u64 r = atomicAdd(&st->cand_recs, 1ULL);
      u32* w = cand + off;
      w[0] = dense[e]; w[1] = keep;
      u32 o = 0;
      for (u32 j = 0; j < n; j++) if (onlisted(ix.onlist_bits, ids[j])) w[2 + o++] = ids[j];
      cand_off[r] = off;
      if (cand_key) cand_key[r] = dense_first[e];
    }
  }
  // sets of more than 64 transcripts (thousands: poly-A and repeat-family classes) are copied by the whole wavefront
  u64 m = __ballot(longset);
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const u64 se = shfl_u64(e, src);
    const u32* sid = ix.ec_ids + ix.ec_off[se];
    const u32 sn = (u32)__shfl((int)n, src, 64);
    u32 keep = 0;
    for (u32 j = lane_id(); j < sn; j += 64) keep += onlisted(ix.onlist_bits, sid[j]);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) keep += (u32)__shfl_xor((int)keep, d, 64);
    if (keep == 0) continue;
    u64 off = 0, r = 0;
    if (lane_id() == 0) { off = atomicAdd(&st->cand_words, (u64)keep + 2); r = atomicAdd(&st->cand_recs, 1ULL); }
    off = shfl_u64(off, 0); r = shfl_u64(r, 0);
    u32* w = cand + off;
    if (lane_id() == 0) { w[0] = dense[se]; w[1] = keep; cand_off[r] = off; if (cand_key) cand_key[r] = dense_first[se]; }
    u32 o = 0;
    for (u32 j0 = 0; j0 < sn; j0 += 64) {
      const u32 j = j0 + lane_id();
      const u32 x = j < sn ? sid[j] : 0u;
      const bool ok = j < sn && onlisted(ix.onlist_bits, x);
      const u64 bm = __ballot(ok);
      if (ok) w[2 + o + __popcll(bm & ((1ULL << lane_id()) - 1ULL))] = x;
      o += (u32)__popcll(bm);
    }
  }
}
// (b) one 16-lane group per distinct tuple (4 tuples per wavefront): intersect the m sorted sets; 16 candidates of the
//     smallest set per step, membership by binary search in the others, survivors compacted with ballot + popcount
//     prefix over the group's 16 bits of the wavefront mask
constexpr int RES_LANES = 16;
constexpr int RES_BLOCK = BLOCK;   // (1024-thread blocks were slower: 5.9 against 3.9 ms, the block-wide allocation barrier waits for the slowest tuple)
__global__ __launch_bounds__(RES_BLOCK) void k_resolve(DevIndex ix, const u32* __restrict__ stream, const TSlot* table,
                                                   const u64* list, u64 n, const u64* __restrict__ slot_off, u32* cand, u64* cand_off,
                                                   u64* cand_key, const DevState* st) {
  // every tuple owns a slot of the candidate stream (offsets = scan of the bounds of k_bound_tuples, behind what
  // k_cand_singles wrote) and record number cand_recs + its index: nothing is allocated here.  (The first version took two
  // same-address atomics per block of 16 tuples -- 250 k of them at ~12 ns each were 3 of the kernel's 3.9 ms.)
  const u64 base_words = st->cand_words, base_recs = st->cand_recs;
  const u64 gid = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / RES_LANES;
  const bool valid = gid < n;
  const int lane = lane_id();
  const int sub = lane & (RES_LANES - 1);
  const int gsh = lane & ~(RES_LANES - 1);  // bit position of this group inside the wavefront mask
  TSlot sl; sl.owner = 0; sl.count = 0; sl.tag = 0; sl.first = ~0ULL;
  u32 m = 0, best = 0, nb = 0;
  const u32* es = nullptr; const u32* base = nullptr;
  // Lane j of the group looks at set j: offset and size of all sets in ONE round of loads (a thread-serial scan of the m
  // sets was 2 m dependent latencies), the smallest by a 16-lane min-reduction (first wins on ties, like the serial scan).
  u64 my_off = 0; u32 my_sz = 0xFFFFFFFFu;
  bool small = false;   // m <= TUPLE_CAP and every set has <= RES_LANES members: the all-pairs path below
  if (valid) {
    const u64 le = list[gid];
    const u64 owner = le >> 32;     // (the record's place in the store rides in the list entry: the slot -- count, first occurrence -- is
    sl = table[le & 0xFFFFFFFFULL];  //  only needed when the record is written, its load runs beside the chain record -> offsets -> members)
    m = stream[owner + 1];
    es = stream + owner + 2;
    if (m <= (u32)RES_LANES) {
      if ((u32)sub < m) { const u32 e = es[sub]; my_off = ix.ec_off[e]; my_sz = (u32)(ix.ec_off[e + 1] - my_off); }
    } else {
      u64 best_sz = ~0ULL;
      for (u32 j = 0; j < m; j++) { u32 e = es[j]; u64 sz = ix.ec_off[e + 1] - ix.ec_off[e]; if (sz < best_sz) { best_sz = sz; best = j; } }
      base = ix.ec_ids + ix.ec_off[es[best]];
      nb = (u32)best_sz;
    }
  }
  {
    u32 key = my_sz, arg = (u32)sub;   // (size, lane) minimum over the group
#pragma unroll
    for (int d = 1; d < RES_LANES; d <<= 1) {
      const u32 k2 = __shfl_xor(key, d, RES_LANES), a2 = __shfl_xor(arg, d, RES_LANES);
      if (k2 < key || (k2 == key && a2 < arg)) { key = k2; arg = a2; }
    }
    const u32 mx_in = ((u32)sub < m && m <= (u32)RES_LANES) ? my_sz : 0u;
    u32 mx = mx_in;
#pragma unroll
    for (int d = 1; d < RES_LANES; d <<= 1) mx = max(mx, __shfl_xor(mx, d, RES_LANES));
    if (valid && m <= (u32)RES_LANES) {
      best = arg; nb = key;
      const u64 boff = __shfl(my_off, (int)best, RES_LANES);
      base = ix.ec_ids + boff;
      small = mx <= (u32)RES_LANES && m <= (u32)TUPLE_CAP;   // (tuples of the overflow kernel can have more sets: serial path)
    }
  }
  if (valid && nb > RES_BIG_MIN) return;   // k_resolve_big's (the whole 16-lane group leaves: nb is uniform in it; no barrier follows)
  u32 total = 0, first_mask = 0, x0 = 0;
  if (small) {
    // all pairs in registers: lane c holds candidate c of the smallest set, lane i member i of every other set (<= 12 loads
    // in flight per lane, ONE round); candidate c survives if every other set has a lane that holds it
    const u32 x = (u32)sub < nb ? base[sub] : 0u;
    u32 y[TUPLE_CAP];
#pragma unroll
    for (int j = 0; j < TUPLE_CAP; j++) {
      const u64 oj = __shfl(my_off, j, RES_LANES); const u32 sj = __shfl(my_sz, j, RES_LANES);
      y[j] = ((u32)j < m && (u32)j != best && (u32)sub < sj) ? ix.ec_ids[oj + sub] : 0xFFFFFFFFu;
    }
    bool okx = (u32)sub < nb && onlisted(ix.onlist_bits, x);
    u32 keep = 0;
    for (u32 c = 0; c < nb; c++) {
      const u32 xc = __shfl(x, (int)c, RES_LANES);
      bool okc = __shfl((int)okx, (int)c, RES_LANES) != 0;
#pragma unroll
      for (int j = 0; j < TUPLE_CAP; j++) {
        if ((u32)j < m && (u32)j != best) {
          const u32 hit = (u32)((__ballot(y[j] == xc) >> gsh) & 0xFFFFu);
          okc = okc && hit != 0;
        }
      }
      keep |= (okc ? 1u : 0u) << c;
    }
    first_mask = keep; x0 = x;
    total = (u32)__popc(keep);
  }
  // classify the candidates of the smallest set: membership in every other set (binary search) and the on-list mask;
  // `mask_of(c0)` is recomputed in the write pass except for the first 16 candidates
  // The sets of the tuple sixteen at a time: lane j of the group fetches set j's offset, size and bitmap slot (one round of loads for sixteen sets
  // -- a lane walking the sets one after the other paid four dependent latencies per set: 16 ms per 30 M stress pairs, whose tuples have dozens of
  // sets), then every candidate lane tests all of the batch's bitmap sets at once and searches the batch's short lists one after the other.
  auto chunk_mask = [&](u32 c0, u32* x_out) -> u32 {
    const u32 c = c0 + sub;
    bool ok = c < nb;
    const u32 x = ok ? base[c] : 0;
    if (ok) ok = onlisted(ix.onlist_bits, x);
    for (u32 j0 = 0; j0 < m; j0 += RES_LANES) {
      if (((__ballot(ok) >> gsh) & 0xFFFFu) == 0u) break;   // (no candidate of the group is left)
      const u32 jj = j0 + sub;
      const bool vj = jj < m && jj != best;
      u64 o = 0; u32 sz = 0, slot = BM_NONE;
      if (vj) { const u32 e = es[jj]; o = ix.ec_off[e]; sz = (u32)(ix.ec_off[e + 1] - o); if (sz > ix.bm_min) slot = ix.ec_bm_slot[e]; }
      u32 w[RES_LANES];
#pragma unroll
      for (int t = 0; t < RES_LANES; t++) {
        const u32 st = __shfl(slot, t, RES_LANES);
        w[t] = (ok && st != BM_NONE) ? ix.bm_words[(u64)st * ix.bm_stride + (x >> 5)] : 0xFFFFFFFFu;
      }
      u32 all = 0xFFFFFFFFu;
#pragma unroll
      for (int t = 0; t < RES_LANES; t++) all &= w[t];
      ok = ok && ((all >> (x & 31)) & 1u);
#pragma unroll
      for (int t = 0; t < RES_LANES; t++) {
        const u32 st = __shfl(slot, t, RES_LANES), zt = __shfl(sz, t, RES_LANES);
        const u64 ot = __shfl(o, t, RES_LANES);
        const bool vt = __shfl((int)vj, t, RES_LANES) != 0;
        if (vt && st == BM_NONE && ok) ok = set_contains(ix.ec_ids + ot, zt, x);
      }
    }
    *x_out = x;
    return (u32)((__ballot(ok) >> gsh) & 0xFFFFu);
  };
  if (valid && !small) {
    for (u32 c0 = 0; c0 < nb; c0 += RES_LANES) {
      u32 x; const u32 gm = chunk_mask(c0, &x);
      if (c0 == 0) { first_mask = gm; x0 = x; }
      total += (u32)__popc(gm);
    }
  }
  if (!valid) return;
  if (total == 0) { if (sub == 0) cand_off[base_recs + gid] = ~0ULL; return; }  // empty intersection: not pseudoaligned (MinCollector.cpp:200-202)
  const u64 out_off = base_words + slot_off[gid];
  if (sub == 0) { cand[out_off] = (u32)sl.count; cand[out_off + 1] = total; cand_off[base_recs + gid] = out_off; if (cand_key) cand_key[base_recs + gid] = sl.first; }
  u32 written = 0;
  for (u32 c0 = 0; c0 < nb; c0 += RES_LANES) {
    u32 x, gm;
    if (c0 == 0) { gm = first_mask; x = x0; } else gm = chunk_mask(c0, &x);
    if ((gm >> sub) & 1u) cand[out_off + 2 + written + __popc(gm & ((1u << sub) - 1))] = x;
    written += (u32)__popc(gm);
  }
}

// (b') tuples whose smallest set is LARGE (more than 16 members): repeat-family and poly-A classes of a real transcriptome -- a few
// hundred to a few thousand transcripts per set, eight or more sets per tuple.  k_resolve's scheme (16 lanes, every candidate of the
// smallest set binary-searched in every other set in global memory, once to count and once to write) costs nb x (m - 1) x log2|set|
// DEPENDENT random reads per tuple: 145 ms for the 0.97 M tuples of 4 M stress pairs.  Here ONE WAVEFRONT takes a tuple and keeps the
// running intersection in LDS: the smallest set is loaded once (coalesced), every other set streams through LDS in tiles of 1024 ids
// (coalesced, independent loads) and the survivors are looked up in the tile by binary search IN LDS -- both lists are sorted, so only the
// survivors inside the tile's id range are looked at -- then compacted; once 64 or fewer survive they are searched in the remaining
// sets directly.  Global traffic is the sum of the sets' sizes, read once, instead of a dependent chain per candidate.
constexpr int RB_WAVES = 4, RB_TILE = 1024, RB_MAXSETS = 256;
constexpr u32 RB_CAND_BIG = 1024, RB_CAND_HUGE = 4096;   // two launches: smallest set of 17..1024 members (8 KB of LDS per wavefront: 20 wavefronts per CU), of more
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int CAND>
__global__ __launch_bounds__(64 * RB_WAVES) void k_resolve_big(DevIndex ix, const u32* __restrict__ stream, const TSlot* table, const u64* list,
                                                               const u32* __restrict__ big_idx, u64 n_big, const u64* __restrict__ slot_off, u32* cand,
                                                               u64* cand_off, u64* cand_key, const DevState* st) {
  __shared__ u32 s_cand_all[RB_WAVES][CAND];
  __shared__ u32 s_tile_all[RB_WAVES][RB_TILE];
  __shared__ u32 s_meta_all[RB_WAVES][4 * RB_MAXSETS];   // offsets (two words), sizes (0xFFFFFFFF = taken) and bitmap slots (BM_NONE = none) of the tuple's sets
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const u64 bi = (u64)blockIdx.x * RB_WAVES + w;
  if (bi >= n_big) return;
  u32* s_cand = s_cand_all[w];
  u32* s_tile = s_tile_all[w];
  u32* s_off = s_meta_all[w];
  u32* s_offh = s_off + RB_MAXSETS;
  u32* s_sz = s_off + 2 * RB_MAXSETS;
  u32* s_bm = s_off + 3 * RB_MAXSETS;
  const u64 base_words = st->cand_words, base_recs = st->cand_recs;
  const u64 gid = big_idx[bi];
  const u64 le = list[gid];
  const u64 owner = le >> 32;
  const u32 m = stream[owner + 1];
  const u32* es = stream + owner + 2;
  const u64 out_off = base_words + slot_off[gid];
  u32 cnt = 0;
  // offset and size of every set in one round of loads per 64 sets, kept in LDS (a pair whose mates run through a repeat or a poly-A
  // stretch has more than a hundred distinct classes); tuples of more than RB_MAXSETS sets and smallest sets beyond the LDS buffer
  // take the plain path below
  const bool lds_meta = m <= (u32)RB_MAXSETS;
  u32 bsz = 0xFFFFFFFFu, bj = 0xFFFFFFFFu;
  for (u32 j = lane; j < m; j += 64) {
    const u32 e = es[j] & kamd::EC_ID_MASK;
    const u64 off = ix.ec_off[e];
    const u32 sz = (u32)(ix.ec_off[e + 1] - off);
    if (lds_meta) { s_off[j] = (u32)off; s_offh[j] = (u32)(off >> 32); s_sz[j] = sz; s_bm[j] = sz > ix.bm_min ? ix.ec_bm_slot[e] : BM_NONE; }
    if (sz < bsz) { bsz = sz; bj = j; }   // (first wins on ties, as in k_resolve)
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 s2 = __shfl_xor(bsz, d, 64), j2 = __shfl_xor(bj, d, 64);
    if (s2 < bsz || (s2 == bsz && j2 < bj)) { bsz = s2; bj = j2; }
  }
  const u32 best = bj, nb = bsz;
  if (!lds_meta || nb > (u32)CAND) {
    // 64 candidates per step, searched in global memory, written at once (the tuple's slot holds nb ids)
    const u32* base = ix.ec_ids + ix.ec_off[es[best] & kamd::EC_ID_MASK];
    for (u32 c0 = 0; c0 < nb; c0 += 64) {
      const u32 c = c0 + lane;
      bool ok = c < nb;
      const u32 x = ok ? base[c] : 0u;
      if (ok) ok = onlisted(ix.onlist_bits, x);
      for (u32 j = 0; j < m; j++) {
        if (j == best) continue;
        const u32 e = es[j] & kamd::EC_ID_MASK;
        if (ok) { const u64 o = ix.ec_off[e]; ok = set_has(ix, e, o, (u32)(ix.ec_off[e + 1] - o), x); }
      }
      const u64 bm = __ballot(ok);
      if (ok) cand[out_off + 2 + cnt + __popcll(bm & ((1ULL << lane) - 1ULL))] = x;
      cnt += (u32)__popcll(bm);
    }
  } else {
    wave_lds_sync();
    const u32* base = ix.ec_ids + ((u64)s_off[best] | ((u64)s_offh[best] << 32));
    wave_lds_sync();
    if (lane == 0) s_sz[best] = 0xFFFFFFFFu;
    // the on-listed members of the smallest set, compacted into LDS
    for (u32 c0 = 0; c0 < nb; c0 += 64) {
      const u32 c = c0 + lane;
      const u32 x = c < nb ? base[c] : 0u;
      const bool ok = c < nb && onlisted(ix.onlist_bits, x);
      const u64 bm = __ballot(ok);
      if (ok) s_cand[cnt + __popcll(bm & ((1ULL << lane) - 1ULL))] = x;
      cnt += (u32)__popcll(bm);
    }
    wave_lds_sync();
    // The sets that exist as bitmaps (more than BM_MIN_MEMBERS members: the poly-A and repeat-family classes -- the same few hundred sets in
    // every such tuple, so their bitmaps sit in L2) first, all of them at once: a survivor is tested against them bit by bit, four tests in
    // flight, and drops out at the first set that lacks it.  Streaming a 3 500-member set through LDS and binary-searching the survivors in
    // it cost 350 k look-ups per poly-A tuple (100 sets of 3 500 members): 52 ms for the 10 043 such tuples of 30 M stress pairs (round 5).
    u32 nbm = 0;
    for (u32 j0 = 0; j0 < m; j0 += 64) {
      const u32 j = j0 + lane;
      const bool has = j < m && s_sz[j] != 0xFFFFFFFFu && s_bm[j] != BM_NONE;
      const u64 bmk = __ballot(has);
      if (has) { s_tile[nbm + __popcll(bmk & ((1ULL << lane) - 1ULL))] = s_bm[j]; s_sz[j] = 0xFFFFFFFFu; }
      nbm += (u32)__popcll(bmk);
    }
    wave_lds_sync();
    if (nbm && cnt) {
      // four survivors per lane and trip, four sets per step: sixteen bit tests in flight (a tuple of a hundred poly-A sets and 3 500 survivors was
      // 55 x 35 dependent steps with one survivor per lane)
      u32 kept = 0;
      for (u32 c0 = 0; c0 < cnt; c0 += 256) {
        u32 x[4]; bool ok[4]; const u32* wp[4]; u32 bit[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u32 c = c0 + 64u * q + lane;
          ok[q] = c < cnt;
          x[q] = ok[q] ? s_cand[c] : 0u;
          wp[q] = ix.bm_words + (x[q] >> 5); bit[q] = 1u << (x[q] & 31);
        }
        for (u32 b = 0; b < nbm && __ballot(ok[0] || ok[1] || ok[2] || ok[3]) != 0ULL; b += 4) {
          const u64 o0 = (u64)s_tile[b] * ix.bm_stride, o1 = (u64)s_tile[min(b + 1, nbm - 1)] * ix.bm_stride;
          const u64 o2 = (u64)s_tile[min(b + 2, nbm - 1)] * ix.bm_stride, o3 = (u64)s_tile[min(b + 3, nbm - 1)] * ix.bm_stride;
          u32 w[4];
#pragma unroll
          for (int q = 0; q < 4; q++) w[q] = ok[q] ? (wp[q][o0] & wp[q][o1] & wp[q][o2] & wp[q][o3]) : 0u;
#pragma unroll
          for (int q = 0; q < 4; q++) ok[q] = ok[q] && (w[q] & bit[q]) != 0u;
        }
        wave_lds_sync();   // (every lane has read its four survivors: the compaction below writes at or below them)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u64 bmk = __ballot(ok[q]);
          if (ok[q]) s_cand[kept + __popcll(bmk & ((1ULL << lane) - 1ULL))] = x[q];
          kept += (u32)__popcll(bmk);
        }
        wave_lds_sync();
      }
      cnt = kept;
    }
    // while more than 64 survive: the smallest set not yet taken streams through LDS (both lists are sorted: only the survivors inside a
    // tile's id range are looked up in it, by binary search in LDS); bit 31 of a survivor marks "found in this set"
    while (cnt > 64) {
      u32 ks = 0xFFFFFFFFu, kj = 0xFFFFFFFFu;
      for (u32 j = lane; j < m; j += 64) { const u32 sz = s_sz[j]; if (sz < ks) { ks = sz; kj = j; } }
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const u32 s2 = __shfl_xor(ks, d, 64), j2 = __shfl_xor(kj, d, 64);
        if (s2 < ks || (s2 == ks && j2 < kj)) { ks = s2; kj = j2; }
      }
      if (ks == 0xFFFFFFFFu) break;   // every set taken
      const u32* B = ix.ec_ids + ((u64)s_off[kj] | ((u64)s_offh[kj] << 32));
      const u32 nB = ks;
      wave_lds_sync();
      if (lane == 0) s_sz[kj] = 0xFFFFFFFFu;
      u32 a = 0;   // survivors below a are smaller than everything still to come of B
      for (u32 t0 = 0; t0 < nB && a < cnt; t0 += RB_TILE) {
        const u32 tn = min((u32)RB_TILE, nB - t0);
        for (u32 i = lane; i < tn; i += 64) s_tile[i] = B[t0 + i];
        wave_lds_sync();
        const u32 lo_v = s_tile[0], hi_v = s_tile[tn - 1];
        u32 l = a, h = cnt;
        while (l < h) { const u32 mid = (l + h) >> 1; if ((s_cand[mid] & 0x7FFFFFFFu) < lo_v) l = mid + 1; else h = mid; }
        const u32 a2 = l;
        h = cnt;
        while (l < h) { const u32 mid = (l + h) >> 1; if ((s_cand[mid] & 0x7FFFFFFFu) <= hi_v) l = mid + 1; else h = mid; }
        const u32 b2 = l;
        for (u32 i = a2 + lane; i < b2; i += 64) {
          const u32 x = s_cand[i] & 0x7FFFFFFFu;
          u32 lo = 0, hi = tn;
          while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (s_tile[mid] < x) lo = mid + 1; else hi = mid; }
          if (lo < tn && s_tile[lo] == x) s_cand[i] = x | 0x80000000u;
        }
        a = b2;
        wave_lds_sync();
      }
      // keep the marked ones (in place: a chunk's writes land at or below its own reads)
      u32 kept = 0;
      for (u32 c0 = 0; c0 < cnt; c0 += 64) {
        const u32 c = c0 + lane;
        const u32 v = c < cnt ? s_cand[c] : 0u;
        const bool ok = c < cnt && (v >> 31);
        const u64 bm = __ballot(ok);
        wave_lds_sync();
        if (ok) s_cand[kept + __popcll(bm & ((1ULL << lane) - 1ULL))] = v & 0x7FFFFFFFu;
        kept += (u32)__popcll(bm);
        wave_lds_sync();
      }
      cnt = kept;
    }
    // 64 or fewer survivors: all (survivor, remaining set) pairs at once -- one binary search in global memory per lane and round
    // instead of one round per set (a read that pseudoaligns collapses to a handful of transcripts after the first intersection:
    // taking the remaining sets one after the other was a chain of a dozen dependent reads per set and tuple)
    wave_lds_sync();
    u32 R = 0;
    for (u32 j0 = 0; j0 < m; j0 += 64) {   // the sets not yet taken, compacted behind the survivors' dead flags
      const u32 j = j0 + lane;
      const bool open = j < m && s_sz[j] != 0xFFFFFFFFu;
      const u64 bm = __ballot(open);
      if (open) s_tile[64 + R + __popcll(bm & ((1ULL << lane) - 1ULL))] = j;
      R += (u32)__popcll(bm);
    }
    if (cnt && R) {
      s_tile[lane] = 0u;   // dead flags of the survivors
      wave_lds_sync();
      const u32 P = cnt * R;
      for (u32 p = lane; p < P; p += 64) {
        const u32 ci = p % cnt, jr = s_tile[64 + p / cnt];
        const u64 off = (u64)s_off[jr] | ((u64)s_offh[jr] << 32);
        if (!set_contains(ix.ec_ids + off, s_sz[jr], s_cand[ci])) s_tile[ci] = 1u;
      }
      wave_lds_sync();
      const u32 x = (u32)lane < cnt ? s_cand[lane] : 0u;
      const bool ok = (u32)lane < cnt && s_tile[lane] == 0u;
      const u64 bm = __ballot(ok);
      wave_lds_sync();
      if (ok) s_cand[__popcll(bm & ((1ULL << lane) - 1ULL))] = x;
      cnt = (u32)__popcll(bm);
      wave_lds_sync();
    }
    for (u32 i = lane; i < cnt; i += 64) cand[out_off + 2 + i] = s_cand[i];
  }
  if (lane == 0) {
    if (cnt == 0) cand_off[base_recs + gid] = ~0ULL;   // empty intersection: not pseudoaligned (MinCollector.cpp:200-202)
    else {
      const TSlot sl = table[le & 0xFFFFFFFFULL];
      cand[out_off] = (u32)sl.count; cand[out_off + 1] = cnt; cand_off[base_recs + gid] = out_off;
      if (cand_key) cand_key[base_recs + gid] = sl.first;
    }
  }
}

// --union: (union of mate 1's sets) & (union of mate 2's sets) of a distinct tuple (entries carry the mate flags), one
// thread per tuple -- a k-way merge in increasing order (kamd_core.h for_each_in_set); the option is rare, the kernel plain
__global__ void k_resolve_union(DevIndex ix, const u32* __restrict__ stream, const TSlot* table, const u64* list, u64 n,
                                const u64* __restrict__ slot_off, u32* cand, u64* cand_off, u64* cand_key, u32* cur_scratch,
                                const DevState* st) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const u64 base_words = st->cand_words, base_recs = st->cand_recs;
  const TSlot sl = table[list[gid] & 0xFFFFFFFFULL];
  const u32 m = stream[sl.owner + 1];
  kamd::EcList ecs; ecs.e = const_cast<u32*>(stream + sl.owner + 2); ecs.cap = (int)m; ecs.n = (int)m; ecs.overflow = false;
  u32 cur_small[TUPLE_CAP];
  u32* cur = m <= (u32)TUPLE_CAP ? cur_small : cur_scratch + gid * (u64)TUPLE_CAP_BIG;   // (long tuples come from the overflow kernel)
  const u64 out_off = base_words + slot_off[gid];
  u32 total = 0;
  for_each_member(ix, ecs, cur, [&](u32 tr) { cand[out_off + 2 + total++] = tr; });
  if (total == 0) { cand_off[base_recs + gid] = ~0ULL; return; }
  cand[out_off] = (u32)sl.count; cand[out_off + 1] = total; cand_off[base_recs + gid] = out_off;
  if (cand_key) cand_key[base_recs + gid] = sl.first;
}

// ------------------------------------------------------------------------------------------------------------------
// exclusive scan of u32 sizes into u64 offsets (three kernels; sizes up to 2^31 elements)
// ------------------------------------------------------------------------------------------------------------------
constexpr int SCAN_ELEMS = 2048;  // per block: 256 threads x 8
__global__ __launch_bounds__(BLOCK) void k_scan_local(const u32* __restrict__ in, u64 n, u64* out, u64* block_sums) {
  __shared__ u64 wsum[BLOCK / 64];
  const u64 b0 = (u64)blockIdx.x * SCAN_ELEMS + (u64)threadIdx.x * 8;
  u64 v[8]; u64 run = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { u64 x = (b0 + j < n) ? in[b0 + j] : 0; v[j] = run; run += x; }
  // scan of per-thread totals across the block
  u64 incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { u64 t = __shfl_up(incl, d, 64); if (lane_id() >= d) incl += t; }
  const int w = threadIdx.x >> 6;
  if (lane_id() == 63) wsum[w] = incl;
  __syncthreads();
  u64 woff = 0;
  for (int j = 0; j < w; j++) woff += wsum[j];
  const u64 excl = woff + incl - run;
#pragma unroll
  for (int j = 0; j < 8; j++) if (b0 + j < n) out[b0 + j] = excl + v[j];
  if (threadIdx.x == BLOCK - 1) block_sums[blockIdx.x] = woff + incl;
}
// exclusive scan of the block sums by ONE block: every thread takes a run of consecutive sums (independent loads), the threads' totals
// are scanned across the block, the run is written back.  (One thread walking the sums serially paid a global-memory round trip per
// element: 40-100 us for the 300-1000 blocks of a step's scans.)
__global__ __launch_bounds__(BLOCK) void k_scan_blocks(u64* block_sums, u64 nblocks, u64* total) {
  __shared__ u64 wsum[BLOCK / 64];
  const u64 per = (nblocks + BLOCK - 1) / BLOCK;
  const u64 b0 = (u64)threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  u64 run = 0;
  for (u64 i = b0; i < b1; i++) run += block_sums[i];
  u64 incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { u64 t = __shfl_up(incl, d, 64); if (lane_id() >= d) incl += t; }
  const int w = threadIdx.x >> 6;
  if (lane_id() == 63) wsum[w] = incl;
  __syncthreads();
  u64 woff = 0;
  for (int j = 0; j < w; j++) woff += wsum[j];
  u64 acc = woff + incl - run;
  for (u64 i = b0; i < b1; i++) { const u64 x = block_sums[i]; block_sums[i] = acc; acc += x; }
  if (threadIdx.x == BLOCK - 1) *total = woff + incl;
}
__global__ __launch_bounds__(BLOCK) void k_scan_add(u64* out, u64 n, const u64* block_sums) {
  const u64 b0 = (u64)blockIdx.x * SCAN_ELEMS + (u64)threadIdx.x * 8;
  const u64 add = block_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 8; j++) if (b0 + j < n) out[b0 + j] += add;
}

// final CSR from the distinct candidate sets
__global__ void k_final_keys(const TSlot* table, const u64* list, u64 n, u64* keys) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = table[list[i]].first;
}
__global__ void k_final_sizes(const u32* __restrict__ cand, const TSlot* table, const u64* list, u64 n, u32* sizes) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sizes[i] = cand[table[list[i]].owner + 1];
}
__global__ void k_final_write(const u32* __restrict__ cand, const TSlot* table, const u64* list, u64 n, const u64* ec_off,
                              u32* ec_ids, u32* counts) {
  const u64 wid = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wid >= n) return;
  const TSlot sl = table[list[wid]];
  const u32 sz = cand[sl.owner + 1];
  const u64 o = ec_off[wid];
  for (u32 j = lane_id(); j < sz; j += 64) ec_ids[o + j] = cand[sl.owner + 2 + j];
  if (lane_id() == 0) counts[wid] = (u32)sl.count;
}
__global__ void k_set_last(u64* ec_off, u64 n, const u64* total) { if (threadIdx.x == 0 && blockIdx.x == 0) ec_off[n] = *total; }

// export of distinct tuple records for the multi-GPU exchange
__global__ void k_tuple_export_size(const u32* __restrict__ stream, const TSlot* table, const u64* list, u64 n, DevState* st) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 b = 0;
  if (i < n) b = (u64)stream[(list[i] >> 32) + 1] + 2;
  b = wave_sum64(b);
  if (lane_id() == 0 && b) atomicAdd(&st->bound_words, b);
}
// records are self-delimiting only from the start of a buffer, so the exporter also emits their word offsets
__global__ void k_tuple_export_offsets(const u32* __restrict__ stream, const TSlot* table, const u64* list, u64 n, u32* out,
                                       u64* out_off, DevState* st) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const TSlot sl = table[list[i] & 0xFFFFFFFFULL];
  const u32 m = stream[sl.owner + 1];
  u64 off = atomicAdd(&st->cand_words, (u64)m + 2);
  u64 r = atomicAdd(&st->cand_recs, 1ULL);
  out[off] = (u32)sl.count; out[off + 1] = m;
  for (u32 j = 0; j < m; j++) out[off + 2 + j] = stream[sl.owner + 2 + j];
  out_off[r] = off;
}

}  // namespace

namespace kamdi {

// exact de-duplication of records [r0, r1) of a record stream into `table` (capacity cap, power of two)
// max_probe != 0 (single-launch form only): returns 2 when some record found no slot within max_probe steps -- table too small
int dedup_records(kamd_ctx* c, const u32* stream, const u64* rec_off, u64 r0, u64 r1, TSlot* table, u64 cap, DBuf& slot_buf,
                  u64* list, int track = 0, const u64* keys = nullptr, u32 max_probe = 0, u64 stream_words = ~0ULL) {
  const u64 n = r1 - r0;
  c->host_state.n_list = 0;
  HIPC(hipMemcpyAsync(&((DevState*)c->state.p)->n_list, &c->host_state.n_list, sizeof(u64), hipMemcpyHostToDevice, c->stream));
  if (n == 0) return 0;
  if (c->tune.dedup_form != 1 && stream_words < 0xFFFFFFFFULL) {   // one launch (record offsets fit the low half of the tag word)
    c->host_state.n_retry = 0;
    HIPC(hipMemcpyAsync(&((DevState*)c->state.p)->n_retry, &c->host_state.n_retry, sizeof(u64), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_rec_dedup, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, c->stream, stream, rec_off, r0, n, table, cap - 1, list, keys, track,
                       max_probe ? max_probe : 0xFFFFFFFFu, (DevState*)c->state.p);
    HIPC(hipGetLastError());
    if (int rc = sync_state(c)) return rc;
    return c->host_state.n_retry ? 2 : 0;
  }
  if (int rc = slot_buf.ensure(r1 * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->retry.ensure(2 * n * sizeof(u64), 0, c->stream)) return rc;
  u64* retry_a = c->retry.as<u64>();
  u64* retry_b = retry_a + n;
  const u64* idx = nullptr;
  u64 count = n;
  for (u64 seed = 1; count; seed++) {
    c->host_state.n_retry = 0;
    HIPC(hipMemcpyAsync(&((DevState*)c->state.p)->n_retry, &c->host_state.n_retry, sizeof(u64), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_rec_insert, dim3(grid_for(count, BLOCK)), dim3(BLOCK), 0, c->stream, stream, rec_off, idx, r0, count,
                       table, cap - 1, seed, slot_buf.as<u64>());
    hipLaunchKernelGGL(k_rec_verify, dim3(grid_for(count, VERIFY_BLOCK)), dim3(VERIFY_BLOCK), 0, c->stream, stream, rec_off, idx, r0, count,
                       table, slot_buf.as<u64>(), retry_b, list, keys, track, (DevState*)c->state.p);
    HIPC(hipGetLastError());
    if (int rc = sync_state(c)) return rc;
    count = c->host_state.n_retry;
    std::swap(retry_a, retry_b);
    idx = retry_a;
    if (seed > 64) return kamd::fail(-101, "dedup_records: tag collisions did not resolve");
  }
  return 0;
}


int exclusive_scan(kamd_ctx* c, const u32* sizes, u64 n, u64* out, u64* d_total) {
  const u64 nblocks = std::max<u64>(1, (n + SCAN_ELEMS - 1) / SCAN_ELEMS);
  if (int rc = c->block_sums.ensure((nblocks + 2) * sizeof(u64), 0, c->stream)) return rc;
  hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nblocks), dim3(BLOCK), 0, c->stream, sizes, n, out, c->block_sums.as<u64>());
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(BLOCK), 0, c->stream, c->block_sums.as<u64>(), nblocks, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nblocks), dim3(BLOCK), 0, c->stream, out, n, c->block_sums.as<u64>());
  HIPC(hipGetLastError());
  return 0;
}

u64 pow2_at_least(u64 x) { u64 p = 1024; while (p < x) p <<= 1; return p; }

// forget the distinct tuples (a new run, or records about to be replaced by merged ones)
int tuples_clear(kamd_ctx* c) {
  if (c->tcap && !c->ttable_clean) {
    hipLaunchKernelGGL(k_table_init, dim3(grid_for(c->tcap, BLOCK)), dim3(BLOCK), 0, c->stream, c->ttable.as<TSlot>(), c->tcap);
    HIPC(hipGetLastError());
    c->ttable_clean = true;
  }
  c->host_state.tl_n = 0; c->host_state.ts_words = 0; c->host_state.tl_fail = 0;
  c->n_distinct_tuples = 0;
  return 0;
}
// the tuple table with `cap` slots; the distinct tuples it held move over (all of them are in the store)
int tuples_resize(kamd_ctx* c, u64 cap) {
  // (list entries pack the slot beside the record's offset, slot | offset << 32 -- k_tup_store --: a table beyond 2^32 slots cannot be addressed)
  if (cap > (1ULL << 32)) return kamd::fail(-101, "absorb_tuples: the tuple table would need more than 2^32 slots");
  const u64 n = c->host_state.tl_n;
  DBuf nu;
  if (int rc = nu.ensure(cap * sizeof(TSlot), 0, c->stream)) return rc;
  hipLaunchKernelGGL(k_table_init, dim3(grid_for(cap, BLOCK)), dim3(BLOCK), 0, c->stream, nu.as<TSlot>(), cap);
  if (n) hipLaunchKernelGGL(k_tup_rehash, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, c->stream, (const u32*)c->tstore.as<u32>(), (const TSlot*)c->ttable.as<TSlot>(),
                            nu.as<TSlot>(), cap - 1, c->list.as<u64>(), n);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(c->stream));
  c->ttable.release();
  c->ttable = nu; c->tcap = cap; c->ttable_clean = n == 0;
  return 0;
}
// Absorb the tuple records of one batch (records 0 .. n-1 of `batch` through rec_off; key_base + r = position of record r in the
// run's input) into the persistent tuple table and store.  The table starts at a quarter of the first batch's records (on config #3
// 21.7 M tuple records collapse to 2.0 M distinct tuples; a table for every record would be gigabytes to clear and to miss in),
// is kept at most half full, and a record that finds no slot within 64 probes makes it grow before that record is tried again.
// first_idx (device, optional): the records to look at are rec_off[first_idx[0 .. n-1]] instead of rec_off[0 .. n-1]
int absorb_tuples(kamd_ctx* c, const u32* batch, const u64* rec_off, u64 n, u64 batch_words, u64 key_base, u64 n_tuple_bound,
                  const u64* first_idx, u32 fixed_stride, u64 item0) {
  if (n == 0) return 0;
  if (batch_words >= 0x7FFFFFF0ULL) return kamd::fail(-1, "kamd_pseudoalign: the record stream of one batch must stay below 2^31 words (use smaller batches)");
  DevState* dst = (DevState*)c->state.p;
  if (!c->ev_ab0) { HIPC(hipEventCreate(&c->ev_ab0)); HIPC(hipEventCreate(&c->ev_ab1)); }
  HIPC(hipEventRecord(c->ev_ab0, c->stream));
  const u64 bound = std::min(n, n_tuple_bound);
  u64 want = pow2_at_least(std::max<u64>(bound / 4, 2 * c->host_state.tl_n) + 16);
  if (c->tcap == 0) {
    c->tcap = want;
    if (int rc = c->ttable.ensure(c->tcap * sizeof(TSlot), 0, c->stream)) return rc;
    hipLaunchKernelGGL(k_table_init, dim3(grid_for(c->tcap, BLOCK)), dim3(BLOCK), 0, c->stream, c->ttable.as<TSlot>(), c->tcap);
    c->ttable_clean = true;
  } else if (2 * c->host_state.tl_n + 16 > c->tcap) {
    if (int rc = tuples_resize(c, pow2_at_least(4 * c->host_state.tl_n + 16))) return rc;
  }
  if (int rc = c->list.ensure((c->host_state.tl_n + bound + 1) * sizeof(u64), c->host_state.tl_n * sizeof(u64), c->stream)) return rc;
  if (int rc = c->retry.ensure(2 * (bound + 1) * sizeof(u64), 0, c->stream)) return rc;
  u64* fail_a = c->retry.as<u64>();
  u64* fail_b = fail_a + bound + 1;
  const u64* idx = first_idx;
  u64 count = n;
  for (int round = 0; count; round++) {
    if (round > 40) return kamd::fail(-101, "absorb_tuples: the tuple table does not settle");
    const u64 tl_before = c->host_state.tl_n;
    c->host_state.tl_fail = 0; c->host_state.bound_words = 0;
    if (int rc = push_state(c)) return rc;
    // (the main pass of a batch -- kernel A's records in fixed slots of at least six words -- four records per thread; retries, overflow
    // items and gathered records one per thread)
    if (!idx && fixed_stride >= 6 && (fixed_stride & 1) == 0)
      hipLaunchKernelGGL(k_tup_absorb4, dim3(grid_for((count + ABS_Q - 1) / ABS_Q, BLOCK)), dim3(BLOCK), 0, c->stream, batch, (const u32*)c->tstore.as<u32>(), count,
                         c->ttable.as<TSlot>(), c->tcap - 1, c->list.as<u64>(), key_base, c->track_order ? 1 : 0, 64u, fail_a, dst, fixed_stride, item0);
    else
    hipLaunchKernelGGL(k_tup_absorb, dim3(grid_for(count, BLOCK)), dim3(BLOCK), 0, c->stream, batch, (const u32*)c->tstore.as<u32>(), rec_off, idx, count,
                       c->ttable.as<TSlot>(), c->tcap - 1, c->list.as<u64>(), key_base,
                       c->track_order ? 1 : 0, 64u, fail_a, dst,
                       fixed_stride, item0);
    HIPC(hipGetLastError());
    c->ttable_clean = false;
    if (int rc = sync_state(c)) return rc;
    const u64 n_new = c->host_state.tl_n - tl_before, new_words = c->host_state.bound_words;
    if (n_new) {
      if (c->host_state.ts_words + new_words >= 0x7FFFFFF0ULL) return kamd::fail(-101, "absorb_tuples: more than 2^31 words of distinct tuples");
      if (int rc = c->tstore.ensure((c->host_state.ts_words + new_words + 2) * sizeof(u32), c->host_state.ts_words * sizeof(u32), c->stream)) return rc;
      hipLaunchKernelGGL(k_tup_store, dim3(grid_for(n_new, BLOCK)), dim3(BLOCK), 0, c->stream, batch, c->tstore.as<u32>(), c->ttable.as<TSlot>(),
                         c->list.as<u64>(), tl_before, n_new, dst);
      HIPC(hipGetLastError());
      c->host_state.ts_words += new_words;   // (k_tup_store advances the device copy by the same amount)
    }
    count = c->host_state.tl_fail;
    if (count) {   // some records found no slot: a table four times the size, then those records again
      if (int rc = tuples_resize(c, c->tcap * 4)) return rc;
      std::swap(fail_a, fail_b);
      idx = fail_b;
    }
  }
  c->n_distinct_tuples = c->host_state.tl_n;
  HIPC(hipEventRecord(c->ev_ab1, c->stream));
  HIPC(hipEventSynchronize(c->ev_ab1));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, c->ev_ab0, c->ev_ab1));
  c->last_absorb_ms += ms;
  return 0;
}

}  // namespace kamdi
extern "C" int kamd_ec_dense_counts(kamd_ctx* c, uint32_t** d_counts, uint64_t* n) {
  if (!c || !d_counts || !n) return kamd::fail(-1, "kamd_ec_dense_counts: null argument");
  *d_counts = c->dense.as<u32>(); *n = c->n_ecs;
  return 0;
}
extern "C" int kamd_ec_tuples_export(kamd_ctx* c, uint64_t* n_words, uint64_t* n_tuples) {
  if (!c || !n_words || !n_tuples) return kamd::fail(-1, "kamd_ec_tuples_export: null argument");
  HIPC(hipSetDevice(c->device));
  if (int rc = sync_state(c)) return rc;
  c->host_state.bound_words = 0;
  if (int rc = push_state(c)) return rc;
  const u64 n = c->n_distinct_tuples;
  if (n) hipLaunchKernelGGL(k_tuple_export_size, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, c->stream, c->tstore.as<u32>(),
                            c->ttable.as<TSlot>(), c->list.as<u64>(), n, (DevState*)c->state.p);
  HIPC(hipGetLastError());
  if (int rc = sync_state(c)) return rc;
  *n_words = c->host_state.bound_words; *n_tuples = n;
  return 0;
}
extern "C" int kamd_ec_tuples_copy(kamd_ctx* c, uint32_t* d_out_words, uint64_t* d_out_rec_off) {
  if (!c || !d_out_words || !d_out_rec_off) return kamd::fail(-1, "kamd_ec_tuples_copy: null argument");
  HIPC(hipSetDevice(c->device));
  const u64 n = c->n_distinct_tuples;
  c->host_state.cand_words = 0; c->host_state.cand_recs = 0;
  if (int rc = push_state(c)) return rc;
  if (n) hipLaunchKernelGGL(k_tuple_export_offsets, dim3(grid_for(n, BLOCK)), dim3(BLOCK), 0, c->stream, c->tstore.as<u32>(),
                            c->ttable.as<TSlot>(), c->list.as<u64>(), n, d_out_words, (u64*)d_out_rec_off, (DevState*)c->state.p);
  HIPC(hipGetLastError());
  return sync_state(c);
}
extern "C" int kamd_ec_tuples_replace(kamd_ctx* c, const uint32_t* d_words, uint64_t n_words, const uint64_t* d_rec_off,
                                      uint64_t n_recs) {
  if (!c) return kamd::fail(-1, "kamd_ec_tuples_replace: null argument");
  if (c->track_order) return kamd::fail(-1, "kamd_ec_tuples_replace: merged records have no input order (kamd_ec_track_order is on)");
  HIPC(hipSetDevice(c->device));
  if (int rc = sync_state(c)) return rc;
  // the distinct tuples of this rank give way to the records of all ranks: a fresh table, the gathered records absorbed as one batch
  // (d_words / d_rec_off are only read)
  if (int rc = tuples_clear(c)) return rc;
  if (int rc = push_state(c)) return rc;
  c->had_overflow_items = true;   // gathered records may hold another rank's long tuples (k_resolve_union's cursor scratch)
  c->finalized = false;
  return absorb_tuples(c, d_words, (const u64*)d_rec_off, n_recs, n_words, 0, n_recs);
}

// explicit transcript-set records (positional filters): plain copies, the records are content-keyed
extern "C" int kamd_ec_explicit_export(kamd_ctx* c, uint64_t* n_words, uint64_t* n_recs) {
  if (!c || !n_words || !n_recs) return kamd::fail(-1, "kamd_ec_explicit_export: null argument");
  HIPC(hipSetDevice(c->device));
  if (int rc = sync_state(c)) return rc;
  *n_words = c->exp_words_done; *n_recs = c->host_state.exp_recs;
  return 0;
}
extern "C" int kamd_ec_explicit_copy(kamd_ctx* c, uint32_t* d_out_words, uint64_t* d_out_rec_off) {
  if (!c) return kamd::fail(-1, "kamd_ec_explicit_copy: null argument");
  HIPC(hipSetDevice(c->device));
  if (c->exp_words_done) HIPC(hipMemcpyAsync(d_out_words, c->exp_stream.p, c->exp_words_done * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
  if (c->host_state.exp_recs) HIPC(hipMemcpyAsync(d_out_rec_off, c->exp_off.p, c->host_state.exp_recs * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int kamd_ec_explicit_replace(kamd_ctx* c, const uint32_t* d_words, uint64_t n_words, const uint64_t* d_rec_off,
                                        uint64_t n_recs) {
  if (!c) return kamd::fail(-1, "kamd_ec_explicit_replace: null argument");
  if (c->track_order) return kamd::fail(-1, "kamd_ec_explicit_replace: merged records have no input order (kamd_ec_track_order is on)");
  HIPC(hipSetDevice(c->device));
  if (int rc = sync_state(c)) return rc;
  if (int rc = c->exp_stream.ensure(std::max<u64>(n_words, 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->exp_off.ensure((n_recs + 1) * sizeof(u64), 0, c->stream)) return rc;
  if (n_words) HIPC(hipMemcpyAsync(c->exp_stream.p, d_words, n_words * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
  if (n_recs) HIPC(hipMemcpyAsync(c->exp_off.p, d_rec_off, n_recs * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
  c->exp_words_done = n_words; c->host_state.exp_recs = n_recs; c->host_state.exp_words = n_words;
  c->finalized = false;
  return push_state(c);
}

// ---- finalize ---------------------------------------------------------------------------------------------------------
extern "C" int kamd_ec_track_order(kamd_ctx* c, int on) {
  if (!c) return kamd::fail(-1, "kamd_ec_track_order: null context");
  if (c->host_state.st_processed != 0) return kamd::fail(-1, "kamd_ec_track_order: call before the first batch (or after kamd_ec_reset)");
  c->track_order = on != 0;
  return 0;
}

extern "C" int kamd_ec_finalize(kamd_ctx* c, kamd_ec_result* out) {
  if (!c) return kamd::fail(-1, "kamd_ec_finalize: null context");
  if (!c->has_index) return kamd::fail(-1, "kamd_ec_finalize: no index uploaded");
  HIPC(hipSetDevice(c->device));
  if (!c->ev_fin0) { HIPC(hipEventCreate(&c->ev_fin0)); HIPC(hipEventCreate(&c->ev_fin1)); }
  HIPC(hipEventRecord(c->ev_fin0, c->stream));
  if (int rc = sync_state(c)) return rc;
  DevState* dst = (DevState*)c->state.p;
  const u64 n_t = c->n_distinct_tuples;   // (the batches' tuple records were absorbed as they came: absorb_tuples)
  // size bound of the candidate stream
  c->host_state.bound_words = 0; c->host_state.cand_words = 0; c->host_state.cand_recs = 0; c->host_state.n_big = 0; c->host_state.n_huge = 0;
  if (int rc = push_state(c)) return rc;
  hipLaunchKernelGGL(k_bound_singles, dim3(grid_for(c->n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, c->ix, c->dense.as<u32>(), dst);
  if (int rc = c->tup_bound.ensure((n_t + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->tup_off.ensure((n_t + 2) * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->tup_big.ensure((n_t + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (n_t) {
    hipLaunchKernelGGL(k_bound_tuples, dim3(grid_for(n_t, BLOCK)), dim3(BLOCK), 0, c->stream, c->ix, c->tstore.as<u32>(),
                       c->ttable.as<TSlot>(), c->list.as<u64>(), n_t, c->tup_bound.as<u32>(), c->tup_big.as<u32>(), dst);
    if (int rc = exclusive_scan(c, c->tup_bound.as<u32>(), n_t, c->tup_off.as<u64>(), c->tup_off.as<u64>() + n_t)) return rc;
  }
  HIPC(hipGetLastError());
  u64 tup_words = 0;
  if (n_t) HIPC(hipMemcpyAsync(&tup_words, c->tup_off.as<u64>() + n_t, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  if (int rc = sync_state(c)) return rc;
  const u64 n_exp_w = c->exp_words_done, n_exp_r = c->host_state.exp_recs;
  const u64 bound = c->host_state.bound_words + n_exp_w;
  const u64 max_cands = c->n_ecs + n_t + n_exp_r;
  if (int rc = c->cand.ensure((bound + 2) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->cand_off.ensure((max_cands + 1) * sizeof(u64), 0, c->stream)) return rc;
  if (c->track_order) if (int rc = c->cand_key.ensure((max_cands + 1) * sizeof(u64), 0, c->stream)) return rc;
  u64* cand_key = c->track_order ? c->cand_key.as<u64>() : nullptr;
  hipLaunchKernelGGL(k_cand_singles, dim3(grid_for(c->n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, c->ix, c->dense.as<u32>(),
                     c->dense_first.as<u64>(), c->cand.as<u32>(), c->cand_off.as<u64>(), cand_key, dst);
  if (n_t && c->ix.union_mode) {
    // (cursors for tuples longer than TUPLE_CAP: only the overflow kernel produces them, so the buffer is sized when it ran)
    const bool big = c->had_overflow_items;
    if (big) if (int rc = c->overflow_scratch.ensure(n_t * (u64)TUPLE_CAP_BIG * sizeof(u32), 0, c->stream)) return rc;
    hipLaunchKernelGGL(k_resolve_union, dim3(grid_for(n_t, BLOCK)), dim3(BLOCK), 0, c->stream, c->ix, c->tstore.as<u32>(),
                       c->ttable.as<TSlot>(), c->list.as<u64>(), n_t, c->tup_off.as<u64>(), c->cand.as<u32>(), c->cand_off.as<u64>(),
                       cand_key, big ? c->overflow_scratch.as<u32>() : nullptr, dst);
  } else if (n_t) {
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(n_t * RES_LANES, RES_BLOCK)), dim3(RES_BLOCK), 0, c->stream, c->ix, c->tstore.as<u32>(),
                       c->ttable.as<TSlot>(), c->list.as<u64>(), n_t, c->tup_off.as<u64>(), c->cand.as<u32>(), c->cand_off.as<u64>(),
                       cand_key, dst);
    // the tuples whose smallest set is large (k_bound_tuples listed them): one wavefront each, the intersection out of LDS
    const u64 n_big = c->host_state.n_big, n_huge = c->host_state.n_huge;
    if (n_big) hipLaunchKernelGGL(k_resolve_big<(int)RB_CAND_BIG>, dim3(grid_for(n_big, RB_WAVES)), dim3(64 * RB_WAVES), 0, c->stream, c->ix, c->tstore.as<u32>(),
                                  c->ttable.as<TSlot>(), c->list.as<u64>(), c->tup_big.as<u32>(), n_big, c->tup_off.as<u64>(), c->cand.as<u32>(),
                                  c->cand_off.as<u64>(), cand_key, dst);
    if (n_huge) hipLaunchKernelGGL(k_resolve_big<(int)RB_CAND_HUGE>, dim3(grid_for(n_huge, RB_WAVES)), dim3(64 * RB_WAVES), 0, c->stream, c->ix, c->tstore.as<u32>(),
                                   c->ttable.as<TSlot>(), c->list.as<u64>(), c->tup_big.as<u32>() + (n_t - n_huge), n_huge, c->tup_off.as<u64>(), c->cand.as<u32>(),
                                   c->cand_off.as<u64>(), cand_key, dst);
    c->last_fin_big = n_big + n_huge;
    if (getenv("KAMD_DEBUG_FIN")) fprintf(stderr, "[kamd] finalize: %llu distinct tuples, smallest set 17..1024: %llu, beyond: %llu\n", (unsigned long long)n_t,
                                          (unsigned long long)n_big, (unsigned long long)n_huge);
  }
  HIPC(hipGetLastError());
  if (int rc = sync_state(c)) return rc;
  // the tuples' slots and record numbers follow what k_cand_singles allocated
  c->host_state.cand_words += tup_words; c->host_state.cand_recs += n_t;
  if (int rc = push_state(c)) return rc;
  if (n_exp_r) {  // sets produced by the positional filters join the candidates
    const u64 wbase = c->host_state.cand_words, rbase = c->host_state.cand_recs;
    hipLaunchKernelGGL(k_copy_words, dim3(grid_for(n_exp_w, BLOCK)), dim3(BLOCK), 0, c->stream, c->exp_stream.as<u32>(), n_exp_w,
                       c->cand.as<u32>() + wbase);
    hipLaunchKernelGGL(k_copy_offsets, dim3(grid_for(n_exp_r, BLOCK)), dim3(BLOCK), 0, c->stream, c->exp_off.as<u64>(), n_exp_r, wbase,
                       c->cand_off.as<u64>() + rbase);
    if (cand_key) hipLaunchKernelGGL(k_copy_offsets, dim3(grid_for(n_exp_r, BLOCK)), dim3(BLOCK), 0, c->stream, c->exp_key.as<u64>(), n_exp_r,
                                     0ULL, cand_key + rbase);
    HIPC(hipGetLastError());
    c->host_state.cand_words = wbase + n_exp_w; c->host_state.cand_recs = rbase + n_exp_r;
  }
  const u64 n_cand = c->host_state.cand_recs;
  // merge equal transcript sets
  c->ccap = pow2_at_least(2 * n_cand + 16);
  if (int rc = c->ctable.ensure(c->ccap * sizeof(TSlot), 0, c->stream)) return rc;
  hipLaunchKernelGGL(k_table_init, dim3(grid_for(c->ccap, BLOCK)), dim3(BLOCK), 0, c->stream, c->ctable.as<TSlot>(), c->ccap);
  if (int rc = c->clist.ensure((n_cand + 1) * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = dedup_records(c, c->cand.as<u32>(), c->cand_off.as<u64>(), 0, n_cand, c->ctable.as<TSlot>(), c->ccap, c->cand_slot,
                             c->clist.as<u64>(), cand_key ? 2 : 0, cand_key, 0, c->host_state.cand_words)) return rc;
  const u64 n_final = c->host_state.n_list;
  if (c->track_order && n_final > 1) {
    // first-occurrence order (what the reference produces at -t 1): sort the distinct sets by the index of the first item
    // that produced them.  Keys are distinct (an item yields one set), so the order is total; the sort runs on the host
    // (n_final is ~1e6 at most) and only when the caller asked for it.
    if (int rc = c->ec_first.ensure(n_final * sizeof(u64), 0, c->stream)) return rc;
    hipLaunchKernelGGL(k_final_keys, dim3(grid_for(n_final, BLOCK)), dim3(BLOCK), 0, c->stream, c->ctable.as<TSlot>(), c->clist.as<u64>(), n_final,
                       c->ec_first.as<u64>());
    std::vector<u64> keys(n_final), slots(n_final), sorted(n_final);
    HIPC(hipMemcpyAsync(keys.data(), c->ec_first.p, n_final * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(slots.data(), c->clist.p, n_final * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    std::vector<u64> perm(n_final);
    for (u64 i = 0; i < n_final; i++) perm[i] = i;
    std::sort(perm.begin(), perm.end(), [&](u64 a, u64 b) { return keys[a] != keys[b] ? keys[a] < keys[b] : slots[a] < slots[b]; });
    for (u64 i = 0; i < n_final; i++) sorted[i] = slots[perm[i]];
    HIPC(hipMemcpyAsync(c->clist.p, sorted.data(), n_final * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  // CSR
  if (int rc = c->sizes.ensure((n_final + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->ec_off.ensure((n_final + 2) * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->ec_counts.ensure((n_final + 1) * sizeof(u32), 0, c->stream)) return rc;
  u64 nnz = 0;
  if (n_final) {
    hipLaunchKernelGGL(k_final_sizes, dim3(grid_for(n_final, BLOCK)), dim3(BLOCK), 0, c->stream, c->cand.as<u32>(), c->ctable.as<TSlot>(),
                       c->clist.as<u64>(), n_final, c->sizes.as<u32>());
    if (int rc = exclusive_scan(c, c->sizes.as<u32>(), n_final, c->ec_off.as<u64>(), &dst->bound_words)) return rc;
    hipLaunchKernelGGL(k_set_last, dim3(1), dim3(64), 0, c->stream, c->ec_off.as<u64>(), n_final, &dst->bound_words);
    if (int rc = sync_state(c)) return rc;
    nnz = c->host_state.bound_words;
    if (int rc = c->ec_ids.ensure((nnz + 1) * sizeof(u32), 0, c->stream)) return rc;
    hipLaunchKernelGGL(k_final_write, dim3(grid_for(n_final * 64, BLOCK)), dim3(BLOCK), 0, c->stream, c->cand.as<u32>(), c->ctable.as<TSlot>(),
                       c->clist.as<u64>(), n_final, c->ec_off.as<u64>(), c->ec_ids.as<u32>(), c->ec_counts.as<u32>());
    HIPC(hipGetLastError());
  } else {
    HIPC(hipMemsetAsync(c->ec_off.p, 0, sizeof(u64), c->stream));
    if (int rc = c->ec_ids.ensure(sizeof(u32), 0, c->stream)) return rc;
  }
  HIPC(hipEventRecord(c->ev_fin1, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  HIPC(hipEventElapsedTime(&c->last_finalize_ms, c->ev_fin0, c->ev_fin1));
  c->last_fin_records = n_t; c->last_fin_stream_words = c->host_state.ts_words; c->last_fin_cand_words = c->host_state.cand_words;
  ++c->ec_generation;
  c->result.n_ecs = n_final; c->result.nnz = nnz;
  c->result.d_ec_off = c->ec_off.as<uint64_t>(); c->result.d_ec_ids = c->ec_ids.as<u32>(); c->result.d_counts = c->ec_counts.as<u32>();
  c->result.n_pseudoaligned = 0;  // filled by kamd_ec_download callers from the counts; kept for ABI symmetry
  c->finalized = true;
  if (out) *out = c->result;
  return 0;
}

extern "C" int kamd_ec_finalize_result(kamd_ctx* c, kamd_ec_result* out) {
  if (!c || !out) return kamd::fail(-1, "kamd_ec_finalize_result: null argument");
  if (!c->finalized) return kamd::fail(-1, "kamd_ec_finalize_result: call kamd_ec_finalize first");
  *out = c->result;
  return 0;
}

extern "C" int kamd_ec_download(kamd_ctx* c, uint64_t* ec_off, uint32_t* ec_ids, uint32_t* counts) {
  if (!c || !c->finalized) return kamd::fail(-1, "kamd_ec_download: call kamd_ec_finalize first");
  HIPC(hipSetDevice(c->device));
  HIPC(hipMemcpyAsync(ec_off, c->ec_off.p, (c->result.n_ecs + 1) * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  if (c->result.nnz) HIPC(hipMemcpyAsync(ec_ids, c->ec_ids.p, c->result.nnz * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  if (c->result.n_ecs) HIPC(hipMemcpyAsync(counts, c->ec_counts.p, c->result.n_ecs * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- ECs supplied by the caller (`quant-tcc`: src/main.cpp:2813-2900 reads the TCC matrix, KmerIndex::loadECsFromFile
// src/KmerIndex.cpp:1561-1600 the EC list; EM_lambda :2989-3000 sets `collection.counts[ec] = count` per sample) ------------------
// The context's EC result becomes the given CSR: kamd_em_run(ctx, NULL...), kamd_bootstrap(_batch) and the plan cache then work as
// after kamd_ec_finalize.  kamd_ec_set_counts replaces only the counts (same matrix: the EM plan of the previous sample is reused).
extern "C" int kamd_ec_upload(kamd_ctx* c, const uint64_t* ec_off, const uint32_t* ec_ids, const uint32_t* counts, uint64_t n_ecs) {
  if (!c || !ec_off || (n_ecs && !ec_ids)) return kamd::fail(-1, "kamd_ec_upload: null argument");
  if (ec_off[0] != 0) return kamd::fail(-1, "kamd_ec_upload: ec_off[0] must be 0");
  std::vector<std::pair<u64, u64>> by_hash((size_t)n_ecs);   // (content hash, class): equal classes end up next to each other
  for (u64 e = 0; e < n_ecs; e++) {
    if (ec_off[e + 1] <= ec_off[e]) return kamd::fail(-1, "kamd_ec_upload: empty equivalence class or offsets not increasing");
    u64 h = kamd::mix64(ec_off[e + 1] - ec_off[e]);
    for (u64 j = ec_off[e]; j < ec_off[e + 1]; j++) {
      if (j > ec_off[e] && ec_ids[j] <= ec_ids[j - 1]) return kamd::fail(-1, "kamd_ec_upload: the transcripts of an equivalence class must be sorted and distinct");
      if (c->has_index && ec_ids[j] >= c->n_targets) return kamd::fail(-1, "kamd_ec_upload: transcript id beyond the targets of the uploaded index");
      h = kamd::mix64(h ^ ec_ids[j]);
    }
    by_hash[(size_t)e] = {h, e};
  }
  // two classes with the same transcripts would be two rows of one set for the EM (the reference keys its classes by content,
  // ecmapinv: src/KmerIndex.cpp:1561-1600) and would race for the singleton slot of a transcript: refused
  std::sort(by_hash.begin(), by_hash.end());
  for (size_t i = 1; i < by_hash.size(); i++) {
    if (by_hash[i].first != by_hash[i - 1].first) continue;
    const u64 a = by_hash[i - 1].second, b = by_hash[i].second;
    if (ec_off[a + 1] - ec_off[a] == ec_off[b + 1] - ec_off[b] && std::equal(ec_ids + ec_off[a], ec_ids + ec_off[a + 1], ec_ids + ec_off[b]))
      return kamd::fail(-1, "kamd_ec_upload: equivalence classes " + std::to_string(std::min(a, b)) + " and " + std::to_string(std::max(a, b)) + " hold the same transcripts");
  }
  HIPC(hipSetDevice(c->device));
  const u64 nnz = ec_off[n_ecs];
  if (int rc = c->ec_off.ensure((n_ecs + 1) * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->ec_ids.ensure(std::max<u64>(nnz, 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->ec_counts.ensure(std::max<u64>(n_ecs, 1) * sizeof(u32), 0, c->stream)) return rc;
  HIPC(hipMemcpyAsync(c->ec_off.p, ec_off, (n_ecs + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  if (nnz) HIPC(hipMemcpyAsync(c->ec_ids.p, ec_ids, nnz * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  if (n_ecs) {
    if (counts) HIPC(hipMemcpyAsync(c->ec_counts.p, counts, n_ecs * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    else HIPC(hipMemsetAsync(c->ec_counts.p, 0, n_ecs * sizeof(u32), c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  ++c->ec_generation;
  c->result.n_ecs = n_ecs; c->result.nnz = nnz;
  c->result.d_ec_off = c->ec_off.as<uint64_t>(); c->result.d_ec_ids = c->ec_ids.as<u32>(); c->result.d_counts = c->ec_counts.as<u32>();
  c->result.n_pseudoaligned = 0;
  c->finalized = true;
  return 0;
}
extern "C" int kamd_ec_set_counts(kamd_ctx* c, const uint32_t* counts) {
  if (!c || !counts) return kamd::fail(-1, "kamd_ec_set_counts: null argument");
  if (!c->finalized) return kamd::fail(-1, "kamd_ec_set_counts: no EC result (call kamd_ec_upload or kamd_ec_finalize first)");
  HIPC(hipSetDevice(c->device));
  if (c->result.n_ecs) HIPC(hipMemcpyAsync(c->ec_counts.p, counts, c->result.n_ecs * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return 0;
}

