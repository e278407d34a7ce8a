// kamd_dev.h -- what the translation units of libkallisto_amd.so share: the device-side views of the index and of the run's state, wavefront
// helpers, the growable device buffer, the context, and the host functions one unit calls in another.
//   kamd_match.hip   kernel A (k_match_v3), k_classify, the straight-line matchers (overflow / explicit sets / fragment lengths), kamd_pseudoalign, kamd_fld_*
//   kamd_ec.hip      exact de-duplication of records and tuples, EC resolution (k_resolve*), kamd_ec_*
//   kamd_em.hip      the EM in all its forms (component-local k_em_sell, hybrid, streamed, CSR), bootstrap
//   kamd_io.hip      FASTQ text in HBM -> packed reads
//   kamd_ctx.hip     context, index upload, tuning, diagnostics, communicators and what runs over them
// The per-item semantics live in kamd_core.h (shared with the CPU emulation used by the tests).
#pragma once

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/kallisto_amd.h"
#include "kamd_core.h"
#include "kamd_fq_core.h"
#include "kamd_host.h"
#include "kamd_em_local.h"
#include "kamd_em_sell.h"

#define HIPC(x)                                                                                   \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) return kamd::fail(-100, std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace kamdi {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int BLOCK = 256;
constexpr int TUPLE_CAP = 12;        // distinct set ids kept in LDS per item; more -> overflow kernel
constexpr int TUPLE_CAP_BIG = 1024;  // per-item capacity of the overflow kernel (global scratch)
constexpr int EXPLICIT_HITS = 256;   // distinct (block, strand) pairs of a mate's hits kept for the per-hit strand filter

struct DevIndex {
  const u64* table; u64 n_buckets;
  int table_layout; u32 tag_q, tag_dsh, tag_w;   // kamd_core.h: LAYOUT_WIDE, or LAYOUT_COMPACT with its shifts
  const u32* slot_block; const u32* slot_dist;
  const u32* uec_ec;
  const u64* ec_off; const u32* ec_ids; const uint8_t* ec_nonempty;
  const u32* uec_ecn;   // uec_ec with bit 31 = "the set is non-empty": one gather instead of two dependent ones in k_classify
  const u32* onlist_bits;
  // the LARGE transcript sets (more than bm_min members: poly-A and repeat-family classes, a few hundred of them) also as bitmaps over the
  // transcripts, bm_stride words each -- what the reference's Roaring containers are for such sets (src/MinCollector.cpp:425-496, `r &= ...`):
  // membership is one bit test instead of a binary search.  ec_bm_slot[e] = the set's bitmap, or BM_NONE.  bm_min = 0xFFFFFFFF: none at all
  const u32* ec_bm_slot; const u32* bm_words; u32 bm_stride, bm_min;
  u64 n_ecs; int k;
  // positional tables (findPosition / strand filters)
  const u64* unitig_blk_off; const u32* unitig_len; const u32* blk_unitig; const u32* blk_lb; const u32* blk_ub; const u32* blk_ec;
  const u64* blk_pos_off; const u32* blk_posw; const uint8_t* blk_sense; const int32_t* target_lens;
  // D-list (second k-mer table + the dummy hit); n_dbuckets == 0: none
  const u64* dtable; u64 n_dbuckets; u64 dummy_slot; u32 dummy_uec; u32 dummy_strand;
  const u32* utext;   // 2-bit text of all unitigs (kamd_core.h: text_canon)
  int no_jump;   // kamd_quant_opts::no_jump of the run (set by the entry points that take the options)
  int union_mode;     // kamd_quant_opts::do_union: per-mate unions instead of intersections (MinCollector.cpp:163-169)
  int comprehensive;  // strand filter per hit (ProcessReads.cpp:62-82): a strand option together with --union / --no-jump
};
// the k-mer table(s) as the per-item logic sees them; partial = match()'s `partial` argument = single-end reads
__host__ __device__ inline kamd::Table make_table(const DevIndex& ix, bool partial) {
  kamd::Table t{(const uint64_t*)ix.table, ix.n_buckets};
  t.layout = (uint8_t)ix.table_layout; t.q = (uint8_t)ix.tag_q; t.dsh = (uint8_t)ix.tag_dsh; t.tagw = (uint8_t)ix.tag_w;
  t.dslots = (const uint64_t*)ix.dtable; t.n_dbuckets = ix.n_dbuckets; t.dummy_uec = ix.dummy_uec; t.dummy_slot = ix.dummy_slot;
  t.dummy_strand = ix.dummy_strand != 0; t.partial = partial && !ix.union_mode;   // --union: match(..., partial = false) (KmerIndex.cpp:1704)
  t.no_jump = ix.no_jump != 0;
  return t;
}

// the filters of processBuffer that depend on the position of the first mapping k-mer (ProcessReads.cpp:1095-1145)
struct FilterDev { int single_overhang, has_mean_fl, fl, strand, comprehensive; };

// device-resident cursors and statistics
struct DevState {
  u64 stream_words, n_recs, n_overflow, n_retry;
  u64 st_processed, st_single, st_multi;
  u64 n_list, bound_words;       // generic append cursor / size bound accumulator
  u64 n_explicit, n_explicit_big; // items whose set was changed by a positional filter (from the main / overflow kernel)
  u64 n_hit_overflow;            // explicit-set pass: reads whose hits touched more than EXPLICIT_HITS blocks
  u64 exp_words, exp_recs;       // explicit transcript-set stream
  u64 cand_words, cand_recs;     // candidate transcript-set stream
  u64 tl_n, ts_words, tl_fail;   // distinct tuples so far (entries of the tuple list), words of the tuple store, records of a batch that found no slot
  u64 n_big, n_huge;             // kamd_ec_finalize: distinct tuples whose smallest set has 17 .. 1024 / more than 1024 members (k_resolve_big's work lists)
};

// counters of kernel A (their own struct: chunks of kernel A run on one stream while another copies DevState to and fro)
struct DevStatsA { u64 probes, bucket_reads, raw_words, text_hits, wave_iters, lane_iters; };

struct TSlot { u64 tag, owner, count, first; };  // first: smallest first-occurrence key of the merged records (candidate table)

// ------------------------------------------------------------------------------------------------------------------
// wavefront helpers (64 lanes)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { u32 t = __shfl_up(v, d, 64); if (lane_id() >= d) v += t; }
  return v;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
  return ((u64)(u32)__shfl((int)(v >> 32), src, 64) << 32) | (u32)__shfl((int)v, src, 64);
}
__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;  // valid in lane 0
}

__device__ __forceinline__ u64 rec_hash(const u32* w, u32 n, u64 seed) {
  u64 h = kamd::mix64(seed ^ (0x9e3779b97f4a7c15ULL * (n + 1)));
  if (n > 64) {
    // long records (candidate sets of thousands of transcripts): four interleaved chains -- one chain is 3 500 dependent multiplies for a poly-A
    // class, and the thread's wavefront waits for it (k_rec_dedup 9.8 ms per 30 M stress pairs).  A record takes this form by its length alone.
    u64 h0 = h, h1 = h ^ 0x9e3779b97f4a7c15ULL, h2 = h ^ 0xc2b2ae3d27d4eb4fULL, h3 = h ^ 0x165667b19e3779f9ULL;
    u32 i = 0;
    for (; i + 4 <= n; i += 4) { h0 = kamd::mix64(h0 ^ w[i]); h1 = kamd::mix64(h1 ^ w[i + 1]); h2 = kamd::mix64(h2 ^ w[i + 2]); h3 = kamd::mix64(h3 ^ w[i + 3]); }
    for (; i < n; i++) h0 = kamd::mix64(h0 ^ w[i]);
    return kamd::mix64(kamd::mix64(h0 ^ (h1 + 1)) ^ kamd::mix64(h2 ^ (h3 + 3))) | 1ULL;
  }
  for (u32 i = 0; i < n; i++) h = kamd::mix64(h ^ w[i]);
  return h | 1ULL;  // 0 is the empty tag
}

__device__ __forceinline__ bool onlisted(const u32* bits, u32 t) { return (bits[t >> 5] >> (t & 31)) & 1u; }
__device__ __forceinline__ bool set_contains(const u32* ids, u32 n, u32 x) {
  u32 lo = 0, hi = n;
  while (lo < hi) { u32 mid = (lo + hi) >> 1; if (ids[mid] < x) lo = mid + 1; else hi = mid; }
  return lo < n && ids[lo] == x;
}
constexpr u32 BM_NONE = 0xFFFFFFFFu;
constexpr u32 BM_MIN_MEMBERS = 128;           // sets with more members get a bitmap (kamd_index_upload) ...
constexpr size_t BM_MAX_BYTES = 256u << 20;   // ... the largest first, while they fit this much HBM
__device__ __forceinline__ bool bitmap_has(const DevIndex& ix, u32 slot, u32 x) { return (ix.bm_words[(u64)slot * ix.bm_stride + (x >> 5)] >> (x & 31)) & 1u; }
// is transcript x a member of index set e (its ids at ec_ids + off, sz of them)?
__device__ __forceinline__ bool set_has(const DevIndex& ix, u32 e, u64 off, u32 sz, u32 x) {
  if (sz > ix.bm_min) { const u32 s = ix.ec_bm_slot[e]; if (s != BM_NONE) return bitmap_has(ix, s, x); }
  return set_contains(ix.ec_ids + off, sz, x);
}
// f(tr) for every on-listed member of the item's transcript set (intersection of its sets, or the per-mate unions intersected
// with --union), in increasing order; thread-serial.  cur: ecs.n words of scratch for the --union merge.
template <class F>
__device__ __forceinline__ void for_each_member(const DevIndex& ix, const kamd::EcList& ecs, u32* cur, F&& f) {
  const kamd::SetTables st{(const uint64_t*)ix.ec_off, ix.ec_ids};
  kamd::for_each_in_set(st, ecs, ix.union_mode != 0, cur, [&](u32 x) { if (onlisted(ix.onlist_bits, x)) f(x); });
}
__device__ __forceinline__ kamd::PosTables pos_tables(const DevIndex& ix) {
  return kamd::PosTables{(const uint64_t*)ix.unitig_blk_off, ix.unitig_len, ix.blk_unitig, ix.blk_lb, ix.blk_ub, ix.blk_ec,
                         (const uint64_t*)ix.blk_pos_off, ix.blk_posw, ix.blk_sense, (const uint64_t*)ix.ec_off, ix.ec_ids,
                         ix.target_lens, ix.k};
}
__device__ __forceinline__ kamd::FirstHit first_hit(const DevIndex& ix, const kamd::MateInfo& m) {
  kamd::FirstHit h; h.valid = m.n_hits > 0;
  h.block = h.valid ? ix.slot_block[m.first_slot] : 0u; h.dist = h.valid ? ix.slot_dist[m.first_slot] : 0u;
  h.strand = m.first_strand; h.pos = m.first_pos;
  return h;
}
__device__ __forceinline__ kamd::FilterCfg item_filter_cfg(const FilterDev& fd, bool paired, const kamd::MateInfo& m0,
                                                           const kamd::MateInfo& m1) {
  kamd::FilterCfg cfg;
  cfg.fraglen = !fd.single_overhang && fd.has_mean_fl && (!paired || m0.n_hits == 0 || m1.n_hits == 0);  // ProcessReads.cpp:1095
  cfg.fl = fd.fl; cfg.strand = fd.strand;
  cfg.comprehensive = fd.comprehensive != 0 && fd.strand != 0;   // (hits1 stays empty: see needs_hit_list)
  return cfg;
}
// per-hit strand filter and mate 2 without hits: the outcome depends on every hit of mate 1, which only the explicit-set pass
// collects (kamd_core.h FilterCfg) -- such items always take that pass
__device__ __forceinline__ bool needs_hit_list(const FilterDev& fd, const kamd::MateInfo& m1) {
  return fd.comprehensive != 0 && fd.strand != 0 && m1.n_hits == 0;
}
// 0 = the filters leave the set unchanged, 1 = they empty it, 2 = they change it (*kept = new size)
__device__ __forceinline__ int filter_outcome(const DevIndex& ix, const FilterDev& fd, bool paired, const kamd::MateInfo& m0,
                                              const kamd::MateInfo& m1, const kamd::EcList& ecs, u32* kept, u32* cur) {
  const kamd::FilterCfg cfg = item_filter_cfg(fd, paired, m0, m1);
  if (!cfg.fraglen && !cfg.strand) return 0;
  if (needs_hit_list(fd, m1)) {   // decided by the explicit-set pass; the record is at most the unfiltered set
    const kamd::SetTables st{(const uint64_t*)ix.ec_off, ix.ec_ids};
    *kept = (u32)kamd::set_size_bound(st, ecs.e, ecs.n, ix.union_mode != 0);
    return 2;
  }
  const kamd::PosTables pt = pos_tables(ix);
  const kamd::FirstHit h0 = first_hit(ix, m0), h1 = first_hit(ix, m1);
  u32 total = 0, keep = 0;
  for_each_member(ix, ecs, cur, [&](u32 tr) { ++total; keep += kamd::keep_transcript(pt, cfg, h0, h1, tr) ? 1u : 0u; });
  *kept = keep;
  return keep == total ? 0 : (keep == 0 ? 1 : 2);
}
struct DBuf {
  void* p = nullptr; size_t cap = 0;
  // grow to at least `bytes`; `keep` bytes of the old contents are preserved
  int ensure(size_t bytes, size_t keep, hipStream_t s) {
    if (bytes <= cap) return 0;
    size_t ncap = std::max(bytes, 2 * cap);   // (doubling: a stream that grows batch by batch is reallocated O(log n) times)
    void* np = nullptr;
    HIPC(hipMalloc(&np, ncap));
    if (keep && p) HIPC(hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, s));
    if (p) { HIPC(hipStreamSynchronize(s)); HIPC(hipFree(p)); }
    p = np; cap = ncap;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

inline unsigned grid_for(u64 n, int block) { return (unsigned)((n + block - 1) / block); }

struct SellCache;
void sell_cache_free(SellCache*);
}  // namespace kamdi
using namespace kamdi;
namespace {   // (two trivial kernels several units launch: a copy per unit)
// append the explicit records to the candidate stream (offsets rebased)
__global__ void k_copy_words(const u32* __restrict__ src, u64 n, u32* dst) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ void k_copy_offsets(const u64* __restrict__ src, u64 n, u64 base, u64* dst) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] + base;
}
}  // namespace

struct kamd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool has_index = false;
  DevIndex ix{};
  std::vector<void*> index_allocs;
  u64 n_ecs = 0, n_targets = 0;
  u32 n_set_bitmaps = 0;   // index sets that also exist as bitmaps (DevIndex::bm_words)
  DBuf dense, stream_buf, rec_off, overflow_items, overflow_scratch, state, rec_slot, retry, ttable, list;
  DBuf cand, cand_off, cand_slot, ctable, clist, sizes, block_sums, tup_bound, tup_off, tup_big;
  DBuf raw2, overflow_left, stats_b;   // the second pass over the items whose class list overflowed: its raw records, what overflows again, its counters
  u64 overflow_second_total = 0;       // since kamd_ec_reset: overflow items the second pass took care of
  // the second pass BESIDE the absorption of the batch's other tuple records (kamd_match.hip, overflow_side_launch): a stream of its own, counters of its
  // own (a DevState that only its kernels touch: push_state / sync_state move the main one wholesale), two pinned images (what it starts from, what it ended with)
  DBuf ov_state; DevState* ov_pin = nullptr; hipStream_t ov_stream = nullptr; hipEvent_t ov_ev_in = nullptr, ov_ev_done = nullptr, ov_ev_t0 = nullptr, ov_ev_t1 = nullptr;
  bool ov_side_pending = false;
  u64 last_fin_big = 0;
  DBuf explicit_items, explicit_items_big, exp_stream, exp_off, exp_scratch, bs_cp, bs_samp, raw, dense_first, exp_key, cand_key, ec_first;
  DBuf ec_off, ec_ids, ec_counts;
  DBuf em_alpha, em_next, em_eff, em_state, em_cn, em_colcnt, em_coloff, em_colrow, em_segoff, em_segt, em_partial, em_a0, em_a1, em_single, em_actflag, em_actpos, em_active;
  DBuf pm_a, pm_b, pm_rank;      // streamed EM: re-layout arenas, scratch of the rows' stable renumbering
  DBuf eml_tmp, ems_tmp, ems_plan, ems_maps;   // component-local EM: set-up scratch, sliced-ELLPACK plan, what a replicate re-uses
  SellCache* sell_cache = nullptr;   // plan of the finalized matrix, kept for bootstrap replicates (allocated on first use)
  u64 ec_generation = 0;         // bumped whenever the finalized EC result is rebuilt (plans of an older result are stale)
  int last_em_plan_cached = 0;
  DBuf fq_tiles, fq_nlpos[2], fq_recs, fq_res, fq_words, fq_len;   // kamd_fastq_unit_pack: scratch and the packed batch it returns
  void* fq_host = nullptr;       // pinned FqResult
  u64 fq_batch_reads = 0; int32_t fq_batch_max_len = 0, fq_batch_files = 0;   // reads parsed into fq_recs since the last kamd_fastq_batch_pack
  DBuf fld_tl, fld_card, fld_scratch, fld_items, fld_cand;
  void* fld_host = nullptr; u64 fld_host_cap = 0;   // pinned staging of kamd_fld_from_batch
  // kamd_fld_prefetch: the first prefix of a batch, launched on a side stream so that it overlaps kernel A
  hipStream_t fld_stream = nullptr; hipEvent_t fld_ev = nullptr, fld_ev_in = nullptr;
  struct { const uint32_t* w = nullptr; const uint16_t* l = nullptr; u64 n = 0; int32_t max_len = 0, strand = 0, so = 0; bool valid = false; } fld_pending;
  // a prefetch that waits for kernel A of the same batch to finish (launched by align_batch behind kernel A: the fragment-length kernels then
  // run beside k_classify / k_tup_absorb, which leave most of the memory system's request rate unused, instead of underneath kernel A, which
  // lives on it)
  struct { const uint32_t* w = nullptr; const uint16_t* l = nullptr; u64 n = 0; int32_t max_len = 0, strand = 0, so = 0, comp = 0; bool valid = false; } fld_deferred;
  DBuf pt_label, pt_flag, pt_len, pt_rowpos, pt_nnzpos, pt_off, pt_ids, pt_counts, pt_wcounts, pt_hist, pt_ck_alpha, pt_ck_a;
  DevState host_state{};
  DevState* state_pin = nullptr;   // pinned staging of the read-backs (sync_state)
  u64 tcap = 0, ccap = 0;        // slots of the tuple table (persistent over the batches of a run) / the candidate table
  DBuf tstore;                   // the tuple store: the distinct tuple records of the run, compact
  bool ttable_clean = false;     // the tuple table holds no entries (just initialised)
  u64 recs_total = 0;            // records (items) of the batches absorbed so far: first-occurrence keys
  u64 multi_before = 0;          // st_multi before the current batch
  u64 n_distinct_tuples = 0;
  float last_absorb_ms = 0.f; hipEvent_t ev_ab0 = nullptr, ev_ab1 = nullptr;
  kamd_ec_result result{};
  bool finalized = false;
  u64 exp_words_done = 0;        // words of the explicit-set stream actually written
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_align_ms = 0.f, last_em_ms = 0.f, last_classify_ms = 0.f;
  hipEvent_t ev2 = nullptr, ev3 = nullptr, ev_fin0 = nullptr, ev_fin1 = nullptr;
  DBuf stats_a;                  // DevStatsA: kernel A's counters
  hipStream_t al_stream = nullptr; hipEvent_t al_ev_in = nullptr, al_ev_out = nullptr; std::vector<hipEvent_t> al_ev_chunk;   // kamd_pseudoalign's side stream (WorkStream)
  float last_finalize_ms = 0.f; u64 last_fin_records = 0, last_fin_stream_words = 0, last_fin_cand_words = 0;
  hipStream_t em_stream = nullptr;
  hipStream_t em_side_stream = nullptr; hipEvent_t em_ev_fork = nullptr, em_ev_join = nullptr;   // component-local EM: the small size class runs beside the large one
  void* em_pin = nullptr; size_t em_pin_bytes = 0;   // component-local EM: pinned, mapped host memory (change counts the kernels publish, result staging)
  DBuf em_clk;                                       // diagnostic phase clocks (KAMD_EM_CLK)
  // hybrid EM (components beyond a workgroup's LDS beside the LDS form): the two sub-matrices, the streamed plan's arenas, its vectors
  DBuf hy_sub, hy_a, hy_b, hy_x, hy_maps;
  DBuf hy_gb[6];   // the blocked form of the oversized components: two set-up scratch arenas, then per direction the entry stream and the pieces' arrays
  hipStream_t hy_giant_stream = nullptr;   // the oversized components' kernels (k_em_sell stays on the context stream)
  hipEvent_t hy_ev_sell = nullptr, hy_ev_giant = nullptr;
  uint64_t last_em_max_comp_nnz = 0, last_em_giant_nnz = 0, last_em_giant_rows = 0, last_em_giant_tr = 0;
  uint32_t last_em_giant_chunks = 0; uint64_t last_em_giant_pieces = 0; int last_em_graph_fallback = 0; float last_em_plan_ms = 0.f;
  float last_merge_ms = 0.f, em_coll_ms = 0.f; uint32_t em_coll_n = 0; hipEvent_t ev_mg0 = nullptr, ev_mg1 = nullptr;   // several ranks: kamd_ec_allreduce (HIP events), the EM's collectives (host wall, the host waits for each)
  const uint32_t* labels_override = nullptr;   // em_local_setup_device takes these component labels instead of computing them (the hybrid's sub-matrix: same components)
  bool em_prefer_hybrid = false;               // the last matrix of this context needed the hybrid: the next plan starts there
  uint64_t overflow_total = 0; float overflow_ms = 0.f; hipEvent_t ev_ov0 = nullptr, ev_ov1 = nullptr;   // since kamd_ec_reset: items of the overflow kernel, its time
  int items_per_wave = 1024, refill_min = 8;   // (copies of tune.*, see apply_tuning)
  kamd_tuning tune{};
  bool track_order = false;  // kamd_ec_track_order: finalize emits the sets in first-occurrence order
  bool had_overflow_items = false;   // some item went through the overflow kernel (tuples of more than TUPLE_CAP sets may exist)
  int n_cus = 0, last_em_k = 0; unsigned last_em_grid = 0, last_em_lds = 0;
  uint64_t last_em_iters = 0, last_em_nnz = 0, last_em_nnz_multi = 0, last_em_nseg = 0, last_em_necs = 0;
  std::vector<struct kamd_comm*> comms;   // communicators bound to this context (detached by kamd_ctx_destroy, so that either may go first)
};

namespace kamdi {
// kamd_ctx.hip
int sync_state(kamd_ctx* c);
int push_state(kamd_ctx* c);
void apply_quant_opts(kamd_ctx* c, const kamd_quant_opts* o);
void apply_tuning(kamd_ctx* c);
void comm_detach_all(kamd_ctx* c);
// kamd_ec.hip
int exclusive_scan(kamd_ctx* c, const u32* sizes, u64 n, u64* out, u64* d_total);
int tuples_clear(kamd_ctx* c);
int absorb_tuples(kamd_ctx* c, const u32* batch, const u64* rec_off, u64 n, u64 batch_words, u64 key_base, u64 n_tuple_bound,
                  const u64* first_idx = nullptr, u32 fixed_stride = 0, u64 item0 = 0);
template <class T>
int upload(kamd_ctx* c, const T* host, size_t n, const T** dev) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  HIPC(hipMalloc(&p, bytes));
  c->index_allocs.push_back(p);
  if (n) HIPC(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
  *dev = reinterpret_cast<const T*>(p);
  return 0;
}
}  // namespace kamdi
