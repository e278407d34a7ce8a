// kamd_ctx.hip -- context, tuning, index upload, diagnostics, communicators (RCCL by dlopen / callbacks), kamd_ec_allreduce, kamd_em_run_comm, kamd_quant_batches
#include "kamd_dev.h"

// ---- tuning --------------------------------------------------------------------------------------------------------------
namespace {
void tuning_defaults(kamd_tuning* t) {
  memset(t, 0, sizeof *t);
  t->text_verify = 1; t->items_per_wave = 1024; t->refill_min = 8; t->lds_pad = -1;
  t->em_form = 3; t->em_local_block = 1024; t->em_group_div = -1; t->em_split_len = 16; t->em_small_nnz = -1; t->em_entries_per_lane = -1; t->em_windowed = 2; t->em_graph = 1; t->em_row_lanes = 4;
  t->em_fin_blocks = 1024; t->dedup_form = 2; t->align_chunks = -1; t->em_reg_slices = 1;
  t->em_hybrid = 1; t->overflow_second_pass = 1; t->em_giant_nnz = -1; t->em_blocked = 1;
}
// 0 = keep; values outside a field's range are ignored
void tuning_merge(kamd_tuning* t, const kamd_tuning& n) {
  if (n.text_verify == 1 || n.text_verify == 2) t->text_verify = n.text_verify;
  if (n.dedup_form == 1 || n.dedup_form == 2) t->dedup_form = n.dedup_form;
  if (n.items_per_wave >= 64) t->items_per_wave = n.items_per_wave;
  if (n.refill_min >= 1 && n.refill_min <= 64) t->refill_min = n.refill_min;
  if (n.lds_pad != 0) t->lds_pad = n.lds_pad < 0 ? -1 : n.lds_pad;
  if (n.em_form >= 1 && n.em_form <= 3) t->em_form = n.em_form;
  if (n.em_local_block == 128 || n.em_local_block == 256 || n.em_local_block == 512 || n.em_local_block == 1024) t->em_local_block = n.em_local_block;
  if (n.em_split_len >= 1 && n.em_split_len <= 64) t->em_split_len = n.em_split_len;
  if (n.em_group_div >= 1 && n.em_group_div <= 1024) t->em_group_div = n.em_group_div;
  else if (n.em_group_div < 0) t->em_group_div = -1;
  if (n.em_small_nnz != 0) t->em_small_nnz = n.em_small_nnz < 0 ? -1 : std::min(n.em_small_nnz, 4096);
  if (n.em_entries_per_lane != 0) t->em_entries_per_lane = n.em_entries_per_lane < 0 ? -1 : n.em_entries_per_lane;
  if (n.em_windowed == 1 || n.em_windowed == 2) t->em_windowed = n.em_windowed;
  if (n.em_graph == 1 || n.em_graph == 2) t->em_graph = n.em_graph;
  if (n.em_row_lanes == 2 || n.em_row_lanes == 4 || n.em_row_lanes == 8) t->em_row_lanes = n.em_row_lanes;
  if (n.em_fin_blocks >= 64) t->em_fin_blocks = n.em_fin_blocks;
  if (n.align_chunks != 0) t->align_chunks = n.align_chunks < 0 ? -1 : std::min(n.align_chunks, 64);
  if (n.em_reg_slices == 1 || n.em_reg_slices == 2) t->em_reg_slices = n.em_reg_slices;
  if (n.em_hybrid == 1 || n.em_hybrid == 2) t->em_hybrid = n.em_hybrid;
  if (n.overflow_second_pass >= 1 && n.overflow_second_pass <= 3) t->overflow_second_pass = n.overflow_second_pass;
  if (n.em_blocked == 1 || n.em_blocked == 2) t->em_blocked = n.em_blocked;
  if (n.em_giant_nnz != 0) t->em_giant_nnz = n.em_giant_nnz < 0 ? -1 : std::max(n.em_giant_nnz, 8);
}
// experiments: the same knobs from ONE environment variable, read once when a context is created --
//   KAMD_TUNE="em_form=streamed,em_graph=0,items_per_wave=512"   (field names of kamd_tuning; on/off fields take 1 / 0; em_form also by name)
// Unknown names are reported on stderr and ignored.
void tuning_from_env(kamd_tuning* t) {
  const char* e = getenv("KAMD_TUNE");
  if (!e || !*e) return;
  kamd_tuning n; memset(&n, 0, sizeof n);
  struct Field { const char* name; int32_t kamd_tuning::*p; bool onoff; };
  static const Field fields[] = {
    {"text_verify", &kamd_tuning::text_verify, true}, {"items_per_wave", &kamd_tuning::items_per_wave, false}, {"refill_min", &kamd_tuning::refill_min, false},
    {"lds_pad", &kamd_tuning::lds_pad, false}, {"em_form", &kamd_tuning::em_form, false}, {"em_entries_per_lane", &kamd_tuning::em_entries_per_lane, false},
    {"em_windowed", &kamd_tuning::em_windowed, true}, {"em_graph", &kamd_tuning::em_graph, true}, {"em_row_lanes", &kamd_tuning::em_row_lanes, false},
    {"em_fin_blocks", &kamd_tuning::em_fin_blocks, false}, {"em_local_block", &kamd_tuning::em_local_block, false}, {"em_group_div", &kamd_tuning::em_group_div, false},
    {"em_split_len", &kamd_tuning::em_split_len, false}, {"dedup_form", &kamd_tuning::dedup_form, false}, {"align_chunks", &kamd_tuning::align_chunks, false},
    {"em_small_nnz", &kamd_tuning::em_small_nnz, false}, {"em_reg_slices", &kamd_tuning::em_reg_slices, true}, {"em_hybrid", &kamd_tuning::em_hybrid, true},
    {"overflow_second_pass", &kamd_tuning::overflow_second_pass, false}, {"em_giant_nnz", &kamd_tuning::em_giant_nnz, false}, {"em_blocked", &kamd_tuning::em_blocked, true}};
  std::string all(e);
  for (size_t pos = 0; pos < all.size();) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    const std::string kv = all.substr(pos, end - pos);
    pos = end + 1;
    const size_t eq = kv.find('=');
    if (eq == std::string::npos) continue;
    const std::string key = kv.substr(0, eq), val = kv.substr(eq + 1);
    bool known = false;
    for (const Field& f : fields) {
      if (key != f.name) continue;
      known = true;
      int32_t v = atoi(val.c_str());
      if (key == "em_form") v = val == "streamed" ? 1 : val == "csr" ? 2 : val == "local" ? 3 : v;
      if (key == "overflow_second_pass" && v == 0) v = 2;   // (1 = beside the absorption, 3 = after it, 0 / 2 = off)
      n.*(f.p) = f.onoff ? (v != 0 ? 1 : 2) : v;
    }
    if (!known) fprintf(stderr, "[kallisto_amd] KAMD_TUNE: unknown field '%s' ignored\n", key.c_str());
  }
  tuning_merge(t, n);
}
}  // namespace
namespace kamdi {
// the options of the run that the per-item logic reads from the device index
void apply_quant_opts(kamd_ctx* c, const kamd_quant_opts* o) {
  c->ix.no_jump = o->no_jump ? 1 : 0;
  c->ix.union_mode = o->do_union ? 1 : 0;
  c->ix.comprehensive = (o->strand != 0 && (o->no_jump || o->do_union)) ? 1 : 0;   // ProcessReads.cpp:1139-1140
}
void apply_tuning(kamd_ctx* c) {
  c->items_per_wave = c->tune.items_per_wave; c->refill_min = c->tune.refill_min;
}
int sync_state(kamd_ctx* c) {
  // through pinned memory: a device-to-host copy into pageable memory is staged by the runtime (a few tens of microseconds per call, and a
  // step reads its sizes back about twenty times); into pinned memory it is one DMA and the wait for it
  if (!c->state_pin) HIPC(hipHostMalloc((void**)&c->state_pin, sizeof(DevState), hipHostMallocDefault));
  HIPC(hipMemcpyAsync(c->state_pin, c->state.p, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  memcpy(&c->host_state, c->state_pin, sizeof(DevState));
  return 0;
}
int push_state(kamd_ctx* c) {
  HIPC(hipMemcpyAsync(c->state.p, &c->host_state, sizeof(DevState), hipMemcpyHostToDevice, c->stream));
  return 0;
}
}  // namespace kamdi

// ======================================================================================================================
// C ABI
// ======================================================================================================================
extern "C" uint32_t kamd_abi_version(void) { return KAMD_ABI_VERSION; }
extern "C" int kamd_ctx_create(int device, void* hip_stream, kamd_ctx** out) {
  if (!out) return kamd::fail(-1, "kamd_ctx_create: null output pointer");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return kamd::fail(-102, "kamd_ctx_create: no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return kamd::fail(-1, "kamd_ctx_create: bad device ordinal");
  HIPC(hipSetDevice(device));
  kamd_ctx* c = new kamd_ctx;
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  // (the launch shapes of k_classify and of kernel A's second pass follow the CU count from the first batch on -- the EM used to look it up, after them)
  { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 256; c->n_cus = v; }
  if (c->state.ensure(sizeof(DevState), 0, c->stream)) { delete c; return -100; }
  memset(&c->host_state, 0, sizeof c->host_state);
  if (push_state(c)) { delete c; return -100; }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev2) != hipSuccess || hipEventCreate(&c->ev3) != hipSuccess) { delete c; return kamd::fail(-100, "hipEventCreate failed"); }
  if (c->stats_a.ensure(sizeof(DevStatsA), 0, c->stream) || hipMemsetAsync(c->stats_a.p, 0, sizeof(DevStatsA), c->stream) != hipSuccess) { delete c; return -100; }
  tuning_defaults(&c->tune);
  tuning_from_env(&c->tune);
  apply_tuning(c);
  *out = c;
  return 0;
}

extern "C" int kamd_ctx_tune(kamd_ctx* c, const kamd_tuning* t) {
  if (!c || !t) return kamd::fail(-1, "kamd_ctx_tune: null argument");
  tuning_merge(&c->tune, *t);
  apply_tuning(c);
  return 0;
}
extern "C" int kamd_ctx_get_tuning(const kamd_ctx* c, kamd_tuning* out) {
  if (!c || !out) return kamd::fail(-1, "kamd_ctx_get_tuning: null argument");
  *out = c->tune;
  return 0;
}

extern "C" void kamd_ctx_destroy(kamd_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  comm_detach_all(c);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev2) (void)hipEventDestroy(c->ev2);
  if (c->ev3) (void)hipEventDestroy(c->ev3);
  if (c->al_stream) { (void)hipStreamSynchronize(c->al_stream); (void)hipStreamDestroy(c->al_stream); }
  if (c->al_ev_in) (void)hipEventDestroy(c->al_ev_in);
  if (c->al_ev_out) (void)hipEventDestroy(c->al_ev_out);
  for (hipEvent_t e : c->al_ev_chunk) if (e) (void)hipEventDestroy(e);
  if (c->ev_fin0) (void)hipEventDestroy(c->ev_fin0);
  if (c->ev_fin1) (void)hipEventDestroy(c->ev_fin1);
  if (c->ev_ab0) (void)hipEventDestroy(c->ev_ab0);
  if (c->ev_ab1) (void)hipEventDestroy(c->ev_ab1);
  if (c->em_stream) (void)hipStreamDestroy(c->em_stream);
  if (c->em_side_stream) { (void)hipStreamSynchronize(c->em_side_stream); (void)hipStreamDestroy(c->em_side_stream); }
  if (c->em_ev_fork) (void)hipEventDestroy(c->em_ev_fork);
  if (c->em_ev_join) (void)hipEventDestroy(c->em_ev_join);
  if (c->em_pin) (void)hipHostFree(c->em_pin);
  if (c->state_pin) (void)hipHostFree(c->state_pin);
  c->em_clk.release();
  if (c->hy_giant_stream) { (void)hipStreamSynchronize(c->hy_giant_stream); (void)hipStreamDestroy(c->hy_giant_stream); }
  if (c->hy_ev_sell) (void)hipEventDestroy(c->hy_ev_sell);
  if (c->hy_ev_giant) (void)hipEventDestroy(c->hy_ev_giant);
  if (c->ev_mg0) (void)hipEventDestroy(c->ev_mg0);
  if (c->ev_mg1) (void)hipEventDestroy(c->ev_mg1);
  if (c->ev_ov0) (void)hipEventDestroy(c->ev_ov0);
  if (c->ev_ov1) (void)hipEventDestroy(c->ev_ov1);
  if (c->ov_stream) { (void)hipStreamSynchronize(c->ov_stream); (void)hipStreamDestroy(c->ov_stream); }
  for (hipEvent_t e : {c->ov_ev_in, c->ov_ev_done, c->ov_ev_t0, c->ov_ev_t1}) if (e) (void)hipEventDestroy(e);
  if (c->ov_pin) (void)hipHostFree(c->ov_pin);
  c->ov_state.release();
  for (DBuf* b : {&c->hy_sub, &c->hy_a, &c->hy_b, &c->hy_x, &c->hy_maps, &c->hy_gb[0], &c->hy_gb[1], &c->hy_gb[2], &c->hy_gb[3], &c->hy_gb[4], &c->hy_gb[5]}) b->release();
  if (c->fld_stream) { (void)hipStreamSynchronize(c->fld_stream); (void)hipStreamDestroy(c->fld_stream); }
  if (c->fld_ev) (void)hipEventDestroy(c->fld_ev);
  if (c->fld_ev_in) (void)hipEventDestroy(c->fld_ev_in);
  if (c->fld_host) (void)hipHostFree(c->fld_host);
  if (c->fq_host) (void)hipHostFree(c->fq_host);
  if (c->sell_cache) sell_cache_free(c->sell_cache);
  for (void* p : c->index_allocs) (void)hipFree(p);
  for (DBuf* b : {&c->dense, &c->stream_buf, &c->rec_off, &c->overflow_items, &c->overflow_scratch, &c->state, &c->rec_slot,
                  &c->retry, &c->ttable, &c->tstore, &c->stats_a, &c->list, &c->cand, &c->cand_off, &c->cand_slot, &c->ctable, &c->tup_bound, &c->tup_off, &c->tup_big, &c->raw2, &c->overflow_left, &c->stats_b, &c->clist, &c->sizes, &c->explicit_items,
                  &c->explicit_items_big, &c->exp_stream, &c->exp_off, &c->exp_scratch, &c->bs_cp, &c->bs_samp, &c->raw, &c->dense_first, &c->exp_key, &c->cand_key, &c->ec_first, &c->pm_a, &c->pm_b, &c->pm_rank, &c->eml_tmp, &c->ems_tmp, &c->ems_plan, &c->ems_maps, &c->fld_tl, &c->fld_card, &c->fld_scratch, &c->fld_items, &c->fld_cand,
                  &c->block_sums, &c->ec_off, &c->ec_ids, &c->ec_counts, &c->em_alpha, &c->em_next, &c->em_eff,
                  &c->em_state, &c->em_cn, &c->em_colcnt, &c->em_coloff, &c->em_colrow,
                  &c->em_segoff, &c->em_segt, &c->em_partial, &c->em_a0, &c->em_a1, &c->em_single, &c->em_actflag, &c->em_actpos, &c->em_active, &c->pt_label, &c->pt_flag, &c->pt_len,
                  &c->pt_rowpos, &c->pt_nnzpos, &c->pt_off, &c->pt_ids, &c->pt_counts, &c->pt_wcounts, &c->pt_hist, &c->pt_ck_alpha,
                  &c->pt_ck_a, &c->fq_tiles, &c->fq_nlpos[0], &c->fq_nlpos[1], &c->fq_recs, &c->fq_res, &c->fq_words, &c->fq_len})
    b->release();
  delete c;
}

namespace {
// bitmap of large set which[blockIdx.x]: one bit per member (kamd_dev.h DevIndex::bm_words)
__global__ void k_bm_fill(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, const u32* __restrict__ which, u32 stride, u32* words) {
  const u32 e = which[blockIdx.x];
  u32* w = words + (u64)blockIdx.x * stride;
  for (u64 j = ec_off[e] + threadIdx.x; j < ec_off[e + 1]; j += blockDim.x) { const u32 x = ec_ids[j]; atomicOr(&w[x >> 5], 1u << (x & 31)); }
}
}  // namespace
extern "C" int kamd_index_upload(kamd_ctx* c, const kamd_index* hix) {
  if (!c || !hix) return kamd::fail(-1, "kamd_index_upload: null argument");
  HIPC(hipSetDevice(c->device));
  kamd_index_view v;
  if (int rc = kamd_index_get_view(hix, &v)) return rc;
  for (void* p : c->index_allocs) (void)hipFree(p);
  c->index_allocs.clear();
  DevIndex d{};
  d.k = v.k; d.n_buckets = v.n_buckets; d.n_ecs = v.n_ecs;
  d.table_layout = (int)v.table_layout; d.tag_q = v.tag_q; d.tag_dsh = v.tag_dsh; d.tag_w = v.tag_w;
  const u64 slots = (v.n_buckets + v.pad_buckets) * v.slots_per_bucket;
  if (int rc = upload(c, (const u64*)v.table, (size_t)(v.n_buckets + v.pad_buckets) * 8, &d.table)) return rc;
  if (int rc = upload(c, v.slot_block, slots, &d.slot_block)) return rc;
  if (int rc = upload(c, v.slot_dist, slots, &d.slot_dist)) return rc;
  if (int rc = upload(c, v.uec_ec, v.n_uec, &d.uec_ec)) return rc;
  if (int rc = upload(c, (const u64*)v.ec_off, v.n_ecs + 1, &d.ec_off)) return rc;
  if (int rc = upload(c, v.ec_ids, v.ec_nnz, &d.ec_ids)) return rc;
  std::vector<uint8_t> ne(v.n_ecs + 1);
  for (u64 e = 0; e < v.n_ecs; e++) ne[e] = v.ec_off[e + 1] > v.ec_off[e];
  if (int rc = upload(c, ne.data(), v.n_ecs, &d.ec_nonempty)) return rc;
  std::vector<u32> ecn(v.n_uec + 1);
  for (u64 u = 0; u < v.n_uec; u++) ecn[u] = v.uec_ec[u] | (ne[v.uec_ec[u]] ? 0x80000000u : 0u);
  if (int rc = upload(c, ecn.data(), v.n_uec, &d.uec_ecn)) return rc;
  if (int rc = upload(c, v.onlist_bits, v.onlist_words, &d.onlist_bits)) return rc;
  {
    // bitmaps of the large transcript sets (kamd_dev.h DevIndex): the largest first while they fit BM_MAX_BYTES
    const u64 n_ids = v.n_targets + v.dlist_size;
    const u32 stride = (u32)((n_ids + 31) / 32);
    std::vector<std::pair<u64, u32>> big;   // (size, set)
    for (u64 e = 0; e < v.n_ecs; e++) { const u64 sz = v.ec_off[e + 1] - v.ec_off[e]; if (sz > BM_MIN_MEMBERS) big.emplace_back(sz, (u32)e); }
    std::sort(big.begin(), big.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
    const size_t fit = stride ? BM_MAX_BYTES / ((size_t)stride * 4) : 0;
    if (big.size() > fit) big.resize(fit);
    d.bm_stride = stride; d.bm_min = big.empty() ? 0xFFFFFFFFu : BM_MIN_MEMBERS;
    d.ec_bm_slot = nullptr; d.bm_words = nullptr;
    if (!big.empty()) {
      // (the bits are set on the device from the sets already uploaded: a stress-like index has 8 000 such sets, 200 MB of bitmaps -- nothing of that
      // is built on the host or crosses the link)
      std::vector<u32> slot(v.n_ecs, BM_NONE), which(big.size());
      for (size_t s = 0; s < big.size(); s++) { slot[big[s].second] = (u32)s; which[s] = big[s].second; }
      const u32* d_which = nullptr; const u32* d_words = nullptr;
      if (int rc = upload(c, slot.data(), slot.size(), &d.ec_bm_slot)) return rc;
      if (int rc = upload(c, which.data(), which.size(), &d_which)) return rc;
      {
        void* p = nullptr;
        HIPC(hipMalloc(&p, big.size() * (size_t)stride * 4));
        c->index_allocs.push_back(p);
        HIPC(hipMemsetAsync(p, 0, big.size() * (size_t)stride * 4, c->stream));
        d_words = (const u32*)p;
      }
      hipLaunchKernelGGL(k_bm_fill, dim3((unsigned)big.size()), dim3(256), 0, c->stream, d.ec_off, d.ec_ids, d_which, stride, const_cast<u32*>(d_words));
      HIPC(hipGetLastError());
      d.bm_words = d_words;
      HIPC(hipStreamSynchronize(c->stream));   // (stack-owned staging buffers)
    }
    c->n_set_bitmaps = (u32)big.size();
  }
  if (int rc = upload(c, (const u64*)v.unitig_blk_off, v.n_unitigs + 1, &d.unitig_blk_off)) return rc;
  if (int rc = upload(c, v.unitig_len, v.n_unitigs, &d.unitig_len)) return rc;
  if (int rc = upload(c, v.blk_unitig, v.n_blocks, &d.blk_unitig)) return rc;
  if (int rc = upload(c, v.blk_lb, v.n_blocks, &d.blk_lb)) return rc;
  if (int rc = upload(c, v.blk_ub, v.n_blocks, &d.blk_ub)) return rc;
  if (int rc = upload(c, v.blk_ec, v.n_blocks, &d.blk_ec)) return rc;
  if (int rc = upload(c, (const u64*)v.blk_pos_off, v.n_blocks + 1, &d.blk_pos_off)) return rc;
  if (int rc = upload(c, v.blk_posw, (size_t)v.blk_pos_off[v.n_blocks], &d.blk_posw)) return rc;
  if (int rc = upload(c, v.blk_sense, (size_t)v.blk_pos_off[v.n_blocks], &d.blk_sense)) return rc;
  if (int rc = upload(c, v.target_lens, v.n_targets + v.dlist_size, &d.target_lens)) return rc;   // incl. the D-list pseudo-targets
  d.dtable = nullptr; d.n_dbuckets = v.n_dbuckets; d.dummy_slot = v.dummy_slot; d.dummy_uec = v.dummy_uec; d.dummy_strand = v.dummy_strand;
  if (v.n_dbuckets) if (int rc = upload(c, (const u64*)v.dtable, (size_t)(v.n_dbuckets + v.dpad_buckets) * 8, &d.dtable)) return rc;
  if (int rc = upload(c, v.utext, (size_t)v.utext_words, &d.utext)) return rc;
  HIPC(hipStreamSynchronize(c->stream));  // `ne` is a stack-owned staging buffer
  c->ix = d; c->has_index = true; c->n_ecs = v.n_ecs; c->n_targets = v.n_targets;
  if (int rc = c->dense.ensure(std::max<u64>(v.n_ecs, 1) * sizeof(u32), 0, c->stream)) return rc;
  HIPC(hipMemsetAsync(c->dense.p, 0, std::max<u64>(v.n_ecs, 1) * sizeof(u32), c->stream));
  if (int rc = c->dense_first.ensure(std::max<u64>(v.n_ecs, 1) * sizeof(u64), 0, c->stream)) return rc;
  HIPC(hipMemsetAsync(c->dense_first.p, 0xFF, std::max<u64>(v.n_ecs, 1) * sizeof(u64), c->stream));
  memset(&c->host_state, 0, sizeof c->host_state);
  c->finalized = false; c->exp_words_done = 0; c->recs_total = 0; c->multi_before = 0; c->last_absorb_ms = 0.f; c->overflow_total = 0; c->overflow_ms = 0.f;
  if (int rc = tuples_clear(c)) return rc;
  HIPC(hipMemsetAsync(c->stats_a.p, 0, sizeof(DevStatsA), c->stream));
  return push_state(c);
}

extern "C" int kamd_ec_reset(kamd_ctx* c) {
  if (!c || !c->has_index) return kamd::fail(-1, "kamd_ec_reset: no context / index");
  HIPC(hipSetDevice(c->device));
  HIPC(hipMemsetAsync(c->dense.p, 0, std::max<u64>(c->n_ecs, 1) * sizeof(u32), c->stream));
  HIPC(hipMemsetAsync(c->dense_first.p, 0xFF, std::max<u64>(c->n_ecs, 1) * sizeof(u64), c->stream));
  memset(&c->host_state, 0, sizeof c->host_state);
  c->finalized = false; c->exp_words_done = 0; c->recs_total = 0; c->multi_before = 0; c->last_absorb_ms = 0.f; c->overflow_total = 0; c->overflow_ms = 0.f; c->overflow_second_total = 0;
  if (c->ov_side_pending) { (void)hipStreamSynchronize(c->ov_stream); c->ov_side_pending = false; }   // (a batch that failed between the side launch and its join)
  if (int rc = tuples_clear(c)) return rc;
  HIPC(hipMemsetAsync(c->stats_a.p, 0, sizeof(DevStatsA), c->stream));
  c->had_overflow_items = false;
  c->fq_batch_reads = 0; c->fq_batch_max_len = 0; c->fq_batch_files = 0;   // (units parsed but never packed belong to the abandoned run)
  return push_state(c);
}


// ---- diagnostics: ceiling of the k-mer table's access pattern ---------------------------------------------------------
// Every lane follows a dependent chain of random 64-byte bucket reads (4 x 16 B, the probe's loads) with nothing in
// between: the rate this reaches at a given occupancy is the practical roofline of kernel A's probe stream.
namespace {
template <int WORDS>   // 8: the whole 64-byte line (4 x 16-byte loads), 1: one 8-byte word of it
__global__ void k_random_lines(const u64* __restrict__ table, u64 n_buckets, int iters, u64* sink) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 x = kamd::mix64(gid + 1);
  u64 acc = 0;
  for (int i = 0; i < iters; i++) {
    const u64 b = __umul64hi(x, n_buckets);
    u64 v;
    if (WORDS == 8) {
      const ulonglong2* bp = (const ulonglong2*)(table + b * 8);
      const ulonglong2 s0 = bp[0], s1 = bp[1], s2 = bp[2], s3 = bp[3];   // (the whole 64-byte bucket)
      v = s0.x ^ s0.y ^ s1.x ^ s1.y ^ s2.x ^ s2.y ^ s3.x ^ s3.y;
    } else v = table[b * 8 + (x & 7)];
    acc ^= v;
    x = kamd::mix64(x ^ v);  // the next address depends on the loaded line
  }
  if (acc == 0x1234567ULL) sink[0] = acc;
}
}  // namespace
// span_mb: the reads fall into the first span_mb MiB of the table (0 = all of it) -- how does the rate depend on the footprint
// (HBM vs the 256 MB MALL vs L2)?  access_bytes: 64 (whole line) or 8 (one word of it).
extern "C" int kamd_debug_random_lines_span(kamd_ctx* c, uint32_t n_blocks, uint32_t block_threads, uint32_t iters, uint32_t span_mb,
                                            uint32_t access_bytes, double* gbytes_per_s, double* mlines_per_s) {
  if (!c || !c->has_index) return kamd::fail(-1, "kamd_debug_random_lines: no context / index");
  HIPC(hipSetDevice(c->device));
  if (int rc = c->sizes.ensure(64, 0, c->stream)) return rc;
  u64 nb = c->ix.n_buckets;
  if (span_mb) nb = std::min<u64>(nb, (u64)span_mb * 1024 * 1024 / 64);
  auto launch = [&](int it) {
    if (access_bytes == 8) hipLaunchKernelGGL(k_random_lines<1>, dim3(n_blocks), dim3(block_threads), 0, c->stream, c->ix.table, nb, it, c->sizes.as<u64>());
    else hipLaunchKernelGGL(k_random_lines<8>, dim3(n_blocks), dim3(block_threads), 0, c->stream, c->ix.table, nb, it, c->sizes.as<u64>());
  };
  launch(8);
  HIPC(hipEventRecord(c->ev0, c->stream));
  launch((int)iters);
  HIPC(hipEventRecord(c->ev1, c->stream));
  HIPC(hipEventSynchronize(c->ev1));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  const double lines = (double)n_blocks * block_threads * iters;
  if (mlines_per_s) *mlines_per_s = lines / (ms * 1e-3) / 1e6;
  if (gbytes_per_s) *gbytes_per_s = lines * 64.0 / (ms * 1e-3) / 1e9;
  return 0;
}
extern "C" int kamd_debug_random_lines(kamd_ctx* c, uint32_t n_blocks, uint32_t block_threads, uint32_t iters, double* gbytes_per_s,
                                       double* mlines_per_s) {
  return kamd_debug_random_lines_span(c, n_blocks, block_threads, iters, 0, 64, gbytes_per_s, mlines_per_s);
}

extern "C" int kamd_profile_get(kamd_ctx* c, kamd_profile* p) {
  if (!c || !p) return kamd::fail(-1, "kamd_profile_get: null argument");
  p->last_align_kernel_ms = c->last_align_ms; p->last_em_ms = c->last_em_ms; p->last_em_iters = c->last_em_iters;
  p->last_classify_ms = c->last_classify_ms; p->kernel_a_version = 3;
  p->last_em_nnz = c->last_em_nnz; p->last_em_nnz_multi = c->last_em_nnz_multi; p->last_em_nseg = c->last_em_nseg; p->last_em_necs = c->last_em_necs;
  p->last_em_k = c->last_em_k; p->last_em_grid = c->last_em_grid; p->last_em_lds = c->last_em_lds; p->last_em_plan_cached = c->last_em_plan_cached;
  p->last_finalize_ms = c->last_finalize_ms; p->last_fin_records = c->last_fin_records; p->last_fin_stream_words = c->last_fin_stream_words;
  p->last_fin_cand_words = c->last_fin_cand_words;
  p->absorb_ms = c->last_absorb_ms; p->n_distinct_tuples = c->n_distinct_tuples; p->tuple_store_words = c->host_state.ts_words; p->tuple_table_slots = c->tcap;
  p->last_em_max_comp_nnz = c->last_em_max_comp_nnz; p->last_em_giant_nnz = c->last_em_giant_nnz; p->last_em_giant_rows = c->last_em_giant_rows;
  p->last_em_giant_tr = c->last_em_giant_tr; p->last_em_giant_chunks = c->last_em_giant_chunks; p->last_em_graph_fallback = c->last_em_graph_fallback; p->last_em_giant_pieces = c->last_em_giant_pieces;
  p->last_em_plan_ms = c->last_em_plan_ms;
  p->n_overflow_items = c->overflow_total; p->overflow_ms = c->overflow_ms;
  p->n_overflow_second_pass = c->overflow_second_total;
  p->last_merge_ms = c->last_merge_ms; p->em_collective_ms = c->em_coll_ms; p->em_collectives = c->em_coll_n;
  return 0;
}

// ======================================================================================================================
// several GPUs: communicators (RCCL loaded at run time, or caller-supplied collectives) and what runs over them
// ======================================================================================================================
namespace {
// the part of rccl.h this file needs (the library is bound at run time: see kamd_comm_create_rccl)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[KAMD_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclSum = 0, kNcclUint8 = 1, kNcclInt32 = 2, kNcclUint32 = 3, kNcclUint64 = 5, kNcclFloat64 = 8 };
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
  if (g_rccl.lib) return 0;
  void* h = nullptr;
  // the copy the process already holds (PyTorch ships its own librccl.so: two copies in one process would each open the GPUs)
  for (const char* n : {"librccl.so", "librccl.so.1"}) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }
  if (!h) if (const char* e = getenv("KAMD_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
  if (!h) for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) return kamd::fail(-106, std::string("kamd_comm: librccl.so could not be loaded (") + (dlerror() ? dlerror() : "not found") + "); set KAMD_RCCL_LIB");
  RcclApi a; a.lib = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.Broadcast = (decltype(a.Broadcast))dlsym(h, "ncclBroadcast");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather || !a.Broadcast)
    return kamd::fail(-106, "kamd_comm: librccl.so lacks an expected entry point");
  g_rccl = a;
  return 0;
}
int rccl_fail(ncclResult_t r, const char* what) {
  return kamd::fail(-107, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error") + " (" + std::to_string(r) + ")");
}
}  // namespace

struct kamd_comm {
  kamd_ctx* ctx = nullptr;              // nullptr once the context has been destroyed (kamd_ctx_destroy detaches its communicators)
  int device = 0;
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;            // RCCL backend
  kamd_comm_callbacks cb{}; void* user = nullptr; bool use_cb = false;
  DBuf tmp_a, tmp_b, tmp_c, tmp_d;      // staging of the record exchange
};

namespace kamdi {
void comm_detach_all(kamd_ctx* c) { for (kamd_comm* m : c->comms) m->ctx = nullptr; c->comms.clear(); }
}  // namespace kamdi
namespace {
int comm_allreduce(kamd_comm* m, void* d_buf, u64 count, int type) {   // type: 0 u32, 1 i32, 2 u64, 3 f64
  if ((m->world == 1 && !m->nccl) || count == 0) return 0;   // (a world of one on RCCL still goes through the library: see the tests)
  kamd_ctx* c = m->ctx;
  if (m->use_cb) {
    HIPC(hipStreamSynchronize(c->stream));
    if (int rc = m->cb.allreduce_sum(m->user, d_buf, count, type)) return kamd::fail(-107, "kamd_comm: the all-reduce callback failed (" + std::to_string(rc) + ")");
    return 0;
  }
  static const int dt[4] = {kNcclUint32, kNcclInt32, kNcclUint64, kNcclFloat64};
  const ncclResult_t r = g_rccl.AllReduce(d_buf, d_buf, (size_t)count, dt[type], kNcclSum, m->nccl, c->stream);
  return r ? rccl_fail(r, "ncclAllReduce") : 0;
}
int comm_allgather(kamd_comm* m, const void* d_send, void* d_recv, u64 bytes) {
  kamd_ctx* c = m->ctx;
  if (m->world == 1 && !m->nccl) { if (bytes) HIPC(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, c->stream)); return 0; }
  if (bytes == 0) return 0;
  if (m->use_cb) {
    HIPC(hipStreamSynchronize(c->stream));
    if (int rc = m->cb.allgather(m->user, d_send, d_recv, bytes)) return kamd::fail(-107, "kamd_comm: the all-gather callback failed (" + std::to_string(rc) + ")");
    return 0;
  }
  const ncclResult_t r = g_rccl.AllGather(d_send, d_recv, (size_t)bytes, kNcclUint8, m->nccl, c->stream);
  return r ? rccl_fail(r, "ncclAllGather") : 0;
}
int comm_broadcast(kamd_comm* m, void* d_buf, u64 bytes, int root) {
  if ((m->world == 1 && !m->nccl) || bytes == 0) return 0;
  kamd_ctx* c = m->ctx;
  if (m->use_cb) {
    HIPC(hipStreamSynchronize(c->stream));
    if (int rc = m->cb.broadcast(m->user, d_buf, bytes, root)) return kamd::fail(-107, "kamd_comm: the broadcast callback failed (" + std::to_string(rc) + ")");
    return 0;
  }
  const ncclResult_t r = g_rccl.Broadcast(d_buf, d_buf, (size_t)bytes, kNcclUint8, root, m->nccl, c->stream);
  return r ? rccl_fail(r, "ncclBroadcast") : 0;
}
// all-gather of variable-length record buffers (words + word offsets of the records): every rank's records concatenated in
// rank order, offsets rebased.  Results in m->tmp_c (words) / m->tmp_d (offsets).
// sizes: {words, records} of every rank when the caller has exchanged them already (kamd_ec_allreduce: one all-gather for all the sizes of a
// merge), else null: exchanged here
int comm_gather_records(kamd_comm* m, const u32* d_words, u64 n_words, const u64* d_off, u64 n_recs, u64* tot_words, u64* tot_recs, const u64* sizes = nullptr) {
  kamd_ctx* c = m->ctx;
  const int W = m->world;
  if (int rc = m->tmp_a.ensure((size_t)(2 + 2 * W) * sizeof(u64), 0, c->stream)) return rc;
  std::vector<u64> all((size_t)2 * W);
  if (sizes) memcpy(all.data(), sizes, all.size() * sizeof(u64));
  else {
    u64 mine[2] = {n_words, n_recs};
    u64* d_sizes = m->tmp_a.as<u64>();
    HIPC(hipMemcpyAsync(d_sizes, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));   // `mine` is a stack buffer
    if (int rc = comm_allgather(m, d_sizes, d_sizes + 2, sizeof mine)) return rc;
    HIPC(hipMemcpyAsync(all.data(), d_sizes + 2, all.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  u64 mw = 1, mr = 1, tw = 0, tr = 0;
  for (int r = 0; r < W; r++) { mw = std::max(mw, all[2 * r]); mr = std::max(mr, all[2 * r + 1]); tw += all[2 * r]; tr += all[2 * r + 1]; }
  // padded send buffers, one all-gather each
  if (int rc = m->tmp_a.ensure((size_t)(2 + 2 * W) * sizeof(u64) + mw * sizeof(u32) + mr * sizeof(u64) + 64, (size_t)(2 + 2 * W) * sizeof(u64), c->stream)) return rc;
  char* sa = (char*)m->tmp_a.p + (((size_t)(2 + 2 * W) * sizeof(u64) + 15) & ~(size_t)15);
  u32* send_w = (u32*)sa; u64* send_o = (u64*)(sa + ((mw * sizeof(u32) + 15) & ~(size_t)15));
  if (int rc = m->tmp_b.ensure((size_t)W * (mw * sizeof(u32) + mr * sizeof(u64)) + 64, 0, c->stream)) return rc;
  u32* recv_w = m->tmp_b.as<u32>(); u64* recv_o = (u64*)((char*)m->tmp_b.p + (((size_t)W * mw * sizeof(u32) + 15) & ~(size_t)15));
  HIPC(hipMemsetAsync(send_w, 0, mw * sizeof(u32), c->stream));
  HIPC(hipMemsetAsync(send_o, 0, mr * sizeof(u64), c->stream));
  if (n_words) HIPC(hipMemcpyAsync(send_w, d_words, n_words * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
  if (n_recs) HIPC(hipMemcpyAsync(send_o, d_off, n_recs * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
  if (int rc = comm_allgather(m, send_w, recv_w, mw * sizeof(u32))) return rc;
  if (int rc = comm_allgather(m, send_o, recv_o, mr * sizeof(u64))) return rc;
  // concatenate
  if (int rc = m->tmp_c.ensure(std::max<u64>(tw, 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = m->tmp_d.ensure(std::max<u64>(tr, 1) * sizeof(u64), 0, c->stream)) return rc;
  u64 bw = 0, br = 0;
  for (int r = 0; r < W; r++) {
    const u64 nw = all[2 * r], nr = all[2 * r + 1];
    if (nw) HIPC(hipMemcpyAsync(m->tmp_c.as<u32>() + bw, recv_w + (size_t)r * mw, nw * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
    if (nr) hipLaunchKernelGGL(k_copy_offsets, dim3(grid_for(nr, BLOCK)), dim3(BLOCK), 0, c->stream, recv_o + (size_t)r * mr, nr, bw, m->tmp_d.as<u64>() + br);
    bw += nw; br += nr;
  }
  HIPC(hipGetLastError());
  *tot_words = tw; *tot_recs = tr;
  return 0;
}
// the stop rule's sum over the ranks (one per chunk of EM rounds); the host waits for it, so its wall time is the collective's cost
int comm_sum_cb(void* user, int32_t* d_counts, int32_t n) {
  kamd_comm* m = (kamd_comm*)user;
  const auto t0 = std::chrono::steady_clock::now();
  int rc = comm_allreduce(m, d_counts, (u64)n, 1);
  if (!rc && m->ctx && hipStreamSynchronize(m->ctx->stream) != hipSuccess) rc = kamd::fail(-100, "kamd_comm: stream error behind an all-reduce");
  if (m->ctx) { m->ctx->em_coll_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); m->ctx->em_coll_n += 1; }
  return rc;
}
}  // namespace

extern "C" int kamd_comm_unique_id(void* id128) {
  if (!id128) return kamd::fail(-1, "kamd_comm_unique_id: null argument");
  if (int rc = rccl_load()) return rc;
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r) return rccl_fail(r, "ncclGetUniqueId");
  memcpy(id128, id.internal, KAMD_COMM_ID_BYTES);
  return 0;
}
extern "C" int kamd_comm_create_rccl(kamd_ctx* c, int32_t rank, int32_t world, const void* id128, kamd_comm** out) {
  if (!c || !out || (world > 1 && !id128)) return kamd::fail(-1, "kamd_comm_create_rccl: null argument");
  if (world < 1 || rank < 0 || rank >= world) return kamd::fail(-1, "kamd_comm_create_rccl: bad rank / world");
  *out = nullptr;
  HIPC(hipSetDevice(c->device));
  kamd_comm* m = new kamd_comm;
  m->ctx = c; m->device = c->device; m->rank = rank; m->world = world;
  if (world > 1 || id128) {
    if (int rc = rccl_load()) { delete m; return rc; }
    ncclUniqueId id; memcpy(id.internal, id128, KAMD_COMM_ID_BYTES);
    const ncclResult_t r = g_rccl.CommInitRank(&m->nccl, world, id, rank);
    if (r) { delete m; return rccl_fail(r, "ncclCommInitRank"); }
  }
  c->comms.push_back(m);
  *out = m;
  return 0;
}
extern "C" int kamd_comm_create_callbacks(kamd_ctx* c, int32_t rank, int32_t world, const kamd_comm_callbacks* cb, void* user, kamd_comm** out) {
  if (!c || !out || !cb || !cb->allreduce_sum || !cb->allgather || !cb->broadcast) return kamd::fail(-1, "kamd_comm_create_callbacks: null argument");
  if (world < 1 || rank < 0 || rank >= world) return kamd::fail(-1, "kamd_comm_create_callbacks: bad rank / world");
  kamd_comm* m = new kamd_comm;
  m->ctx = c; m->device = c->device; m->rank = rank; m->world = world; m->cb = *cb; m->user = user; m->use_cb = true;
  c->comms.push_back(m);
  *out = m;
  return 0;
}
extern "C" int kamd_comm_info(const kamd_comm* m, int32_t* rank, int32_t* world, int32_t* ranks_seen, int32_t* backend) {
  if (!m) return kamd::fail(-1, "kamd_comm_info: null argument");
  if (rank) *rank = m->rank;
  if (world) *world = m->world;
  if (backend) *backend = m->nccl ? 1 : m->use_cb ? 2 : 0;
  if (ranks_seen) {
    *ranks_seen = m->world;
    if (m->nccl && g_rccl.CommCount) {
      int n = 0;
      const ncclResult_t r = g_rccl.CommCount(m->nccl, &n);
      if (r) return rccl_fail(r, "ncclCommCount");
      *ranks_seen = n;
    }
  }
  return 0;
}
extern "C" void kamd_comm_destroy(kamd_comm* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->ctx) {
    (void)hipStreamSynchronize(m->ctx->stream);
    auto& v = m->ctx->comms;
    v.erase(std::remove(v.begin(), v.end(), m), v.end());
  } else (void)hipDeviceSynchronize();   // the context went first: its stream handle is no longer ours to touch
  if (m->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->nccl);
  for (DBuf* b : {&m->tmp_a, &m->tmp_b, &m->tmp_c, &m->tmp_d}) b->release();
  delete m;
}
extern "C" int kamd_ec_allreduce(kamd_ctx* c, kamd_comm* m) {
  if (!c || !m || m->ctx != c) return kamd::fail(-1, "kamd_ec_allreduce: null argument / communicator of another context");
  if (m->world == 1 && !m->nccl) return 0;
  if (c->track_order) return kamd::fail(-1, "kamd_ec_allreduce: merged records have no input order (kamd_ec_track_order is on)");
  HIPC(hipSetDevice(c->device));
  if (!c->ev_mg0) { HIPC(hipEventCreate(&c->ev_mg0)); HIPC(hipEventCreate(&c->ev_mg1)); }
  HIPC(hipEventRecord(c->ev_mg0, c->stream));
  // ONE fixed-size all-gather carries everything the ranks have to tell each other before the payloads move (round 4 took a host round trip
  // per item): the --union flag -- a rank that was handed no batch never saw the options of the run, but resolves the merged records like
  // everybody else --, and the sizes of its tuple records and of its explicit-set records
  uint64_t nw = 0, nt = 0, ew = 0, er = 0; u64 tw = 0, tr = 0;
  if (int rc = kamd_ec_tuples_export(c, &nw, &nt)) return rc;
  if (int rc = kamd_ec_explicit_export(c, &ew, &er)) return rc;
  const int W = m->world;
  std::vector<u64> tsz((size_t)2 * W), esz((size_t)2 * W);
  {
    if (int rc = m->tmp_a.ensure((size_t)(1 + W) * 5 * sizeof(u64), 0, c->stream)) return rc;
    u64 mine[5] = {c->ix.union_mode ? 1ULL : 0ULL, nw, nt, ew, er};
    u64* d_sizes = m->tmp_a.as<u64>();
    HIPC(hipMemcpyAsync(d_sizes, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));   // `mine` is a stack buffer
    if (int rc = comm_allgather(m, d_sizes, d_sizes + 5, sizeof mine)) return rc;
    std::vector<u64> all((size_t)5 * W);
    HIPC(hipMemcpyAsync(all.data(), d_sizes + 5, all.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (int r = 0; r < W; r++) {
      if (all[5 * r]) c->ix.union_mode = 1;
      tsz[2 * r] = all[5 * r + 1]; tsz[2 * r + 1] = all[5 * r + 2]; esz[2 * r] = all[5 * r + 3]; esz[2 * r + 1] = all[5 * r + 4];
    }
  }
  // (a) one all-reduce of the dense count vector over the index's transcript sets
  if (int rc = comm_allreduce(m, c->dense.p, c->n_ecs, 0)) return rc;
  // (b) the de-duplicated tuple records of every rank
  DBuf w, o;
  if (int rc = w.ensure(std::max<u64>(std::max(nw, ew), 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = o.ensure(std::max<u64>(std::max(nt, er), 1) * sizeof(u64), 0, c->stream)) { w.release(); return rc; }
  int rc = kamd_ec_tuples_copy(c, w.as<u32>(), o.as<uint64_t>());
  if (!rc) rc = comm_gather_records(m, w.as<u32>(), nw, o.as<u64>(), nt, &tw, &tr, tsz.data());
  if (!rc) rc = kamd_ec_tuples_replace(c, m->tmp_c.as<u32>(), tw, m->tmp_d.as<uint64_t>(), tr);
  // (c) explicit-set records (positional filters): content-keyed, simply concatenated; nothing moves when no rank has any
  u64 e_all = 0;
  for (int r = 0; r < W; r++) e_all += esz[2 * r + 1];
  if (!rc && e_all) {
    rc = kamd_ec_explicit_copy(c, w.as<u32>(), o.as<uint64_t>());
    if (!rc) rc = comm_gather_records(m, w.as<u32>(), ew, o.as<u64>(), er, &tw, &tr, esz.data());
    if (!rc) rc = kamd_ec_explicit_replace(c, m->tmp_c.as<u32>(), tw, m->tmp_d.as<uint64_t>(), tr);
  }
  if (!rc && hipEventRecord(c->ev_mg1, c->stream) != hipSuccess) rc = kamd::fail(-100, "kamd_ec_allreduce: event error");
  if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = kamd::fail(-100, "kamd_ec_allreduce: stream error");
  if (!rc) { float ms = 0.f; if (hipEventElapsedTime(&ms, c->ev_mg0, c->ev_mg1) == hipSuccess) c->last_merge_ms = ms; }
  w.release(); o.release();
  return rc;
}
extern "C" int kamd_comm_broadcast_host(kamd_ctx* c, kamd_comm* m, void* buf, uint64_t bytes, int32_t root) {
  if (!c || !m || !buf) return kamd::fail(-1, "kamd_comm_broadcast_host: null argument");
  if ((m->world == 1 && !m->nccl) || bytes == 0) return 0;
  HIPC(hipSetDevice(c->device));
  if (int rc = m->tmp_a.ensure(bytes, 0, c->stream)) return rc;
  HIPC(hipMemcpyAsync(m->tmp_a.p, buf, bytes, hipMemcpyHostToDevice, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  if (int rc = comm_broadcast(m, m->tmp_a.p, bytes, root)) return rc;
  HIPC(hipMemcpyAsync(buf, m->tmp_a.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int kamd_comm_sum_u64_host(kamd_ctx* c, kamd_comm* m, uint64_t* values, uint64_t count) {
  if (!c || !m || !values) return kamd::fail(-1, "kamd_comm_sum_u64_host: null argument");
  if ((m->world == 1 && !m->nccl) || count == 0) return 0;
  HIPC(hipSetDevice(c->device));
  if (int rc = m->tmp_a.ensure(count * sizeof(u64), 0, c->stream)) return rc;
  HIPC(hipMemcpyAsync(m->tmp_a.p, values, count * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  if (int rc = comm_allreduce(m, m->tmp_a.p, count, 2)) return rc;
  HIPC(hipMemcpyAsync(values, m->tmp_a.p, count * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int kamd_em_run_comm(kamd_ctx* c, kamd_comm* m, const double* eff_lens, uint64_t n_targets, uint32_t n_iter, uint32_t min_rounds,
                                double* alpha, double* alpha_before_zeroes, int32_t* rounds) {
  if (!c || !m || !eff_lens || !alpha) return kamd::fail(-1, "kamd_em_run_comm: null argument");
  if (m->world == 1) return kamd_em_run(c, nullptr, nullptr, nullptr, nullptr, 0, eff_lens, n_targets, n_iter, min_rounds, alpha, alpha_before_zeroes, rounds);
  std::vector<double> abz_local;
  double* abz = alpha_before_zeroes;
  if (!abz) { abz_local.assign(n_targets, 0.0); abz = abz_local.data(); }
  c->em_coll_ms = 0.f; c->em_coll_n = 0;
  if (int rc = kamd_em_run_partitioned(c, (uint32_t)m->rank, (uint32_t)m->world, comm_sum_cb, m, eff_lens, n_targets, n_iter, min_rounds, alpha, abz, rounds)) return rc;
  const auto t_fin = std::chrono::steady_clock::now();
  // every transcript is non-zero on exactly one rank: the sum over the ranks is the result
  HIPC(hipSetDevice(c->device));
  if (int rc = m->tmp_a.ensure(2 * n_targets * sizeof(double), 0, c->stream)) return rc;
  double* d = m->tmp_a.as<double>();
  HIPC(hipMemcpyAsync(d, alpha, n_targets * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemcpyAsync(d + n_targets, abz, n_targets * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  if (int rc = comm_allreduce(m, d, 2 * n_targets, 3)) return rc;
  HIPC(hipMemcpyAsync(alpha, d, n_targets * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipMemcpyAsync(abz, d + n_targets, n_targets * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->em_coll_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_fin).count(); c->em_coll_n += 1;
  return 0;
}

// ---- the whole quant flow over batches resident in HBM (src/main.cpp:2654-2730: ProcessReads -> fragment lengths -> EMAlgorithm) ----
// What a caller that already holds packed reads on the device does with the entry points above, in one call and without a
// language binding's overhead between the stages: every batch pseudoaligned in order (the fragment-length sample taken from the
// first 10 000 qualifying pairs on the way), the EC state merged over the communicator's ranks if there is one, the classes
// resolved, effective lengths, the EM, TPM.
extern "C" int kamd_quant_batches(kamd_ctx* c, const kamd_quant_opts* o, const kamd_batch* batches, uint64_t n_batches, const int32_t* target_lens,
                                  uint64_t n_targets, kamd_comm* comm, kamd_quant_out* out) {
  if (!c || !o || (n_batches && !batches) || !target_lens || !out || !out->flens || !out->eff_lens || !out->est_counts)
    return kamd::fail(-1, "kamd_quant_batches: null argument");
  if (comm && comm->ctx != c) return kamd::fail(-1, "kamd_quant_batches: communicator of another context");
  if (n_targets != c->n_targets) return kamd::fail(-1, "kamd_quant_batches: n_targets differs from the uploaded index");
  const bool multi = comm && (comm->world > 1 || comm->nccl);
  const bool rank0 = !comm || comm->rank == 0;
  const bool estimate = o->paired && o->fld == 0.0;
  memset(out->flens, 0, KAMD_MAX_FRAG_LEN * sizeof(uint32_t));
  uint64_t used = 0, n_proc = 0;
  for (uint64_t b = 0; b < n_batches; b++) {
    const kamd_batch& B = batches[b];
    if (b == 0 && estimate && rank0) if (int rc = kamd_fld_prefetch(c, o, B.d_words, B.d_len, B.n_items, B.max_len)) return rc;   // runs underneath kernel A
    if (int rc = kamd_pseudoalign(c, o, B.d_words, B.d_len, B.n_items, B.max_len)) return rc;
    n_proc += B.n_items;
    // the first 10000 qualifying pairs of the input in order, carried across batches until the sample is full
    // (src/ProcessReads.cpp:981-1008: tlencount persists from batch to batch); rank 0's reads when several ranks run
    if (estimate && used < 10000 && rank0) if (int rc = kamd_fld_from_batch(c, o, B.d_words, B.d_len, B.n_items, B.max_len, out->flens, &used)) return rc;
  }
  if (multi) {
    if (int rc = kamd_comm_sum_u64_host(c, comm, &n_proc, 1)) return rc;
    if (estimate) {
      // The sample is the first 10000 qualifying pairs of the INPUT; the ranks hold consecutive blocks of it (rank 0 the first).
      // Rank 0's reads nearly always fill it; when they do not, the next rank continues the same sample over its own reads, and so
      // on -- what one process reading all of the input does (src/ProcessReads.cpp:981-1008).  Every rank ends with the same sample.
      std::vector<uint32_t> pack(KAMD_MAX_FRAG_LEN + 2);
      for (int r = 0; r < comm->world; r++) {
        if (r > 0 && comm->rank == r)
          for (uint64_t b = 0; b < n_batches && used < 10000; b++)
            if (int rc = kamd_fld_from_batch(c, o, batches[b].d_words, batches[b].d_len, batches[b].n_items, batches[b].max_len, out->flens, &used)) return rc;
        memcpy(pack.data(), out->flens, KAMD_MAX_FRAG_LEN * sizeof(uint32_t));
        pack[KAMD_MAX_FRAG_LEN] = (uint32_t)used; pack[KAMD_MAX_FRAG_LEN + 1] = (uint32_t)(used >> 32);
        if (int rc = kamd_comm_broadcast_host(c, comm, pack.data(), pack.size() * sizeof(uint32_t), r)) return rc;
        memcpy(out->flens, pack.data(), KAMD_MAX_FRAG_LEN * sizeof(uint32_t));
        used = (uint64_t)pack[KAMD_MAX_FRAG_LEN] | ((uint64_t)pack[KAMD_MAX_FRAG_LEN + 1] << 32);
        if (used >= 10000) break;
      }
    }
    if (int rc = kamd_ec_allreduce(c, comm)) return rc;
  }
  // the effective lengths only need the fragment-length sample: a quarter of a millisecond of host arithmetic, done while the device resolves the classes
  std::vector<double> mft(KAMD_MAX_FRAG_LEN);
  std::thread eff_thread([&] {
    if (o->fld == 0.0) kamd_mean_frag_lens_trunc(out->flens, mft.data());
    else kamd_trunc_gaussian_fld(0, KAMD_MAX_FRAG_LEN, o->fld, o->sd, mft.data());
    kamd_eff_lens(target_lens, n_targets, mft.data(), out->eff_lens);
  });
  const int fin_rc = kamd_ec_finalize(c, nullptr);
  eff_thread.join();
  if (fin_rc) return fin_rc;
  int32_t rounds = 0;
  if (multi) { if (int rc = kamd_em_run_comm(c, comm, out->eff_lens, n_targets, 10000, 50, out->est_counts, out->alpha_before_zeroes, &rounds)) return rc; }
  else if (int rc = kamd_em_run(c, nullptr, nullptr, nullptr, nullptr, 0, out->eff_lens, n_targets, 10000, 50, out->est_counts, out->alpha_before_zeroes, &rounds)) return rc;
  if (out->tpm) kamd_counts_to_tpm(out->est_counts, out->eff_lens, n_targets, out->tpm);
  out->n_processed = n_proc; out->em_rounds = rounds;
  return 0;
}

