// kamd_em.hip -- the EM (EMAlgorithm::run) in all its forms, connected components, bootstrap (Bootstrap::run_em, Multinomial::sample)
#include "kamd_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Kernel B: EM (EMAlgorithm::run, src/EMAlgorithm.h:95-223), FP64, three launches per round and no atomics on the data:
//   k_em_rows   one EC row per 4-lane group: denom_e = sum_t alpha[t] * w[e,t]  ->  cn[e] = counts[e] / denom_e
//   k_em_seg    the transposed (transcript-major) copy of the matrix, cut into segments of <= 64 entries:
//               partial[s] = sum_{e in segment} (w[e,t] * alpha[t]) * cn[e]   (+ counts[e] for singleton rows)
//   k_em_final  next[t] = sum of t's segment sums and the convergence test of :176-199
// (the loop control of :202-221 is evaluated by every block from the previous round's record -- see EmState)
// alpha is double-buffered (round i reads A[i&1], writes A[(i+1)&1]); the clamp of the final round (:212-221) is applied
// on read, so the unclamped buffer is alpha_before_zeroes_.
// ------------------------------------------------------------------------------------------------------------------
// Loop control without a control kernel.  Two records, indexed by the parity of the launch: the kernels of a round read
// the record the PREVIOUS round left (iter, final flag, number of transcripts that still changed) and every block derives
// from it -- identically -- what the reference's loop would do next (:202-221); block 0 of k_em_rows publishes that as this
// round's record, k_em_final accumulates the round's change count into it.  Nothing is read and written in the same launch.
struct EmState {
  int iter;         // round index i this record's round ran as (-1: before the first round)
  int chcount;      // transcripts with next > 1e-2 that moved by more than 1 % in that round (:177-179)
  int final_round;  // finalRound: the round read the clamped alpha and was the last one
  int done;
  int rounds;       // i at exit ("ran for i rounds")
  int force_final;  // host request (partitioned EM): the next round is the final round
  int pad[2];
};
struct EmNow { int it, fin, done, rounds; };
__host__ __device__ inline EmNow em_next_round(const EmState& prev, int n_iter, int min_rounds, bool spec) {
  EmNow n; n.it = prev.iter; n.fin = prev.final_round; n.done = prev.done; n.rounds = prev.rounds;
  if (n.done) return n;                                                              // (fin keeps telling how the loop ended)
  if (prev.final_round) { n.done = 1; n.rounds = prev.iter; return n; }              // :207-209 (break: i is not incremented)
  const bool stopEM = !spec && prev.chcount == 0 && prev.iter > min_rounds;          // :202-205
  n.it = prev.iter + 1;
  n.fin = (stopEM || prev.force_final) ? 1 : 0;                                      // :212-221 (clamp applied on read)
  if (n.it >= n_iter) { n.done = 1; n.rounds = n_iter; }                             // the loop ran out
  return n;
}
// Algebra used by the kernels.  With a_t = alpha_t / eff_len_t the reference's row pass
//     denom_e = sum_t alpha_t * (wc_e / eff_t) = wc_e * S_e,  S_e = sum_t a_t          (:152-154, weights.cpp:236)
// and its update  (w_et * alpha_t) * (count_e / denom_e) = a_t * g_e,  g_e = count_e / S_e   (:161-164; wc_e cancels,
// which is why bootstraps may keep the original weights), so that
//     next_t = count(singleton row of t) + a_t * sum_{multi rows e containing t} g_e .
// Per nnz this streams 4 B (id) + gathers 8 B in each pass instead of 20 B; the FP64 rounding differs from the
// reference's operation order at the 1e-16 level (tolerance of the path: 1e-4), the round count is unchanged.
__device__ __forceinline__ double em_clamped(const double* alpha, const double* a, u32 t, int clamp) {
  if (clamp && alpha[t] < 1e-7 / 10.0) return 0.0;  // alpha_limit/10 (:217-219), applied on read in the final round
  return a[t];
}

// column counts of the multi-transcript rows + the singleton row count of every transcript
__global__ void k_em_prepare(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, const u32* __restrict__ counts,
                             u64 n_ecs, u32* col_cnt, double* single) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 a = 0, b = 0;
  if (e < n_ecs) { a = ec_off[e]; b = ec_off[e + 1]; }
  if (b - a == 1) single[ec_ids[a]] = (double)counts[e];  // :119-123 (a transcript has at most one singleton set)
  const bool longrow = b - a > 64;
  if (b - a >= 2 && !longrow) for (u64 j = a; j < b; j++) atomicAdd(&col_cnt[ec_ids[j]], 1u);
  u64 m = __ballot(longrow);   // (rows of thousands of transcripts: the wavefront takes them together)
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const u64 ra = shfl_u64(a, src), rb = shfl_u64(b, src);
    for (u64 j = ra + (u64)lane_id(); j < rb; j += 64) atomicAdd(&col_cnt[ec_ids[j]], 1u);
  }
}
__global__ void k_em_transpose(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs,
                               const u64* __restrict__ col_off, u32* col_fill, u32* col_row) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ecs) return;
  const u64 a = ec_off[e], b = ec_off[e + 1];
  if (b - a == 1) return;
  for (u64 j = a; j < b; j++) {
    const u32 t = ec_ids[j];
    col_row[col_off[t] + atomicAdd(&col_fill[t], 1u)] = (u32)e;
  }
}
// alpha_ = 1/T (:38).  A transcript that is in no EC at all gets next = 0 in every round and is never read by another
// transcript's update, so it is kept at 0 from the start and left out of k_em_final's work list (same outputs).
__global__ void k_em_active(u64 n_tr, const u32* __restrict__ col_cnt, const double* __restrict__ single, u32* flag) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_tr) flag[t] = (col_cnt[t] != 0u || single[t] != 0.0) ? 1u : 0u;
}
__global__ void k_em_init(u64 n_tr, const double* __restrict__ eff, const u32* __restrict__ flag, const u64* __restrict__ pos,
                          double* alpha0, double* alpha1, double* a0, double* a1, u32* active) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tr) return;
  const double al = flag[t] ? 1.0 / (double)n_tr : 0.0;
  alpha0[t] = al; a0[t] = al / eff[t]; alpha1[t] = 0.0; a1[t] = 0.0;
  if (flag[t]) active[pos[t]] = (u32)t;
}

template <int EM_ROW_LANES>
__global__ __launch_bounds__(BLOCK) void k_em_rows(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids,
                                                   const u32* __restrict__ counts, const u32* __restrict__ wcounts, u64 n_ecs,
                                                   const double* __restrict__ alpha0, const double* __restrict__ alpha1,
                                                   const double* __restrict__ a0, const double* __restrict__ a1,
                                                   double* __restrict__ g, EmState* st, int parity, int n_iter, int min_rounds,
                                                   int* spec_hist) {
  const EmState prev = st[parity ^ 1];
  const EmNow now = em_next_round(prev, n_iter, min_rounds, spec_hist != nullptr);
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // publish this round's record (chcount starts at 0 for k_em_final)
    EmState r; r.iter = now.it; r.chcount = 0; r.final_round = now.fin; r.done = now.done; r.rounds = now.rounds; r.force_final = 0;
    r.pad[0] = r.pad[1] = 0;
    st[parity] = r;
    if (spec_hist && !prev.done && prev.iter >= 0) spec_hist[prev.iter] = prev.chcount;
  }
  if (now.done) return;
  const int odd = now.it & 1;
  const double* alpha = odd ? alpha1 : alpha0;
  const double* av = odd ? a1 : a0;
  const int clamp = now.fin;
  const u64 e = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / EM_ROW_LANES;
  const int sub = threadIdx.x & (EM_ROW_LANES - 1);
  double S = 0.0;
  u64 a = 0, b = 0;
  if (e < n_ecs) { a = ec_off[e]; b = ec_off[e + 1]; }
  if (b - a > 1) {
    if (clamp) for (u64 j = a + sub; j < b; j += EM_ROW_LANES) S += em_clamped(alpha, av, ec_ids[j], 1);
    else for (u64 j = a + sub; j < b; j += EM_ROW_LANES) S += av[ec_ids[j]];
  }
#pragma unroll
  for (int d = 1; d < EM_ROW_LANES; d <<= 1) S += __shfl_xor(S, d, 64);
  if (e < n_ecs && sub == 0) {
    const u32 cnt = counts[e];
    // rows the reference skips contribute nothing: count 0 (:133-135) or denom = wc*S below denorm_min, i.e. zero (:156-158)
    g[e] = (b - a == 1 || cnt == 0 || (double)wcounts[e] * S < 4.9406564584124654e-324) ? 0.0 : (double)cnt / S;
  }
}

// The transcript-major pass is balanced over nnz, not over transcripts (a highly expressed transcript sits in thousands
// of ECs): every column is cut into segments of at most EM_SEG entries; k_em_seg reduces one segment per 16-lane group
// (4 independent loads in flight per lane), k_em_final adds a transcript's segment sums in a fixed order.
constexpr int EM_SEG = 64;
constexpr int EM_SEG_LANES = 16;
constexpr int EM_FIN_LANES = 4;
__global__ void k_em_nseg(const u32* __restrict__ col_cnt, u64 n_tr, u32* nseg) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_tr) nseg[t] = (col_cnt[t] + EM_SEG - 1) / EM_SEG;
}
__global__ void k_em_segsetup(const u64* __restrict__ col_off, const u64* __restrict__ seg_off, u64 n_tr, u32* seg_t) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tr) return;
  for (u64 s = seg_off[t]; s < seg_off[t + 1]; s++) seg_t[s] = (u32)t;
}
__global__ __launch_bounds__(BLOCK) void k_em_seg(const u64* __restrict__ col_off, const u64* __restrict__ seg_off,
                                                  const u32* __restrict__ seg_t, u64 n_seg, const u32* __restrict__ col_row,
                                                  const double* __restrict__ g, double* __restrict__ partial, const EmState* st,
                                                  int parity, int n_iter, int min_rounds, int spec) {
  if (em_next_round(st[parity ^ 1], n_iter, min_rounds, spec != 0).done) return;
  const u64 sidx = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / EM_SEG_LANES;
  const int sub = threadIdx.x & (EM_SEG_LANES - 1);
  double acc = 0.0;
  if (sidx < n_seg) {
    const u32 t = seg_t[sidx];
    const u64 begin = col_off[t] + (sidx - seg_off[t]) * EM_SEG;
    const u64 end = min(col_off[t + 1], begin + EM_SEG);
    u32 r[EM_SEG / EM_SEG_LANES];
#pragma unroll
    for (int i = 0; i < EM_SEG / EM_SEG_LANES; i++) {
      const u64 j = begin + sub + (u64)i * EM_SEG_LANES;
      r[i] = j < end ? col_row[j] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < EM_SEG / EM_SEG_LANES; i++) if (r[i] != 0xFFFFFFFFu) acc += g[r[i]];
  }
#pragma unroll
  for (int d = 1; d < EM_SEG_LANES; d <<= 1) acc += __shfl_xor(acc, d, 64);
  if (sidx < n_seg && sub == 0) partial[sidx] = acc;
}
__global__ __launch_bounds__(BLOCK) void k_em_final(const u64* __restrict__ seg_off, const double* __restrict__ partial,
                                                    const double* __restrict__ single, const double* __restrict__ eff,
                                                    const u32* __restrict__ active, u64 n_tr,
                                                    double* alpha0, double* alpha1, double* a0, double* a1, EmState* st, int parity,
                                                    int n_iter, int min_rounds, int spec) {
  const EmNow now = em_next_round(st[parity ^ 1], n_iter, min_rounds, spec != 0);
  if (now.done) return;
  const int it = now.it, odd = it & 1;
  const double* alpha = odd ? alpha1 : alpha0;
  const double* av = odd ? a1 : a0;
  double* next = odd ? alpha0 : alpha1;
  double* anext = odd ? a0 : a1;
  const int clamp = now.fin;
  const int sub = threadIdx.x & (EM_FIN_LANES - 1);
  __shared__ int blk_ch;
  if (threadIdx.x == 0) blk_ch = 0;
  __syncthreads();
  int ch = 0;
  // grid-stride over groups of BLOCK / EM_FIN_LANES transcripts: the convergence counter costs one atomic per block
  for (u64 t0 = (u64)blockIdx.x * (BLOCK / EM_FIN_LANES); t0 < n_tr; t0 += (u64)gridDim.x * (BLOCK / EM_FIN_LANES)) {
    const u64 ai = t0 + threadIdx.x / EM_FIN_LANES;  // n_tr = number of active transcripts
    const bool ok = ai < n_tr;
    const u64 t = ok ? active[ai] : 0;
    double acc = 0.0;
    if (ok) for (u64 s = seg_off[t] + sub; s < seg_off[t + 1]; s += EM_FIN_LANES) acc += partial[s];
#pragma unroll
    for (int d = 1; d < EM_FIN_LANES; d <<= 1) acc += __shfl_xor(acc, d, 64);
    if (ok && sub == 0) {
      double al = alpha[t];
      if (clamp && al < 1e-7 / 10.0) al = 0.0;
      const double at = clamp ? em_clamped(alpha, av, (u32)t, 1) : av[t];
      const double nx = single[t] + at * acc;
      if (nx > 1e-2 && (fabs(nx - al) / nx) > 1e-2) ++ch;          // :177-179
      next[t] = nx;
      anext[t] = nx / eff[t];
    }
  }
  const u64 bal = __ballot(ch != 0);
  int wsum = ch;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) wsum += __shfl_down(wsum, d, 64);
  if (lane_id() == 0 && bal) atomicAdd(&blk_ch, wsum);
  __syncthreads();
  if (threadIdx.x == 0 && blk_ch) atomicAdd(&st[parity].chcount, blk_ch);
}
// ------------------------------------------------------------------------------------------------------------------
// Kernel B, streamed form: two launches per round (rows, columns) over a re-laid-out matrix.  At the sizes of a
// transcriptome (a few 1e6 non-zeros) a round is bound by dependent-load latency and launch count, not by bytes: the CSR
// form above chases row offsets -> ids -> a[] (three dependent latencies per kernel, three kernels).  Re-layout, once per run:
//   * only rows with >= 2 transcripts and only transcripts that occur in such a row are kept ("m-space", compact ids);
//     singleton rows are the constant single[] term of their transcript
//   * both directions of the matrix are FLAGGED STREAMS: entry = index | PM_END on the last entry of a row (column).
//     No offset arrays are read in the loop.  A wavefront owns one chunk of 64 x K consecutive entries (each lane K
//     consecutive ones: K/4 16-byte loads, then K independent 8-byte gathers -- two dependent memory latencies per launch),
//     reduces them with a lane-local pass + ONE segmented wavefront scan, stages the segment sums in LDS and finishes them
//     lane-parallel (coalesced constants and stores, no divergent divisions)
//   * a segment that crosses into a chunk from the left is re-read by that chunk if the part outside is short (<= 256
//     entries, loads issued together with the chunk's own), else completed by a small fix-up launch from the chunks'
//     left/right partial sums in chunk order (deterministic summation, no floating-point atomics); that launch only exists
//     when such segments do
// A persistent single-launch form with grid barriers was built and measured first (scratch/em_persistent_kernel_attempt):
// one grid barrier costs 4.9-6.3 us on 256 CUs (kamd_debug_grid_barrier), a kernel boundary ~1.7 us, so launches win.
// ------------------------------------------------------------------------------------------------------------------
constexpr u32 PM_END = 0x80000000u;
constexpr u32 PM_NONE = 0xFFFFFFFFu;
constexpr u32 PM_HEAVY = 0x80000000u;
constexpr u32 PM_LONG = 0x40000000u;      // head word: the crossing segment has more than 64 * PM_HEAD entries before the chunk, at most PM_LOOP_MAX: the chunk re-reads them in a loop
constexpr u32 PM_LOOP_MAX = 32768;
constexpr int PM_BLOCK = 256;         // 4 wavefronts = 4 chunks per block
constexpr int PM_HEAD = 4;            // a crossing segment with <= 64 * PM_HEAD entries before the chunk is re-read by the chunk
constexpr int PM_LDS_SLOTS = 512;     // segment sums staged per wavefront and window
struct PmSide {
  const u32* stream;     // [n_chunks * 64 * K] index | PM_END; the tail padding points at a sentinel whose value is 0; 64 * PM_HEAD
                         // readable entries (any valid index) in front of it
  const u32* seg_base;   // [n_chunks] segment that contains the chunk's first entry
  const u32* head;       // [n_chunks] entries of that segment before the chunk: 0, 1..64*PM_HEAD (re-read), PM_HEAVY (partials)
  const u32* fix_first;  // [n_chunks] heavy crossing segment that ENDS in this chunk: the chunk it started in, else PM_NONE
  const u32* lane_word;  // [n_chunks * 64] per lane of the chunk: segment ends below the lane | ends in the lane << 12 | distance to the
                         // nearest lane at or below it that holds an end (lane + 1: none) << 18 -- what the passes would otherwise
                         // recount from the END flags every round
  double* lp;            // [n_chunks] sum of the chunk's entries up to its first segment end (all of them if there is none)
  double* rp;            // [n_chunks] sum of the entries after the chunk's last segment end
  u32 n_chunks;
};
struct PmArgs {
  PmSide rows, cols;
  const u64* cw;           // [R] count | weight count << 32 of the kept rows
  double* g;               // [R + 1]; g[R] = 0 is the sentinel the column stream's padding points at
  double* alpha0; double* alpha1; double* a0; double* a1;   // [M + 1]; a*[M] = 0 is the row stream's sentinel
  double* ac0; double* ac1;   // [M + 1] a with the final round's clamp (alpha < alpha_limit / 10 -> 0, EMAlgorithm.h:212-221): what the final round reads
  const double* single; const double* eff;   // [M]
  u32 R, M;
  int n_iter, min_rounds;
  EmState* st;             // the two parity-indexed loop-control records (see EmState)
  int* spec_hist;          // partitioned EM: per-round change counts of this rank (the stop rule is applied by the host); else null
};

// Wavefront scans on the DPP data path (row_shr within the rows of 16 lanes, then row_bcast:15 / row_bcast:31 across
// rows): six VALU steps instead of six LDS round trips (__shfl_up is ds_bpermute_b32, ~100+ cycles each, and the steps of
// a scan depend on each other).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 pm_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double pm_dpp(double v) {
  const u32 lo = pm_dpp<CTRL, ROW_MASK>((u32)__double2loint(v)), hi = pm_dpp<CTRL, ROW_MASK>((u32)__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ u32 pm_scan_incl(u32 x) {   // inclusive prefix sum over the 64 lanes
  x += pm_dpp<0x111, 0xF>(x); x += pm_dpp<0x112, 0xF>(x); x += pm_dpp<0x114, 0xF>(x); x += pm_dpp<0x118, 0xF>(x);
  x += pm_dpp<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3
  x += pm_dpp<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
  return x;
}
// segmented inclusive sum: lane l gets the sum over lanes (l - reach, l] where reach = distance to the nearest segment
// start at or below l (0: the lane starts a segment itself; l + 1: none below)
__device__ __forceinline__ double pm_scan_seg(double y, int reach, int lane) {
  const int r = lane & 15;   // position inside the row
  { const double t = pm_dpp<0x111, 0xF>(y); if (reach >= 1 && r >= 1) y += t; }
  { const double t = pm_dpp<0x112, 0xF>(y); if (reach >= 2 && r >= 2) y += t; }
  { const double t = pm_dpp<0x114, 0xF>(y); if (reach >= 4 && r >= 4) y += t; }
  { const double t = pm_dpp<0x118, 0xF>(y); if (reach >= 8 && r >= 8) y += t; }
  // rows 1 and 3 take the total of the row before them if their run reaches back past the row's first lane
  { const double t = pm_dpp<0x142, 0xA>(y); if ((lane & 16) && reach > r) y += t; }
  // rows 2 and 3 take lane 31's value if their run reaches back past lane 32
  { const double t = pm_dpp<0x143, 0xC>(y); if (lane >= 32 && reach > lane - 32) y += t; }
  return y;
}
__device__ __forceinline__ double pm_wave_sum(double x) {   // sum over the 64 lanes, in every lane
  x += pm_dpp<0x111, 0xF>(x); x += pm_dpp<0x112, 0xF>(x); x += pm_dpp<0x114, 0xF>(x); x += pm_dpp<0x118, 0xF>(x);
  x += pm_dpp<0x142, 0xA>(x); x += pm_dpp<0x143, 0xC>(x);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
// what is done with a finished segment sum: load() fetches the segment's constants (issued before the sums are known,
// coalesced: consecutive lanes finish consecutive segments), finish() consumes them with the sum
struct PmRowEmit {   // g_e = count_e / S_e; rows the reference skips get 0: count 0 (:133-135), denom below denorm_min (:156-158)
  const u64* cw; double* g;
  using Ctx = u64;
  __device__ __forceinline__ Ctx load(u32 r) const { return cw[r]; }
  __device__ __forceinline__ void finish(u32 r, const Ctx& w, double S) const {
    const u32 cnt = (u32)w, wc = (u32)(w >> 32);
    g[r] = (cnt == 0 || (double)wc * S < 4.9406564584124654e-324) ? 0.0 : (double)cnt / S;
  }
};
struct PmColEmit {   // next_t = single_t + a_t * sum_e g_e and the convergence test of :176-199
  const double* alpha_cur; const double* a_cur; const double* single; const double* eff;
  double* alpha_nx; double* a_nx; double* ac_nx; int* ch; int clamp;   // a_cur is the clamped copy in the final round
  struct Ctx { double al, at, sg, ef; };
  __device__ __forceinline__ Ctx load(u32 m) const { return Ctx{alpha_cur[m], a_cur[m], single[m], eff[m]}; }
  __device__ __forceinline__ void finish(u32 m, const Ctx& x, double acc) const {
    const double al = (clamp && x.al < 1e-7 / 10.0) ? 0.0 : x.al;
    const double nx = x.sg + x.at * acc;
    if (nx > 1e-2 && (fabs(nx - al) / nx) > 1e-2) ++*ch;
    const double an = nx / x.ef;
    alpha_nx[m] = nx;
    a_nx[m] = an;
    ac_nx[m] = nx < 1e-7 / 10.0 ? 0.0 : an;
  }
};

// one wavefront, one chunk: value of entry j = src[index_j]; every segment that ends in the chunk is finished here unless it
// crossed in from far to the left (PM_HEAVY: pm_fix)
// (in two steps so that the kernels can issue the chunk's loads BEFORE they use the round's loop-control record: both are
// first-touch reads of a cold L2, ~2 us each, overlapped instead of chained)
template <int K>
struct PmWave { u32 id[K]; u32 hraw[PM_HEAD]; u32 sb, hd, lw; };
template <int K>
__device__ __forceinline__ void pm_wave_load(const PmSide& s, u32 c, PmWave<K>& w) {
  const uint4* p = reinterpret_cast<const uint4*>(s.stream + (u64)c * (64 * K) + lane_id() * K);
#pragma unroll
  for (int q = 0; q < K / 4; q++) { const uint4 x = p[q]; w.id[4 * q] = x.x; w.id[4 * q + 1] = x.y; w.id[4 * q + 2] = x.z; w.id[4 * q + 3] = x.w; }
  // the 64 * PM_HEAD entries before the chunk (the stream has that much padding in front): which of them belong to the
  // segment that crosses into the chunk is known once head[c] is here, their addresses do not depend on it
  const u32* hb = s.stream + (u64)c * (64 * K) + lane_id();
#pragma unroll
  for (int i = 0; i < PM_HEAD; i++) w.hraw[i] = *(hb - 64 * (i + 1));
  w.sb = s.seg_base[c];
  w.hd = s.head[c];
  w.lw = s.lane_word[(u64)c * 64 + lane_id()];
}
// where the value of an entry comes from (the blocked passes of an oversized component read a block of the vector out of LDS instead)
struct PmSrcGlobal {
  const double* __restrict__ p;
  __device__ __forceinline__ double operator()(u32 id) const { return p[id & ~PM_END]; }
};
template <int K, int PRE, bool WIN, class Emit, class Src>
__device__ __forceinline__ void pm_wave_pass(const PmSide& s, u32 c, PmWave<K>& w, const Src& src, double* lds, const Emit& em) {
  const int lane = lane_id();
  u32 (&id)[K] = w.id;
  const u32 sb = w.sb;
  const u32 hd = w.hd;
  const bool heavy = (hd & PM_HEAVY) != 0;
  const bool longhead = !heavy && (hd & PM_LONG) != 0;
  const u32 hlen = heavy ? 0u : (hd & ~PM_LONG);
  // lane l holds the entries 64 * (i + 1) - l before the chunk
  // A LONG head (a row / column of hundreds to thousands of entries that crosses into the chunk: repeat-family and poly-A classes, hub
  // transcripts) is re-read in a loop, 64 entries per trip -- the trips are independent, so the chunk pays one more latency and a few
  // dozen issue slots, where the fix-up launch it replaces cost a kernel boundary and a launch per direction and round (4.7 + 4.6 us
  // each on the stress workload: 4 launches per round -> 2).  The entries are summed lane-strided, then across the lanes: a fixed order.
  auto head_sum = [&](void) -> double {
    double hv[PM_HEAD];
#pragma unroll
    for (int i = 0; i < PM_HEAD; i++) hv[i] = src(w.hraw[i]);
    double hs = 0.0;
#pragma unroll
    for (int i = 0; i < PM_HEAD; i++) hs += (u32)(64 * (i + 1) - lane) <= hlen ? hv[i] : 0.0;
    if (longhead) {
      const u32* before = s.stream + (u64)c * (64 * K);   // entry i of the head (counted backwards from the chunk) at before[-1 - i]
      u32 i = 64 * PM_HEAD + (u32)lane;
      for (; i + 192 < hlen; i += 256) {   // four independent gathers per trip
        const u32 i0 = before[-1 - (long)i], i1 = before[-1 - (long)(i + 64)], i2 = before[-1 - (long)(i + 128)], i3 = before[-1 - (long)(i + 192)];
        const double v0 = src(i0), v1 = src(i1), v2 = src(i2), v3 = src(i3);
        hs += v0; hs += v1; hs += v2; hs += v3;
      }
      for (; i < hlen; i += 64) hs += src(before[-1 - (long)i]);
    }
    return hs;
  };
  const u32 skip = heavy ? 1u : 0u;   // a heavy crossing segment's end is finished by pm_fix
  // (the END flag stays in id[k]: its sign is the "this entry ends a segment" test of the loops below)
  if constexpr (!WIN) {
    const u32 ebase = w.lw & 0xFFFu, ne = (w.lw >> 12) & 0x3Fu;
    const int reach = (int)(w.lw >> 18);
    const u32 n_ends = (u32)__builtin_amdgcn_readlane((int)(ebase + ne), 63);
    // every segment end of the chunk has its own LDS slot (all real chunks): no window tests, and the lane's first end is
    // staged like the others and completed in place once the carry is known -- 6 instructions per entry instead of ~25
    double v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = src(id[k]);
    double hsum = head_sum();
    typename Emit::Ctx pre[PRE];
#pragma unroll
    for (int i = 0; i < PRE; i++) { const u32 j = skip + lane + 64 * i; if (j < n_ends) pre[i] = em.load(sb + j); }
    if (hlen) hsum = pm_wave_sum(hsum);
    double run = 0.0;
    double* slot = lds + ebase;
#pragma unroll
    for (int k = 0; k < K; k++) {
      run += v[k];
      if ((int32_t)id[k] < 0) { *slot++ = run; run = 0.0; }
    }
    const double y = pm_scan_seg(run, reach, lane);
    double carry = pm_dpp<0x138, 0xF>(y);   // wave_shr:1 (lane 0 gets 0)
    if (ebase == 0) carry += hsum;
    if (ne) {   // the lane's first end: what the lanes below it hold of that segment comes first in the sum
      const double first = carry + lds[ebase];
      lds[ebase] = first;
      if (heavy && ebase == 0) s.lp[c] = first;
    }
    if (lane == 63) { if (n_ends == 0) { s.lp[c] = y; s.rp[c] = y; } else s.rp[c] = y; }
    if (skip >= n_ends) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < PRE; i++) { const u32 t = skip + lane + 64 * i; if (t < n_ends) em.finish(sb + t, pre[i], lds[t]); }
    for (u32 t = skip + lane + 64 * PRE; t < n_ends; t += 128) {
      const u32 t1 = t + 64;
      const bool h1 = t1 < n_ends;
      const typename Emit::Ctx x0 = em.load(sb + t), x1 = h1 ? em.load(sb + t1) : x0;
      em.finish(sb + t, x0, lds[t]);
      if (h1) em.finish(sb + t1, x1, lds[t1]);
    }
    return;
  } else {
  u32 ne = 0;
#pragma unroll
  for (int k = 0; k < K; k++) ne += id[k] >> 31;
  const u32 incl = pm_scan_incl(ne);   // segment ends in this and the lower lanes
  const u32 ebase = incl - ne;
  const u32 n_ends = (u32)__builtin_amdgcn_readlane((int)incl, 63);
  const u64 heads = __ballot(ne > 0);
  // (one window unless the chunk holds more segment ends than the wavefront's share of LDS; then the pass is repeated)
  for (u32 w0 = skip; w0 == skip || w0 < n_ends; w0 += PM_LDS_SLOTS) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = src(id[k]);
    double hsum = head_sum();
    typename Emit::Ctx pre[PRE];
#pragma unroll
    for (int i = 0; i < PRE; i++) { const u32 j = w0 + lane + 64 * i; if (j < n_ends) pre[i] = em.load(sb + j); }
    if (hlen) hsum = pm_wave_sum(hsum);
    // lane-local pass in entry order: a segment that ends after an earlier end of the same lane is complete -> staged in LDS
    // at its local index; the sum up to the lane's first end waits for the carry; the open tail feeds the wavefront scan
    double run = 0.0, first_part = 0.0;
    bool got = false;
    u32 j = ebase;
#pragma unroll
    for (int k = 0; k < K; k++) {
      run += v[k];
      if ((int32_t)id[k] < 0) {
        if (!got) { first_part = run; got = true; }
        else if (j >= w0 && j < w0 + PM_LDS_SLOTS) lds[j - w0] = run;
        ++j; run = 0.0;
      }
    }
    double y = run;   // segmented inclusive scan; a lane that holds a segment end starts a new run with its tail
    // lane may add the partial sum of lane - d iff no lane in (lane - d, lane] holds an end, i.e. iff d <= its distance to the
    // nearest end at or below it (the lane itself: 0; none: lane + 1 -- which also covers the lane >= d test)
    const u64 below = heads & ((2ULL << lane) - 1ULL);
    const int reach = below ? lane - (63 - __clzll((long long)below)) : lane + 1;
    y = pm_scan_seg(y, reach, lane);
    double carry = pm_dpp<0x138, 0xF>(y);   // wave_shr:1 (lane 0 gets 0)
    if (ebase == 0) carry += hsum;   // the chunk's first end also gets the re-read head
    if (got && ebase >= w0 && ebase < w0 + PM_LDS_SLOTS) lds[ebase - w0] = carry + first_part;
    if (w0 == skip) {  // partial sums for pm_fix: a chunk without any end lies wholly inside one segment (its sum is both)
      if (lane == 63) { if (n_ends == 0) { s.lp[c] = y; s.rp[c] = y; } else s.rp[c] = y; }
      if (heavy && got && ebase == 0) s.lp[c] = carry + first_part;
    }
    if (w0 >= n_ends) break;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u32 wn = min(n_ends - w0, (u32)PM_LDS_SLOTS);
#pragma unroll
    for (int i = 0; i < PRE; i++) { const u32 t = lane + 64 * i; if (t < wn) em.finish(sb + w0 + t, pre[i], lds[t]); }
    for (u32 t = lane + 64 * PRE; t < wn; t += 128) {   // beyond the prefetched contexts: two segments per lane and trip
      const u32 t1 = t + 64;
      const bool h1 = t1 < wn;
      const typename Emit::Ctx x0 = em.load(sb + w0 + t), x1 = h1 ? em.load(sb + w0 + t1) : x0;
      em.finish(sb + w0 + t, x0, lds[t]);
      if (h1) em.finish(sb + w0 + t1, x1, lds[t1]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  }
}
// a heavy crossing segment that ends in chunk c: right partial of the chunk it started in + the chunks wholly inside it +
// this chunk's left partial, in chunk order; one thread per chunk
template <class Emit>
__device__ __forceinline__ void pm_fix(const PmSide& s, u32 c, const Emit& em) {
  if (c >= s.n_chunks) return;
  const u32 f = s.fix_first[c];
  if (f == PM_NONE) return;
  const u32 seg = s.seg_base[c];
  const typename Emit::Ctx cx = em.load(seg);
  double S = s.rp[f];
  for (u32 k = f + 1; k < c; k++) S += s.lp[k];
  S += s.lp[c];
  em.finish(seg, cx, S);
}
// change counter of the round: one atomic per block
__device__ __forceinline__ void pm_count_changes(int ch, int* lds_ch, EmState* rec) {
  if (threadIdx.x == 0) *lds_ch = 0;
  __syncthreads();
  if (__ballot(ch != 0)) {
    int wsum = ch;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wsum += __shfl_down(wsum, d, 64);
    if (lane_id() == 0) atomicAdd(lds_ch, wsum);
  }
  __syncthreads();
  if (threadIdx.x == 0 && *lds_ch) atomicAdd(&rec->chcount, *lds_ch);
}

// rows launch (first of the round: block 0 publishes the round's loop-control record, like k_em_rows)
template <int K, int PRE, bool WIN>
__global__ __launch_bounds__(PM_BLOCK) void k_pm_rows_pass(PmArgs A, int parity) {
  __shared__ double lds_sums[(PM_BLOCK / 64) * PM_LDS_SLOTS];
  const EmState prev = A.st[parity ^ 1];
  const u32 c = __builtin_amdgcn_readfirstlane(blockIdx.x * (PM_BLOCK / 64) + (threadIdx.x >> 6));
  PmWave<K> w;
  if (c < A.rows.n_chunks) pm_wave_load<K>(A.rows, c, w);
  const EmNow now = em_next_round(prev, A.n_iter, A.min_rounds, A.spec_hist != nullptr);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    EmState r; r.iter = now.it; r.chcount = 0; r.final_round = now.fin; r.done = now.done; r.rounds = now.rounds; r.force_final = 0;
    r.pad[0] = r.pad[1] = 0;
    A.st[parity] = r;
    if (A.spec_hist && !prev.done && prev.iter >= 0) A.spec_hist[prev.iter] = prev.chcount;
  }
  if (now.done) return;
  const int odd = now.it & 1;
  if (c >= A.rows.n_chunks) return;
  const PmRowEmit em{A.cw, A.g};
  const double* a_cur = now.fin ? (odd ? A.ac1 : A.ac0) : (odd ? A.a1 : A.a0);
  pm_wave_pass<K, PRE, WIN>(A.rows, c, w, PmSrcGlobal{a_cur}, lds_sums + (threadIdx.x >> 6) * PM_LDS_SLOTS, em);
}
template <int K, int PRE, bool WIN>
__global__ __launch_bounds__(PM_BLOCK) void k_pm_cols_pass(PmArgs A, int parity) {
  __shared__ double lds_sums[(PM_BLOCK / 64) * PM_LDS_SLOTS];
  __shared__ int lds_ch;
  const EmState prev = A.st[parity ^ 1];
  const u32 c = __builtin_amdgcn_readfirstlane(blockIdx.x * (PM_BLOCK / 64) + (threadIdx.x >> 6));
  PmWave<K> w;
  if (c < A.cols.n_chunks) pm_wave_load<K>(A.cols, c, w);
  const EmNow now = em_next_round(prev, A.n_iter, A.min_rounds, A.spec_hist != nullptr);
  if (now.done) return;
  const int odd = now.it & 1;
  int ch = 0;
  const double* a_cur = now.fin ? (odd ? A.ac1 : A.ac0) : (odd ? A.a1 : A.a0);
  const PmColEmit em{odd ? A.alpha1 : A.alpha0, a_cur, A.single, A.eff, odd ? A.alpha0 : A.alpha1, odd ? A.a0 : A.a1, odd ? A.ac0 : A.ac1, &ch, now.fin};
  if (c < A.cols.n_chunks) pm_wave_pass<K, PRE, WIN>(A.cols, c, w, PmSrcGlobal{A.g}, lds_sums + (threadIdx.x >> 6) * PM_LDS_SLOTS, em);
  pm_count_changes(ch, &lds_ch, &A.st[parity]);
}
// fix-up launches (only enqueued when a direction has heavy crossing segments)
__global__ __launch_bounds__(PM_BLOCK) void k_pm_rows_fix(PmArgs A, int parity) {
  if (em_next_round(A.st[parity ^ 1], A.n_iter, A.min_rounds, A.spec_hist != nullptr).done) return;
  const PmRowEmit em{A.cw, A.g};
  pm_fix(A.rows, blockIdx.x * PM_BLOCK + threadIdx.x, em);
}
__global__ __launch_bounds__(PM_BLOCK) void k_pm_cols_fix(PmArgs A, int parity) {
  __shared__ int lds_ch;
  const EmNow now = em_next_round(A.st[parity ^ 1], A.n_iter, A.min_rounds, A.spec_hist != nullptr);
  if (now.done) return;
  const int odd = now.it & 1;
  int ch = 0;
  const double* a_cur = now.fin ? (odd ? A.ac1 : A.ac0) : (odd ? A.a1 : A.a0);
  const PmColEmit em{odd ? A.alpha1 : A.alpha0, a_cur, A.single, A.eff, odd ? A.alpha0 : A.alpha1, odd ? A.a0 : A.a1, odd ? A.ac0 : A.ac1, &ch, now.fin};
  pm_fix(A.cols, blockIdx.x * PM_BLOCK + threadIdx.x, em);
  pm_count_changes(ch, &lds_ch, &A.st[parity]);
}

// ---- one-time re-layout for the streamed form -------------------------------------------------------------------------
// Kept rows are renumbered by their smallest transcript id (a counting sort): rows of one gene become neighbours, and since
// the isoforms of a gene are neighbours in transcript space too, the 8-byte gathers of a wavefront fall into few cache
// lines (the passes are bound by the L2 request rate of uncoalesced gathers, not by bytes).
__global__ void k_pm_flags(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs, const u32* __restrict__ col_cnt, u64 n_tr,
                           u32* hist, u32* mflag) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_ecs) { const u64 a = ec_off[i]; if (ec_off[i + 1] - a >= 2) atomicAdd(&hist[ec_ids[a]], 1u); }   // sets are sorted: first = smallest
  if (i < n_tr) mflag[i] = col_cnt[i] ? 1u : 0u;
}
// The new number of a kept row: rows ordered by the bucket of their smallest transcript (4096 buckets over the transcripts: rows of one gene
// become neighbours, the gathers of a wavefront fall into few lines and a row block of the blocked form holds few genes' rows), rows of one
// bucket by a 48-bit hash of their CONTENT (the set of transcripts; equal sets do not occur).  The numbering -- and with it the order of every
// floating-point sum of the streamed and blocked forms -- is therefore a function of the matrix alone, whatever order kamd_ec_finalize's
// atomics emitted the classes in (until round 5 the place inside a group came from an atomic counter: abundances of an oversized component
// differed in the last bits from run to run).  A least-significant-digit radix sort of the rows by that 60-bit key, 12 bits per pass; a pass
// is a STABLE counting sort: one wavefront per tile of PM_RANK_TILE rows counts its rows per digit (k_pm_radix_hist), a scan over (digit, tile)
// gives every tile's first place in every digit, k_pm_radix_place walks the tile in order, 64 rows at a time.  Rows that are not kept (fewer
// than two transcripts) carry the largest key and end up behind the R kept ones.
constexpr int PM_RANK_BUCKETS = 4096, PM_RANK_TILE = 1024, PM_RANK_PASSES = 5;
__global__ void k_pm_rowkey(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs, u64 n_tr, u64* key) {
  const u64 e = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / 8;   // 8 lanes per row
  const int sub = threadIdx.x & 7;
  if (e >= n_ecs) return;
  const u64 a = ec_off[e], b = ec_off[e + 1];
  u64 h = 0;
  for (u64 j = a + sub; j < b; j += 8) h += kamd::mix64((u64)ec_ids[j] + 0x9e3779b97f4a7c15ULL);   // (a sum: the same for any order of the members)
  h += shfl_u64(h, lane_id() ^ 1); h += shfl_u64(h, lane_id() ^ 2); h += shfl_u64(h, lane_id() ^ 4);
  if (sub == 0) key[e] = b - a >= 2 ? (((u64)ec_ids[a] * PM_RANK_BUCKETS / n_tr) << 48) | (kamd::mix64(h) >> 16) : ~0ULL;
}
__device__ __forceinline__ u32 pm_radix_digit(u64 k, int pass) { return (u32)(k >> (12 * pass)) & (PM_RANK_BUCKETS - 1); }
// order_in == null: the identity (first pass)
__global__ __launch_bounds__(64) void k_pm_radix_hist(const u64* __restrict__ key, const u32* __restrict__ order_in, u64 n, int pass, u32 n_tiles, u32* hist) {
  __shared__ u32 s_cnt[PM_RANK_BUCKETS];
  const int lane = lane_id();
  for (int i = lane; i < PM_RANK_BUCKETS; i += 64) s_cnt[i] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const u64 p0 = (u64)blockIdx.x * PM_RANK_TILE;
  for (u32 i = lane; i < (u32)PM_RANK_TILE; i += 64) {
    const u64 p = p0 + i;
    if (p < n) atomicAdd(&s_cnt[pm_radix_digit(key[order_in ? order_in[p] : p], pass)], 1u);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int i = lane; i < PM_RANK_BUCKETS; i += 64) hist[(u64)i * n_tiles + blockIdx.x] = s_cnt[i];
}
__global__ __launch_bounds__(64) void k_pm_radix_place(const u64* __restrict__ key, const u32* __restrict__ order_in, u64 n, int pass, u32 n_tiles,
                                                       const u64* __restrict__ first, u32* order_out) {
  __shared__ u32 s_cnt[PM_RANK_BUCKETS];   // rows of the tile placed so far, per digit
  const int lane = lane_id();
  for (int i = lane; i < PM_RANK_BUCKETS; i += 64) s_cnt[i] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const u64 p0 = (u64)blockIdx.x * PM_RANK_TILE;
  for (u32 i0 = 0; i0 < (u32)PM_RANK_TILE; i0 += 64) {
    const u64 p = p0 + i0 + lane;
    const bool in = p < n;
    const u32 e = in ? (order_in ? order_in[p] : (u32)p) : 0u;
    const u32 dg = in ? pm_radix_digit(key[e], pass) : 0xFFFFFFFFu;
    // the lanes of one digit, in lane order: one leader per distinct digit and trip
    u64 todo = __ballot(in);
    u32 before = 0, same_total = 0;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const u32 ld = (u32)__shfl((int)dg, leader, 64);
      const u64 same = __ballot(dg == ld);
      if (dg == ld) { before = (u32)__popcll(same & ((1ULL << lane) - 1ULL)); same_total = (u32)__popcll(same); }
      todo &= ~same;
    }
    if (in) order_out[first[(u64)dg * n_tiles + blockIdx.x] + s_cnt[dg] + before] = e;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (in && before == 0) s_cnt[dg] += same_total;   // (the first lane of every digit)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
// the sorted order -> the kept rows' new numbers and lengths (the R kept rows come first)
__global__ void k_pm_rank_emit(const u64* __restrict__ ec_off, const u32* __restrict__ order, u64 R, u64* rpos, u32* len_sorted) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const u32 e = order[r];
  rpos[e] = r;
  len_sorted[r] = (u32)(ec_off[e + 1] - ec_off[e]);
}
// kept row e -> entries of the row stream, its count word and offset; 8 lanes per row
__global__ void k_pm_rows(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, const u32* __restrict__ counts,
                          const u32* __restrict__ wcounts, u64 n_ecs, const u64* __restrict__ rpos, const u64* __restrict__ roff,
                          const u64* __restrict__ mpos, u32* stream, u64* cw) {
  const u64 e = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  const int sub = threadIdx.x & 7;
  if (e >= n_ecs) return;
  const u64 a = ec_off[e], b = ec_off[e + 1];
  if (b - a < 2) return;
  const u64 r = rpos[e], base = roff[r];
  for (u64 j = a + sub; j < b; j += 8) stream[base + (j - a)] = (u32)mpos[ec_ids[j]] | (j + 1 == b ? PM_END : 0u);
  if (sub == 0) cw[r] = (u64)counts[e] | ((u64)wcounts[e] << 32);
}
__global__ void k_pm_cols(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs, const u64* __restrict__ rpos,
                          const u64* __restrict__ col_off, u32* col_fill, u32* stream) {
  const u64 e = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  const int sub = threadIdx.x & 7;
  if (e >= n_ecs) return;
  const u64 a = ec_off[e], b = ec_off[e + 1];
  if (b - a < 2) return;
  const u32 r = (u32)rpos[e];
  for (u64 j = a + sub; j < b; j += 8) {
    const u32 t = ec_ids[j];
    const u64 p = col_off[t] + atomicAdd(&col_fill[t], 1u);
    stream[p] = r | (p + 1 == col_off[t + 1] ? PM_END : 0u);
  }
}
__global__ void k_pm_fill(u32* p, u64 a, u64 b, u32 v) {
  const u64 i = a + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b) p[i] = v;
}
// m-space vectors; alpha_ = 1/T for every transcript (:38)
__global__ void k_pm_minit(u64 n_tr, const u32* __restrict__ mflag, const u64* __restrict__ mpos, const u64* __restrict__ col_off,
                           const double* __restrict__ single, const double* __restrict__ eff, u64 M, u64* coff, double* single_m,
                           double* eff_m, double* alpha0, double* alpha1, double* a0, double* a1, double* ac0, double* ac1) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) { if (coff) coff[M] = col_off[n_tr]; alpha0[M] = alpha1[M] = a0[M] = a1[M] = ac0[M] = ac1[M] = 0.0; }
  if (t >= n_tr || !mflag[t]) return;
  const u64 m = mpos[t];
  if (coff) coff[m] = col_off[t];   // (null: a cached plan gets new counts and effective lengths, its layout stays)
  single_m[m] = single[t]; eff_m[m] = eff[t];
  const double al = 1.0 / (double)n_tr;
  alpha0[m] = al; a0[m] = al / eff[t]; alpha1[m] = 0.0; a1[m] = 0.0;
  ac0[m] = al < 1e-7 / 10.0 ? 0.0 : al / eff[t]; ac1[m] = 0.0;
}
// per chunk: the segment its first entry belongs to (binary search in the segment offsets) and how it is completed
__global__ void k_pm_chunks(const u64* __restrict__ off, u64 n_seg, u64 nz, u32 chunk, u32 n_chunks, u32* seg_base, u32* head,
                            u32* fix_first, u32* n_fix, u32* max_ends, int no_long) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const u64 c0 = (u64)c * chunk, c1 = c0 + chunk;
  u64 lo = 0, hi = n_seg;  // largest s in [0, n_seg) with off[s] <= c0 (off[0] = 0, offsets strictly increase)
  while (hi - lo > 1) { const u64 mid = (lo + hi) / 2; if (off[mid] <= c0) lo = mid; else hi = mid; }
  const u64 hs = off[lo], se = off[lo + 1];
  const u64 hl = c0 - hs;  // entries of the segment before the chunk
  // (a long head is only re-read by the chunk the segment ENDS in: a chunk wholly inside the segment would gather its hl preceding entries every
  // round and throw the sum away -- O(L^2 / chunk) wasted gathers per long segment; such chunks are marked heavy and need nothing but their own sum)
  const bool is_long = hl > 64ULL * PM_HEAD && hl <= (u64)PM_LOOP_MAX && !no_long && se <= c1;
  const bool is_heavy = hl > 64ULL * PM_HEAD && !is_long;
  seg_base[c] = (u32)lo;
  head[c] = hl == 0 ? 0u : (is_heavy ? PM_HEAVY : (is_long ? (PM_LONG | (u32)hl) : (u32)hl));
  const bool fix = is_heavy && se <= c1;
  fix_first[c] = fix ? (u32)(hs / chunk) : PM_NONE;
  if (fix) atomicAdd(n_fix, 1u);
  const u64 ce = c1 < nz ? c1 : nz;   // segment ends inside the chunk: segments lo .. l2 - 1, l2 = largest s in [0, n_seg] with off[s] <= ce
  u64 l2 = lo; hi = n_seg + 1;
  while (hi - l2 > 1) { const u64 mid = (l2 + hi) / 2; if (off[mid] <= ce) l2 = mid; else hi = mid; }
  if (l2 - lo > (u64)PM_LDS_SLOTS) atomicMax(max_ends, (u32)(l2 - lo));
}
// per lane of every chunk: PmSide::lane_word; one wavefront per chunk, any K
__global__ __launch_bounds__(PM_BLOCK) void k_pm_lanes(const u32* __restrict__ stream, u32 k, u32 n_chunks, u32* lane_word) {
  const u32 c = blockIdx.x * (PM_BLOCK / 64) + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = lane_id();
  const u32* p = stream + (u64)c * (64 * k) + (u64)lane * k;
  u32 ne = 0;
  for (u32 i = 0; i < k; i++) ne += p[i] >> 31;
  const u32 incl = pm_scan_incl(ne);
  const u64 heads = __ballot(ne > 0);
  const u64 below = heads & ((2ULL << lane) - 1ULL);
  const u32 reach = below ? (u32)(lane - (63 - __clzll((long long)below))) : (u32)(lane + 1);
  lane_word[(u64)c * 64 + lane] = (incl - ne) | (ne << 12) | (reach << 18);
}
// back to transcript space: both buffers (result and alpha_before_zeroes are picked by the caller).  A transcript that
// only has a singleton set holds that count in every buffer from round 1 on; one that is in no set stays 0.
__global__ void k_pm_scatter(u64 n_tr, const u32* __restrict__ mflag, const u64* __restrict__ mpos, const double* __restrict__ single,
                             const double* __restrict__ am0, const double* __restrict__ am1, double* out0, double* out1) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tr) return;
  if (mflag[t]) { const u64 m = mpos[t]; out0[t] = am0[m]; out1[t] = am1[m]; }
  else { out0[t] = single[t]; out1[t] = single[t]; }
}
// ------------------------------------------------------------------------------------------------------------------
// EM over several GPUs: the EC x transcript matrix is block diagonal over the connected components of the
// transcript/EC graph (gene families), and the EM update never crosses a component, so each rank runs the unchanged
// EM on the components it owns -- no per-round collective.  Components: min-label propagation along rows + pointer
// jumping.  Ownership: hash(label) mod world.
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_cc_init(u32* label, u64 n) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) label[t] = (u32)t;
}
// Connected components of the transcript / EC graph as a lock-free union-find in ONE pass over the rows (round 2 propagated minimum
// labels row by row and jumped pointers until nothing changed: three or four iterations, each with a read-back of the "changed"
// flag).  parent[x] <= x always: a union hooks the LARGER root under the smaller one with a compare-and-swap, so the root of a
// component is its smallest transcript id -- the label the plan builder expects -- and paths only lead downwards (no cycles, whatever
// the interleaving).  The walks use plain loads, which an XCD's L2 may serve with a value from before another XCD's update -- any value
// parent[x] ever had is an ancestor of x, so a stale walk only ends at an ancestor that is no longer a root; the compare-and-swap
// (executed at the memory side) then fails and returns the node's present parent, a strictly smaller id, and the union goes on from there.
__device__ __forceinline__ u32 cc_find(u32* parent, u32 x) {
  u32 p = parent[x];
  while (p != x) {
    const u32 gp = parent[p];
    if (gp != p) parent[x] = gp;   // path halving (gp is an ancestor of x; x is no root, so no compare-and-swap on it can succeed any more)
    x = p; p = gp;
  }
  return x;
}
__device__ __forceinline__ void cc_union_pair(u32* parent, u32 x, u32 y) {
  u32 r0 = cc_find(parent, x), r1 = cc_find(parent, y);
  while (r0 != r1) {
    const bool first_hi = r0 > r1;
    const u32 hi = first_hi ? r0 : r1, lo = first_hi ? r1 : r0;
    const u32 old = atomicCAS(parent + hi, hi, lo);
    if (old == hi) break;                          // hooked: hi was a root, lo is a node of the other tree (parent < child still holds)
    if (first_hi) r0 = old; else r1 = old;         // hi has a parent by now: go on from it
  }
}
// one thread per row; a row of more than 64 transcripts (repeat-family and poly-A classes hold thousands: one thread walking such a row
// was 1.3 ms of the kernel on the stress workload) is taken by the whole wavefront afterwards, 64 members per trip -- unions commute
__global__ void k_cc_union(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs, u32* parent) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 a = 0, b = 0;
  if (e < n_ecs) { a = ec_off[e]; b = ec_off[e + 1]; }
  const bool longrow = b - a > 64;
  if (b - a >= 2 && !longrow) {
    u32 r0 = cc_find(parent, ec_ids[a]);
    for (u64 j = a + 1; j < b; j++) {
      u32 r1 = cc_find(parent, ec_ids[j]);
      while (r0 != r1) {
        const bool first_hi = r0 > r1;
        const u32 hi = first_hi ? r0 : r1, lo = first_hi ? r1 : r0;
        const u32 old = atomicCAS(parent + hi, hi, lo);
        if (old == hi) { r0 = lo; break; }          // hooked: hi was a root, lo is a node of the other tree (parent < child still holds)
        if (first_hi) r0 = old; else r1 = old;       // hi has a parent by now: go on from it
      }
    }
  }
  u64 m = __ballot(longrow);
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const u64 ra = shfl_u64(a, src), rb = shfl_u64(b, src);
    const u32 t0 = ec_ids[ra];
    for (u64 j = ra + 1 + (u64)lane_id(); j < rb; j += 64) cc_union_pair(parent, t0, ec_ids[j]);
  }
}
__global__ void k_cc_flatten(u32* label, u64 n) {   // label[t] <- root of t (behind the kernel boundary plain loads see everything)
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u32 l = label[t];
  for (;;) { const u32 ll = label[l]; if (ll == l) break; l = ll; }
  label[t] = l;
}
// component labels of the matrix into c->pt_label (label = smallest transcript id of the component; transcripts in no row label themselves)
int cc_labels(kamd_ctx* c, const u64* d_ec_off, const u32* d_ec_ids, u64 n_ecs, u64 T);
__device__ __forceinline__ u32 cc_owner(u32 label, u32 world) { return (u32)(kamd::mix64((u64)label + 0x51ULL) % world); }
__global__ void k_part_sizes(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs, const u32* __restrict__ label,
                             u32 rank, u32 world, u32* row_flag, u32* row_len) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ecs) return;
  const u64 a = ec_off[e], b = ec_off[e + 1];
  const bool mine = b > a && cc_owner(label[ec_ids[a]], world) == rank;
  row_flag[e] = mine ? 1u : 0u;
  row_len[e] = mine ? (u32)(b - a) : 0u;
}
__global__ void k_part_copy(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, const u32* __restrict__ counts,
                            const u32* __restrict__ wcounts, u64 n_ecs, const u32* __restrict__ row_flag,
                            const u64* __restrict__ row_pos, const u64* __restrict__ nnz_pos, u64* out_off, u32* out_ids,
                            u32* out_counts, u32* out_wcounts) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ecs || !row_flag[e]) return;
  const u64 r = row_pos[e], o = nnz_pos[e];
  const u64 a = ec_off[e], b = ec_off[e + 1];
  out_off[r] = o;
  for (u64 j = a; j < b; j++) out_ids[o + (j - a)] = ec_ids[j];
  out_counts[r] = counts[e];
  out_wcounts[r] = wcounts[e];
}

}  // namespace

// ---- EM ----------------------------------------------------------------------------------------------------------------
namespace {
struct EmPartition { uint32_t rank = 0, world = 1; kamd_em_sum_cb cb = nullptr; void* user = nullptr; };
int em_run_impl(kamd_ctx* c, const uint64_t* d_ec_off, const uint32_t* d_ec_ids, const uint32_t* d_counts,
                const uint32_t* d_weight_counts, uint64_t n_ecs, const double* eff_lens, uint64_t n_targets, uint32_t n_iter,
                uint32_t min_rounds, double* alpha, double* alpha_before_zeroes, int32_t* rounds, const EmPartition& part);
}  // namespace

extern "C" int kamd_em_run(kamd_ctx* c, const uint64_t* d_ec_off, const uint32_t* d_ec_ids, const uint32_t* d_counts,
                           const uint32_t* d_weight_counts, uint64_t n_ecs, const double* eff_lens, uint64_t n_targets,
                           uint32_t n_iter, uint32_t min_rounds, double* alpha, double* alpha_before_zeroes, int32_t* rounds) {
  return em_run_impl(c, d_ec_off, d_ec_ids, d_counts, d_weight_counts, n_ecs, eff_lens, n_targets, n_iter, min_rounds, alpha,
                     alpha_before_zeroes, rounds, EmPartition{});
}

extern "C" int kamd_em_run_partitioned(kamd_ctx* c, uint32_t rank, uint32_t world, kamd_em_sum_cb sum_cb, void* user,
                                       const double* eff_lens, uint64_t n_targets, uint32_t n_iter, uint32_t min_rounds,
                                       double* alpha, double* alpha_before_zeroes, int32_t* rounds) {
  if (world == 0 || rank >= world) return kamd::fail(-1, "kamd_em_run_partitioned: bad rank / world");
  if (world > 1 && !sum_cb) return kamd::fail(-1, "kamd_em_run_partitioned: a sum callback is required when world > 1");
  EmPartition p; p.rank = rank; p.world = world; p.cb = sum_cb; p.user = user;
  return em_run_impl(c, nullptr, nullptr, nullptr, nullptr, 0, eff_lens, n_targets, n_iter, min_rounds, alpha, alpha_before_zeroes,
                     rounds, p);
}

namespace {
// ---- streamed EM: re-layout of the matrix (once per run) and the per-round launches -------------------------------------
struct Carver {  // sub-allocations of one device arena, 256-byte aligned
  size_t off = 0;
  size_t take(size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; }
};
struct PmPlan {
  PmArgs args{};
  int k = 0;                 // entries per lane
  bool windowed = false;     // some chunk holds more segment ends than a wavefront's LDS slots (or KAMD_EM_WINDOWED=1): the general pass
  u32 n_chunks = 0;
  u32 n_fix[2] = {0, 0};     // heavy crossing segments per direction (0: no fix-up launch)
  const u32* mflag = nullptr; const u64* mpos = nullptr;   // transcript -> m-space
  const u64* roff = nullptr; const u64* coff = nullptr;    // [R + 1] / [M + 1] entry offsets of the rows / columns (the hybrid picks its hot targets by length)
  u32* rs = nullptr; u32* cs = nullptr; u64 nzpad = 0;     // the two entry streams (nzpad entries each)
  const u64* rpos = nullptr;                               // row of the source matrix -> kept row (rows of fewer than two transcripts: unset)
};
constexpr int PM_KS[] = {8, 12, 16, 20, 24, 28, 32};
template <int K>
void pm_launch_round(const PmPlan& P, hipStream_t s, int parity) {
  const unsigned grid = grid_for(P.n_chunks, PM_BLOCK / 64);
  constexpr int PRE_R = (K + 5) / 6 < 2 ? 2 : (K + 5) / 6;   // 64 * PRE >= ~chunk / 6 row ends
  constexpr int PRE_C = K / 16 + 1;                           // 64 * PRE >= ~chunk / 16 column ends
  if (P.windowed) hipLaunchKernelGGL((k_pm_rows_pass<K, PRE_R, true>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, parity);
  else hipLaunchKernelGGL((k_pm_rows_pass<K, PRE_R, false>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, parity);
  if (P.n_fix[0]) hipLaunchKernelGGL(k_pm_rows_fix, dim3(grid_for(P.n_chunks, PM_BLOCK)), dim3(PM_BLOCK), 0, s, P.args, parity);
  if (P.windowed) hipLaunchKernelGGL((k_pm_cols_pass<K, PRE_C, true>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, parity);
  else hipLaunchKernelGGL((k_pm_cols_pass<K, PRE_C, false>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, parity);
  if (P.n_fix[1]) hipLaunchKernelGGL(k_pm_cols_fix, dim3(grid_for(P.n_chunks, PM_BLOCK)), dim3(PM_BLOCK), 0, s, P.args, parity);
}
void pm_enqueue_round(const PmPlan& P, hipStream_t s, int parity) {
  switch (P.k) {
    case 8: pm_launch_round<8>(P, s, parity); break;
    case 12: pm_launch_round<12>(P, s, parity); break;
    case 16: pm_launch_round<16>(P, s, parity); break;
    case 20: pm_launch_round<20>(P, s, parity); break;
    case 24: pm_launch_round<24>(P, s, parity); break;
    case 28: pm_launch_round<28>(P, s, parity); break;
    default: pm_launch_round<32>(P, s, parity); break;
  }
}

// returns 0 = plan built (the rounds can be enqueued with pm_enqueue_round), 1 = not applicable (the caller uses the CSR
// form), < 0 = error.  Needs col_cnt / em_coloff / em_single / em_eff of the caller (k_em_prepare + scan) and a zeroed col_fill.
int em_streamed_setup(kamd_ctx* c, const u64* ec_off, const u32* ec_ids, const u32* counts, const u32* wcounts, u64 n_ecs, u64 T,
                      const u32* col_cnt, u32* col_fill, PmPlan* P, DBuf* arena_a = nullptr, DBuf* arena_b = nullptr) {
  if (n_ecs == 0) return 1;
  DBuf& ar_a = arena_a ? *arena_a : c->pm_a;   // (the hybrid keeps the streamed plan of the oversized components in arenas of its own: pm_a / pm_b
  DBuf& ar_b = arena_b ? *arena_b : c->pm_b;   //  hold the LDS form's plan and vectors at the same time)
  if (c->n_cus == 0) {
    int v = 0;
    HIPC(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device));
    c->n_cus = v > 0 ? v : 256;
  }
  // stage 1: which rows / transcripts are kept, and their new numbers (rows: counting sort by smallest transcript)
  Carver s1;
  const size_t o_hist = s1.take((T + 1) * 4), o_fill = s1.take((T + 1) * 4), o_mflag = s1.take(T * 4);
  const size_t o_start = s1.take((T + 2) * 8), o_rpos = s1.take((n_ecs + 1) * 8), o_mpos = s1.take((T + 1) * 8);
  const size_t o_len = s1.take((n_ecs + 1) * 4), o_roff = s1.take((n_ecs + 2) * 8);
  if (int rc = ar_a.ensure(s1.off, 0, c->stream)) return rc;
  char* b1 = (char*)ar_a.p;
  u32* hist = (u32*)(b1 + o_hist); u32* mflag = (u32*)(b1 + o_mflag);
  u64* start = (u64*)(b1 + o_start); u64* rpos = (u64*)(b1 + o_rpos); u64* mpos = (u64*)(b1 + o_mpos);
  u32* len_sorted = (u32*)(b1 + o_len); u64* roff = (u64*)(b1 + o_roff);
  (void)o_fill;
  HIPC(hipMemsetAsync(hist, 0, (o_mflag - o_hist), c->stream));
  hipLaunchKernelGGL(k_pm_flags, dim3(grid_for(std::max(n_ecs, T), BLOCK)), dim3(BLOCK), 0, c->stream, ec_off, ec_ids, n_ecs, col_cnt, T, hist,
                     mflag);
  if (int rc = exclusive_scan(c, hist, T, start, start + T)) return rc;
  if (int rc = exclusive_scan(c, mflag, T, mpos, mpos + T)) return rc;
  u64 R = 0, NZ = 0, M = 0;
  HIPC(hipMemcpyAsync(&R, start + T, 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipMemcpyAsync(&M, mpos + T, 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipMemcpyAsync(&NZ, c->em_coloff.as<u64>() + T, 8, hipMemcpyDeviceToHost, c->stream));   // non-zeros of the kept rows
  HIPC(hipStreamSynchronize(c->stream));
  if (NZ == 0 || R == 0 || M == 0 || R >= 0x7FFFFFF0ULL || M >= 0x7FFFFFF0ULL || NZ >= (1ULL << 40)) return 1;
  {
    if (n_ecs >= 0xFFFFFFF0ULL) return 1;
    const u32 n_tiles = (u32)((n_ecs + PM_RANK_TILE - 1) / PM_RANK_TILE);
    const u64 n_cells = (u64)PM_RANK_BUCKETS * n_tiles;
    Carver rv;
    const size_t o_key = rv.take(n_ecs * 8), o_oa = rv.take(n_ecs * 4), o_ob = rv.take(n_ecs * 4), o_h = rv.take(n_cells * 4), o_f = rv.take((n_cells + 2) * 8);
    if (int rc = c->pm_rank.ensure(rv.off, 0, c->stream)) return rc;
    char* rb = (char*)c->pm_rank.p;
    u64* key = (u64*)(rb + o_key); u32* ord[2] = {(u32*)(rb + o_oa), (u32*)(rb + o_ob)}; u32* rhist = (u32*)(rb + o_h); u64* rfirst = (u64*)(rb + o_f);
    hipLaunchKernelGGL(k_pm_rowkey, dim3(grid_for(n_ecs * 8, BLOCK)), dim3(BLOCK), 0, c->stream, ec_off, ec_ids, n_ecs, T, key);
    const u32* in = nullptr;
    for (int pass = 0; pass < PM_RANK_PASSES; pass++) {
      u32* out = ord[pass & 1];
      hipLaunchKernelGGL(k_pm_radix_hist, dim3(n_tiles), dim3(64), 0, c->stream, (const u64*)key, in, n_ecs, pass, n_tiles, rhist);
      if (int rc = exclusive_scan(c, rhist, n_cells, rfirst, rfirst + n_cells)) return rc;
      hipLaunchKernelGGL(k_pm_radix_place, dim3(n_tiles), dim3(64), 0, c->stream, (const u64*)key, in, n_ecs, pass, n_tiles, (const u64*)rfirst, out);
      in = out;
    }
    hipLaunchKernelGGL(k_pm_rank_emit, dim3(grid_for(R, BLOCK)), dim3(BLOCK), 0, c->stream, ec_off, in, R, rpos, len_sorted);
    HIPC(hipGetLastError());
  }
  if (int rc = exclusive_scan(c, len_sorted, R, roff, roff + R)) return rc;   // roff[R] = NZ
  // entries per lane: the smallest K whose chunks fit the chip in one go at 12 wavefronts per CU (measured on config #3:
  // 28.0 us per round at K = 24 / 3038 chunks against 32.7 at K = 20 / 3646 and 32.5 at K = 16 / 4557, same box)
  int K = PM_KS[sizeof(PM_KS) / sizeof(PM_KS[0]) - 1];
  bool k_forced = false;
  if (c->tune.em_entries_per_lane > 0) { for (int k : PM_KS) if (k == c->tune.em_entries_per_lane) { K = k; k_forced = true; } }
  if (!k_forced) for (int k : PM_KS) if ((NZ + 64ULL * k - 1) / (64ULL * k) <= (u64)c->n_cus * 12) { K = k; break; }
  const u32 chunk = 64u * (u32)K;
  const u64 n_chunks64 = (NZ + chunk - 1) / chunk;
  if (n_chunks64 >= 0x7FFFFFF0ULL) return 1;
  const u32 n_chunks = (u32)n_chunks64;
  const u64 nzpad = (u64)n_chunks * chunk;
  // stage 2
  Carver s2;
  const size_t front = 64 * PM_HEAD * 4;   // the passes read up to 64 * PM_HEAD entries before a chunk, also before chunk 0
  const size_t o_rs = s2.take(front + nzpad * 4), o_cs = s2.take(front + nzpad * 4);
  size_t o_meta[2][5];
  size_t o_lane[2];
  for (int s = 0; s < 2; s++) {
    for (int j = 0; j < 3; j++) o_meta[s][j] = s2.take((size_t)n_chunks * 4);
    for (int j = 3; j < 5; j++) o_meta[s][j] = s2.take((size_t)n_chunks * 8);
    o_lane[s] = s2.take((size_t)n_chunks * 64 * 4);
  }
  const size_t o_cw = s2.take(R * 8), o_coff = s2.take((M + 1) * 8), o_g = s2.take((R + 1) * 8);
  size_t o_vec[6];
  for (int j = 0; j < 6; j++) o_vec[j] = s2.take((M + 1) * 8);
  const size_t o_single = s2.take(M * 8), o_eff = s2.take(M * 8), o_nfix = s2.take(64);
  if (int rc = ar_b.ensure(s2.off, 0, c->stream)) return rc;
  char* b2 = (char*)ar_b.p;
  u32* rs = (u32*)(b2 + o_rs + front); u32* cs = (u32*)(b2 + o_cs + front);
  HIPC(hipMemsetAsync(b2 + o_rs, 0, front, c->stream));
  HIPC(hipMemsetAsync(b2 + o_cs, 0, front, c->stream));
  u64* cw = (u64*)(b2 + o_cw); u64* coff = (u64*)(b2 + o_coff);
  double* g = (double*)(b2 + o_g);
  double* vec[6]; for (int j = 0; j < 6; j++) vec[j] = (double*)(b2 + o_vec[j]);
  double* single_m = (double*)(b2 + o_single); double* eff_m = (double*)(b2 + o_eff);
  HIPC(hipMemsetAsync(b2 + o_nfix, 0, 64, c->stream));
  HIPC(hipMemsetAsync(g + R, 0, 8, c->stream));
  const u64* col_off = c->em_coloff.as<u64>();
  hipLaunchKernelGGL(k_pm_rows, dim3(grid_for(n_ecs * 8, BLOCK)), dim3(BLOCK), 0, c->stream, ec_off, ec_ids, counts, wcounts, n_ecs, rpos, roff,
                     mpos, rs, cw);
  hipLaunchKernelGGL(k_pm_cols, dim3(grid_for(n_ecs * 8, BLOCK)), dim3(BLOCK), 0, c->stream, ec_off, ec_ids, n_ecs, rpos, col_off, col_fill, cs);
  if (nzpad > NZ) {
    hipLaunchKernelGGL(k_pm_fill, dim3(grid_for(nzpad - NZ, BLOCK)), dim3(BLOCK), 0, c->stream, rs, NZ, nzpad, (u32)M);
    hipLaunchKernelGGL(k_pm_fill, dim3(grid_for(nzpad - NZ, BLOCK)), dim3(BLOCK), 0, c->stream, cs, NZ, nzpad, (u32)R);
  }
  hipLaunchKernelGGL(k_pm_minit, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, T, mflag, mpos, col_off, c->em_single.as<double>(),
                     c->em_eff.as<double>(), M, coff, single_m, eff_m, vec[0], vec[1], vec[2], vec[3], vec[4], vec[5]);
  PmSide sides[2];
  for (int s = 0; s < 2; s++) {
    PmSide& d = sides[s];
    d.stream = s == 0 ? rs : cs;
    u32* sb = (u32*)(b2 + o_meta[s][0]); u32* hd = (u32*)(b2 + o_meta[s][1]); u32* ff = (u32*)(b2 + o_meta[s][2]);
    d.seg_base = sb; d.head = hd; d.fix_first = ff;
    d.lp = (double*)(b2 + o_meta[s][3]); d.rp = (double*)(b2 + o_meta[s][4]);
    d.n_chunks = n_chunks;
    u32* lw = (u32*)(b2 + o_lane[s]);
    d.lane_word = lw;
    hipLaunchKernelGGL(k_pm_lanes, dim3(grid_for(n_chunks, PM_BLOCK / 64)), dim3(PM_BLOCK), 0, c->stream, d.stream, chunk / 64, n_chunks, lw);
    hipLaunchKernelGGL(k_pm_chunks, dim3(grid_for(n_chunks, BLOCK)), dim3(BLOCK), 0, c->stream, s == 0 ? roff : coff, s == 0 ? R : M, NZ, chunk,
                       n_chunks, sb, hd, ff, (u32*)(b2 + o_nfix) + s, (u32*)(b2 + o_nfix) + 2, getenv("KAMD_EM_NO_LONG_HEADS") ? 1 : 0);
  }
  HIPC(hipGetLastError());
  PmArgs& A = P->args;
  A.rows = sides[0]; A.cols = sides[1];
  A.cw = cw; A.g = g;
  A.alpha0 = vec[0]; A.alpha1 = vec[1]; A.a0 = vec[2]; A.a1 = vec[3]; A.ac0 = vec[4]; A.ac1 = vec[5];
  A.single = single_m; A.eff = eff_m;
  A.R = (u32)R; A.M = (u32)M;
  A.st = (EmState*)c->em_state.p;
  u32 plan_words[3] = {0, 0, 0};   // heavy crossings per direction, largest number of segment ends in a chunk (if above the LDS slots)
  HIPC(hipMemcpyAsync(plan_words, b2 + o_nfix, sizeof plan_words, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  P->n_fix[0] = plan_words[0]; P->n_fix[1] = plan_words[1];
  P->windowed = plan_words[2] > (u32)PM_LDS_SLOTS;
  if (c->tune.em_windowed == 1) P->windowed = true;
  P->k = K; P->n_chunks = n_chunks; P->mflag = mflag; P->mpos = mpos; P->rpos = rpos;
  P->roff = roff; P->coff = coff; P->rs = rs; P->cs = cs; P->nzpad = nzpad;
  c->last_em_nnz_multi = NZ; c->last_em_nseg = n_chunks; c->last_em_necs = n_ecs; c->last_em_k = K;
  c->last_em_grid = grid_for(n_chunks, PM_BLOCK / 64);
  return 0;
}

// ---- component-local EM (kamd_em_local.h) -------------------------------------------------------------------------------------
// One workgroup per group of connected components; the group's whole state lives in LDS for the rounds of a launch.  The plan
// is built on the device (em_local_setup_device) as a CSR in both directions and converted to the sliced-ELLPACK layout the
// kernel iterates (kamd_em_sell.h).  (Two earlier kernels that iterated the CSR directly -- one thread per row, then 8 / 16
// lanes per row / column -- were instruction-bound, ~7 000 wavefront instructions per group and round, and were removed.)
struct EmLocalDev {
  const u32* row_base; const u32* tr_base; const u64* nz_base;
  const u32* row_ptr; const u32* col_ptr; const uint16_t* row_tr; const uint16_t* col_row;
  const u64* cw; const double* single; const double* eff;
  const u32* tr_id = nullptr;   // (device-built plans only)
};
constexpr int EML_MAX_ROUNDS = 64;   // rounds per launch (the per-group change history of a launch lives in LDS)
// ---- component-local EM over the sliced-ELLPACK layout (kamd_em_sell.h) ---------------------------------------------------
// One workgroup per group, the whole group in LDS for the rounds of a launch.  A wavefront takes whole slices: lane l walks
// the entries of its segment at stream[j * 64 + l] (u16 local index -> gather of an FP64 value, padding points at a zero slot),
// j < width of the slice -- no row pointers, no cross-lane reduction except in the few slices that hold split segments (one
// segmented DPP scan).  The lane that completes a segment finishes it at once (g = count / S for a row; next alpha, the
// convergence test and a = alpha / eff for a transcript), so a round has two block barriers.
struct EmSellDev {
  const u32* row_base; const u32* tr_base; const u32* rslice_base; const u32* cslice_base; const u64* rell_base; const u64* cell_base;
  const u32* rdesc; const u32* cdesc; const uint16_t* rell; const uint16_t* cell;
  const u64* cw; const double* single; const double* eff;
};
constexpr int EMS_MAX_BLOCK = 1024;
// sum over the lane's entries of one slice; e: the lane's first 64-bit word of four u16 indices (kamd_em_sell.h: word q of the lane at e + q * 64)
// EXP != 0: timing experiments (KAMD_EM_EXP, results are garbage): 1 = gathers at conflict-free addresses (indices still loaded),
// 3 = no gathers (indices only); 4 (no divisions) and 5 (no block barriers) are applied by the caller
template <int EXP>
__device__ __forceinline__ double ems_slice_sum(const u64* e, u32 width, const double* src, u32 n_src) {
  const u32 cf = (u32)lane_id() % n_src;   // (experiments only)
  u32 sink = 0;
  auto val_at = [&](u32 ix) -> double {
    if constexpr (EXP == 1) { sink += ix; return src[cf]; }
    else if constexpr (EXP == 3) return __hiloint2double(0x3ff00000, (int)ix);
    else return src[ix];
  };
  double S = 0.0;
  u32 j = 0;
  for (; j + 8 <= width; j += 8) {
    const u64 w0 = e[(size_t)(j >> 2) * 64], w1 = e[(size_t)((j >> 2) + 1) * 64];
    const u32 a0 = (u32)w0, a1 = (u32)(w0 >> 32), b0 = (u32)w1, b1 = (u32)(w1 >> 32);
    const double v0 = val_at(a0 & 0xFFFFu), v1 = val_at(a0 >> 16), v2 = val_at(a1 & 0xFFFFu), v3 = val_at(a1 >> 16);
    const double v4 = val_at(b0 & 0xFFFFu), v5 = val_at(b0 >> 16), v6 = val_at(b1 & 0xFFFFu), v7 = val_at(b1 >> 16);
    S += v0; S += v1; S += v2; S += v3; S += v4; S += v5; S += v6; S += v7;
  }
  if (j < width) {   // 1 .. 7 entries left: one or two words; a word's trips beyond the width hold padding (the zero slot) and are not read
    const u32 rem = width - j;
    const u64 w0 = e[(size_t)(j >> 2) * 64];
    const u64 w1 = rem > 4 ? e[(size_t)((j >> 2) + 1) * 64] : 0ULL;
    const u32 a0 = (u32)w0, a1 = (u32)(w0 >> 32), b0 = (u32)w1, b1 = (u32)(w1 >> 32);
    const double v0 = val_at(a0 & 0xFFFFu);
    const double v1 = rem > 1 ? val_at(a0 >> 16) : 0.0;
    const double v2 = rem > 2 ? val_at(a1 & 0xFFFFu) : 0.0;
    const double v3 = rem > 3 ? val_at(a1 >> 16) : 0.0;
    S += v0;
    if (rem > 1) S += v1;
    if (rem > 2) S += v2;
    if (rem > 3) S += v3;
    if (rem > 4) {
      const double v4 = val_at(b0 & 0xFFFFu);
      const double v5 = rem > 5 ? val_at(b0 >> 16) : 0.0;
      const double v6 = rem > 6 ? val_at(b1 & 0xFFFFu) : 0.0;
      S += v4;
      if (rem > 5) S += v5;
      if (rem > 6) S += v6;
    }
  }
  if constexpr (EXP != 0) { if (sink == 0xFFFFFFFFu) S += 1.0; }
  return S;
}
// One team of NW wavefronts iterates one group out of its own piece of LDS.  WAVE_TEAM: the team is a single wavefront (several
// teams share a workgroup), so the two hand-overs of a round -- g after the rows pass, a / alpha after the columns pass -- need no
// block barrier: a wavefront's LDS operations complete in order, the fence only keeps the compiler from moving them.
// alpha_in / a_in -> alpha_out / a_out (the same vectors, or the other half of a ping-pong pair: the input then stays what it was, the
// checkpoint a speculative chunk is replayed from).  clk: diagnostic, phase clocks of every wavefront in round EMS_CLK_ROUND.
constexpr int EMS_CLK_ROUND = 8, EMS_CLK_WORDS = 16;
template <bool WAVE_TEAM, int EXP = 0>
__device__ __forceinline__ void ems_group_rounds(const EmSellDev& P, u32 g, unsigned char* smem, u32 tid, u32 nthr, const double* alpha, const double* a,
                                                 double* alpha_out, double* a_out, int n_rounds, int clamp, int* s_hist, long long* clk = nullptr) {
  namespace L = kamd_em_sell;
  const int lane = lane_id();
  const u32 wv = WAVE_TEAM ? 0u : (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), NW = WAVE_TEAM ? 1u : nthr >> 6;
  const u32 r0 = P.row_base[g], nR = P.row_base[g + 1] - r0, t0 = P.tr_base[g], nT = P.tr_base[g + 1] - t0;
  const u32 rs0 = P.rslice_base[g], nrs = P.rslice_base[g + 1] - rs0, cs0 = P.cslice_base[g], ncs = P.cslice_base[g + 1] - cs0;
  const u64 re0 = P.rell_base[g], ce0 = P.cell_base[g];
  const u32 nru = (u32)(P.rell_base[g + 1] - re0), ncu = (u32)(P.cell_base[g + 1] - ce0);
  auto team_sync = [] {
    if (WAVE_TEAM || EXP == 5) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    else __syncthreads();
  };
  // the layout kamd_em_sell::group_bytes() prices
  double* s_al0 = reinterpret_cast<double*>(smem);
  double* s_a0 = s_al0 + (nT + 1); double* s_al1 = s_a0 + (nT + 1); double* s_a1 = s_al1 + (nT + 1);
  double* s_single = s_a1 + (nT + 1); double* s_eff = s_single + nT; double* s_g = s_eff + nT;
  u64* s_cw = reinterpret_cast<u64*>(s_g + (nR + 1));
  u32* s_rdesc = reinterpret_cast<u32*>(s_cw + nR); u32* s_cdesc = s_rdesc + 2 * nrs;
  uint16_t* s_rell = reinterpret_cast<uint16_t*>(s_cdesc + 2 * ncs); uint16_t* s_cell = s_rell + ((nru + 1) & ~1u);
  for (u32 i = tid; i < nT; i += nthr) {
    double al = alpha[t0 + i], av = a[t0 + i];
    if (clamp && al < 1e-7 / 10.0) { al = 0.0; av = 0.0; }   // the final round reads alpha < alpha_limit / 10 as 0 (:212-221)
    s_al0[i] = al; s_a0[i] = av; s_single[i] = P.single[t0 + i]; s_eff[i] = P.eff[t0 + i];
  }
  for (u32 i = tid; i < nR; i += nthr) s_cw[i] = P.cw[r0 + i];
  for (u32 i = tid; i < 2 * nrs; i += nthr) s_rdesc[i] = P.rdesc[2 * (u64)rs0 + i];
  for (u32 i = tid; i < 2 * ncs; i += nthr) s_cdesc[i] = P.cdesc[2 * (u64)cs0 + i];
  // streams: a padding entry becomes the index of the zero slot (metadata words never hold 0xFFFF halves)
  for (u32 i = tid; i < nru; i += nthr) { const uint16_t v = P.rell[re0 + i]; s_rell[i] = v == (uint16_t)L::SELL_PAD ? (uint16_t)nT : v; }
  for (u32 i = tid; i < ncu; i += nthr) { const uint16_t v = P.cell[ce0 + i]; s_cell[i] = v == (uint16_t)L::SELL_PAD ? (uint16_t)nR : v; }
  if (tid == 0) { s_al0[nT] = s_a0[nT] = s_al1[nT] = s_a1[nT] = 0.0; s_g[nR] = 0.0; }
  team_sync();
  double* al = s_al0; double* av = s_a0; double* aln = s_al1; double* avn = s_a1;
  for (int r = 0; r < n_rounds; r++) {
    const bool tick = clk && r == EMS_CLK_ROUND && lane == 0;
    long long w0 = 0;
    if (tick) { clk[0] = clock64(); w0 = wall_clock64(); }
    // rows: S_e over the row's transcripts, then g_e = count_e / S_e (rows the reference skips get 0: count 0, :133-135;
    // denom below denorm_min, :156-158)
    for (u32 s = wv; s < nrs; s += NW) {
      const u32 d0 = s_rdesc[2 * s], d1 = s_rdesc[2 * s + 1];
      const bool meta = (d0 & L::DESC_META) != 0;
      const u32 off = d0 & ~L::DESC_META, width = d1 & 0xFFFFu;
      const bool tk = tick && s == wv;
      if (tk) clk[12] = clock64();
      double S = ems_slice_sum<EXP>(reinterpret_cast<const u64*>(s_rell + off + (meta ? 2 * L::SELL_META_WORDS : 0u)) + lane, width, av, nT);
      if (tk) { clk[13] = clock64(); clk[14] = (long long)(width | (meta ? 0x10000u : 0u)); }
      u32 seg = (d1 >> 16) + (u32)lane; bool fin = seg < nR;
      if (meta) {
        const u32 w = reinterpret_cast<const u32*>(s_rell + off)[lane];
        S = pm_scan_seg(S, (int)((w >> 16) & 0x7Fu), lane);
        seg = w & 0xFFFFu; fin = (w & L::META_ACTIVE) && (w & L::META_LAST);
      }
      if (fin) {
        const u64 cwv = s_cw[seg];
        const u32 cnt = (u32)cwv, wc = (u32)(cwv >> 32);
        if constexpr (EXP == 4) s_g[seg] = (double)cnt * S;
        else s_g[seg] = (cnt == 0 || (double)wc * S < 4.9406564584124654e-324) ? 0.0 : (double)cnt / S;
      }
      if (tk) clk[15] = clock64();
    }
    if (tick) clk[1] = clock64();
    team_sync();
    if (tick) clk[2] = clock64();
    // columns: next_t = single_t + a_t * sum of g over the transcript's rows, and the convergence test of :176-199
    int ch = 0;
    for (u32 s = wv; s < ncs; s += NW) {
      const u32 d0 = s_cdesc[2 * s], d1 = s_cdesc[2 * s + 1];
      const bool meta = (d0 & L::DESC_META) != 0;
      const u32 off = d0 & ~L::DESC_META, width = d1 & 0xFFFFu;
      const bool tk = tick && s == wv;
      if (tk) clk[8] = clock64();
      double acc = ems_slice_sum<EXP>(reinterpret_cast<const u64*>(s_cell + off + (meta ? 2 * L::SELL_META_WORDS : 0u)) + lane, width, s_g, nR);
      if (tk) { clk[9] = clock64(); clk[10] = (long long)(width | (meta ? 0x10000u : 0u)); }
      u32 seg = (d1 >> 16) + (u32)lane; bool fin = seg < nT;
      if (meta) {
        const u32 w = reinterpret_cast<const u32*>(s_cell + off)[lane];
        acc = pm_scan_seg(acc, (int)((w >> 16) & 0x7Fu), lane);
        seg = w & 0xFFFFu; fin = (w & L::META_ACTIVE) && (w & L::META_LAST);
      }
      bool chg = false;
      if (fin) {
        const double at = av[seg], cur = al[seg];
        const double nx = s_single[seg] + at * acc;
        // :177-179 `fabs(next - alpha) / next > 1e-2`, without the division unless the quotient is within 1e-7 of the threshold (the
        // quotient's rounding error is 1e-16: outside that band the product test decides the same way as the reference's quotient)
        const double dd = fabs(nx - cur);
        bool moved = dd > 1.0000001e-2 * nx;
        if (dd > 0.9999999e-2 * nx && !moved) moved = dd / nx > 1e-2;
        chg = nx > 1e-2 && moved;
        aln[seg] = nx;
        if constexpr (EXP == 4) avn[seg] = nx * s_eff[seg];
        else avn[seg] = nx / s_eff[seg];
      }
      ch += __popcll(__ballot(chg));   // (wave-uniform: a scalar count, no cross-lane reduction)
      if (tk) clk[11] = clock64();
    }
    if (ch && lane == 0) atomicAdd(&s_hist[r], ch);
    if (tick) clk[3] = clock64();
    team_sync();
    if (tick) { clk[4] = clock64(); clk[5] = (long long)(nrs | (ncs << 16)); clk[6] = (long long)(nru | ((u64)ncu << 32)); clk[7] = wall_clock64() - w0; }
    double* t1 = al; al = aln; aln = t1;
    double* t2 = av; av = avn; avn = t2;
  }
  for (u32 i = tid; i < nT; i += nthr) { alpha_out[t0 + i] = al[i]; a_out[t0 + i] = av[i]; }
}
// ---- the same rounds with every wavefront's first slices of each direction held in registers ------------------------------------------
// Between two barriers a wavefront of k_em_sell used to walk a chain of dependent LDS trips for its slice: descriptor -> index words ->
// gathers -> (metadata word -> scan) -> segment constants -> store, and most of the 16 wavefronts had nothing to do while the slowest
// set the pace (profiles/r03_em_phase_clocks.txt).  Nothing on that chain but the gathered values changes from round to round: a
// wavefront owns the same slices for the whole launch.  So the index words of a slice (W x 64 bits per lane: em_split_len <= 4 W
// entries), its metadata and the constants of the segment a lane finishes (count and weight count of a row; single, eff and the lane's
// own alpha / a of a transcript) are loaded ONCE per launch, and a pass is: issue the gathers, add, (segmented scan), finish, store --
// one LDS round trip.  With the slices in registers the kernel is bound by the vector instructions it issues (a wave64 instruction
// occupies its SIMD for four clocks; profiles/README.md, round 4: SIMD clocks = 4 x instructions explains the phase clocks), so the
// rest of this form is an instruction diet:
//   * the 16-bit indices are turned into LDS BYTE ADDRESSES of the gathered value when they are loaded (the arrays lie in the first
//     64 KB of the workgroup's LDS, checked per group): a gather is `and` / `shift` + ds_read_b64 with the array's base in the
//     instruction's offset field instead of unpack + shift-add + read;
//   * the segmented scan of a slice with split segments runs only the steps some lane of the slice needs (a mask computed once per
//     launch: most split segments span two or three lanes, a step costs seven instructions);
//   * NS slices per wavefront and direction live in registers (groups of ~10 000 entries have 17-22 row slices for 16 wavefronts).
// Slices beyond those take the LDS path.  Two index words per lane (split length 8) form the sums exactly as the LDS form does; the
// wider forms add each batch of sixteen values as a balanced tree.
template <int NB>
__device__ __forceinline__ void ems_reg_gather(const u64* w, const unsigned char* lds0, double* v) {
#pragma unroll
  for (int q = 0; q < NB; q++) {
    u32 lo = (u32)w[q], hi = (u32)(w[q] >> 32);
    // (opaque to the optimiser: otherwise the gather addresses, which do not change from round to round, are hoisted out of the
    // round loop into registers of their own and the kernel spills)
    asm volatile("" : "+v"(lo), "+v"(hi));
    v[4 * q] = *reinterpret_cast<const double*>(lds0 + (lo & 0xFFFFu)); v[4 * q + 1] = *reinterpret_cast<const double*>(lds0 + (lo >> 16));
    v[4 * q + 2] = *reinterpret_cast<const double*>(lds0 + (hi & 0xFFFFu)); v[4 * q + 3] = *reinterpret_cast<const double*>(lds0 + (hi >> 16));
  }
}
// TREE: batches of four words summed as balanced trees (the forms with 128 registers); else batches of two words added in entry order --
// what the LDS form computes, bit for bit, inside 64 registers
template <int NQ, bool TREE>
__device__ __forceinline__ double ems_reg_sum(const u64* w, const unsigned char* lds0, double S = 0.0) {
  if constexpr (!TREE) {
    constexpr int NB = NQ < 2 ? NQ : 2;
    double v[4 * NB];
    ems_reg_gather<NB>(w, lds0, v);
#pragma unroll
    for (int i = 0; i < 4 * NB; i++) S += v[i];   // (entries beyond the slice's width point at the zero slot: + 0.0 changes no bit of a sum >= 0)
    if constexpr (NQ > 2) return ems_reg_sum<NQ - 2, false>(w + 2, lds0, S);
    else return S;
  } else {
    constexpr int NB = NQ < 4 ? NQ : 4;
    double v[4 * NB];
    ems_reg_gather<NB>(w, lds0, v);
#pragma unroll
    for (int st = 1; st < 4 * NB; st *= 2)
#pragma unroll
      for (int i = 0; i + st < 4 * NB; i += 2 * st) v[i] += v[i + st];
    if constexpr (NQ > 4) return ems_reg_sum<NQ - 4, true>(w + 4, lds0, S + v[0]);
    else return S + v[0];
  }
}
template <int W, bool TREE, int NQ = 1>
__device__ __forceinline__ double ems_reg_slice_sum(const u64* w, u32 nq, const unsigned char* lds0) {   // nq: words in use, wave-uniform, 1 .. W
  if constexpr (NQ >= W) return ems_reg_sum<W, TREE>(w, lds0);
  else { if (nq <= (u32)NQ) return ems_reg_sum<NQ, TREE>(w, lds0); return ems_reg_slice_sum<W, TREE, NQ + 1>(w, nq, lds0); }
}
// pm_scan_seg with the steps no lane of the wavefront needs left out (steps: bit i set = step i has a lane that adds; wave-uniform)
__device__ __forceinline__ u32 pm_scan_seg_steps(int reach, int lane) {
  const int r = lane & 15;
  u32 m = 0;
  if (__ballot(reach >= 1 && r >= 1)) m |= 1u;
  if (__ballot(reach >= 2 && r >= 2)) m |= 2u;
  if (__ballot(reach >= 4 && r >= 4)) m |= 4u;
  if (__ballot(reach >= 8 && r >= 8)) m |= 8u;
  if (__ballot((lane & 16) && reach > r)) m |= 16u;
  if (__ballot(lane >= 32 && reach > lane - 32)) m |= 32u;
  return m;
}
__device__ __forceinline__ double pm_scan_seg_masked(double y, int reach, int lane, u32 steps) {
  const int r = lane & 15;
  if (steps & 1u) { const double t = pm_dpp<0x111, 0xF>(y); if (reach >= 1 && r >= 1) y += t; }
  if (steps & 2u) { const double t = pm_dpp<0x112, 0xF>(y); if (reach >= 2 && r >= 2) y += t; }
  if (steps & 4u) { const double t = pm_dpp<0x114, 0xF>(y); if (reach >= 4 && r >= 4) y += t; }
  if (steps & 8u) { const double t = pm_dpp<0x118, 0xF>(y); if (reach >= 8 && r >= 8) y += t; }
  if (steps & 16u) { const double t = pm_dpp<0x142, 0xA>(y); if ((lane & 16) && reach > r) y += t; }
  if (steps & 32u) { const double t = pm_dpp<0x143, 0xC>(y); if (lane >= 32 && reach > lane - 32) y += t; }
  return y;
}
// the LDS path of ems_group_rounds_reg (slices beyond a wavefront's registers): one index word at a time -- few registers, rarely run
__device__ __forceinline__ double ems_slice_sum_narrow(const u64* e, u32 width, const double* src) {
  double S = 0.0;
  for (u32 j = 0; j < width; j += 4) {   // (entries beyond the width point at the zero slot)
    const u64 w = e[(size_t)(j >> 2) * 64];
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    const double v0 = src[lo & 0xFFFFu], v1 = src[lo >> 16], v2 = src[hi & 0xFFFFu], v3 = src[hi >> 16];
    S += v0; S += v1; S += v2; S += v3;
  }
  return S;
}
template <int W, int NS>
__device__ __forceinline__ void ems_group_rounds_reg(const EmSellDev& P, u32 g, unsigned char* smem, u32 tid, u32 nthr, const double* alpha, const double* a,
                                                     double* alpha_out, double* a_out, int n_rounds, int clamp, int* s_hist, long long* clk = nullptr) {
  namespace L = kamd_em_sell;
  constexpr bool TREE = W >= 8 || (W == 4 && NS == 2);   // (the forms that run with 128 registers, one workgroup per CU)
  const int lane = lane_id();
  const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), NW = nthr >> 6;
  const u32 r0 = P.row_base[g], nR = P.row_base[g + 1] - r0, t0 = P.tr_base[g], nT = P.tr_base[g + 1] - t0;
  const u32 rs0 = P.rslice_base[g], nrs = P.rslice_base[g + 1] - rs0, cs0 = P.cslice_base[g], ncs = P.cslice_base[g + 1] - cs0;
  const u64 re0 = P.rell_base[g], ce0 = P.cell_base[g];
  const u32 nru = (u32)(P.rell_base[g + 1] - re0), ncu = (u32)(P.cell_base[g + 1] - ce0);
  // the layout kamd_em_sell::group_bytes() prices, in another order: the two gathered arrays (a, then g) come first, so that their byte
  // addresses fit 16 bits for every group of up to ~8 000 rows + transcripts (what is left of the priced bytes stays unused: alpha and
  // a are updated in place here -- the rows pass only reads a, the columns pass reads and writes a[t] / alpha[t] in the one lane that
  // finishes transcript t, and a barrier lies between the passes on either side)
  double* s_a0 = reinterpret_cast<double*>(smem);
  double* s_g = s_a0 + (nT + 1);
  double* s_al0 = s_g + (nR + 1);
  double* s_single = s_al0 + (nT + 1); double* s_eff = s_single + nT;
  u64* s_cw = reinterpret_cast<u64*>(s_eff + nT);
  u32* s_rdesc = reinterpret_cast<u32*>(s_cw + nR); u32* s_cdesc = s_rdesc + 2 * nrs;
  uint16_t* s_rell = reinterpret_cast<uint16_t*>(s_cdesc + 2 * ncs); uint16_t* s_cell = s_rell + ((nru + 1) & ~1u);
  const u32 a_off = 0u, g_off = (nT + 1) * 8u;
  const bool near = g_off + (nR + 1) * 8u <= 0x10000u;   // both gathered arrays inside the first 64 KB: byte addresses fit the 16-bit entries
  for (u32 i = tid; i < nT; i += nthr) {
    double al = alpha[t0 + i], av = a[t0 + i];
    if (clamp && al < 1e-7 / 10.0) { al = 0.0; av = 0.0; }   // the final round reads alpha < alpha_limit / 10 as 0 (:212-221)
    s_al0[i] = al; s_a0[i] = av; s_single[i] = P.single[t0 + i]; s_eff[i] = P.eff[t0 + i];
  }
  for (u32 i = tid; i < nR; i += nthr) s_cw[i] = P.cw[r0 + i];
  for (u32 i = tid; i < 2 * nrs; i += nthr) s_rdesc[i] = P.rdesc[2 * (u64)rs0 + i];
  for (u32 i = tid; i < 2 * ncs; i += nthr) s_cdesc[i] = P.cdesc[2 * (u64)cs0 + i];
  for (u32 i = tid; i < nru; i += nthr) { const uint16_t v = P.rell[re0 + i]; s_rell[i] = v == (uint16_t)L::SELL_PAD ? (uint16_t)nT : v; }
  for (u32 i = tid; i < ncu; i += nthr) { const uint16_t v = P.cell[ce0 + i]; s_cell[i] = v == (uint16_t)L::SELL_PAD ? (uint16_t)nR : v; }
  if (tid == 0) { s_al0[nT] = s_a0[nT] = 0.0; s_g[nR] = 0.0; }
  __syncthreads();
  // ---- this wavefront's own slices: everything that does not change from round to round goes into registers ----
  struct Own { u64 w[W]; u32 nq, seg, steps; int reach; bool has, meta, fin; };
  auto take = [&](const u32* desc, const uint16_t* ell, u32 s, u32 n_slices, u32 n_segs, u32 zero, u32 base_off) {
    Own o;
    o.has = near && s < n_slices; o.nq = 1; o.seg = 0; o.reach = 0; o.meta = false; o.fin = false; o.steps = 0;
    if (o.has && (((desc[2 * s + 1] & 0xFFFFu) + 3u) >> 2) > (u32)W) o.has = false;   // a slice wider than the registers hold: the LDS path takes it
    const u64 padw = (u64)(base_off + zero * 8u) * 0x0001000100010001ULL;
#pragma unroll
    for (int q = 0; q < W; q++) o.w[q] = padw;
    if (o.has) {
      const u32 d0 = desc[2 * s], d1 = desc[2 * s + 1];
      o.meta = (d0 & L::DESC_META) != 0;
      const u32 off = d0 & ~L::DESC_META, width = d1 & 0xFFFFu;
      o.nq = (u32)__builtin_amdgcn_readfirstlane((int)((width + 3u) >> 2));
      const u64* e = reinterpret_cast<const u64*>(ell + off + (o.meta ? 2 * L::SELL_META_WORDS : 0u)) + lane;
#pragma unroll
      for (int q = 0; q < W; q++) if ((u32)q < o.nq) {
        // index -> byte address of the value: four 16-bit fields of a word, no carry between them (base + 8 x index < 2^16)
        const u64 ix4 = e[(size_t)q * 64];
        o.w[q] = ((ix4 & 0x1FFF1FFF1FFF1FFFULL) << 3) + (u64)base_off * 0x0001000100010001ULL;
      }
      o.seg = (d1 >> 16) + (u32)lane; o.fin = o.seg < n_segs;
      if (o.meta) {
        const u32 w = reinterpret_cast<const u32*>(ell + off)[lane];
        o.reach = (int)((w >> 16) & 0x7Fu); o.seg = w & 0xFFFFu; o.fin = (w & L::META_ACTIVE) && (w & L::META_LAST);
        o.steps = pm_scan_seg_steps(o.reach, lane);
      }
      if (!o.fin) o.seg = 0;
    }
    return o;
  };
  Own ro[NS], co[NS];
  u32 r_cnt[NS], r_wc[NS];
  double c_single[NS], c_eff[NS], c_at[NS], c_cur[NS];
  u32 r_next = wv, c_next = wv;   // first slice of this wavefront on the LDS path
#pragma unroll
  for (int k = 0; k < NS; k++) {
    ro[k] = take(s_rdesc, s_rell, wv + (u32)k * NW, nrs, nR, nT, a_off);
    co[k] = take(s_cdesc, s_cell, wv + (u32)k * NW, ncs, nT, nR, g_off);
    // (register slices are a prefix of the wavefront's slices: the LDS path starts behind the last one held)
    if (ro[k].has && r_next == wv + (u32)k * NW) r_next += NW; else ro[k].has = false;
    if (co[k].has && c_next == wv + (u32)k * NW) c_next += NW; else co[k].has = false;
    r_cnt[k] = 0; r_wc[k] = 0; c_single[k] = 0.0; c_eff[k] = 1.0; c_at[k] = 0.0; c_cur[k] = 0.0;
    if (ro[k].has && ro[k].fin) { const u64 cwv = s_cw[ro[k].seg]; r_cnt[k] = (u32)cwv; r_wc[k] = (u32)(cwv >> 32); }
    if (co[k].has && co[k].fin) { c_single[k] = s_single[co[k].seg]; c_eff[k] = s_eff[co[k].seg]; c_at[k] = s_a0[co[k].seg]; c_cur[k] = s_al0[co[k].seg]; }
  }
  double* const al = s_al0; double* const av = s_a0;
  const unsigned char* const lds0 = smem;
  for (int r = 0; r < n_rounds; r++) {
    const bool tick = clk && r == EMS_CLK_ROUND && lane == 0;
    long long w0 = 0;
    if (tick) { clk[0] = clock64(); w0 = wall_clock64(); }
    // rows: S_e over the row's transcripts, then g_e = count_e / S_e (rows the reference skips get 0: count 0, :133-135;
    // denom below denorm_min, :156-158)
#pragma unroll
    for (int k = 0; k < NS; k++) {
      if (ro[k].has) {
        const bool tk = tick && k == 0;
        if (tk) clk[12] = clock64();
        double S = ems_reg_slice_sum<W, TREE>(ro[k].w, ro[k].nq, lds0);
        if (tk) { clk[13] = clock64(); clk[14] = (long long)((ro[k].nq * 4u) | (ro[k].meta ? 0x10000u : 0u)); }
        if (ro[k].meta) S = pm_scan_seg_masked(S, ro[k].reach, lane, ro[k].steps);
        if (ro[k].fin) s_g[ro[k].seg] = (r_cnt[k] == 0 || (double)r_wc[k] * S < 4.9406564584124654e-324) ? 0.0 : (double)r_cnt[k] / S;
        if (tk) clk[15] = clock64();
      }
    }
    for (u32 s = r_next; s < nrs; s += NW) {
      const u32 d0 = s_rdesc[2 * s], d1 = s_rdesc[2 * s + 1];
      const bool meta = (d0 & L::DESC_META) != 0;
      const u32 off = d0 & ~L::DESC_META, width = d1 & 0xFFFFu;
      double S = ems_slice_sum_narrow(reinterpret_cast<const u64*>(s_rell + off + (meta ? 2 * L::SELL_META_WORDS : 0u)) + lane, width, av);
      u32 seg = (d1 >> 16) + (u32)lane; bool fin = seg < nR;
      if (meta) {
        const u32 w = reinterpret_cast<const u32*>(s_rell + off)[lane];
        S = pm_scan_seg(S, (int)((w >> 16) & 0x7Fu), lane);
        seg = w & 0xFFFFu; fin = (w & L::META_ACTIVE) && (w & L::META_LAST);
      }
      if (fin) {
        const u64 cwv = s_cw[seg];
        const u32 cnt = (u32)cwv, wc = (u32)(cwv >> 32);
        s_g[seg] = (cnt == 0 || (double)wc * S < 4.9406564584124654e-324) ? 0.0 : (double)cnt / S;
      }
    }
    if (tick) clk[1] = clock64();
    __syncthreads();
    if (tick) clk[2] = clock64();
    // columns: next_t = single_t + a_t * sum of g over the transcript's rows, and the convergence test of :176-199
    int ch = 0;
    auto finish_col = [&](u32 seg, double at, double cur, double single, double eff, double acc, double& nx_out, double& a_out_v) {
      const double nx = single + at * acc;
      // :177-179 `fabs(next - alpha) / next > 1e-2`, without the division unless the quotient is within 1e-7 of the threshold (the
      // quotient's rounding error is 1e-16: outside that band the product test decides the same way as the reference's quotient)
      const double dd = fabs(nx - cur);
      bool moved = dd > 1.0000001e-2 * nx;
      if (dd > 0.9999999e-2 * nx && !moved) moved = dd / nx > 1e-2;
      nx_out = nx; a_out_v = nx / eff;
      al[seg] = nx; av[seg] = a_out_v;
      return nx > 1e-2 && moved;
    };
#pragma unroll
    for (int k = 0; k < NS; k++) {
      if (co[k].has) {
        const bool tk = tick && k == 0;
        if (tk) clk[8] = clock64();
        double acc = ems_reg_slice_sum<W, TREE>(co[k].w, co[k].nq, lds0);
        if (tk) { clk[9] = clock64(); clk[10] = (long long)((co[k].nq * 4u) | (co[k].meta ? 0x10000u : 0u)); }
        if (co[k].meta) acc = pm_scan_seg_masked(acc, co[k].reach, lane, co[k].steps);
        bool chg = false;
        if (co[k].fin) chg = finish_col(co[k].seg, c_at[k], c_cur[k], c_single[k], c_eff[k], acc, c_cur[k], c_at[k]);
        ch += __popcll(__ballot(chg));
        if (tk) clk[11] = clock64();
      }
    }
    for (u32 s = c_next; s < ncs; s += NW) {
      const u32 d0 = s_cdesc[2 * s], d1 = s_cdesc[2 * s + 1];
      const bool meta = (d0 & L::DESC_META) != 0;
      const u32 off = d0 & ~L::DESC_META, width = d1 & 0xFFFFu;
      double acc = ems_slice_sum_narrow(reinterpret_cast<const u64*>(s_cell + off + (meta ? 2 * L::SELL_META_WORDS : 0u)) + lane, width, s_g);
      u32 seg = (d1 >> 16) + (u32)lane; bool fin = seg < nT;
      if (meta) {
        const u32 w = reinterpret_cast<const u32*>(s_cell + off)[lane];
        acc = pm_scan_seg(acc, (int)((w >> 16) & 0x7Fu), lane);
        seg = w & 0xFFFFu; fin = (w & L::META_ACTIVE) && (w & L::META_LAST);
      }
      bool chg = false;
      if (fin) { double nx, an; chg = finish_col(seg, av[seg], al[seg], s_single[seg], s_eff[seg], acc, nx, an); }
      ch += __popcll(__ballot(chg));
    }
    if (ch && lane == 0) atomicAdd(&s_hist[r], ch);
    if (tick) clk[3] = clock64();
    __syncthreads();
    if (tick) { clk[4] = clock64(); clk[5] = (long long)(nrs | (ncs << 16)); clk[6] = (long long)(nru | ((u64)ncu << 32)); clk[7] = wall_clock64() - w0; }
  }
  for (u32 i = tid; i < nT; i += nthr) { alpha_out[t0 + i] = al[i]; a_out[t0 + i] = av[i]; }
}
// The stop rule of EMAlgorithm::run (:202-205) on the change counts of the PREVIOUS chunk of rounds: a chunk that was launched
// speculatively behind the one the run stops in has nothing to do (its input stays the checkpoint the host replays from).
// Block 0 also hands the previous chunk's counts to the host (pinned, mapped memory the host polls: no stream synchronisation per chunk).
struct EmsPrev { const int* hist; int n; int base; int min_rounds; int* host_hist; int* host_seq; int seq; };
__device__ __forceinline__ bool ems_prev_stopped(const EmsPrev& v, int* s_flag) {
  if (!v.hist) return false;
  if (threadIdx.x < 64) {
    const int i = (int)threadIdx.x;
    const int h = i < v.n ? v.hist[i] : 1;
    const bool stop = i < v.n && h == 0 && v.base + i > v.min_rounds;
    const u64 m = __ballot(stop);
    if (blockIdx.x == 0 && v.host_hist) {
      if (i < v.n) v.host_hist[i] = h;
      __threadfence_system();
      if (i == 0) __hip_atomic_store(v.host_seq, v.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i == 0) *s_flag = m != 0;
  }
  __syncthreads();
  return *s_flag != 0;
}
// one workgroup per group (the large size class, or every group when there is only one class): groups g_first + blockIdx.x
// W > 0: the register-resident form (ems_group_rounds_reg) for split lengths up to 4 W; W = 0: everything out of LDS
// (two workgroups of 16 wavefronts per CU = 8 wavefronts per SIMD = 64 VGPRs: the register-resident forms are held to that)
template <bool CLK, int EXP = 0, int W = 0, int NS = 1>
__global__ __launch_bounds__(EMS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu((W >= 8 || (W == 4 && NS == 2)) ? 4 : 8))) void k_em_sell(EmSellDev P, u32 g_first, const double* alpha, const double* a, double* alpha_out, double* a_out,
                                                           int n_rounds, int clamp, int* hist, EmsPrev prev, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ems_smem[];
  __shared__ int s_hist[EML_MAX_ROUNDS];
  __shared__ int s_stop;
  if (ems_prev_stopped(prev, &s_stop)) return;
  if (threadIdx.x < EML_MAX_ROUNDS) s_hist[threadIdx.x] = 0;
  __syncthreads();
  long long* my_clk = CLK ? clk + ((size_t)blockIdx.x * (EMS_MAX_BLOCK / 64) + (threadIdx.x >> 6)) * EMS_CLK_WORDS : nullptr;
  if constexpr (W > 0) ems_group_rounds_reg<W, NS>(P, g_first + blockIdx.x, ems_smem, threadIdx.x, blockDim.x, alpha, a, alpha_out, a_out, n_rounds, clamp, s_hist, my_clk);
  else ems_group_rounds<false, EXP>(P, g_first + blockIdx.x, ems_smem, threadIdx.x, blockDim.x, alpha, a, alpha_out, a_out, n_rounds, clamp, s_hist, my_clk);
  __syncthreads();
  if (hist && (int)threadIdx.x < n_rounds && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}
// one WAVEFRONT per group (the small size class): the wavefronts of a workgroup run their groups independently -- no block barrier
// inside the rounds, so a CU holds a few dozen groups at different points of their rounds and the LDS pipe always has work
constexpr int EMS_WAVE_TEAMS = 4;   // groups per workgroup
__global__ __launch_bounds__(64 * EMS_WAVE_TEAMS) void k_em_sell_wave(EmSellDev P, u32 n_small, u32 team_bytes, const double* alpha, const double* a,
                                                                        double* alpha_out, double* a_out, int n_rounds, int clamp, int* hist, EmsPrev prev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ems_smem[];
  __shared__ int s_hist[EML_MAX_ROUNDS];
  __shared__ int s_stop;
  if (ems_prev_stopped(prev, &s_stop)) return;
  if (threadIdx.x < EML_MAX_ROUNDS) s_hist[threadIdx.x] = 0;
  __syncthreads();
  const u32 team = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const u32 g = blockIdx.x * EMS_WAVE_TEAMS + team;
  if (g < n_small) ems_group_rounds<true>(P, g, ems_smem + (size_t)team * team_bytes, (u32)lane_id(), 64u, alpha, a, alpha_out, a_out, n_rounds, clamp, s_hist);
  __syncthreads();
  if (hist && (int)threadIdx.x < n_rounds && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}
// ---- conversion of the device-built CSR plan (em_local_setup_device) into the sliced-ELLPACK layout --------------------------
struct SellBuild {
  // CSR plan (device)
  const u32* row_base; const u32* tr_base; const u32* row_ptr; const u32* col_ptr; const uint16_t* row_tr; const uint16_t* col_row; const u64* nz_base;
  const u64* cw; const double* single; const double* eff; const u32* tr_id; u32 n_groups; u32 R; u32 M; u32 cap;
  // scratch
  u32* rlen; u32* clen; u32* rnew; u32* cnew; u32* rlane; u32* clane; u32* rvl; u32* cvl; u32* gsz;   // gsz: 4 words per group
  // SELL plan (device)
  const u32* rslice_base; const u32* cslice_base; const u64* rell_base; const u64* cell_base;
  u32* rdesc; u32* cdesc; uint16_t* rell; uint16_t* cell; u64* cw_new; double* single_new; double* eff_new; u32* tr_id_new;
};
__device__ __forceinline__ u32 sell_group_of(const u32* base, u32 n_groups, u32 i) {   // last g with base[g] <= i
  u32 lo = 0, hi = n_groups;
  while (hi - lo > 1) { const u32 mid = (lo + hi) / 2; if (base[mid] <= i) lo = mid; else hi = mid; }
  return lo;
}
__global__ void k_sell_lens(SellBuild B) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B.R) { const u32 g = sell_group_of(B.row_base, B.n_groups, i); B.rlen[i] = B.row_ptr[(u64)i + g + 1] - B.row_ptr[(u64)i + g]; }
  if (i < B.M) { const u32 g = sell_group_of(B.tr_base, B.n_groups, i); B.clen[i] = B.col_ptr[(u64)i + g + 1] - B.col_ptr[(u64)i + g]; }
}
// kamd_em_sell::layout_group for one direction of one group by ONE WAVEFRONT (the header's version is what one thread -- or the CPU
// emulation -- runs; one thread per group left the chip idle: 1009 threads, ~1 ms per pass).  Same layout rules: split segments
// first, in the caller's order, never straddling a slice; then the others by decreasing length; a slice's width is its longest
// lane; slices that hold split lanes carry a word of metadata per lane.  Segments of EQUAL length keep the caller's order (as on the host).
constexpr int SELL_BUILD_WAVES = 4;    // wavefronts (group directions) per block
struct SellWaveScratch { u32 hist[kamd_em_sell::SELL_LANES + 1], start[kamd_em_sell::SELL_LANES + 1], cur[kamd_em_sell::SELL_LANES + 1]; };
template <class Sink>
__device__ kamd_em_sell::LayoutSize sell_layout_wave(const u32* __restrict__ len, u32 n, u32 cap, Sink& sink, SellWaveScratch& S) {
  namespace L = kamd_em_sell;
  const int lane = lane_id();
  for (u32 b = lane; b <= L::SELL_LANES; b += 64) { S.hist[b] = 0; S.cur[b] = 0; }
  __builtin_amdgcn_wave_barrier();
  // 1. histogram of the unsplit lengths; the split segments are laid out as they are met (their order is the caller's)
  u32 n_split = 0, pos = 0;            // pos: next free absolute lane of the split part (wave-uniform)
  u32 off = 0;                         // u16 offset of the slice that is open (wave-uniform)
  u32 width = 0;                       // its width so far
  u32 mw = 0;                          // this lane's metadata word in the open slice
  u32 closed = 0;                      // slices closed so far
  for (u32 c0 = 0; c0 < n; c0 += 64) {
    const u32 i = c0 + lane;
    const u32 l = i < n ? len[i] : 0;
    if (i < n && l <= cap) atomicAdd(&S.hist[l], 1u);
    u64 m = __ballot(i < n && l > cap);
    while (m) {
      const int b = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      const u32 lb = (u32)__shfl((int)l, b, 64);
      const u32 nv = L::seg_lanes(lb, cap), vl = L::seg_vlen(lb, cap);
      if ((pos % L::SELL_LANES) + nv > L::SELL_LANES) {   // does not fit: close the slice, the rest of its lanes stay inactive
        sink.meta(off, lane, mw);
        if (lane == 0) sink.slice(closed, off | L::DESC_META, width);
        off += 2 * L::SELL_META_WORDS + L::quad_width(width) * L::SELL_LANES; ++closed; width = 0; mw = 0;
        pos = (pos / L::SELL_LANES + 1) * L::SELL_LANES;
      }
      if (Sink::wants_segments && lane == b) sink.seg(c0 + b, n_split, pos, nv, vl);
      const u32 rel = pos % L::SELL_LANES;
      if ((u32)lane >= rel && (u32)lane < rel + nv) {
        const u32 v = (u32)lane - rel;
        mw = n_split | (v << 16) | (v + 1 == nv ? L::META_LAST : 0u) | L::META_ACTIVE;
      }
      width = vl > width ? vl : width;
      pos += nv; ++n_split;
      if (pos % L::SELL_LANES == 0) {                        // exactly full
        sink.meta(off, lane, mw);
        if (lane == 0) sink.slice(closed, off | L::DESC_META, width);
        off += 2 * L::SELL_META_WORDS + L::quad_width(width) * L::SELL_LANES; ++closed; width = 0; mw = 0;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  const u32 split_lanes = pos;
  // 2. start lane of every length (decreasing), by lane 0; 64 LDS reads
  if (lane == 0) { u32 c = split_lanes; for (u32 b = L::SELL_LANES; b >= 1; b--) { S.start[b] = c; c += S.hist[b]; } S.start[0] = c; }
  __builtin_amdgcn_wave_barrier();
  const u32 total_lanes = S.start[0];
  if (Sink::wants_segments) {
    for (u32 c0 = 0; c0 < n; c0 += 64) {
      const u32 i = c0 + lane;
      // equal lengths keep the caller's order (the canonical numbering of kamd_em_local.h: neighbours in it are neighbours in a slice):
      // a lane's place inside its length class = the class's count so far + the lanes below it in this step with the same length
      const u32 l = i < n ? len[i] : 0u;
      const bool act = i < n && l <= cap;
      u64 same = __ballot(act);
#pragma unroll
      for (int b = 0; b < 7; b++) { const u64 bal = __ballot((l >> b) & 1u); same &= ((l >> b) & 1u) ? bal : ~bal; }
      const u32 below = (u32)__popcll(same & ((1ULL << lane) - 1ULL)), cls = (u32)__popcll(same);
      u32 base = 0;
      if (act) base = S.cur[l];
      __builtin_amdgcn_wave_barrier();
      if (act) {
        const u32 p = S.start[l] + base + below;
        sink.seg(i, n_split + (p - split_lanes), p, 1u, l);
        if (below + 1 == cls) S.cur[l] = base + cls;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // 3. the slice the split part left open takes the longest unsplit segments; then the plain slices, 64 at a time
  auto len_at = [&](u32 p) -> u32 {   // length of the unsplit segment at absolute lane p (split_lanes <= p < total_lanes)
    u32 b = L::SELL_LANES;
    while (b > 1 && !(p >= S.start[b] && p < S.start[b] + S.hist[b])) --b;
    return b;
  };
  const u32 n_slices = (total_lanes + L::SELL_LANES - 1) / L::SELL_LANES;
  if (split_lanes % L::SELL_LANES) {
    const u32 lo = closed * L::SELL_LANES, p = lo + (u32)lane;
    if (split_lanes < total_lanes) { const u32 b = len_at(split_lanes); width = b > width ? b : width; }
    if (p >= split_lanes && p < total_lanes) mw = (n_split + (p - split_lanes)) | L::META_LAST | L::META_ACTIVE;
    sink.meta(off, lane, mw);
    if (lane == 0) sink.slice(closed, off | L::DESC_META, width);
    off += 2 * L::SELL_META_WORDS + L::quad_width(width) * L::SELL_LANES; ++closed;
  }
  for (u32 s0 = closed; s0 < n_slices; s0 += 64) {
    const u32 si = s0 + (u32)lane;
    u32 w = 0;
    if (si < n_slices) w = len_at(si * L::SELL_LANES);
    const u32 mine = L::quad_width(w) * L::SELL_LANES;
    const u32 incl = wave_incl_scan(mine);
    if (si < n_slices) sink.slice(si, off + (incl - mine), w | ((n_split + (si * L::SELL_LANES - split_lanes)) << 16));
    off += (u32)__shfl((int)incl, 63, 64);
  }
  return L::LayoutSize{n_slices, off};
}
__global__ __launch_bounds__(64 * SELL_BUILD_WAVES) void k_sell_sizes(SellBuild B) {
  __shared__ SellWaveScratch scr[SELL_BUILD_WAVES];
  const u32 w = (blockIdx.x * blockDim.x + threadIdx.x) / 64, g = w >> 1;
  if (g >= B.n_groups) return;
  kamd_em_sell::NullSink ns;
  const kamd_em_sell::LayoutSize z = (w & 1) ? sell_layout_wave(B.clen + B.tr_base[g], B.tr_base[g + 1] - B.tr_base[g], B.cap, ns, scr[threadIdx.x / 64])
                                             : sell_layout_wave(B.rlen + B.row_base[g], B.row_base[g + 1] - B.row_base[g], B.cap, ns, scr[threadIdx.x / 64]);
  if (lane_id() == 0) { B.gsz[4 * g + 2 * (w & 1)] = z.n_slices; B.gsz[4 * g + 2 * (w & 1) + 1] = z.n_u16; }
}
struct SellDevSink {
  static const bool wants_segments = true;
  u32* new_id; u32* lane; u32* vlen; u32* desc; uint16_t* ell; u32 seg0; u32 desc0; u64 ell0;
  __device__ void seg(u32 old, u32 id, u32 ln, u32, u32 vl) const { new_id[seg0 + old] = id; lane[seg0 + old] = ln; vlen[seg0 + old] = vl; }
  __device__ void slice(u32 i, u32 d0, u32 d1) const { desc[2 * (u64)(desc0 + i)] = d0; desc[2 * (u64)(desc0 + i) + 1] = d1; }
  __device__ void meta(u32 off, u32 l, u32 w) const { ell[ell0 + off + 2 * l] = (uint16_t)w; ell[ell0 + off + 2 * l + 1] = (uint16_t)(w >> 16); }
};
__global__ __launch_bounds__(64 * SELL_BUILD_WAVES) void k_sell_layout(SellBuild B) {
  __shared__ SellWaveScratch scr[SELL_BUILD_WAVES];
  const u32 w = (blockIdx.x * blockDim.x + threadIdx.x) / 64, g = w >> 1;
  if (g >= B.n_groups) return;
  if (w & 1) {
    SellDevSink sc{B.cnew, B.clane, B.cvl, B.cdesc, B.cell, B.tr_base[g], B.cslice_base[g], B.cell_base[g]};
    sell_layout_wave(B.clen + B.tr_base[g], B.tr_base[g + 1] - B.tr_base[g], B.cap, sc, scr[threadIdx.x / 64]);
  } else {
    SellDevSink sr{B.rnew, B.rlane, B.rvl, B.rdesc, B.rell, B.row_base[g], B.rslice_base[g], B.rell_base[g]};
    sell_layout_wave(B.rlen + B.row_base[g], B.row_base[g + 1] - B.row_base[g], B.cap, sr, scr[threadIdx.x / 64]);
  }
}
// entries with the other direction's new ids, and the per-segment constants in the new order; one thread per old segment
__global__ void k_sell_entries(SellBuild B) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B.R) {
    const u32 g = sell_group_of(B.row_base, B.n_groups, i), r0 = B.row_base[g], t0 = B.tr_base[g];
    B.cw_new[r0 + B.rnew[i]] = B.cw[i];
    const u32* rd = B.rdesc + 2 * (u64)B.rslice_base[g];
    const uint16_t* src = B.row_tr + B.nz_base[g];
    const u32 b = B.row_ptr[(u64)i + g], e = B.row_ptr[(u64)i + g + 1];
    const u32 ln = B.rlane[i], vl = B.rvl[i];
    for (u32 j = b, q = 0; j < e; j++, q++)
      B.rell[B.rell_base[g] + kamd_em_sell::entry_pos(rd, ln, vl, q)] = (uint16_t)B.cnew[t0 + src[j]];
  }
  if (i < B.M) {
    const u32 g = sell_group_of(B.tr_base, B.n_groups, i), r0 = B.row_base[g], t0 = B.tr_base[g];
    const u32 m = t0 + B.cnew[i];
    B.single_new[m] = B.single[i]; B.eff_new[m] = B.eff[i]; B.tr_id_new[m] = B.tr_id[i];
    const u32* cd = B.cdesc + 2 * (u64)B.cslice_base[g];
    const uint16_t* src = B.col_row + B.nz_base[g];
    const u32 b = B.col_ptr[(u64)i + g], e = B.col_ptr[(u64)i + g + 1];
    const u32 ln = B.clane[i], vl = B.cvl[i];
    for (u32 j = b, q = 0; j < e; j++, q++)
      B.cell[B.cell_base[g] + kamd_em_sell::entry_pos(cd, ln, vl, q)] = (uint16_t)B.rnew[r0 + src[j]];
  }
}
__global__ void k_eml_init(double* alpha, double* a, const double* __restrict__ eff_m, u64 M, double a0) {   // alpha_ = 1/T (:38)
  const u64 m = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (m < M) { alpha[m] = a0; a[m] = a0 / eff_m[m]; }
}
// the data-parallel steps of kamd_em_local.h, one thread per index
// step F (kamd_em_local.h step_tr_f) with the slot hand-out aggregated per wavefront: the transcripts of a gene are neighbours in
// transcript space and belong to one group, so the 64 lanes of a wavefront used to queue 64 atomics on ONE group counter (~12 ns each at
// the memory side: 163 us for config #3's 192 669 transcripts); here the lanes that share the leader's group take consecutive slots from
// one atomic.  The order inside a group changes, the result does not: step F2 ranks the members (canonical numbering).
__device__ __forceinline__ void eml_step_tr_f_wave(u64 t, const kamd_em_local::BuildArgs& A) {
  const bool act = t < A.T && A.in_multi[t];
  const u32 g = act ? kamd_em_local::eml_group_of(A, A.label[t]) : 0xFFFFFFFFu;
  u64 m = __ballot(act);
  const u64 lt = (1ULL << lane_id()) - 1ULL;
  while (m) {
    const int leader = __ffsll((long long)m) - 1;
    const u32 lg = (u32)__shfl((int)g, leader, 64);
    const u64 same = __ballot(act && g == lg);
    u32 base = 0;
    if (lane_id() == leader) base = atomicAdd(&A.tr_fill[lg], (u32)__popcll(same));
    base = (u32)__shfl((int)base, leader, 64);
    if (act && g == lg) A.tmp_tr_id[(u64)A.tr_base[lg] + base + (u32)__popcll(same & lt)] = (u32)t;
    m &= ~same;
  }
}
// step B (step_tr_b) the same way: one atomic per (wavefront, component) instead of one per transcript
__device__ __forceinline__ void eml_step_tr_b_wave(u64 t, const kamd_em_local::BuildArgs& A) {
  const bool act = t < A.T && A.in_multi[t];
  const u32 r = act ? A.label[t] : 0xFFFFFFFFu;
  u64 m = __ballot(act);
  while (m) {
    const int leader = __ffsll((long long)m) - 1;
    const u32 lr = (u32)__shfl((int)r, leader, 64);
    const u64 same = __ballot(act && r == lr);
    if (lane_id() == leader) atomicAdd(&A.c_tr[lr], (u32)__popcll(same));
    m &= ~same;
  }
}
// step G (step_rows_g) with the row slots of a group reserved per WORKGROUP: the rows of a group arrive from everywhere in EC order, so a wavefront
// cannot aggregate them, and one returning atomic per row on the groups' few hundred counters serializes at the memory side (config #3: 618 012
// rows on 505 counters, 193 us).  Here a workgroup counts its 4 096 rows per group in LDS, takes one range per (workgroup, group) and hands the
// slots of the range out by the LDS ranks.  As with steps B / F the arrival order changes and the result does not (step G2 ranks the rows).
constexpr int RG_BLOCK = 1024, RG_PER = 4, RG_BINS = 8192;
__global__ __launch_bounds__(RG_BLOCK) void k_eml_rows_g(kamd_em_local::BuildArgs A) {
  __shared__ u32 bin[RG_BINS];   // rows of this workgroup in the group, then the first slot of its range
  const u32 tid = threadIdx.x;
  for (u32 i = tid; i < A.n_groups; i += RG_BLOCK) bin[i] = 0;
  __syncthreads();
  u32 g[RG_PER], loc[RG_PER]; u64 key[RG_PER];
  const u64 e0 = (u64)blockIdx.x * (RG_BLOCK * RG_PER) + tid;
#pragma unroll
  for (int j = 0; j < RG_PER; j++) {
    const u64 e = e0 + (u64)j * RG_BLOCK;
    g[j] = 0xFFFFFFFFu; loc[j] = 0; key[j] = 0;
    if (e >= A.n_ecs) continue;
    const u64 a = A.ec_off[e], b = A.ec_off[e + 1];
    if (b - a < 2) continue;
    const u32 first = A.ec_ids[a];
    g[j] = kamd_em_local::eml_group_of(A, A.label[first]);
    loc[j] = atomicAdd(&bin[g[j]], 1u);
    u64 h = 0x9E3779B97F4A7C15ULL;   // (the row's content key, as step_rows_g computes it)
    for (u64 q = a; q < b; q++) { h ^= A.ec_ids[q]; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 29; }
    key[j] = ((u64)A.local_of[first] << 48) | (h >> 16);
  }
  __syncthreads();
  for (u32 i = tid; i < A.n_groups; i += RG_BLOCK) { const u32 n = bin[i]; if (n) bin[i] = atomicAdd(&A.row_fill[i], n); }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RG_PER; j++) {
    if (g[j] == 0xFFFFFFFFu) continue;
    const u32 rn = A.row_base[g[j]] + bin[g[j]] + loc[j];
    A.row_key[rn] = key[j];
    A.row_e[rn] = (u32)(e0 + (u64)j * RG_BLOCK);
  }
}
template <int S>
__global__ void k_eml_step(kamd_em_local::BuildArgs A, u64 n) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (S == 1) { eml_step_tr_b_wave(i < n ? i : ~0ULL, A); return; }
  if constexpr (S == 3) { eml_step_tr_f_wave(i < n ? i : ~0ULL, A); return; }
  if (i >= n) return;
  if constexpr (S == 0) kamd_em_local::step_rows_a(i, A);
  else if constexpr (S == 1) kamd_em_local::step_tr_b(i, A);
  else if constexpr (S == 2) kamd_em_local::step_root_d(i, A);
  else if constexpr (S == 3) kamd_em_local::step_tr_f(i, A);
  else if constexpr (S == 4) kamd_em_local::step_rows_g(i, A);
  else if constexpr (S == 5) kamd_em_local::step_rows_i(i, A);
  else if constexpr (S == 6) kamd_em_local::step_m_j(i, kamd_em_local::eml_group_of_slot(A, i), A);
  else if constexpr (S == 7) kamd_em_local::step_group_j(i, A);
  else if constexpr (S == 9) kamd_em_local::step_root_c(i, A);
  else if constexpr (S == 10) kamd_em_local::step_m_f2(i, kamd_em_local::eml_group_of_slot(A, i), A);
  else if constexpr (S == 11) kamd_em_local::step_slot_g2(i, kamd_em_local::eml_group_of_row_slot(A, i), A);
  else if constexpr (S == 12) kamd_em_local::step_ent_k2(i, A);
  else kamd_em_local::step_rows_k(i, A);
}
// ---- the second half of the plan (kamd_em_local.h steps I, J, K, K2 and the two scans between them) as ONE kernel, a workgroup per
// group: after G2 a group's rows are consecutive and in their final order, its transcripts occupy a consecutive range of m-space, so
// row offsets, column counts, column offsets, the transposed entries and their canonical order are all local to the group -- LDS
// atomics and block scans instead of 2 x 4.7 M memory-side atomics (0.25 ms each), two device-wide scans and the ranking pass.
// LDS: row offsets [nR + 1] | column offsets [nT + 1] | column cursors [nT] (u32), then per transposed entry its row and its column (u16).
__device__ __forceinline__ void eml_block_excl_scan(u32* a, u32 n, u32* s_w) {   // in place; a[n] receives the total; s_w: one word per wavefront
  const u32 nthr = blockDim.x, tid = threadIdx.x;
  const u32 per = (n + nthr - 1) / nthr;
  const u32 b0 = tid * per < n ? tid * per : n, b1 = b0 + per < n ? b0 + per : n;
  u32 run = 0;
  for (u32 i = b0; i < b1; i++) run += a[i];
  u32 incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(incl, d, 64); if (lane_id() >= d) incl += t; }
  const u32 w = tid >> 6;
  if (lane_id() == 63) s_w[w] = incl;
  __syncthreads();
  u32 woff = 0;
  for (u32 j = 0; j < w; j++) woff += s_w[j];
  u32 acc = woff + incl - run;
  for (u32 i = b0; i < b1; i++) { const u32 x = a[i]; a[i] = acc; acc += x; }
  if (tid == nthr - 1) a[n] = woff + incl;
  __syncthreads();
}
// steps F2 / G2 (the canonical numbering) with the group's keys in LDS: one workgroup per group loads the keys its members were handed
// out in arrival order and every thread ranks its members against them (all lanes read the same LDS word: a broadcast) -- the
// per-member kernels walked the group's keys in global memory, a few hundred dependent-latency loads per member
__global__ __launch_bounds__(BLOCK) void k_eml_rank_tr(kamd_em_local::BuildArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_smem[];
  u32* s_t = reinterpret_cast<u32*>(gb_smem);
  const u32 g = blockIdx.x, lo = A.tr_base[g], n = A.tr_base[g + 1] - lo;
  for (u32 i = threadIdx.x; i < n; i += blockDim.x) s_t[i] = A.tmp_tr_id[lo + i];
  __syncthreads();
  for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
    const u32 t = s_t[i];
    u32 rank = 0;
    u32 j = 0;
    for (; j + 8 <= n; j += 8) {
      u32 k[8];
#pragma unroll
      for (int q = 0; q < 8; q++) k[q] = s_t[j + q];
#pragma unroll
      for (int q = 0; q < 8; q++) rank += k[q] < t ? 1u : 0u;
    }
    for (; j < n; j++) rank += s_t[j] < t ? 1u : 0u;
    A.local_of[t] = rank;
    const u64 m = (u64)lo + rank;
    A.tr_id[m] = t; A.single[m] = A.single_all[t]; A.eff_m[m] = A.eff[t];
  }
}
// (1024 lanes per group: the ranking is quadratic in the group's rows -- 1 240 of them in the groups of ~9 300 entries round 4 made the
// default -- and a workgroup of 256 lanes took 255 us for it)
constexpr int EML_RANK_BLOCK = 1024;
__global__ __launch_bounds__(EML_RANK_BLOCK) void k_eml_rank_rows(kamd_em_local::BuildArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_smem[];
  const u32 g = blockIdx.x, lo = A.row_base[g], n = A.row_base[g + 1] - lo;
  u64* s_k = reinterpret_cast<u64*>(gb_smem);
  u32* s_e = reinterpret_cast<u32*>(s_k + n);
  for (u32 i = threadIdx.x; i < n; i += blockDim.x) { s_k[i] = A.row_key[lo + i]; s_e[i] = A.row_e[lo + i]; }
  __syncthreads();
  for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
    const u64 key = s_k[i];
    const u32 e = s_e[i];
    u32 rank = 0;
    u32 j = 0;
    for (; j + 8 <= n; j += 8) {   // eight independent LDS reads per trip (one at a time the loop waits out the LDS latency 600 times per row)
      u64 k[8];
#pragma unroll
      for (int q = 0; q < 8; q++) k[q] = s_k[j + q];
      bool tie = false;
#pragma unroll
      for (int q = 0; q < 8; q++) { rank += k[q] < key ? 1u : 0u; tie = tie || k[q] == key; }
      if (tie) {   // equal keys (the row itself, or a 48-bit hash collision): the EC index decides
#pragma unroll
        for (int q = 0; q < 8; q++) rank += (k[q] == key && s_e[j + q] < e) ? 1u : 0u;
      }
    }
    for (; j < n; j++) { const u64 kj = s_k[j]; rank += (kj < key || (kj == key && s_e[j] < e)) ? 1u : 0u; }
    const u32 rn = lo + rank;
    A.row_new[e] = rn;
    A.row_e_final[rn] = e;
    A.len_new[rn] = (u32)(A.ec_off[(u64)e + 1] - A.ec_off[e]);
    A.cw[rn] = (u64)A.counts[e] | ((u64)(A.wcounts ? A.wcounts[e] : A.counts[e]) << 32);
  }
}
__global__ __launch_bounds__(BLOCK) void k_eml_group_build(kamd_em_local::BuildArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_smem[];
  __shared__ u32 s_w[BLOCK / 64];
  const u32 g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const u32 r0 = A.row_base[g], nR = A.row_base[g + 1] - r0, t0 = A.tr_base[g], nT = A.tr_base[g + 1] - t0;
  const u64 z0 = A.nz_base[g];
  const u32 nnz = (u32)(A.nz_base[g + 1] - z0);
  u32* s_roff = reinterpret_cast<u32*>(gb_smem);
  u32* s_coff = s_roff + (nR + 1);
  u32* s_cur = s_coff + (nT + 1);
  uint16_t* s_erow = reinterpret_cast<uint16_t*>(s_cur + nT);
  uint16_t* s_ecol = s_erow + ((nnz + 1) & ~1u);
  // row offsets (relative to the group's first entry)
  for (u32 r = tid; r < nR; r += nthr) s_roff[r] = A.len_new[r0 + r];
  for (u32 l = tid; l <= nT; l += nthr) s_coff[l] = 0;
  for (u32 l = tid; l < nT; l += nthr) s_cur[l] = 0;
  __syncthreads();
  eml_block_excl_scan(s_roff, nR, s_w);
  for (u32 r = tid; r <= nR; r += nthr) A.row_ptr[(u64)r0 + r + g] = s_roff[r];   // (row r0 + r of group g sits at row_ptr[r0 + r + g]; the last one closes the group)
  // the rows' entries (local transcript ids) and the column counts
  for (u32 r = tid; r < nR; r += nthr) {
    const u32 e = A.row_e_final[r0 + r];
    const u64 a = A.ec_off[e];
    const u32 len = s_roff[r + 1] - s_roff[r], at = s_roff[r];
    for (u32 j = 0; j < len; j++) {
      const u32 l = A.local_of[A.ec_ids[a + j]];
      A.row_tr[z0 + at + j] = (uint16_t)l;
      atomicAdd(&s_coff[l], 1u);
    }
  }
  __syncthreads();
  eml_block_excl_scan(s_coff, nT, s_w);
  for (u32 l = tid; l <= nT; l += nthr) A.col_ptr[(u64)t0 + l + g] = s_coff[l];
  // the transposed entries in arrival order ...
  for (u32 r = tid; r < nR; r += nthr) {
    const u32 len = s_roff[r + 1] - s_roff[r], at = s_roff[r];
    for (u32 j = 0; j < len; j++) {
      const u32 l = A.row_tr[z0 + at + j];   // (written by this thread above)
      const u32 p = s_coff[l] + atomicAdd(&s_cur[l], 1u);
      s_erow[p] = (uint16_t)r; s_ecol[p] = (uint16_t)l;
    }
  }
  __syncthreads();
  // ... and in their final one: a column's entries by row (a row occurs once in a column)
  for (u32 p = tid; p < nnz; p += nthr) {
    const u32 l = s_ecol[p], v = s_erow[p];
    const u32 lo = s_coff[l], hi = s_coff[l + 1];
    u32 rank = 0;
    u32 q = lo;
    for (; q + 8 <= hi; q += 8) {
      u32 k[8];
#pragma unroll
      for (int i = 0; i < 8; i++) k[i] = s_erow[q + i];
#pragma unroll
      for (int i = 0; i < 8; i++) rank += k[i] < v ? 1u : 0u;
    }
    for (; q < hi; q++) rank += s_erow[q] < v ? 1u : 0u;
    A.col_row[z0 + lo + rank] = (uint16_t)v;
  }
}
// the largest connected component (entries, rows, transcripts: three maxima) -- what decides at once whether a matrix can take the
// component-local form as a whole; stats[0..2], zeroed by the caller
struct CompStats { u32 max_nnz, max_rows, max_tr, pad; };
__global__ void k_comp_stats(const u32* __restrict__ c_nnz, const u32* __restrict__ c_rows, const u32* __restrict__ c_tr, u64 T, u32* stats) {
  // (grid-stride over at most COMP_STATS_BLOCKS workgroups, one atomic per workgroup and none if its maximum is not above the published one: one
  // atomic per wavefront on three words of one line was 3 010 serialized atomics per word for config #3's 192 669 transcripts -- 105 us)
  u32 v0 = 0, v1 = 0, v2 = 0;
  for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < T; r += (u64)gridDim.x * blockDim.x) {
    v0 = max(v0, c_nnz[r]);
    if (c_rows) v1 = max(v1, c_rows[r]);
    if (c_tr) v2 = max(v2, c_tr[r]);
  }
  __shared__ u32 red[3][BLOCK / 64];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { v0 = max(v0, (u32)__shfl_down(v0, d, 64)); v1 = max(v1, (u32)__shfl_down(v1, d, 64)); v2 = max(v2, (u32)__shfl_down(v2, d, 64)); }
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0) { red[0][w] = v0; red[1][w] = v1; red[2][w] = v2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) { v0 = max(v0, red[0][i]); v1 = max(v1, red[1][i]); v2 = max(v2, red[2][i]); }
    if (v0 > __atomic_load_n(&stats[0], __ATOMIC_RELAXED)) atomicMax(&stats[0], v0);
    if (v1 > __atomic_load_n(&stats[1], __ATOMIC_RELAXED)) atomicMax(&stats[1], v1);
    if (v2 > __atomic_load_n(&stats[2], __ATOMIC_RELAXED)) atomicMax(&stats[2], v2);
  }
}
constexpr unsigned COMP_STATS_BLOCKS = 256;
int cc_labels(kamd_ctx* c, const u64* d_ec_off, const u32* d_ec_ids, u64 n_ecs, u64 T) {
  if (int rc = c->pt_label.ensure((T + 1) * sizeof(u32), 0, c->stream)) return rc;
  hipLaunchKernelGGL(k_cc_init, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, c->pt_label.as<u32>(), T);
  if (n_ecs) hipLaunchKernelGGL(k_cc_union, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, d_ec_off, d_ec_ids, n_ecs, c->pt_label.as<u32>());
  hipLaunchKernelGGL(k_cc_flatten, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, c->pt_label.as<u32>(), T);
  HIPC(hipGetLastError());
  return 0;
}
// The plan built on the device: component labels by the kernels the partitioned EM uses, then the steps of
// kamd_em_local.h with scans in between.  The host only sees the per-group sizes (budget check, bases) and, for the final
// scatter, tr_id and the singleton counts.  0 = ok (P holds the host part, *dev the device part), 1 = not applicable.
// small_limit != 0: two size classes (kamd_em_local.h BuildArgs): components of at most small_limit entries in groups of about
// target_small entries (P->n_small of them, first), the others in groups of about `target` entries.
int em_local_setup_device(kamd_ctx* c, const u64* d_ec_off, const u32* d_ec_ids, const u32* d_counts, const u32* d_wcounts, u64 n_ecs, u64 nnz,
                          const double* eff_lens, u64 T, u64 budget, u64 target, kamd_em_local::Plan* P, EmLocalDev* dev,
                          kamd_em_local::BuildArgs* args_out = nullptr, u32 small_limit = 0, u64 target_small = 0, bool host_maps = true,
                          CompStats* comp_out = nullptr) {
  namespace L = kamd_em_local;
  if (nnz >= (1ULL << 32) || T >= 0xFFFFFFF0ULL || n_ecs >= 0xFFFFFFF0ULL) return 1;
  // component labels (smallest transcript id of the component): one lock-free union-find pass, cc_labels
  if (int rc = c->pt_label.ensure((T + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->pt_hist.ensure(64, 0, c->stream)) return rc;
  if (!c->labels_override) if (int rc = cc_labels(c, d_ec_off, d_ec_ids, n_ecs, T)) return rc;
  // scratch: per transcript / per root ...
  Carver t1;
  const size_t o_inm = t1.take(T + 8), o_sall = t1.take(T * 8 + 8), o_cn = t1.take(T * 4 + 8), o_cr = t1.take(T * 4 + 8), o_ct = t1.take(T * 4 + 8);
  const size_t o_cum = t1.take((T + 2) * 8), o_loc = t1.take(T * 4 + 8), o_eff = t1.take(T * 8 + 8), o_rnew = t1.take(n_ecs * 4 + 8);
  const size_t o_cs = t1.take(small_limit ? T * 4 + 8 : 8), o_cb = t1.take(small_limit ? T * 4 + 8 : 8), o_cumb = t1.take(small_limit ? (T + 2) * 8 : 8);
  // ... and per group: at most nnz / target + 2 groups (the real number is known after the scan below)
  const u64 ng_max = nnz / std::max<u64>(1, target) + 2 + (small_limit ? nnz / std::max<u64>(1, target_small) + 2 : 0);
  const size_t o_gr = t1.take(ng_max * 4 + 8), o_gt = t1.take(ng_max * 4 + 8), o_gn = t1.take(ng_max * 4 + 8), o_rf = t1.take(ng_max * 4 + 8),
               o_tf = t1.take(ng_max * 4 + 8);
  if (int rc = c->eml_tmp.ensure(t1.off, 0, c->stream)) return rc;
  char* tb = (char*)c->eml_tmp.p;
  HIPC(hipMemsetAsync(tb, 0, o_cum, c->stream));   // in_multi, single_all, c_nnz, c_rows, c_tr
  HIPC(hipMemcpyAsync(tb + o_eff, eff_lens, T * 8, hipMemcpyHostToDevice, c->stream));
  L::BuildArgs A{};
  A.ec_off = (const uint64_t*)d_ec_off; A.ec_ids = d_ec_ids; A.counts = d_counts; A.wcounts = d_wcounts; A.n_ecs = n_ecs;
  A.eff = (const double*)(tb + o_eff); A.T = T; A.label = c->labels_override ? c->labels_override : c->pt_label.as<u32>(); A.target_nnz = std::max<u64>(1, target);
  A.in_multi = (uint8_t*)(tb + o_inm); A.single_all = (double*)(tb + o_sall); A.c_nnz = (u32*)(tb + o_cn); A.c_rows = (u32*)(tb + o_cr);
  A.c_tr = (u32*)(tb + o_ct); A.cum_nnz = (const uint64_t*)(tb + o_cum); A.local_of = (u32*)(tb + o_loc); A.row_new = (u32*)(tb + o_rnew);
  hipLaunchKernelGGL(k_eml_step<0>, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, A, n_ecs);
  hipLaunchKernelGGL(k_eml_step<1>, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, A, T);
  // the largest component, read back with the entry count below (same synchronisation)
  CompStats cst{};
  HIPC(hipMemsetAsync(c->pt_hist.p, 0, sizeof(CompStats), c->stream));
  hipLaunchKernelGGL(k_comp_stats, dim3(std::min<unsigned>(grid_for(T, BLOCK), COMP_STATS_BLOCKS)), dim3(BLOCK), 0, c->stream, A.c_nnz, A.c_rows, A.c_tr, T, (u32*)c->pt_hist.p);
  HIPC(hipMemcpyAsync(&cst, c->pt_hist.p, sizeof(CompStats), hipMemcpyDeviceToHost, c->stream));
  u64 NZ = 0;
  u32 ng = 0;
  P->n_small = 0;
  if (small_limit) {
    A.small_limit = small_limit; A.c_small = (u32*)(tb + o_cs); A.c_big = (u32*)(tb + o_cb);
    hipLaunchKernelGGL(k_eml_step<9>, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, A, T);
    if (int rc = exclusive_scan(c, A.c_small, T, (u64*)(tb + o_cum), (u64*)(tb + o_cum) + T)) return rc;
    if (int rc = exclusive_scan(c, A.c_big, T, (u64*)(tb + o_cumb), (u64*)(tb + o_cumb) + T)) return rc;
    u64 nz2[2] = {0, 0};
    HIPC(hipMemcpyAsync(&nz2[0], (u64*)(tb + o_cum) + T, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(&nz2[1], (u64*)(tb + o_cumb) + T, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    NZ = nz2[0] + nz2[1];
    if (comp_out) *comp_out = cst;
    if (NZ == 0) return 1;
    A.target_big = A.target_nnz; A.target_nnz = std::max<u64>(1, target_small); A.cum_big = (const uint64_t*)(tb + o_cumb);
    A.ng_small = nz2[0] ? (u32)((nz2[0] - 1) / A.target_nnz + 1) : 0u;
    ng = A.ng_small + (nz2[1] ? (u32)((nz2[1] - 1) / A.target_big + 1) : 0u);
    P->n_small = A.ng_small;
  } else {
    if (int rc = exclusive_scan(c, A.c_nnz, T, (u64*)(tb + o_cum), (u64*)(tb + o_cum) + T)) return rc;
    HIPC(hipMemcpyAsync(&NZ, (u64*)(tb + o_cum) + T, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (comp_out) *comp_out = cst;
    if (NZ == 0) return 1;
    ng = (u32)((NZ - 1) / A.target_nnz + 1);
  }
  if (comp_out) *comp_out = cst;
  if ((u64)ng > ng_max) return kamd::fail(-105, "kamd_em_run: component-local EM: more groups than entries allow");
  A.n_groups = ng;
  HIPC(hipMemsetAsync(tb + o_gr, 0, t1.off - o_gr, c->stream));
  A.g_rows = (u32*)(tb + o_gr); A.g_tr = (u32*)(tb + o_gt); A.g_nnz = (u32*)(tb + o_gn); A.row_fill = (u32*)(tb + o_rf); A.tr_fill = (u32*)(tb + o_tf);
  hipLaunchKernelGGL(k_eml_step<2>, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, A, T);
  std::vector<u32> g_rows(ng), g_tr(ng), g_nnz(ng);
  HIPC(hipMemcpyAsync(g_rows.data(), A.g_rows, ng * 4, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipMemcpyAsync(g_tr.data(), A.g_tr, ng * 4, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipMemcpyAsync(g_nnz.data(), A.g_nnz, ng * 4, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  P->n_groups = ng; P->T = T; P->max_group_bytes = 0;
  P->row_base.assign(ng + 1, 0); P->tr_base.assign(ng + 1, 0); P->nz_base.assign(ng + 1, 0);
  for (u32 g = 0; g < ng; g++) {
    const u64 gb = L::group_bytes(g_nnz[g], g_rows[g], g_tr[g]);
    if (g_rows[g] > 65535 || g_tr[g] > 65535 || gb > budget) return 1;
    P->max_group_bytes = std::max<uint64_t>(P->max_group_bytes, gb);
    P->row_base[g + 1] = P->row_base[g] + g_rows[g]; P->tr_base[g + 1] = P->tr_base[g] + g_tr[g]; P->nz_base[g + 1] = P->nz_base[g] + g_nnz[g];
  }
  const u64 R = P->row_base[ng], M = P->tr_base[ng];
  if (P->nz_base[ng] != NZ) return kamd::fail(-105, "kamd_em_run: component-local EM: group sizes do not add up");
  // the plan's arrays + the scratch that depends on R and M
  Carver pv;
  const size_t p_rb = pv.take((ng + 1) * 4), p_tb = pv.take((ng + 1) * 4), p_zb = pv.take((ng + 1) * 8);
  const size_t p_rp = pv.take((R + ng) * 4 + 8), p_cp = pv.take((M + ng) * 4 + 8), p_rt = pv.take(NZ * 2 + 8), p_cr = pv.take(NZ * 2 + 8);
  const size_t p_cw = pv.take(R * 8 + 8), p_sg = pv.take(M * 8 + 8), p_ef = pv.take(M * 8 + 8), p_id = pv.take(M * 4 + 8);
  const size_t p_len = pv.take(R * 4 + 8), p_rabs = pv.take((R + 2) * 8), p_cc = pv.take(M * 4 + 8), p_cf = pv.take(M * 4 + 8), p_cabs = pv.take((M + 2) * 8);
  const size_t p_tt = pv.take(M * 4 + 8), p_rk = pv.take(R * 8 + 8), p_crt = pv.take(NZ * 2 + 8), p_ec = pv.take(NZ * 4 + 8), p_re = pv.take(R * 4 + 8), p_ref = pv.take(R * 4 + 8);   // scratch of the canonical numbering
  if (int rc = c->pm_a.ensure(pv.off, 0, c->stream)) return rc;
  char* pb = (char*)c->pm_a.p;
  HIPC(hipMemcpyAsync(pb + p_rb, P->row_base.data(), (ng + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemcpyAsync(pb + p_tb, P->tr_base.data(), (ng + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemcpyAsync(pb + p_zb, P->nz_base.data(), (ng + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemsetAsync(pb + p_cc, 0, p_cabs - p_cc, c->stream));   // col_cnt, col_fill
  A.row_base = (const u32*)(pb + p_rb); A.tr_base = (const u32*)(pb + p_tb); A.nz_base = (const uint64_t*)(pb + p_zb);
  A.row_ptr = (u32*)(pb + p_rp); A.col_ptr = (u32*)(pb + p_cp); A.row_tr = (uint16_t*)(pb + p_rt); A.col_row = (uint16_t*)(pb + p_cr);
  A.cw = (uint64_t*)(pb + p_cw); A.single = (double*)(pb + p_sg); A.eff_m = (double*)(pb + p_ef); A.tr_id = (u32*)(pb + p_id);
  A.len_new = (u32*)(pb + p_len); A.row_abs = (const uint64_t*)(pb + p_rabs); A.col_cnt = (u32*)(pb + p_cc); A.col_fill = (u32*)(pb + p_cf);
  A.col_abs = (const uint64_t*)(pb + p_cabs);
  A.tmp_tr_id = (u32*)(pb + p_tt); A.row_key = (uint64_t*)(pb + p_rk); A.col_row_tmp = (uint16_t*)(pb + p_crt); A.ent_col = (u32*)(pb + p_ec); A.row_e = (u32*)(pb + p_re); A.row_e_final = (u32*)(pb + p_ref);
  // the canonical numbering: members are handed slots by atomic cursors (steps F, G), then ranked inside their group -- out of LDS
  // when the group's keys fit (always, for groups the EM kernel can hold), by the per-member step kernels otherwise
  size_t rk_lds = 0;
  for (u32 g = 0; g < ng; g++) rk_lds = std::max(rk_lds, std::max((size_t)g_tr[g] * 4, (size_t)g_rows[g] * 12) + 16);
  const bool lds_rank = rk_lds <= 150 * 1024;
  hipLaunchKernelGGL(k_eml_step<3>, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, A, T);
  if (lds_rank) {
    HIPC(hipFuncSetAttribute((const void*)k_eml_rank_tr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rk_lds));
    HIPC(hipFuncSetAttribute((const void*)k_eml_rank_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rk_lds));
    hipLaunchKernelGGL(k_eml_rank_tr, dim3(ng), dim3(BLOCK), rk_lds, c->stream, A);
  } else hipLaunchKernelGGL(k_eml_step<10>, dim3(grid_for(M, BLOCK)), dim3(BLOCK), 0, c->stream, A, M);
  if (A.n_groups <= (u32)RG_BINS) hipLaunchKernelGGL(k_eml_rows_g, dim3(grid_for(n_ecs, RG_BLOCK * RG_PER)), dim3(RG_BLOCK), 0, c->stream, A);
  else hipLaunchKernelGGL(k_eml_step<4>, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, A, n_ecs);
  if (lds_rank) hipLaunchKernelGGL(k_eml_rank_rows, dim3(ng), dim3(EML_RANK_BLOCK), rk_lds, c->stream, A);
  else hipLaunchKernelGGL(k_eml_step<11>, dim3(grid_for(R, BLOCK)), dim3(BLOCK), 0, c->stream, A, R);
  // steps I, J, K, K2 and their scans: one workgroup per group out of LDS (k_eml_group_build); a group too large for that -- none
  // that the EM kernel could hold -- takes the steps one by one
  size_t gb_lds = 0;
  for (u32 g = 0; g < ng; g++)
    gb_lds = std::max(gb_lds, ((size_t)g_rows[g] + 1 + 2 * (size_t)g_tr[g] + 1) * 4 + (((size_t)g_nnz[g] + 1) & ~(size_t)1) * 4 + 16);
  if (gb_lds <= 150 * 1024) {
    HIPC(hipFuncSetAttribute((const void*)k_eml_group_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gb_lds));
    hipLaunchKernelGGL(k_eml_group_build, dim3(ng), dim3(BLOCK), gb_lds, c->stream, A);
  } else {
    if (int rc = exclusive_scan(c, A.len_new, R, (u64*)(pb + p_rabs), (u64*)(pb + p_rabs) + R)) return rc;
    hipLaunchKernelGGL(k_eml_step<5>, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, A, n_ecs);
    if (int rc = exclusive_scan(c, A.col_cnt, M, (u64*)(pb + p_cabs), (u64*)(pb + p_cabs) + M)) return rc;
    hipLaunchKernelGGL(k_eml_step<6>, dim3(grid_for(M, BLOCK)), dim3(BLOCK), 0, c->stream, A, M);
    hipLaunchKernelGGL(k_eml_step<7>, dim3(grid_for(ng, BLOCK)), dim3(BLOCK), 0, c->stream, A, (u64)ng);
    hipLaunchKernelGGL(k_eml_step<8>, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, A, n_ecs);
    hipLaunchKernelGGL(k_eml_step<12>, dim3(grid_for(NZ, BLOCK)), dim3(BLOCK), 0, c->stream, A, NZ);
  }
  HIPC(hipGetLastError());
  // what the host needs for the final scatter (a single rank scatters on the device: em_sell_drive_async)
  if (host_maps) {
    P->tr_id.resize(M); P->single_all.resize(T);
    if (M) HIPC(hipMemcpyAsync(P->tr_id.data(), A.tr_id, M * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(P->single_all.data(), A.single_all, T * 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  *dev = EmLocalDev{(const u32*)(pb + p_rb), (const u32*)(pb + p_tb), (const u64*)(pb + p_zb), (const u32*)(pb + p_rp), (const u32*)(pb + p_cp),
                    (const uint16_t*)(pb + p_rt), (const uint16_t*)(pb + p_cr), (const u64*)(pb + p_cw), (const double*)(pb + p_sg),
                    (const double*)(pb + p_ef), (const u32*)(pb + p_id)};
  if (args_out) *args_out = A;
  return 0;
}
// ---- what a second EM on the same matrix re-uses (bootstrap replicates: only the counts change, EMAlgorithm.h:46) ---------------
// row_final[e]: position of EC e's count word in the plan (multi-transcript rows), SELL_NONE otherwise; mslot[t]: slot of
// transcript t in the plan's transcript vectors, SELL_NONE if it is in no multi-transcript row
constexpr u32 SELL_NONE = 0xFFFFFFFFu;
__global__ void k_sell_maps(kamd_em_local::BuildArgs A, SellBuild B, u32* row_final, u32* mslot) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < A.n_ecs) {
    const u64 a = A.ec_off[i], b = A.ec_off[i + 1];
    u32 rf = SELL_NONE;
    if (b - a >= 2) {
      const u32 rn = A.row_new[i];
      const u32 g = sell_group_of(B.row_base, B.n_groups, rn);
      rf = B.row_base[g] + B.rnew[rn];
    }
    row_final[i] = rf;
  }
  if (i < A.T) {
    u32 ms = SELL_NONE;
    if (A.in_multi[i]) {
      const u32 g = kamd_em_local::eml_group_of(A, A.label[i]);
      const u32 old_m = B.tr_base[g] + A.local_of[i];
      ms = B.tr_base[g] + B.cnew[old_m];
    }
    mslot[i] = ms;
  }
}
// new counts / effective lengths into a cached plan
__global__ void k_sell_refresh(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, const u32* __restrict__ counts,
                               const u32* __restrict__ wcounts, u64 n_ecs, const double* __restrict__ eff, u64 T,
                               const u32* __restrict__ row_final, const u32* __restrict__ mslot, u64* cw, double* single_m,
                               double* eff_m, double* single_all) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_ecs) {
    const u64 a = ec_off[i], b = ec_off[i + 1];
    if (b - a == 1) {   // a transcript has at most one singleton set (:119-123)
      const u32 t = ec_ids[a];
      single_all[t] = (double)counts[i];
      if (mslot[t] != SELL_NONE) single_m[mslot[t]] = (double)counts[i];
    } else if (b - a >= 2) cw[row_final[i]] = (u64)counts[i] | ((u64)wcounts[i] << 32);
  }
  if (i < T && mslot[i] != SELL_NONE) eff_m[mslot[i]] = eff[i];
}

// ---- hybrid EM: the connected components that do not fit a workgroup's LDS, iterated BESIDE the component-local form ----------------
// A real transcriptome has repeat families and poly-A classes: equivalence classes of hundreds to thousands of transcripts that chain
// unrelated genes into ONE connected component with half of the matrix's entries (the reference ships --ec-max-size for them,
// src/main.cpp:2151).  EMAlgorithm::run (src/EMAlgorithm.h:112-223) costs the same per entry whatever the graph looks like; the LDS form
// cannot hold such a component.  So the matrix is split by component: what fits keeps k_em_sell, the oversized components become a
// sub-matrix in the flagged-stream layout of the streamed form (em_streamed_setup) and are iterated by two launches per round on a
// stream of their own, on compute units the LDS kernel's stream is masked away from.  Both sides speak the protocol of EmSellGpu::launch:
// n rounds from an input state that stays intact (the checkpoint) to an output state, per-round change counts added to ONE history, the
// stop rule of the previous chunk applied by every kernel (a chunk queued behind the one the run stops in does nothing).  The kernel
// bodies are the streamed form's (pm_wave_load / pm_wave_pass / pm_fix); only the loop control differs: no EmState, the round's
// vectors are kernel arguments, and what changes from chunk to chunk sits in a descriptor in device memory so that the 64 rounds of
// a chunk are ONE hipGraph replayed for every chunk (4 launches per round at 3.5 us of host time each would otherwise bind the host).
struct GiDesc { int* hist; int stopped; int pad; };   // this chunk's per-round change counts (null: not wanted); the run stopped in the previous chunk
// one launch in front of a chunk: where its change counts go, and the stop rule of EMAlgorithm::run (:202-205) on the counts of the PREVIOUS
// chunk, evaluated ONCE here -- every kernel of the chunk then reads one word (a chunk queued behind the one the run stops in does nothing)
__global__ void k_gi_set_desc(GiDesc* d, EmsPrev prev, int* hist) {
  const int i = (int)threadIdx.x;
  bool stop = false;
  if (prev.hist && i < 64) { const int h = i < prev.n ? prev.hist[i] : 1; stop = i < prev.n && h == 0 && prev.base + i > prev.min_rounds; }
  const u64 m = __ballot(stop);
  if (i == 0) { d->hist = hist; d->stopped = m != 0 ? 1 : 0; d->pad = 0; }
}
__device__ __forceinline__ bool gi_stopped(const GiDesc* d, int*) { return d->stopped != 0; }
struct GiColEmit {   // PmColEmit without the clamped copy (the final round's clamp is a pass of its own, k_gi_clamp)
  const double* alpha_cur; const double* a_cur; const double* single; const double* eff; double* alpha_nx; double* a_nx; int* ch; int clamp;
  struct Ctx { double al, at, sg, ef; };
  __device__ __forceinline__ Ctx load(u32 m) const { return Ctx{alpha_cur[m], a_cur[m], single[m], eff[m]}; }
  __device__ __forceinline__ void finish(u32 m, const Ctx& x, double acc) const {
    const double al = (clamp && x.al < 1e-7 / 10.0) ? 0.0 : x.al;
    const double nx = x.sg + x.at * acc;
    if (nx > 1e-2 && (fabs(nx - al) / nx) > 1e-2) ++*ch;
    alpha_nx[m] = nx;
    a_nx[m] = nx / x.ef;
  }
};
__device__ __forceinline__ void gi_count_changes(int ch, int* lds_ch, int* slot) {
  if (threadIdx.x == 0) *lds_ch = 0;
  __syncthreads();
  if (__ballot(ch != 0)) {
    int wsum = ch;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wsum += __shfl_down(wsum, d, 64);
    if (lane_id() == 0) atomicAdd(lds_ch, wsum);
  }
  __syncthreads();
  if (threadIdx.x == 0 && *lds_ch && slot) atomicAdd(slot, *lds_ch);
}
template <int K, int PRE, bool WIN>
__global__ __launch_bounds__(PM_BLOCK) void k_gi_rows(PmArgs A, const double* a_src, const GiDesc* desc) {
  __shared__ double lds_sums[(PM_BLOCK / 64) * PM_LDS_SLOTS];
  __shared__ int s_stop;
  const u32 c = __builtin_amdgcn_readfirstlane(blockIdx.x * (PM_BLOCK / 64) + (threadIdx.x >> 6));
  PmWave<K> w;
  if (c < A.rows.n_chunks) pm_wave_load<K>(A.rows, c, w);
  if (gi_stopped(desc, &s_stop)) return;
  if (c >= A.rows.n_chunks) return;
  const PmRowEmit em{A.cw, A.g};
  pm_wave_pass<K, PRE, WIN>(A.rows, c, w, PmSrcGlobal{a_src}, lds_sums + (threadIdx.x >> 6) * PM_LDS_SLOTS, em);
}
template <int K, int PRE, bool WIN>
__global__ __launch_bounds__(PM_BLOCK) void k_gi_cols(PmArgs A, const double* al_src, const double* a_src, double* al_dst, double* a_dst, int round, int clamp,
                                                      const GiDesc* desc) {
  __shared__ double lds_sums[(PM_BLOCK / 64) * PM_LDS_SLOTS];
  __shared__ int lds_ch;
  __shared__ int s_stop;
  const u32 c = __builtin_amdgcn_readfirstlane(blockIdx.x * (PM_BLOCK / 64) + (threadIdx.x >> 6));
  PmWave<K> w;
  if (c < A.cols.n_chunks) pm_wave_load<K>(A.cols, c, w);
  if (gi_stopped(desc, &s_stop)) return;
  int ch = 0;
  const GiColEmit em{al_src, a_src, A.single, A.eff, al_dst, a_dst, &ch, clamp};
  if (c < A.cols.n_chunks) pm_wave_pass<K, PRE, WIN>(A.cols, c, w, PmSrcGlobal{A.g}, lds_sums + (threadIdx.x >> 6) * PM_LDS_SLOTS, em);
  int* hist = desc->hist;
  gi_count_changes(ch, &lds_ch, hist ? hist + round : nullptr);
}
__global__ __launch_bounds__(PM_BLOCK) void k_gi_rows_fix(PmArgs A, const GiDesc* desc) {
  __shared__ int s_stop;
  if (gi_stopped(desc, &s_stop)) return;
  const PmRowEmit em{A.cw, A.g};
  pm_fix(A.rows, blockIdx.x * PM_BLOCK + threadIdx.x, em);
}
__global__ __launch_bounds__(PM_BLOCK) void k_gi_cols_fix(PmArgs A, const double* al_src, const double* a_src, double* al_dst, double* a_dst, int round, int clamp,
                                                          const GiDesc* desc) {
  __shared__ int lds_ch;
  __shared__ int s_stop;
  if (gi_stopped(desc, &s_stop)) return;
  int ch = 0;
  const GiColEmit em{al_src, a_src, A.single, A.eff, al_dst, a_dst, &ch, clamp};
  pm_fix(A.cols, blockIdx.x * PM_BLOCK + threadIdx.x, em);
  int* hist = desc->hist;
  gi_count_changes(ch, &lds_ch, hist ? hist + round : nullptr);
}
// the final round reads a with the clamp applied (alpha < alpha_limit / 10 -> 0, :212-221); ac[M] = 0 stays the row stream's sentinel
__global__ void k_gi_clamp(const double* __restrict__ al, const double* __restrict__ a, double* ac, u32 M, const GiDesc* desc) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) ac[i] = al[i] < 1e-7 / 10.0 ? 0.0 : a[i];
  else if (i == M) ac[i] = 0.0;
}
__global__ void k_gi_zero_tail(double* p0, double* p1, double* p2, double* p3, double* p4, u32 M) {   // sentinels of the scratch vectors
  if (threadIdx.x == 0 && blockIdx.x == 0) { p0[M] = 0.0; p1[M] = 0.0; p2[M] = 0.0; p3[M] = 0.0; p4[M] = 0.0; }
}
// the oversized components' transcripts back in transcript space (behind k_em_scatter, which left them at 0)
__global__ void k_gi_scatter(u64 T, const u32* __restrict__ mflag, const u64* __restrict__ mpos, const double* __restrict__ fin, const double* __restrict__ before,
                             int have_final, double* out_alpha, double* out_abz) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T || !mflag[t]) return;
  const u64 m = mpos[t];
  out_alpha[t] = fin[m];
  out_abz[t] = have_final ? before[m] : 0.0;
}
// ---- the oversized component in 2-D blocks: gathers out of LDS -------------------------------------------------------------------------
// k_gi_rows / k_gi_cols gather one FP64 value per entry through the vector memory pipeline: a 64-byte line travels from L2 for 8 useful
// bytes, and a round of a 14 M-entry component runs at L2's line rate (39 + 46 us, round 5).  Here the gathered vector is cut into blocks of
// GB_B values that a workgroup holds in LDS: the entries of a direction are re-sorted by (block of the gathered index, segment), so that a
// workgroup loads ONE block, then streams the 16-bit in-block indices of its share of that block's entries and gathers out of LDS.  A
// segment (row / column) restricted to a block -- cut again where it crosses a chunk of 64 x GB_K entries -- is a PIECE; a piece's sum goes
// to its slot of a partial-sum array in which the pieces of a segment are neighbours, in a fixed order.  Nothing is combined by atomics:
//   k_gb_pass<0>  rows: block of a (copied into LDS)                              -> partial sums of the rows
//   k_gb_pass<1>  columns: block of g, COMPUTED while it is loaded from the rows' partial sums (g_e = count_e / S_e), -> partial sums of the columns
//   k_gb_finish   per transcript: next alpha from its partial sums, the convergence test, a = alpha / eff
// three launches per round, every sum in a fixed order (bit-reproducible given the numbering of the rows).
// Values per block, per direction.  Every workgroup fetches its whole block through a cold L2, and what the columns pass fetches per row of its
// block are the row's partial sums: a long row (a poly-A class of 3 500 transcripts) has one per block of a it crosses, and the row blocks that
// hold the long rows are the ones most workgroups work on -- so the blocks of a (rows pass) are LARGE, few pieces per long row, and the
// blocks of g (columns pass) small (measured at 8 M stress pairs, columns pass: 32 us with 2 048-value blocks of a, 22 with 8 192).
constexpr int GB_SHIFT_A = 13, GB_SHIFT_G = 11;             // blocks of a: 8 192 values (64 KB of LDS); blocks of g: 2 048 rows
constexpr int GB_B = 1 << GB_SHIFT_A;                       // (the larger of the two: LDS is sized for it, and index GB_B is the zero slot the padding reads)
constexpr int GB_PF = 4;                                   // chunks of a wavefront requested together (8 was measured no faster: 15.9 against 14.7 us for the rows pass at 30 M stress pairs)
constexpr int GB_K = 8, GB_CH = 64 * GB_K;                  // entries per lane and chunk (one 16-byte load), entries per chunk
constexpr int GB_WAVES = 16, GB_THREADS = 64 * GB_WAVES;   // one workgroup per compute unit: 64 KB of block + 16 x 4 KB of piece sums
constexpr uint16_t GB_END = 0x8000u, GB_PAD = (uint16_t)GB_B;   // entry = in-block index | GB_END on the last entry of a piece; padding reads the zero slot
struct GbSide {
  const uint16_t* stream;   // [n_chunks * GB_CH]; the chunks of a block are consecutive, a block starts on a chunk boundary
  const u32* seg_base;      // [n_chunks] pieces that end in earlier chunks (= id of the chunk's first piece)
  const u32* lane_word;     // [n_chunks * 64] as PmSide::lane_word
  const u32* piece_slot;    // [n_pieces] where a piece's sum goes in `part`
  double* part;             // [n_pieces]
  const u32* slot_base;     // [n_seg + 1] the slots of a segment's pieces: slot_base[s] .. slot_base[s + 1]
  const u32* wg_desc;       // [2 * n_wg] per workgroup: its work items wg_items[first .. end)
  const u32* wg_items;      // [4 * n_items] block, first chunk, end chunk, rotation of the block's load: a workgroup loads the block, then its wavefronts take the chunks
  u32 n_wg, n_chunks, n_pieces, n_seg, n_tgt;
};
struct GbArgs { GbSide rows, cols; const u64* cw; const double* single; const double* eff; u32 R, M; };
template <int DIR>
__global__ __launch_bounds__(GB_THREADS) void k_gb_pass(GbArgs A, const double* a_src, const GiDesc* desc) {
  extern __shared__ __attribute__((aligned(16))) double gb_lds[];
  double* s_val = gb_lds;                     // [GB_B + 1] the block of the gathered vector; [GB_B] = 0 is what the padding reads
  double* s_sum = gb_lds + (GB_B + 2);        // [GB_WAVES][GB_CH] the sums of the pieces that end in a wavefront's chunk
  if (desc->stopped) return;
  const GbSide& S = DIR == 0 ? A.rows : A.cols;
  constexpr int SH = DIR == 0 ? GB_SHIFT_A : GB_SHIFT_G, BV = 1 << SH;   // values of a block in this direction
  const int lane = lane_id();
  const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 it0 = S.wg_desc[2 * blockIdx.x], it1 = S.wg_desc[2 * blockIdx.x + 1];
  for (u32 it = it0; it < it1; it++) {
  const u32 blk = S.wg_items[4 * it], c0 = S.wg_items[4 * it + 1], c1 = S.wg_items[4 * it + 2], rot = S.wg_items[4 * it + 3];
  const u32 base = blk << SH;
  if (it != it0) __syncthreads();   // (the block and the piece sums of the item before are done with)
  // the wavefront's first GB_PF chunks (all of them, at the sizes of a transcriptome) are on their way while the block is loaded: loading the
  // block is up to three dependent memory latencies from a cold L2, and what follows it is then LDS work only
  u32 c = c0 + wv;
  uint4 raw[GB_PF]; u32 sb[GB_PF], ps[GB_PF];
  auto request = [&](u32 cf) {
#pragma unroll
    for (int i = 0; i < GB_PF; i++) {
      const u32 cc = cf + (u32)i * GB_WAVES;
      raw[i] = make_uint4(0, 0, 0, 0); sb[i] = 0;
      if (cc < c1) { raw[i] = reinterpret_cast<const uint4*>(S.stream + (u64)cc * GB_CH)[lane]; sb[i] = S.seg_base[cc]; }
    }
#pragma unroll
    for (int i = 0; i < GB_PF; i++) ps[i] = cf + (u32)i * GB_WAVES < c1 ? S.piece_slot[sb[i] + lane] : 0u;   // (the slots of a chunk's first 64 pieces; the array is padded by 64 entries)
  };
  request(c);
  constexpr int PER = BV / GB_THREADS;   // values of the block per thread
  if (DIR == 0) {
    double v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) { const u32 i = (threadIdx.x + (u32)k * GB_THREADS + rot) & (BV - 1); v[k] = base + i < A.M ? a_src[base + i] : 0.0; }
#pragma unroll
    for (int k = 0; k < PER; k++) s_val[(threadIdx.x + (u32)k * GB_THREADS + rot) & (BV - 1)] = v[k];
  } else {
    // g of the block's rows from the rows' partial sums (in slot order); rows the reference skips get 0: count 0 (:133-135), denom below
    // denorm_min (:156-158).  THREE dependent loads for the whole block (every one of them reads what the launch before wrote on other XCDs:
    // ~2 us each): all offsets and count words; the first four partial sums of every row; then the rows of more than four pieces -- long
    // rows cut at block and chunk boundaries, a few hundred per block -- one per thread out of a list in LDS, all of a row's slots at once.
    const GbSide& Rw = A.rows;
    auto g_of = [](u64 w, double S) -> double {
      const u32 cnt = (u32)w, wc = (u32)(w >> 32);
      return (cnt == 0 || (double)wc * S < 4.9406564584124654e-324) ? 0.0 : (double)cnt / S;
    };
    u32* s_ln = reinterpret_cast<u32*>(s_sum);          // [0] = rows listed; then per listed row {index in the block, first slot, slots} ...
    u64* s_lw = reinterpret_cast<u64*>(s_sum) + 4096;   // ... and its count word (the piece sums' region, 64 KB, is idle while the block is loaded)
    constexpr u32 LIST_CAP = 2048;                       // (rows beyond that are summed by their own thread, slot after slot)
    u32 s0[PER], s1[PER]; u64 w[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const u32 r = base + ((threadIdx.x + (u32)k * GB_THREADS + rot) & (BV - 1));
      const bool in = r < A.R;
      s0[k] = in ? Rw.slot_base[r] : 0u; s1[k] = in ? Rw.slot_base[r + 1] : 0u; w[k] = in ? A.cw[r] : 0ULL;
    }
    // (ONE load per row here: the first slot.  Rows of several pieces -- one in six -- go to the list: a predicated load per further slot and row
    // made every wavefront request the same lines four times over; the columns pass issued four times the L2 requests of the rows pass,
    // profiles/r06_gb_counters.txt)
    double p0[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) p0[k] = s1[k] > s0[k] ? Rw.part[s0[k]] : 0.0;
    // the rows of several pieces are listed IN ROW ORDER (ballot + per-wavefront counts, no atomic cursor): neighbours in the list are neighbours
    // in the partial-sum array, so the list's loads fall into few lines
    u32* s_wc = s_ln + 1 + 3 * LIST_CAP;   // [PER * GB_WAVES] listed rows per (k, wavefront)
    u64 mm[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      mm[k] = __ballot(s1[k] - s0[k] > 1u);
      if (lane == 0) s_wc[k * GB_WAVES + wv] = (u32)__popcll(mm[k]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const u32 np = s1[k] - s0[k], i = (threadIdx.x + (u32)k * GB_THREADS + rot) & (BV - 1);
      if (np > 1u) {
        u32 at = (u32)__popcll(mm[k] & ((1ULL << lane) - 1ULL));
        for (u32 q = 0; q < (u32)k * GB_WAVES + wv; q++) at += s_wc[q];
        if (at < LIST_CAP) { s_ln[1 + 3 * at] = i; s_ln[2 + 3 * at] = s0[k]; s_ln[3 + 3 * at] = np; s_lw[at] = w[k]; }
        else {
          double S = p0[k];
          for (u32 q = s0[k] + 1; q < s1[k]; q++) S += Rw.part[q];
          s_val[i] = g_of(w[k], S);
        }
      } else s_val[i] = np ? g_of(w[k], p0[k]) : 0.0;
    }
    if (threadIdx.x == 0) { u32 tot = 0; for (u32 q = 0; q < (u32)PER * GB_WAVES; q++) tot += s_wc[q]; s_ln[0] = tot; }
    __syncthreads();
    const u32 n_long = min(s_ln[0], LIST_CAP);
    for (u32 q = threadIdx.x; q < n_long; q += GB_THREADS) {
      const u32 i = s_ln[1 + 3 * q], f = s_ln[2 + 3 * q], np = s_ln[3 + 3 * q];
      double S = 0.0;
      for (u32 j0 = 0; j0 < np; j0 += 16) {
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = j0 + j < np ? Rw.part[f + j0 + j] : 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) if (j0 + j < np) S += x[j];
      }
      s_val[i] = g_of(s_lw[q], S);
    }
    __syncthreads();   // (the list lies where the wavefronts' piece sums go)
  }
  if (threadIdx.x == 0) s_val[GB_B] = 0.0;
  __syncthreads();
  double* sums = s_sum + (size_t)wv * GB_CH;
  for (; c < c1; c += GB_PF * GB_WAVES) {
#pragma unroll
    for (int i = 0; i < GB_PF; i++) {
      const u32 cc = c + (u32)i * GB_WAVES;
      if (cc >= c1) break;
      const uint4 rw = raw[i];
      const u32 e[GB_K] = {rw.x & 0xFFFFu, rw.x >> 16, rw.y & 0xFFFFu, rw.y >> 16, rw.z & 0xFFFFu, rw.z >> 16, rw.w & 0xFFFFu, rw.w >> 16};
      double v[GB_K];
#pragma unroll
      for (int k = 0; k < GB_K; k++) v[k] = s_val[e[k] & 0x3FFFu];
      // pieces that end in the lanes below / in this lane, and the distance to the nearest lane at or below that holds an end: from the END flags (a
      // stored word per lane and chunk -- PmSide::lane_word -- was a quarter of the bytes a launch pulls through the cold L2s)
      const u32 ne = ((rw.x >> 15) & 1u) + (rw.x >> 31) + ((rw.y >> 15) & 1u) + (rw.y >> 31) + ((rw.z >> 15) & 1u) + (rw.z >> 31) + ((rw.w >> 15) & 1u) + (rw.w >> 31);
      const u32 incl = pm_scan_incl(ne);
      const u32 ebase = incl - ne;
      const u64 heads = __ballot(ne > 0);
      const u64 below = heads & ((2ULL << lane) - 1ULL);
      const int reach = below ? lane - (63 - __clzll((long long)below)) : lane + 1;
      const u32 n_ends = (u32)__builtin_amdgcn_readlane((int)incl, 63);
      double run = 0.0;
      double* slot = sums + ebase;
#pragma unroll
      for (int k = 0; k < GB_K; k++) {
        run += v[k];
        if (e[k] & GB_END) { *slot++ = run; run = 0.0; }
      }
      const double y = pm_scan_seg(run, reach, lane);
      const double carry = pm_dpp<0x138, 0xF>(y);   // wave_shr:1 (lane 0 gets 0): what the lanes below hold of the lane's first piece
      if (ne) sums[ebase] += carry;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if ((u32)lane < n_ends) S.part[ps[i]] = sums[lane];
      for (u32 t = lane + 64; t < n_ends; t += 64) S.part[S.piece_slot[sb[i] + t]] = sums[t];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (c + GB_PF * GB_WAVES < c1) request(c + GB_PF * GB_WAVES);
  }
  }
}
// next alpha of every transcript of the oversized components from the columns' partial sums (GiColEmit::finish)
__global__ __launch_bounds__(BLOCK) void k_gb_finish(GbArgs A, const double* al_src, const double* a_src, double* al_dst, double* a_dst, int round, int clamp,
                                                     const GiDesc* desc) {
  __shared__ int lds_ch;
  if (desc->stopped) return;
  const u32 m = blockIdx.x * BLOCK + threadIdx.x;
  int ch = 0;
  if (m < A.M) {
    const GbSide& C = A.cols;
    const u32 s0 = C.slot_base[m], s1 = C.slot_base[m + 1];
    const double xal = al_src[m], xat = a_src[m], sg = A.single[m], ef = A.eff[m];
    double acc = C.part[s0];
    for (u32 q = s0 + 1; q < s1; q += 4) {   // (four loads in flight, added in slot order)
      const double p0 = C.part[q], p1 = q + 1 < s1 ? C.part[q + 1] : 0.0, p2 = q + 2 < s1 ? C.part[q + 2] : 0.0, p3 = q + 3 < s1 ? C.part[q + 3] : 0.0;
      acc += p0; if (q + 1 < s1) acc += p1; if (q + 2 < s1) acc += p2; if (q + 3 < s1) acc += p3;
    }
    const double al = (clamp && xal < 1e-7 / 10.0) ? 0.0 : xal;
    const double nx = sg + xat * acc;
    if (nx > 1e-2 && (fabs(nx - al) / nx) > 1e-2) ++ch;
    al_dst[m] = nx;
    a_dst[m] = nx / ef;
  }
  int* hist = desc->hist;
  gi_count_changes(ch, &lds_ch, hist ? hist + round : nullptr);
}
// ---- plan of the blocked form (once per matrix) ----
// entries of segment s: ids[off[s] .. off[s + 1]) (index | PM_END); 8 lanes per segment
// (a lane takes a contiguous eighth of the segment and adds once per change of block: a long row, whose entries are sorted, is a handful of
// atomics instead of one per entry on the same few counters)
__global__ void k_gb_count(const u64* __restrict__ off, const u32* __restrict__ ids, u64 n_seg, int shift, u32* cnt) {
  const u64 s = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  const u64 sub = threadIdx.x & 7;
  if (s >= n_seg) return;
  const u64 a = off[s], len = off[s + 1] - a;
  const u64 j0 = a + len * sub / 8, j1 = a + len * (sub + 1) / 8;
  u32 run = 0, cur = 0xFFFFFFFFu;
  for (u64 j = j0; j < j1; j++) {
    const u32 b = (ids[j] & ~PM_END) >> shift;
    if (b != cur) { if (run) atomicAdd(&cnt[(u64)cur * n_seg + s], run); cur = b; run = 0; }
    ++run;
  }
  if (run) atomicAdd(&cnt[(u64)cur * n_seg + s], run);
}
// ORDERED: the entries of a segment come in increasing index order (rows: the transcripts of a class are sorted), so the place of an entry
// in its run follows from its place in the segment -- the run is born sorted; else the places are handed out by an atomic and the run is
// sorted afterwards (columns: the streamed plan's column entries lie in the order of its own atomics); info[pos] = the run of the entry there
__global__ void k_gb_block_starts(const u64* __restrict__ sub_off, u64 n_seg, u64 n_blk, u64* out) {   // where every block's runs begin (out[n_blk]: the total)
  const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= n_blk) out[b] = sub_off[b * n_seg];
}
template <bool ORDERED>
__global__ void k_gb_scatter(const u64* __restrict__ off, const u32* __restrict__ ids, u64 n_seg, const u32* __restrict__ cnt, const u64* __restrict__ sub_off,
                             const u64* __restrict__ blk_shift, int shift, u32* fill, uint16_t* stream, u32* info) {
  const u64 s = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  const int sub = threadIdx.x & 7;
  if (s >= n_seg) return;
  const u64 o0 = off[s];
  for (u64 j = o0 + sub; j < off[s + 1]; j += 8) {
    const u32 x = ids[j] & ~PM_END, b = x >> shift;
    const u64 key = (u64)b * n_seg + s;
    u64 pos;
    if (ORDERED) {
      u64 before = 0;
      for (u32 q = 0; q < b; q++) before += cnt[(u64)q * n_seg + s];
      pos = sub_off[key] + blk_shift[b] + (j - o0 - before);
    } else {
      pos = sub_off[key] + blk_shift[b] + atomicAdd(&fill[key], 1u);
      info[pos] = (u32)key;
    }
    stream[pos] = (uint16_t)(x & ((1u << shift) - 1u));
  }
}
// every (block, segment) run in increasing index order (the indices of a run are distinct): the place of an entry = the entries of its run
// that are smaller.  k_gb_sort_small: one thread per entry for runs of up to GB_SORT_SMALL entries; k_gb_sort_big: one wavefront per longer
// run, tiles of the run staged in LDS.  Both write the sorted copy `out` (every position of the stream exactly once between them).
constexpr int GB_SORT_SMALL = 64;
__global__ __launch_bounds__(BLOCK) void k_gb_sort_small(const uint16_t* __restrict__ stream, const u32* __restrict__ info, const u32* __restrict__ cnt, const u64* __restrict__ sub_off,
                                                         const u64* __restrict__ blk_shift, u64 n_seg, u64 n_pos, uint16_t* out) {
  const u64 p = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n_pos) return;
  const uint16_t x = stream[p];
  if (x == GB_PAD) { out[p] = x; return; }
  const u64 key = info[p];
  const u32 len = cnt[key];
  if (len > (u32)GB_SORT_SMALL) return;   // k_gb_sort_big's
  const u64 st = sub_off[key] + blk_shift[key / n_seg];
  u32 rank = 0;
  for (u32 k = 0; k < len; k++) rank += stream[st + k] < x ? 1u : 0u;
  out[st + rank] = x;
}
__global__ __launch_bounds__(BLOCK) void k_gb_sort_big(const uint16_t* __restrict__ stream, const u32* __restrict__ cnt, const u64* __restrict__ sub_off, const u64* __restrict__ blk_shift,
                                                       u64 n_seg, u64 n_keys, uint16_t* out) {
  // the indices of a run are distinct and below GB_B: the run as a bitmap in LDS (one bit per index), read back in increasing order -- a
  // constant few hundred cycles per run whatever its length (ranking by counting was n^2 / 64 compares per wavefront: 8 ms at 8 M stress pairs)
  constexpr int WORDS = GB_B / 32, PER_LANE = WORDS / 64;
  __shared__ u32 s_bm[BLOCK / 64][WORDS];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const u64 key = (u64)blockIdx.x * BLOCK + threadIdx.x;
  const u32 len = key < n_keys ? cnt[key] : 0u;
  const u64 start = len ? sub_off[key] + blk_shift[key / n_seg] : 0ULL;
  u64 big = __ballot(len > (u32)GB_SORT_SMALL);
  u32* bm = s_bm[wv];
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const u32 n = (u32)__shfl((int)len, src, 64);
    const u64 st = shfl_u64(start, src);
    const uint16_t* p = stream + st;
#pragma unroll
    for (int q = 0; q < PER_LANE; q++) bm[lane * PER_LANE + q] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (u32 i = lane; i < n; i += 64) { const u32 x = p[i]; atomicOr(&bm[x >> 5], 1u << (x & 31)); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u32 w[PER_LANE], pc = 0;
#pragma unroll
    for (int q = 0; q < PER_LANE; q++) { w[q] = bm[lane * PER_LANE + q]; pc += (u32)__popc(w[q]); }
    u32 o = pm_scan_incl(pc) - pc;
#pragma unroll
    for (int q = 0; q < PER_LANE; q++) {
      u32 word = w[q];
      while (word) { const int bit = __ffs((int)word) - 1; word &= word - 1; out[st + o++] = (uint16_t)((lane * PER_LANE + q) * 32 + bit); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
// per (block, segment) run: its pieces (one per chunk it touches): END flags, the pieces' slots (seg-major numbering) at the positions of their last entries
__global__ void k_gb_piece_count(const u32* __restrict__ cnt, const u64* __restrict__ sub_off, const u64* __restrict__ blk_shift, u64 n_seg, u32 n_blk, u32* pc) {
  const u64 key = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= n_seg * (u64)n_blk) return;
  const u32 len = cnt[key];
  const u64 b = key / n_seg, s = key % n_seg;
  u32 np = 0;
  if (len) { const u64 st = sub_off[key] + blk_shift[b]; np = (u32)((st + len - 1) / GB_CH - st / GB_CH) + 1u; }
  pc[s * n_blk + b] = np;
}
__global__ void k_gb_piece_ends(const u32* __restrict__ cnt, const u64* __restrict__ sub_off, const u64* __restrict__ blk_shift, u64 n_seg, u32 n_blk,
                                const u64* __restrict__ slot_scan, uint16_t* stream, u32* end_slot) {
  const u64 key = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= n_seg * (u64)n_blk) return;
  const u32 len = cnt[key];
  if (!len) return;
  const u64 b = key / n_seg, s = key % n_seg;
  const u64 st = sub_off[key] + blk_shift[b], last = st + len - 1;
  u32 slot = (u32)slot_scan[s * n_blk + b];
  for (u64 c = st / GB_CH; c <= last / GB_CH; c++) {
    const u64 p = min((c + 1) * GB_CH - 1, last);
    stream[p] |= GB_END;
    end_slot[p] = slot++;
  }
}
__global__ void k_gb_slot_base(const u64* __restrict__ slot_scan, u64 n_seg, u32 n_blk, u64 total, u32* slot_base) {
  const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_seg) slot_base[s] = (u32)slot_scan[s * n_blk];
  else if (s == n_seg) slot_base[s] = (u32)total;
}
// per chunk (one wavefront): PmSide-style lane words and the number of pieces that end in it
__global__ __launch_bounds__(BLOCK) void k_gb_lanes(const uint16_t* __restrict__ stream, u32 n_chunks, u32* lane_word, u32* chunk_ends) {
  const u32 c = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = lane_id();
  const uint16_t* p = stream + (u64)c * GB_CH + (u64)lane * GB_K;
  u32 ne = 0;
  for (int i = 0; i < GB_K; i++) ne += (p[i] & GB_END) ? 1u : 0u;
  const u32 incl = pm_scan_incl(ne);
  const u64 heads = __ballot(ne > 0);
  const u64 below = heads & ((2ULL << lane) - 1ULL);
  const u32 reach = below ? (u32)(lane - (63 - __clzll((long long)below))) : (u32)(lane + 1);
  lane_word[(u64)c * 64 + lane] = (incl - ne) | (ne << 12) | (reach << 18);
  if (lane == 63) chunk_ends[c] = incl;
}
__global__ __launch_bounds__(BLOCK) void k_gb_piece_slot(const uint16_t* __restrict__ stream, const u32* __restrict__ lane_word, const u64* __restrict__ seg_base64, u32 n_chunks,
                                                         const u32* __restrict__ end_slot, u32* piece_slot, u32* seg_base) {
  const u32 c = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = lane_id();
  const u64 p0 = (u64)c * GB_CH + (u64)lane * GB_K;
  u32 id = (u32)seg_base64[c] + (lane_word[(u64)c * 64 + lane] & 0xFFFu);
  for (int i = 0; i < GB_K; i++) if (stream[p0 + i] & GB_END) piece_slot[id++] = end_slot[p0 + i];
  if (lane == 0) seg_base[c] = (u32)seg_base64[c];
}
struct GbPlan { GbArgs args{}; bool valid = false; };
struct GiantPart {
  PmPlan plan;
  double* G_al[2] = {nullptr, nullptr}; double* G_a[2] = {nullptr, nullptr};   // the ping-pong state of the chunks (chunk k: [k & 1] -> [(k + 1) & 1])
  double* S_al[2] = {nullptr, nullptr}; double* S_a[2] = {nullptr, nullptr};   // what the rounds inside a chunk alternate between
  double* ac = nullptr;
  GiDesc* desc = nullptr;
  hipStream_t stream = nullptr; hipEvent_t ev = nullptr;
  hipGraph_t graph[2] = {nullptr, nullptr}; hipGraphExec_t gexec[2] = {nullptr, nullptr};
  bool use_graph = true;
  u64 nnz = 0, rows = 0;
  GbPlan gb;   // the same rounds in 2-D blocks with the gathers out of LDS (gb.valid), three launches per round instead of k_gi_rows / k_gi_cols
  void drop_graphs() {
    for (int i = 0; i < 2; i++) {
      if (gexec[i]) (void)hipGraphExecDestroy(gexec[i]);
      if (graph[i]) (void)hipGraphDestroy(graph[i]);
      gexec[i] = nullptr; graph[i] = nullptr;
    }
  }
};
// builds one direction of the blocked form from the streamed plan's segments (off, ids); 0 = ok, 1 = not applicable, < 0 = error
int gb_setup_side(kamd_ctx* c, const u64* off, const u32* ids, u64 n_seg, u64 n_tgt, int dir, GbSide* out) {
  const int shift = dir == 0 ? GB_SHIFT_A : GB_SHIFT_G;
  const u64 n_blk = (n_tgt + (1ULL << shift) - 1) >> shift;
  const u64 n_keys = n_blk * n_seg;
  if (n_blk == 0 || n_seg == 0 || n_keys >= 0x7FFFFFF0ULL || n_blk > 4096) return 1;
  DBuf& t1 = c->hy_gb[0]; DBuf& t2 = c->hy_gb[1]; DBuf& ps = c->hy_gb[2 + dir]; DBuf& pq = c->hy_gb[4 + dir];
  Carver v1;
  const size_t o_cnt = v1.take(n_keys * 4), o_fill = v1.take(n_keys * 4), o_sub = v1.take((n_keys + 2) * 8), o_shift = v1.take(n_blk * 8);
  const size_t o_pc = v1.take(n_keys * 4), o_sscan = v1.take((n_keys + 2) * 8);
  if (int rc = t1.ensure(v1.off, 0, c->stream)) return rc;
  char* b1 = (char*)t1.p;
  u32* cnt = (u32*)(b1 + o_cnt); u32* fill = (u32*)(b1 + o_fill); u64* sub_off = (u64*)(b1 + o_sub); u64* blk_shift = (u64*)(b1 + o_shift);
  u32* pc = (u32*)(b1 + o_pc); u64* slot_scan = (u64*)(b1 + o_sscan);
  HIPC(hipMemsetAsync(cnt, 0, o_sub - o_cnt, c->stream));   // cnt + fill
  hipLaunchKernelGGL(k_gb_count, dim3(grid_for(n_seg * 8, BLOCK)), dim3(BLOCK), 0, c->stream, off, ids, n_seg, shift, cnt);
  if (int rc = exclusive_scan(c, cnt, n_keys, sub_off, sub_off + n_keys)) return rc;
  std::vector<u64> start(n_blk + 1);
  hipLaunchKernelGGL(k_gb_block_starts, dim3(grid_for(n_blk + 1, BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)sub_off, n_seg, n_blk, slot_scan);   // (slot_scan: scratch until the pieces are counted)
  HIPC(hipMemcpyAsync(start.data(), slot_scan, (n_blk + 1) * 8, hipMemcpyDeviceToHost, c->stream));   // (entry n_blk: the total)
  HIPC(hipStreamSynchronize(c->stream));
  std::vector<u64> shift_h(n_blk), chunk0(n_blk + 1, 0);
  for (u64 b = 0; b < n_blk; b++) {
    const u64 len = start[b + 1] - start[b];
    chunk0[b + 1] = chunk0[b] + (len + GB_CH - 1) / GB_CH;
    shift_h[b] = chunk0[b] * GB_CH - start[b];   // (never negative: the padding only moves blocks up)
  }
  const u64 n_chunks = chunk0[n_blk];
  if (n_chunks == 0 || n_chunks >= 0x7FFFFFF0ULL / 64) return 1;
  const u64 nzpad = n_chunks * GB_CH;
  HIPC(hipMemcpyAsync(blk_shift, shift_h.data(), n_blk * 8, hipMemcpyHostToDevice, c->stream));
  // Work lists.  ONE workgroup per compute unit at most (128 KB of LDS each: one more than the units would run as a second round of the launch
  // and double its time); every block at least one workgroup, the rest dealt by the blocks' share of the chunks (largest remainders first), so
  // that a workgroup loads ONE block (loading a block is three dependent memory latencies in the columns pass: a second one per workgroup was
  // measured at +7 us per launch).  Workgroup i runs on XCD i % 8 (round-robin dispatch): the workgroups of a block get indices of one residue
  // while that residue has free slots, so that a block's values are fetched into few of the eight L2s (they start cold at every launch).
  std::vector<u32> desc, items;
  {
    std::vector<u64> parts(n_blk, 0), rem(n_blk, 0);
    u64 used = 0, nonempty = 0;
    for (u64 b = 0; b < n_blk; b++) if (chunk0[b + 1] > chunk0[b]) ++nonempty;
    const u64 budget = std::max<u64>(nonempty, std::min<u64>((u64)c->n_cus, n_chunks));
    for (u64 b = 0; b < n_blk; b++) {
      const u64 nc = chunk0[b + 1] - chunk0[b];
      if (!nc) continue;
      parts[b] = std::min<u64>(nc, std::max<u64>(1, nc * budget / n_chunks));
      rem[b] = nc * budget % n_chunks;
      used += parts[b];
    }
    while (used > budget) {   // (blocks that were lifted to one workgroup: take from the largest)
      u64 best = 0; for (u64 b = 1; b < n_blk; b++) if (parts[b] > parts[best]) best = b;
      if (parts[best] <= 1) break;
      --parts[best]; --used;
    }
    while (used < budget) {
      u64 best = n_blk; for (u64 b = 0; b < n_blk; b++) if (parts[b] && parts[b] < chunk0[b + 1] - chunk0[b] && (best == n_blk || rem[b] > rem[best])) best = b;
      if (best == n_blk) break;
      ++parts[best]; rem[best] = 0; ++used;
    }
    // consecutive workgroup indices per block: round-robin dispatch spreads a block's workgroups over the eight XCDs (grouping them on one XCD
    // was measured SLOWER, 29 against 22 us: they all read the same lines at the same time).  For the same reason every workgroup of a block
    // starts loading the block at a different place (the item's fourth word: a rotation of the index).
    for (u64 b = 0; b < n_blk; b++) {
      const u64 nc = chunk0[b + 1] - chunk0[b];
      for (u64 p = 0; p < parts[b]; p++) {
        desc.push_back((u32)(items.size() / 4));
        items.push_back((u32)b); items.push_back((u32)(chunk0[b] + nc * p / parts[b])); items.push_back((u32)(chunk0[b] + nc * (p + 1) / parts[b]));
        items.push_back((u32)((p * (1ULL << shift) / parts[b]) & ~63ULL));
        desc.push_back((u32)(items.size() / 4));
      }
    }
  }
  const u32 n_wg = (u32)(desc.size() / 2);
  Carver vs;
  const size_t o_stream = vs.take(nzpad * 2 + 64), o_lane = vs.take(n_chunks * 64 * 4), o_sb = vs.take(n_chunks * 4), o_wg = vs.take(desc.size() * 4 + 16), o_it = vs.take(items.size() * 4 + 16);
  if (int rc = ps.ensure(vs.off, 0, c->stream)) return rc;
  char* bs = (char*)ps.p;
  uint16_t* stream = (uint16_t*)(bs + o_stream); u32* lane_word = (u32*)(bs + o_lane); u32* seg_base = (u32*)(bs + o_sb); u32* wg_desc = (u32*)(bs + o_wg); u32* wg_items = (u32*)(bs + o_it);
  Carver v2;
  const size_t o_tmp = v2.take(nzpad * 2 + 64), o_es = v2.take(nzpad * 4), o_ce = v2.take(n_chunks * 4), o_sb64 = v2.take((n_chunks + 2) * 8);
  if (int rc = t2.ensure(v2.off, 0, c->stream)) return rc;
  char* b2 = (char*)t2.p;
  uint16_t* tmp = (uint16_t*)(b2 + o_tmp); u32* end_slot = (u32*)(b2 + o_es); u32* chunk_ends = (u32*)(b2 + o_ce); u64* seg_base64 = (u64*)(b2 + o_sb64);
  HIPC(hipMemsetD16Async((hipDeviceptr_t)stream, (unsigned short)GB_PAD, nzpad, c->stream));
  HIPC(hipMemcpyAsync(wg_desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice, c->stream));
  if (!items.empty()) HIPC(hipMemcpyAsync(wg_items, items.data(), items.size() * 4, hipMemcpyHostToDevice, c->stream));
  if (dir == 0) {
    hipLaunchKernelGGL(k_gb_scatter<true>, dim3(grid_for(n_seg * 8, BLOCK)), dim3(BLOCK), 0, c->stream, off, ids, n_seg, (const u32*)cnt, (const u64*)sub_off, (const u64*)blk_shift,
                       shift, fill, stream, (u32*)nullptr);
  } else {
    u32* info = end_slot;   // (the pieces' slots are written there later: the runs' keys are done with by then)
    hipLaunchKernelGGL(k_gb_scatter<false>, dim3(grid_for(n_seg * 8, BLOCK)), dim3(BLOCK), 0, c->stream, off, ids, n_seg, (const u32*)cnt, (const u64*)sub_off, (const u64*)blk_shift,
                       shift, fill, stream, info);
    hipLaunchKernelGGL(k_gb_sort_small, dim3(grid_for(nzpad, BLOCK)), dim3(BLOCK), 0, c->stream, (const uint16_t*)stream, (const u32*)info, (const u32*)cnt, (const u64*)sub_off,
                       (const u64*)blk_shift, n_seg, nzpad, tmp);
    hipLaunchKernelGGL(k_gb_sort_big, dim3(grid_for(n_keys, BLOCK)), dim3(BLOCK), 0, c->stream, (const uint16_t*)stream, (const u32*)cnt, (const u64*)sub_off, (const u64*)blk_shift,
                       n_seg, n_keys, tmp);
    HIPC(hipMemcpyAsync(stream, tmp, nzpad * 2, hipMemcpyDeviceToDevice, c->stream));
  }
  hipLaunchKernelGGL(k_gb_piece_count, dim3(grid_for(n_keys, BLOCK)), dim3(BLOCK), 0, c->stream, (const u32*)cnt, (const u64*)sub_off, (const u64*)blk_shift, n_seg, (u32)n_blk, pc);
  HIPC(hipGetLastError());
  if (int rc = exclusive_scan(c, pc, n_keys, slot_scan, slot_scan + n_keys)) return rc;
  u64 n_pieces = 0;
  HIPC(hipMemcpyAsync(&n_pieces, slot_scan + n_keys, 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));   // (desc / shift are host vectors: the copies above have completed too)
  if (n_pieces == 0 || n_pieces >= 0x7FFFFFF0ULL) return 1;
  Carver vq;
  const size_t o_ps = vq.take((n_pieces + 64) * 4), o_part = vq.take(n_pieces * 8), o_slb = vq.take((n_seg + 2) * 4);
  if (int rc = pq.ensure(vq.off, 0, c->stream)) return rc;
  char* bq = (char*)pq.p;
  u32* piece_slot = (u32*)(bq + o_ps); double* part = (double*)(bq + o_part); u32* slot_base = (u32*)(bq + o_slb);
  hipLaunchKernelGGL(k_gb_piece_ends, dim3(grid_for(n_keys, BLOCK)), dim3(BLOCK), 0, c->stream, (const u32*)cnt, (const u64*)sub_off, (const u64*)blk_shift, n_seg, (u32)n_blk,
                     (const u64*)slot_scan, stream, end_slot);
  hipLaunchKernelGGL(k_gb_slot_base, dim3(grid_for(n_seg + 1, BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)slot_scan, n_seg, (u32)n_blk, n_pieces, slot_base);
  hipLaunchKernelGGL(k_gb_lanes, dim3(grid_for(n_chunks, BLOCK / 64)), dim3(BLOCK), 0, c->stream, (const uint16_t*)stream, (u32)n_chunks, lane_word, chunk_ends);
  HIPC(hipGetLastError());
  if (int rc = exclusive_scan(c, chunk_ends, n_chunks, seg_base64, seg_base64 + n_chunks)) return rc;
  hipLaunchKernelGGL(k_gb_piece_slot, dim3(grid_for(n_chunks, BLOCK / 64)), dim3(BLOCK), 0, c->stream, (const uint16_t*)stream, (const u32*)lane_word, (const u64*)seg_base64,
                     (u32)n_chunks, (const u32*)end_slot, piece_slot, seg_base);
  HIPC(hipMemsetAsync(part, 0, n_pieces * 8, c->stream));
  HIPC(hipMemsetAsync(piece_slot + n_pieces, 0, 64 * 4, c->stream));
  HIPC(hipGetLastError());
  out->stream = stream; out->seg_base = seg_base; out->lane_word = lane_word; out->piece_slot = piece_slot; out->part = part; out->slot_base = slot_base; out->wg_desc = wg_desc; out->wg_items = wg_items;
  out->n_wg = n_wg; out->n_chunks = (u32)n_chunks; out->n_pieces = (u32)n_pieces; out->n_seg = (u32)n_seg; out->n_tgt = (u32)n_tgt;
  return 0;
}
constexpr size_t GB_LDS_BYTES = (size_t)(GB_B + 1) * 8 + 8 + (size_t)GB_WAVES * GB_CH * 8;
// the blocked form of the oversized components on top of their streamed plan (G.plan); 0 = G.gb is valid, 1 = not applicable, < 0 = error
int gb_setup(kamd_ctx* c, GiantPart& G) {
  G.gb.valid = false;
  const PmPlan& P = G.plan;
  const PmArgs& A = P.args;
  if (P.nzpad == 0 || A.R < 2 || A.M < 2) return 1;
  GbArgs ga{};
  if (int rc = gb_setup_side(c, P.roff, P.rs, A.R, A.M, 0, &ga.rows)) return rc;
  if (int rc = gb_setup_side(c, P.coff, P.cs, A.M, A.R, 1, &ga.cols)) return rc;
  ga.cw = A.cw; ga.single = A.single; ga.eff = A.eff; ga.R = A.R; ga.M = A.M;
  HIPC(hipFuncSetAttribute((const void*)k_gb_pass<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GB_LDS_BYTES));
  HIPC(hipFuncSetAttribute((const void*)k_gb_pass<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GB_LDS_BYTES));
  G.gb.args = ga; G.gb.valid = true;
  if (getenv("KAMD_DEBUG_FIN")) fprintf(stderr, "[kamd] blocked EM: R %u M %u nnz %llu; rows: %u chunks, %u pieces, %u workgroups; columns: %u chunks, %u pieces, %u workgroups\n", A.R, A.M,
                                        (unsigned long long)P.nzpad, ga.rows.n_chunks, ga.rows.n_pieces, ga.rows.n_wg, ga.cols.n_chunks, ga.cols.n_pieces, ga.cols.n_wg);
  return 0;
}
void gb_round(const GiantPart& G, hipStream_t s, int round, int clamp, const double* al_src, const double* a_src, double* al_dst, double* a_dst) {
  const GbArgs& A = G.gb.args;
  hipLaunchKernelGGL(k_gb_pass<0>, dim3(A.rows.n_wg), dim3(GB_THREADS), GB_LDS_BYTES, s, A, a_src, G.desc);
  hipLaunchKernelGGL(k_gb_pass<1>, dim3(A.cols.n_wg), dim3(GB_THREADS), GB_LDS_BYTES, s, A, a_src, G.desc);
  hipLaunchKernelGGL(k_gb_finish, dim3(grid_for(A.M, BLOCK)), dim3(BLOCK), 0, s, A, al_src, a_src, al_dst, a_dst, round, clamp, G.desc);
}
template <int K>
void gi_round(const GiantPart& G, hipStream_t s, int round, int clamp, const double* al_src, const double* a_src, double* al_dst, double* a_dst) {
  const PmPlan& P = G.plan;
  const unsigned grid = grid_for(P.n_chunks, PM_BLOCK / 64);
  constexpr int PRE_R = (K + 5) / 6 < 2 ? 2 : (K + 5) / 6;
  constexpr int PRE_C = K / 16 + 1;
  if (P.windowed) hipLaunchKernelGGL((k_gi_rows<K, PRE_R, true>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, a_src, G.desc);
  else hipLaunchKernelGGL((k_gi_rows<K, PRE_R, false>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, a_src, G.desc);
  if (P.n_fix[0]) hipLaunchKernelGGL(k_gi_rows_fix, dim3(grid_for(P.n_chunks, PM_BLOCK)), dim3(PM_BLOCK), 0, s, P.args, G.desc);
  if (P.windowed) hipLaunchKernelGGL((k_gi_cols<K, PRE_C, true>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, al_src, a_src, al_dst, a_dst, round, clamp, G.desc);
  else hipLaunchKernelGGL((k_gi_cols<K, PRE_C, false>), dim3(grid), dim3(PM_BLOCK), 0, s, P.args, al_src, a_src, al_dst, a_dst, round, clamp, G.desc);
  if (P.n_fix[1]) hipLaunchKernelGGL(k_gi_cols_fix, dim3(grid_for(P.n_chunks, PM_BLOCK)), dim3(PM_BLOCK), 0, s, P.args, al_src, a_src, al_dst, a_dst, round, clamp, G.desc);
}
// n rounds from state [pin] to state [pout]; the rounds in between alternate between the two scratch states, so the input stays intact
void gi_enqueue_rounds(const GiantPart& G, hipStream_t s, int n, int clamp, int pin, int pout) {
  const u32 M = G.plan.args.M;
  for (int i = 0; i < n; i++) {
    const double* al_src = i == 0 ? G.G_al[pin] : G.S_al[(i - 1) & 1];
    const double* a_src = i == 0 ? G.G_a[pin] : G.S_a[(i - 1) & 1];
    double* al_dst = i == n - 1 ? G.G_al[pout] : G.S_al[i & 1];
    double* a_dst = i == n - 1 ? G.G_a[pout] : G.S_a[i & 1];
    if (clamp) {
      hipLaunchKernelGGL(k_gi_clamp, dim3(grid_for((u64)M + 1, BLOCK)), dim3(BLOCK), 0, s, al_src, a_src, G.ac, M, G.desc);
      a_src = G.ac;
    }
    if (G.gb.valid) { gb_round(G, s, i, clamp, al_src, a_src, al_dst, a_dst); continue; }
    switch (G.plan.k) {
      case 8: gi_round<8>(G, s, i, clamp, al_src, a_src, al_dst, a_dst); break;
      case 12: gi_round<12>(G, s, i, clamp, al_src, a_src, al_dst, a_dst); break;
      case 16: gi_round<16>(G, s, i, clamp, al_src, a_src, al_dst, a_dst); break;
      case 20: gi_round<20>(G, s, i, clamp, al_src, a_src, al_dst, a_dst); break;
      case 24: gi_round<24>(G, s, i, clamp, al_src, a_src, al_dst, a_dst); break;
      case 28: gi_round<28>(G, s, i, clamp, al_src, a_src, al_dst, a_dst); break;
      default: gi_round<32>(G, s, i, clamp, al_src, a_src, al_dst, a_dst); break;
    }
  }
}
// one chunk on the oversized components' stream (the caller has made that stream wait for the context stream and joins it afterwards)
int gi_launch_chunk(GiantPart& G, int n, int clamp, int pin, int pout, int* d_h, const EmsPrev& prev) {
  hipLaunchKernelGGL(k_gi_set_desc, dim3(1), dim3(64), 0, G.stream, G.desc, prev, d_h);
  if (G.use_graph && n == EML_MAX_ROUNDS && !clamp && pin != pout) {
    if (!G.gexec[pin]) {
      // a capture that cannot be completed must not leave the stream capturing or a half-built graph behind: the chunk then goes out as
      // plain launches (what use_graph == false does), for this and every later chunk of the plan
      const bool inject = getenv("KAMD_DEBUG_GRAPH_FAIL") != nullptr;   // (tests: the graph of the SECOND parity cannot be instantiated; read on every capture -- off the hot path)
      hipError_t e = hipStreamBeginCapture(G.stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        gi_enqueue_rounds(G, G.stream, n, 0, pin, pout);
        e = hipStreamEndCapture(G.stream, &G.graph[pin]);   // (ends the capture on failure too)
      }
      if (e == hipSuccess && inject && G.gexec[pin ^ 1]) e = hipErrorOutOfMemory;
      if (e == hipSuccess) e = hipGraphInstantiate(&G.gexec[pin], G.graph[pin], nullptr, nullptr, 0);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        G.drop_graphs();
        G.use_graph = false;
      }
    }
    if (G.use_graph) { HIPC(hipGraphLaunch(G.gexec[pin], G.stream)); }
    else gi_enqueue_rounds(G, G.stream, n, clamp, pin, pout);
  } else gi_enqueue_rounds(G, G.stream, n, clamp, pin, pout);
  HIPC(hipGetLastError());
  return 0;
}
}  // namespace
namespace kamdi {
struct SellCache {
  bool valid = false;
  const u64* d_ec_off = nullptr; const u32* d_ec_ids = nullptr; u64 n_ecs = 0, nnz = 0, T = 0, generation = 0;
  int split_len = 0, group_div = 0, small_nnz = 0;
  bool host_maps = false;   // P.tr_id / P.single_all were read back (the host-side scatter of several ranks needs them)
  kamd_em_sell::Plan P;      // host part (tr_id, single_all, bases)
  EmSellDev dev{};           // device part, in ctx->ems_plan
  u32* row_final = nullptr; u32* mslot = nullptr; double* single_all = nullptr; double* d_eff = nullptr;   // in ctx->ems_maps
  bool hybrid = false;       // G holds the streamed plan of the components beyond a workgroup's LDS; P / dev the others
  GiantPart G;
  // what a cached hybrid plan needs to take new counts (a bootstrap replicate): the split of the rows into the two sub-matrices (arrays in ctx->hy_sub)
  struct HyKeep {
    const u32* flag_s = nullptr; const u32* flag_g = nullptr; const u64* rpos_s = nullptr; const u64* rpos_g = nullptr;
    const u64* off_s = nullptr; const u64* off_g = nullptr; const u32* ids_s = nullptr; const u32* ids_g = nullptr;
    u32* cnt_s = nullptr; u32* cnt_g = nullptr; u32* wc_s = nullptr; u32* wc_g = nullptr;
    u64 n_s = 0, n_g = 0; bool no_groups = false;
  } hy;
  int giant_nnz_t = 0, blocked_t = 0, hybrid_t = 0;   // tuning the cached plan was built under
  ~SellCache() { G.drop_graphs(); }
};
}  // namespace kamdi
namespace {

// ---- the sliced-ELLPACK form: device plan + backend of kamd_em_local::run --------------------------------------------------
__global__ void k_em_publish(EmsPrev prev);
struct EmSellGpu {
  kamd_ctx* c; const kamd_em_sell::Plan& P; EmSellDev dev{}; u64 M = 0; size_t lds = 0, team_bytes = 0; int block = 256;
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int reg_words = 0;   // index words per lane and direction the kernel keeps in registers (0: the form that reads everything from LDS)
  double* d_alpha = nullptr; double* d_a = nullptr; double* d_ck_alpha = nullptr; double* d_ck_a = nullptr; int* d_hist = nullptr; double* d_out = nullptr;
  std::vector<double> h_alpha; int err = 0;
  const EmPartition* part = nullptr;   // several ranks: the change counts of a chunk are summed over them before the host reads them
  GiantPart* gi = nullptr;             // hybrid: the oversized components, iterated on gi->stream beside the groups
  hipEvent_t ev_sell = nullptr;
  int gi_par(const double* al) const { return al == d_alpha ? 0 : 1; }   // which half of the ping-pong pair a vector of the groups is
  EmSellGpu(kamd_ctx* ctx, const kamd_em_sell::Plan& p) : c(ctx), P(p) {}
  int setup(int hist_ints, const double* d_eff_new, u64 T_out);
  // (the in-place interface of kamd_em_local::run works on state 0 of the oversized side's ping-pong pair, state 1 is its checkpoint)
  void checkpoint() {
    if (err) return;
    if (hipMemcpyAsync(d_ck_alpha, d_alpha, M * 8, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_ck_a, d_a, M * 8, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) err = -104;
    if (gi && !err) {
      const size_t gb = ((size_t)gi->plan.args.M + 1) * 8;
      if (hipMemcpyAsync(gi->G_al[1], gi->G_al[0], gb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
          hipMemcpyAsync(gi->G_a[1], gi->G_a[0], gb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) err = -104;
    }
  }
  void restore() {
    if (err) return;
    if (hipMemcpyAsync(d_alpha, d_ck_alpha, M * 8, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_a, d_ck_a, M * 8, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) err = -104;
    if (gi && !err) {
      const size_t gb = ((size_t)gi->plan.args.M + 1) * 8;
      if (hipMemcpyAsync(gi->G_al[0], gi->G_al[1], gb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
          hipMemcpyAsync(gi->G_a[0], gi->G_a[1], gb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) err = -104;
    }
  }
  std::vector<double> h_gi_cur, h_gi_prev; int host_reads = 0;   // the oversized side's alpha at the last two host_alpha() calls (final / before the final round)
  // n rounds from (al_in, a_in) to (al_out, a_out) for every group; change counts of the rounds added to d_h (device, zeroed by the caller) if given
  int launch(int n, int clamp, const double* al_in, const double* a_in, double* al_out, double* a_out, int* d_h, const EmsPrev& prev, long long* clk = nullptr) {
    if (n <= 0 || n > EML_MAX_ROUNDS) return -104;
    // the two size classes run side by side: small groups (one wavefront each) on a second stream, forked from and joined to the context stream
    const u32 n_big = P.n_groups - P.n_small;
    const bool fork = P.n_small && n_big;
    hipStream_t ss = c->stream;   // where k_em_sell runs
    if ((fork || gi) && hipEventRecord(ev_fork, c->stream) != hipSuccess) return -104;
    if (fork && hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess) return -104;
    if (gi) {
      // nobody else hands the previous chunk's counts to the host when there is no group at all
      if (!n_big && !P.n_small && prev.hist && prev.host_hist) hipLaunchKernelGGL(k_em_publish, dim3(1), dim3(64), 0, c->stream, prev);
      if (hipStreamWaitEvent(gi->stream, ev_fork, 0) != hipSuccess) return -104;
      if (ss != c->stream && hipStreamWaitEvent(ss, ev_fork, 0) != hipSuccess) return -104;
      // the oversized components first: their kernels are many and short, the groups' one launch then fills the compute units left to it
      if (int rc = gi_launch_chunk(*gi, n, clamp, gi_par(al_in), gi_par(al_out), d_h, prev)) return rc;
      if (hipEventRecord(gi->ev, gi->stream) != hipSuccess) return -104;
    }
    if (n_big) {
#define KAMD_EMS_LAUNCH(C, E) hipLaunchKernelGGL((k_em_sell<C, E>), dim3(n_big), dim3(block), lds, ss, dev, P.n_small, al_in, a_in, al_out, a_out, n, clamp, d_h, prev, clk)
#define KAMD_EMS_LAUNCH_REG(C, W, NS) hipLaunchKernelGGL((k_em_sell<C, 0, W, NS>), dim3(n_big), dim3(block), lds, ss, dev, P.n_small, al_in, a_in, al_out, a_out, n, clamp, d_h, prev, clk)
      if (reg_words == 2) { if (clk) KAMD_EMS_LAUNCH_REG(true, 2, 1); else KAMD_EMS_LAUNCH_REG(false, 2, 1); }
      else if (reg_words == 4) { if (clk) KAMD_EMS_LAUNCH_REG(true, 4, 2); else KAMD_EMS_LAUNCH_REG(false, 4, 2); }
      else if (reg_words == 8) { if (clk) KAMD_EMS_LAUNCH_REG(true, 8, 1); else KAMD_EMS_LAUNCH_REG(false, 8, 1); }
      else if (clk) KAMD_EMS_LAUNCH(true, 0);
      else KAMD_EMS_LAUNCH(false, 0);
#undef KAMD_EMS_LAUNCH
#undef KAMD_EMS_LAUNCH_REG
    }
    EmsPrev prev_w = prev;
    if (n_big) prev_w.host_hist = nullptr;   // (one kernel reports the previous chunk's counts to the host: the workgroup kernel if it runs)
    if (P.n_small) hipLaunchKernelGGL(k_em_sell_wave, dim3((P.n_small + EMS_WAVE_TEAMS - 1) / EMS_WAVE_TEAMS), dim3(64 * EMS_WAVE_TEAMS), (size_t)EMS_WAVE_TEAMS * team_bytes,
                                      fork ? side : c->stream, dev, P.n_small, (u32)team_bytes, al_in, a_in, al_out, a_out, n, clamp, d_h, prev_w);
    if (hipGetLastError() != hipSuccess) return -104;
    if (fork && (hipEventRecord(ev_join, side) != hipSuccess || hipStreamWaitEvent(c->stream, ev_join, 0) != hipSuccess)) return -104;
    if (gi) {
      if (ss != c->stream && (hipEventRecord(ev_sell, ss) != hipSuccess || hipStreamWaitEvent(c->stream, ev_sell, 0) != hipSuccess)) return -104;
      if (hipStreamWaitEvent(c->stream, gi->ev, 0) != hipSuccess) return -104;
    }
    return 0;
  }
  // the backend interface of kamd_em_local::run (several ranks): in place, the host reads the counts after every chunk
  void run(int n, int clamp, int* hist) {
    if (err || n <= 0) return;
    if (n > EML_MAX_ROUNDS) { err = -104; return; }
    if (hist && hipMemsetAsync(d_hist, 0, (size_t)n * sizeof(int), c->stream) != hipSuccess) { err = -104; return; }
    if (int rc = launch(n, clamp, d_alpha, d_a, d_alpha, d_a, hist ? d_hist : nullptr, EmsPrev{})) { err = rc; return; }
    if (hist && part) {
      if (hipStreamSynchronize(c->stream) != hipSuccess) { err = -104; return; }
      if (part->cb(part->user, (int32_t*)d_hist, n)) { err = -103; return; }
    }
    if (hist && (hipMemcpyAsync(hist, d_hist, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                 hipStreamSynchronize(c->stream) != hipSuccess)) err = -104;
  }
  const std::vector<double>& host_alpha() {
    h_alpha.resize(M);
    if (gi && !err) {
      h_gi_prev.swap(h_gi_cur);
      h_gi_cur.resize(gi->plan.args.M);
      if (hipMemcpyAsync(h_gi_cur.data(), gi->G_al[0], (size_t)gi->plan.args.M * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) err = -104;
    }
    ++host_reads;
    if (!err && ((M && hipMemcpyAsync(h_alpha.data(), d_alpha, M * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
                 hipStreamSynchronize(c->stream) != hipSuccess)) err = -104;
    return h_alpha;
  }
};
int EmSellGpu::setup(int hist_ints, const double* d_eff_new, u64 T_out) {
  Carver sv;
  // the two alpha vectors (and the two a vectors) lie back to back: a ping-pong pair, and one copy brings both to the host
  const u64 Mp = M ? M : 1;   // (a plan without groups -- the hybrid with everything on its streamed side -- still has two distinct halves: gi_par tells them apart)
  const size_t o_al = sv.take(2 * Mp * 8 + 8), o_a = sv.take(2 * Mp * 8 + 8), o_h = sv.take((size_t)hist_ints * 4 + 8), o_out = sv.take(2 * T_out * 8 + 8);
  if (int rc = c->pm_b.ensure(sv.off, 0, c->stream)) return rc;
  char* sb = (char*)c->pm_b.p;
  d_alpha = (double*)(sb + o_al); d_a = (double*)(sb + o_a); d_ck_alpha = d_alpha + Mp; d_ck_a = d_a + Mp; d_hist = (int*)(sb + o_h); d_out = (double*)(sb + o_out);
  if (M) hipLaunchKernelGGL(k_eml_init, dim3(grid_for(M, BLOCK)), dim3(BLOCK), 0, c->stream, d_alpha, d_a, d_eff_new, M, 1.0 / (double)P.T);
  HIPC(hipGetLastError());
  lds = (size_t)P.max_group_bytes;
  // the register-resident form holds the index words of a wavefront's slices in registers: 2 / 4 / 8 words for split lengths up to
  // 8 / 16 / 32 (a longer split length, a timing experiment or KAMD_EM_REG=0 take the form that reads everything from LDS)
  reg_words = 0;
  {
    const int cap = std::min(64, std::max(1, c->tune.em_split_len));
    if (c->tune.em_reg_slices == 1) reg_words = cap <= 8 ? 2 : cap <= 16 ? 4 : cap <= 32 ? 8 : 0;
  }
  if (P.n_groups > P.n_small) {
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<false, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<true, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<false, 0, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<true, 0, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<false, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPC(hipFuncSetAttribute((const void*)k_em_sell<true, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (gi && !c->em_ev_fork) {
    HIPC(hipEventCreateWithFlags(&c->em_ev_fork, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&c->em_ev_join, hipEventDisableTiming));
  }
  if (gi) ev_fork = c->em_ev_fork;
  if (P.n_small) {
    team_bytes = ((size_t)P.max_small_bytes + 15) & ~(size_t)15;
    HIPC(hipFuncSetAttribute((const void*)k_em_sell_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(EMS_WAVE_TEAMS * team_bytes)));
    if (!c->em_side_stream) {
      HIPC(hipStreamCreateWithFlags(&c->em_side_stream, hipStreamNonBlocking));
      if (!c->em_ev_fork) {
        HIPC(hipEventCreateWithFlags(&c->em_ev_fork, hipEventDisableTiming));
        HIPC(hipEventCreateWithFlags(&c->em_ev_join, hipEventDisableTiming));
      }
    }
    side = c->em_side_stream; ev_fork = c->em_ev_fork; ev_join = c->em_ev_join;
  }
  return 0;
}
// back to transcript space on the device: a transcript outside m-space keeps its singleton count from round 1 on (0 if in no set)
__global__ void k_em_scatter(const u32* __restrict__ mslot, const double* __restrict__ single_all, const double* __restrict__ fin,
                             const double* __restrict__ before, u64 T, int have_final, double* out_alpha, double* out_abz) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const u32 ms = mslot[t];
  const double sa = single_all[t];
  out_alpha[t] = ms != 0xFFFFFFFFu ? fin[ms] : sa;
  out_abz[t] = have_final ? (ms != 0xFFFFFFFFu ? before[ms] : sa) : 0.0;
}
// hands the change counts of the LAST chunk a run can have to the host (no chunk follows it that would)
__global__ void k_em_publish(EmsPrev prev) {
  __shared__ int s_stop;
  (void)ems_prev_stopped(prev, &s_stop);
}
// The loop control of EMAlgorithm::run (:112-223) for ONE rank, without a stream synchronisation per chunk of rounds: the chunks
// ping-pong between two copies of (alpha, a) -- the input of a chunk IS the checkpoint it is replayed from --, chunk k + 1 is queued
// while chunk k runs, evaluates the stop rule on chunk k's change counts itself (and returns at once if the run stops in k) and
// publishes them to pinned host memory, which the host polls.  Then: replay up to the stop round, the final round with the clamp
// (:212-221), the scatter to transcript space, ONE copy to the host.
int em_sell_drive_async(kamd_ctx* c, EmSellGpu& B, const SellCache& K, u64 T, int n_iter, int min_rounds, double* alpha_out, double* abz_out, int* rounds_out) {
  const int CH = EML_MAX_ROUNDS;
  if (n_iter <= 0) {   // no round at all: the initial vector (alpha_ = 1/T), no final round
    hipLaunchKernelGGL(k_em_scatter, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, K.mslot, K.single_all, B.d_alpha, B.d_alpha, T, 0, B.d_out, B.d_out + T);
    if (B.gi) hipLaunchKernelGGL(k_gi_scatter, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, T, B.gi->plan.mflag, B.gi->plan.mpos, B.gi->G_al[0], B.gi->G_al[0], 0,
                                 B.d_out, B.d_out + T);
    HIPC(hipGetLastError());
    std::vector<double> tmp(2 * T);
    HIPC(hipMemcpyAsync(tmp.data(), B.d_out, 2 * T * 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    memcpy(alpha_out, tmp.data(), T * 8);
    if (abz_out) memcpy(abz_out, tmp.data() + T, T * 8);
    *rounds_out = 0;
    return 0;
  }
  const int n_chunks = std::max(1, (n_iter + CH - 1) / CH);
  // pinned, mapped host memory: change counts per chunk | sequence word | result staging
  const size_t hist_ints = (size_t)(n_chunks + 1) * CH;
  const size_t need = hist_ints * 4 + 256 + 2 * T * 8 + 256;
  if (c->em_pin_bytes < need) {
    if (c->em_pin) { HIPC(hipStreamSynchronize(c->stream)); HIPC(hipHostFree(c->em_pin)); c->em_pin = nullptr; c->em_pin_bytes = 0; }
    HIPC(hipHostMalloc(&c->em_pin, need, hipHostMallocMapped));
    c->em_pin_bytes = need;
  }
  int* h_hist = (int*)c->em_pin;
  int* h_seq = h_hist + hist_ints;
  double* h_out = (double*)((char*)c->em_pin + hist_ints * 4 + 256);
  void* dp = nullptr;
  HIPC(hipHostGetDevicePointer(&dp, c->em_pin, 0));
  int* d_hh = (int*)dp; int* d_seq = d_hh + hist_ints;
  __atomic_store_n(h_seq, 0, __ATOMIC_RELEASE);
  HIPC(hipMemsetAsync(B.d_hist, 0, (size_t)n_chunks * CH * sizeof(int), c->stream));
  double* al[2] = {B.d_alpha, B.d_ck_alpha};
  double* av[2] = {B.d_a, B.d_ck_a};
  auto prev_of = [&](int k) {   // what chunk k (or the publisher behind the last chunk) is told about chunk k - 1
    if (k == 0) return EmsPrev{};
    const int pb = (k - 1) * CH;
    return EmsPrev{B.d_hist + (size_t)(k - 1) * CH, std::min(CH, n_iter - pb), pb, min_rounds, d_hh + (size_t)(k - 1) * CH, d_seq, k};
  };
  // diagnostic: KAMD_EM_CLK=<file> -> phase clocks of every wavefront in one round of the first chunk
  const char* clk_path = getenv("KAMD_EM_CLK");
  long long* d_clk = nullptr; size_t clk_words = 0;
  if (clk_path && *clk_path && B.P.n_groups > B.P.n_small) {
    clk_words = (size_t)(B.P.n_groups - B.P.n_small) * (EMS_MAX_BLOCK / 64) * EMS_CLK_WORDS;
    if (int rc = c->em_clk.ensure(clk_words * 8, 0, c->stream)) return rc;
    d_clk = c->em_clk.as<long long>();
    HIPC(hipMemsetAsync(d_clk, 0, clk_words * 8, c->stream));
  }
  int launched = 0, decided = 0, stop = -1;
  bool published_last = false;
  for (;;) {
    while (launched < n_chunks && launched < decided + 2) {
      const int k = launched, base = k * CH, n = std::min(CH, n_iter - base);
      if (int rc = B.launch(n, 0, al[k & 1], av[k & 1], al[(k + 1) & 1], av[(k + 1) & 1], B.d_hist + (size_t)k * CH, prev_of(k), k == 0 ? d_clk : nullptr))
        return kamd::fail(rc, "kamd_em_run: the component-local EM failed to launch");
      ++launched;
    }
    if (launched == n_chunks && decided == n_chunks - 1 && !published_last) {
      hipLaunchKernelGGL(k_em_publish, dim3(1), dim3(64), 0, c->stream, prev_of(n_chunks));
      HIPC(hipGetLastError());
      published_last = true;
    }
    // wait until the counts of chunk `decided` have arrived
    {
      const auto t0 = std::chrono::steady_clock::now();
      for (u64 spin = 1;; spin++) {
        if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) >= decided + 1) break;
        if ((spin & 4095) == 0) {
          const hipError_t q = hipStreamQuery(c->stream);
          if (q == hipSuccess) {   // everything queued has run
            if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) >= decided + 1) break;
            return kamd::fail(-104, "kamd_em_run: the change counts of a chunk never arrived");
          }
          if (q != hipErrorNotReady) return kamd::fail(-104, std::string("kamd_em_run: ") + hipGetErrorString(q));
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return kamd::fail(-104, "kamd_em_run: timed out waiting for a chunk of rounds");
        }
        __builtin_ia32_pause();
      }
    }
    const int base = decided * CH, n = std::min(CH, n_iter - base);
    const int* hh = h_hist + (size_t)decided * CH;
    for (int i = 0; i < n; i++) if (hh[i] == 0 && base + i > min_rounds) { stop = base + i; break; }   // :202-205
    if (stop >= 0) break;
    if (++decided == n_chunks) break;   // the loop ran out: no final round
  }
  const double* fin; const double* before;
  int rounds;
  const bool have_final = stop >= 0;
  if (have_final) {
    const int s = decided, base = s * CH;
    // (chunk s + 1, queued behind s, saw the stop and left its output -- the input of s -- alone)
    if (int rc = B.launch(stop - base + 1, 0, al[s & 1], av[s & 1], al[(s + 1) & 1], av[(s + 1) & 1], nullptr, EmsPrev{})) return kamd::fail(rc, "kamd_em_run: replay failed to launch");
    if (int rc = B.launch(1, 1, al[(s + 1) & 1], av[(s + 1) & 1], al[s & 1], av[s & 1], nullptr, EmsPrev{})) return kamd::fail(rc, "kamd_em_run: final round failed to launch");
    before = al[(s + 1) & 1]; fin = al[s & 1];   // what the final round read is alpha_before_zeroes_
    rounds = stop + 1;
  } else {
    fin = al[n_chunks & 1]; before = fin;
    rounds = n_iter;
  }
  hipLaunchKernelGGL(k_em_scatter, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, K.mslot, K.single_all, fin, before, T, have_final ? 1 : 0, B.d_out, B.d_out + T);
  if (B.gi) hipLaunchKernelGGL(k_gi_scatter, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, T, B.gi->plan.mflag, B.gi->plan.mpos, B.gi->G_al[B.gi_par(fin)],
                               B.gi->G_al[B.gi_par(before)], have_final ? 1 : 0, B.d_out, B.d_out + T);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(h_out, B.d_out, 2 * T * 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  memcpy(alpha_out, h_out, T * 8);
  if (abz_out) memcpy(abz_out, h_out + T, T * 8);
  if (d_clk) {
    std::vector<long long> hc(clk_words);
    HIPC(hipMemcpy(hc.data(), d_clk, clk_words * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(clk_path, "wb")) {
      const long long hdr[4] = {(long long)(B.P.n_groups - B.P.n_small), EMS_MAX_BLOCK / 64, EMS_CLK_WORDS, B.block};
      fwrite(hdr, 8, 4, f); fwrite(hc.data(), 8, clk_words, f); fclose(f);
    }
  }
  *rounds_out = rounds;
  return 0;
}
// 0 = plan built (P: the host part; *dev: the device part), 1 = not applicable, < 0 = error
int em_sell_setup_device(kamd_ctx* c, const u64* d_ec_off, const u32* d_ec_ids, const u32* d_counts, const u32* d_wcounts, u64 n_ecs, u64 nnz,
                         const double* eff_lens, u64 T, u64 lds_budget, u64 target, kamd_em_sell::Plan* P, EmSellDev* dev, SellCache* cache,
                         u32 small_limit = 0, u64 target_small = 0, u64 small_budget = 0, bool host_maps = true, CompStats* comp_out = nullptr) {
  namespace S = kamd_em_sell;
  kamd_em_local::Plan C;
  EmLocalDev cd{};
  kamd_em_local::BuildArgs A{};
  if (int rc = em_local_setup_device(c, d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, nnz, eff_lens, T, ~0ULL, target, &C, &cd, &A, small_limit, target_small, host_maps, comp_out)) return rc;
  const u32 ng = C.n_groups;
  const u64 R = C.row_base[ng], M = C.tr_base[ng];
  if (R >= 0xFFFFFFF0ULL || M >= 0xFFFFFFF0ULL) return 1;
  for (u32 g = 0; g < ng; g++)
    if (C.row_base[g + 1] - C.row_base[g] >= S::SELL_PAD || C.tr_base[g + 1] - C.tr_base[g] >= S::SELL_PAD) return 1;
  // scratch: lengths, new ids, lanes, per-group sizes
  Carver tv;
  const size_t o_rlen = tv.take(R * 4 + 8), o_clen = tv.take(M * 4 + 8), o_rnew = tv.take(R * 4 + 8), o_cnew = tv.take(M * 4 + 8);
  const size_t o_rlane = tv.take(R * 4 + 8), o_clane = tv.take(M * 4 + 8), o_rvl = tv.take(R * 4 + 8), o_cvl = tv.take(M * 4 + 8), o_gsz = tv.take((size_t)ng * 16 + 16);
  if (int rc = c->ems_tmp.ensure(tv.off, 0, c->stream)) return rc;
  char* tb = (char*)c->ems_tmp.p;
  SellBuild B{};
  B.row_base = cd.row_base; B.tr_base = cd.tr_base; B.row_ptr = cd.row_ptr; B.col_ptr = cd.col_ptr; B.row_tr = cd.row_tr; B.col_row = cd.col_row;
  B.nz_base = cd.nz_base; B.cw = cd.cw; B.single = cd.single; B.eff = cd.eff; B.tr_id = cd.tr_id; B.n_groups = ng; B.R = (u32)R; B.M = (u32)M;
  B.cap = (u32)std::min(64, std::max(1, c->tune.em_split_len));
  B.rlen = (u32*)(tb + o_rlen); B.clen = (u32*)(tb + o_clen); B.rnew = (u32*)(tb + o_rnew); B.cnew = (u32*)(tb + o_cnew);
  B.rlane = (u32*)(tb + o_rlane); B.clane = (u32*)(tb + o_clane); B.rvl = (u32*)(tb + o_rvl); B.cvl = (u32*)(tb + o_cvl); B.gsz = (u32*)(tb + o_gsz);
  const u64 nseg = std::max(R, M);
  hipLaunchKernelGGL(k_sell_lens, dim3(grid_for(nseg, BLOCK)), dim3(BLOCK), 0, c->stream, B);
  hipLaunchKernelGGL(k_sell_sizes, dim3(grid_for(2 * (u64)ng, SELL_BUILD_WAVES)), dim3(64 * SELL_BUILD_WAVES), 0, c->stream, B);
  std::vector<u32> gsz((size_t)ng * 4);
  HIPC(hipMemcpyAsync(gsz.data(), B.gsz, (size_t)ng * 16, hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  P->n_groups = ng; P->n_small = C.n_small; P->T = T; P->row_base = C.row_base; P->tr_base = C.tr_base; P->single_all = C.single_all;
  P->rslice_base.assign(ng + 1, 0); P->cslice_base.assign(ng + 1, 0); P->rell_base.assign(ng + 1, 0); P->cell_base.assign(ng + 1, 0);
  P->max_group_bytes = 0; P->max_small_bytes = 0;
  for (u32 g = 0; g < ng; g++) {
    const u64 gb = S::group_bytes(C.row_base[g + 1] - C.row_base[g], C.tr_base[g + 1] - C.tr_base[g], gsz[4 * g], gsz[4 * g + 2], gsz[4 * g + 1], gsz[4 * g + 3]);
    if (g < P->n_small) {
      if (gb > small_budget) return 2;   // a small group that does not fit a wavefront's share of the LDS: the caller drops the size classes
      P->max_small_bytes = std::max<uint64_t>(P->max_small_bytes, gb);
    } else {
      if (gb > lds_budget) return 1;
      P->max_group_bytes = std::max<uint64_t>(P->max_group_bytes, gb);
    }
    P->rslice_base[g + 1] = P->rslice_base[g] + gsz[4 * g]; P->rell_base[g + 1] = P->rell_base[g] + gsz[4 * g + 1];
    P->cslice_base[g + 1] = P->cslice_base[g] + gsz[4 * g + 2]; P->cell_base[g + 1] = P->cell_base[g] + gsz[4 * g + 3];
  }
  const u64 nrs = P->rslice_base[ng], ncs = P->cslice_base[ng], nru = P->rell_base[ng], ncu = P->cell_base[ng];
  Carver pv;
  const size_t p_rsb = pv.take((ng + 1) * 4), p_csb = pv.take((ng + 1) * 4), p_reb = pv.take((ng + 1) * 8), p_ceb = pv.take((ng + 1) * 8);
  const size_t p_rd = pv.take(nrs * 8 + 8), p_cdesc = pv.take(ncs * 8 + 8), p_re = pv.take(nru * 2 + 8), p_ce = pv.take(ncu * 2 + 8);
  const size_t p_cw = pv.take(R * 8 + 8), p_sg = pv.take(M * 8 + 8), p_ef = pv.take(M * 8 + 8), p_id = pv.take(M * 4 + 8);
  if (int rc = c->ems_plan.ensure(pv.off, 0, c->stream)) return rc;
  char* pb = (char*)c->ems_plan.p;
  HIPC(hipMemcpyAsync(pb + p_rsb, P->rslice_base.data(), (ng + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemcpyAsync(pb + p_csb, P->cslice_base.data(), (ng + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemcpyAsync(pb + p_reb, P->rell_base.data(), (ng + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemcpyAsync(pb + p_ceb, P->cell_base.data(), (ng + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemsetAsync(pb + p_re, 0xFF, nru * 2, c->stream));
  HIPC(hipMemsetAsync(pb + p_ce, 0xFF, ncu * 2, c->stream));
  B.rslice_base = (const u32*)(pb + p_rsb); B.cslice_base = (const u32*)(pb + p_csb); B.rell_base = (const u64*)(pb + p_reb); B.cell_base = (const u64*)(pb + p_ceb);
  B.rdesc = (u32*)(pb + p_rd); B.cdesc = (u32*)(pb + p_cdesc); B.rell = (uint16_t*)(pb + p_re); B.cell = (uint16_t*)(pb + p_ce);
  B.cw_new = (u64*)(pb + p_cw); B.single_new = (double*)(pb + p_sg); B.eff_new = (double*)(pb + p_ef); B.tr_id_new = (u32*)(pb + p_id);
  hipLaunchKernelGGL(k_sell_layout, dim3(grid_for(2 * (u64)ng, SELL_BUILD_WAVES)), dim3(64 * SELL_BUILD_WAVES), 0, c->stream, B);
  hipLaunchKernelGGL(k_sell_entries, dim3(grid_for(nseg, BLOCK)), dim3(BLOCK), 0, c->stream, B);
  HIPC(hipGetLastError());
  if (host_maps) {
    P->tr_id.resize(M);
    if (M) HIPC(hipMemcpyAsync(P->tr_id.data(), B.tr_id_new, M * 4, hipMemcpyDeviceToHost, c->stream));
  }
  // the group bases live in the CSR plan's arena (pm_a), which other EM forms reuse: copy them next to the plan
  Carver bv;
  const size_t b_rb = bv.take((ng + 1) * 4), b_tb = bv.take((ng + 1) * 4), b_rf = bv.take(n_ecs * 4 + 8), b_ms = bv.take(T * 4 + 8), b_sa = bv.take(T * 8 + 8),
               b_ef = bv.take(T * 8 + 8);
  if (int rc = c->ems_maps.ensure(bv.off, 0, c->stream)) return rc;
  char* mb = (char*)c->ems_maps.p;
  HIPC(hipMemcpyAsync(mb + b_rb, cd.row_base, (ng + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
  HIPC(hipMemcpyAsync(mb + b_tb, cd.tr_base, (ng + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_sell_maps, dim3(grid_for(std::max<u64>(n_ecs, T), BLOCK)), dim3(BLOCK), 0, c->stream, A, B, (u32*)(mb + b_rf), (u32*)(mb + b_ms));
  HIPC(hipMemcpyAsync(mb + b_sa, A.single_all, T * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(c->stream));
  *dev = EmSellDev{(const u32*)(mb + b_rb), (const u32*)(mb + b_tb), B.rslice_base, B.cslice_base, B.rell_base, B.cell_base, B.rdesc, B.cdesc, B.rell, B.cell,
                   B.cw_new, B.single_new, B.eff_new};
  if (cache) { cache->row_final = (u32*)(mb + b_rf); cache->mslot = (u32*)(mb + b_ms); cache->single_all = (double*)(mb + b_sa); cache->d_eff = (double*)(mb + b_ef); }
  return 0;
}
}  // namespace
namespace kamdi { void sell_cache_free(SellCache* k) { delete k; } }
namespace {
// ---- hybrid plan: split the matrix by component size, the LDS form on what fits, the streamed layout on the rest ---------------------
__global__ void k_hy_comp_nnz(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs, const u32* __restrict__ label, u32* c_nnz) {
  const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 root = 0xFFFFFFFFu, len = 0;
  if (e < n_ecs) {
    const u64 a = ec_off[e], b = ec_off[e + 1];
    if (b - a >= 2) { root = label[ec_ids[a]]; len = (u32)(b - a); }
  }
  // the rows of one oversized component are tens of thousands of adds to ONE address (~12 ns each at the memory side): the lanes of a
  // wavefront that share the leader's component add once; the others (gene-sized components: all different) take the plain atomic
  const bool act = len != 0;
  const u64 am = __ballot(act);
  if (!am) return;
  const int leader = __ffsll((long long)am) - 1;
  const u32 lroot = (u32)__shfl((int)root, leader, 64);
  const bool with_leader = act && root == lroot;
  u32 v = with_leader ? len : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += (u32)__shfl_down((int)v, d, 64);
  if (lane_id() == 0 && v) atomicAdd(&c_nnz[lroot], v);
  if (act && !with_leader) atomicAdd(&c_nnz[root], len);
}
// a row (singleton rows included: they are the constant term of a transcript of that component) goes with its component
__global__ void k_hy_sizes(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, u64 n_ecs, const u32* __restrict__ label,
                           const u32* __restrict__ c_nnz, u32 lim, u32* flag_s, u32* len_s, u32* flag_g, u32* len_g) {
  const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ecs) return;
  const u64 a = ec_off[e], b = ec_off[e + 1];
  const bool any = b > a;
  const bool giant = any && c_nnz[label[ec_ids[a]]] > lim;
  flag_s[e] = any && !giant ? 1u : 0u; len_s[e] = any && !giant ? (u32)(b - a) : 0u;
  flag_g[e] = giant ? 1u : 0u; len_g[e] = giant ? (u32)(b - a) : 0u;
}
// the maps of a plan without groups (every multi-transcript row went to the oversized side): no slot, the singleton counts
__global__ void k_hy_single(const u64* __restrict__ ec_off, const u32* __restrict__ ec_ids, const u32* __restrict__ counts, u64 n_ecs, double* single_all) {
  const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ecs) return;
  const u64 a = ec_off[e];
  if (ec_off[e + 1] - a == 1) single_all[ec_ids[a]] = (double)counts[e];
}
// the oversized components' kernels run on a stream of their own beside k_em_sell's (no CU mask: reserving compute units for them was measured
// slower in round 5 -- the streamed side is the critical path and finds free units as soon as the groups' launch of a chunk has drained)
int hy_streams(kamd_ctx* c) {
  if (!c->hy_giant_stream) {
    HIPC(hipStreamCreateWithFlags(&c->hy_giant_stream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&c->hy_ev_giant, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&c->hy_ev_sell, hipEventDisableTiming));
  }
  return 0;
}
// the component-local plan of a sub-matrix, with a bounded search for the group size (the cut only helps while no single component is
// the problem).  0 = built, 1 = not applicable, < 0 = error
int em_sell_plan_search(kamd_ctx* c, const u64* d_ec_off, const u32* d_ec_ids, const u32* d_counts, const u32* d_wcounts, u64 n_ecs, u64 nnz,
                        const double* eff_lens, u64 T, u64 lds_budget, int n_cus, bool multi, SellCache& K, CompStats* cst) {
  u32 small = c->tune.em_small_nnz > 0 ? (u32)c->tune.em_small_nnz : 0u;
  // em_group_div < 0 (the default): as few groups per CU as the LDS allows -- one workgroup of 16 wavefronts per CU iterating a group
  // of up to ~10 000 entries (12 bytes of LDS per entry) beats two half-sized ones: every wavefront then owns a slice in both passes
  // of a round, and a round costs the same few LDS round trips whatever the size (measured on config #3, profiles/README.md round 4:
  // 505 groups 8.5 ms, 757 groups 9.6, 1009 groups 10.1).  The group count stays a little under a multiple of the CU count.
  const bool auto_div = c->tune.em_group_div < 0;
  const u64 ncu = (u64)std::max(1, n_cus);
  const u64 div0 = auto_div ? std::max<u64>(1, (nnz + ncu * 10000 - 1) / (ncu * 10000)) : (u64)c->tune.em_group_div;
  int prc = 1, tries = 0;
  for (u64 div = div0; div <= 1024 && prc >= 1; ) {
    const u64 target = std::max<u64>(1024, (nnz + ncu * div - 1) / (ncu * div));
    prc = em_sell_setup_device(c, d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, nnz, eff_lens, T, lds_budget, target, &K.P, &K.dev, &K, small, small,
                               24 * 1024, multi, cst);
    if (prc == 2) { small = 0; continue; }   // (same cut again, one size class)
    if (prc <= 0 || target == 1024) break;
    // a component that cannot fit a workgroup whatever the cut (kamd_em_sell.h group_bytes: 4 bytes of index per entry and direction pair,
    // 16 per row, 48 per transcript at the least; 16-bit local indices): stop at once -- every further cut repeats a full device set-up
    if (cst && ((u64)cst->max_nnz * 4 + (u64)cst->max_rows * 16 > lds_budget || cst->max_rows > 65000 || cst->max_tr > 65000)) break;
    if (++tries >= 5) break;
    div = auto_div && tries == 1 ? div + 1 : div * 2;   // one step of the fine search (two half-sized components at a group's end), then geometric
  }
  return prc == 2 ? 1 : prc;
}
// 0 = K holds the hybrid plan (K.hybrid, K.G), 1 = not applicable, < 0 = error
int em_hybrid_setup(kamd_ctx* c, const u64* d_ec_off, const u32* d_ec_ids, const u32* d_counts, const u32* d_wcounts, u64 n_ecs, u64 nnz,
                    const double* eff_lens, u64 T, u64 lds_budget, SellCache& K, bool first_try = false, bool host_maps = false) {
  if (n_ecs == 0 || nnz >= (1ULL << 32) || T >= 0xFFFFFFF0ULL || n_ecs >= 0xFFFFFFF0ULL) return 1;
  GiantPart& G = K.G;
  G.drop_graphs();
  // component labels and the entries of every component
  if (int rc = cc_labels(c, d_ec_off, d_ec_ids, n_ecs, T)) return rc;
  Carver cv;
  const size_t o_lab = cv.take(T * 4 + 8);   // the labels of the WHOLE matrix (the plan of the side that fits computes its own into pt_label)
  const size_t o_cn = cv.take(T * 4 + 8), o_fs = cv.take(n_ecs * 4 + 8), o_ls = cv.take(n_ecs * 4 + 8), o_fg = cv.take(n_ecs * 4 + 8), o_lg = cv.take(n_ecs * 4 + 8);
  const size_t o_rs = cv.take((n_ecs + 2) * 8), o_zs = cv.take((n_ecs + 2) * 8), o_rg = cv.take((n_ecs + 2) * 8), o_zg = cv.take((n_ecs + 2) * 8);
  // the two sub-matrices (between them every row of the matrix once)
  const size_t o_off_s = cv.take((n_ecs + 2) * 8), o_off_g = cv.take((n_ecs + 2) * 8), o_ids_s = cv.take(nnz * 4 + 8), o_ids_g = cv.take(nnz * 4 + 8);
  const size_t o_cnt_s = cv.take(n_ecs * 4 + 8), o_cnt_g = cv.take(n_ecs * 4 + 8), o_wc_s = cv.take(n_ecs * 4 + 8), o_wc_g = cv.take(n_ecs * 4 + 8);
  if (int rc = c->hy_sub.ensure(cv.off, 0, c->stream)) return rc;
  char* hb = (char*)c->hy_sub.p;
  u32* c_nnz = (u32*)(hb + o_cn);
  u32* flag_s = (u32*)(hb + o_fs); u32* len_s = (u32*)(hb + o_ls); u32* flag_g = (u32*)(hb + o_fg); u32* len_g = (u32*)(hb + o_lg);
  u64* rpos_s = (u64*)(hb + o_rs); u64* zpos_s = (u64*)(hb + o_zs); u64* rpos_g = (u64*)(hb + o_rg); u64* zpos_g = (u64*)(hb + o_zg);
  u64* off_s = (u64*)(hb + o_off_s); u64* off_g = (u64*)(hb + o_off_g); u32* ids_s = (u32*)(hb + o_ids_s); u32* ids_g = (u32*)(hb + o_ids_g);
  u32* cnt_s = (u32*)(hb + o_cnt_s); u32* cnt_g = (u32*)(hb + o_cnt_g); u32* wc_s = (u32*)(hb + o_wc_s); u32* wc_g = (u32*)(hb + o_wc_g);
  u32* label = (u32*)(hb + o_lab);
  HIPC(hipMemcpyAsync(label, c->pt_label.p, T * 4, hipMemcpyDeviceToDevice, c->stream));
  HIPC(hipMemsetAsync(c_nnz, 0, T * 4, c->stream));
  hipLaunchKernelGGL(k_hy_comp_nnz, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, d_ec_off, d_ec_ids, n_ecs, label, c_nnz);
  HIPC(hipGetLastError());
  {
    CompStats cs{};
    if (int rc = c->pt_hist.ensure(64, 0, c->stream)) return rc;
    HIPC(hipMemsetAsync(c->pt_hist.p, 0, sizeof(CompStats), c->stream));
    hipLaunchKernelGGL(k_comp_stats, dim3(std::min<unsigned>(grid_for(T, BLOCK), COMP_STATS_BLOCKS)), dim3(BLOCK), 0, c->stream, c_nnz, (const u32*)nullptr, (const u32*)nullptr, T, (u32*)c->pt_hist.p);
    HIPC(hipMemcpyAsync(&cs, c->pt_hist.p, sizeof(CompStats), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    c->last_em_max_comp_nnz = cs.max_nnz;
    if (cs.max_nnz == 0) return 1;
  }
  u32 lim = c->tune.em_giant_nnz > 0 ? (u32)c->tune.em_giant_nnz : 6000u;
  if (c->last_em_max_comp_nnz <= lim && first_try) return 1;   // (asked first because the last matrix needed it: this one has nothing oversized at the first limit)
  for (int attempt = 0; attempt < 4; attempt++, lim = std::max(lim / 2, 8u)) {
    hipLaunchKernelGGL(k_hy_sizes, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, d_ec_off, d_ec_ids, n_ecs, label, c_nnz, lim,
                       flag_s, len_s, flag_g, len_g);
    if (int rc = exclusive_scan(c, flag_s, n_ecs, rpos_s, rpos_s + n_ecs)) return rc;
    if (int rc = exclusive_scan(c, len_s, n_ecs, zpos_s, zpos_s + n_ecs)) return rc;
    if (int rc = exclusive_scan(c, flag_g, n_ecs, rpos_g, rpos_g + n_ecs)) return rc;
    if (int rc = exclusive_scan(c, len_g, n_ecs, zpos_g, zpos_g + n_ecs)) return rc;
    u64 tot[4] = {0, 0, 0, 0};   // rows / entries of the side that fits, of the oversized side
    HIPC(hipMemcpyAsync(&tot[0], rpos_s + n_ecs, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(&tot[1], zpos_s + n_ecs, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(&tot[2], rpos_g + n_ecs, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(&tot[3], zpos_g + n_ecs, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    const u64 n_s = tot[0], nnz_s = tot[1], n_g = tot[2], nnz_g = tot[3];
    if (n_g == 0) continue;   // nothing above this limit, and the whole matrix did not fit: a lower limit
    hipLaunchKernelGGL(k_part_copy, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, flag_s, rpos_s, zpos_s,
                       off_s, ids_s, cnt_s, wc_s);
    hipLaunchKernelGGL(k_part_copy, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, flag_g, rpos_g, zpos_g,
                       off_g, ids_g, cnt_g, wc_g);
    HIPC(hipMemcpyAsync(off_s + n_s, zpos_s + n_ecs, 8, hipMemcpyDeviceToDevice, c->stream));
    HIPC(hipMemcpyAsync(off_g + n_g, zpos_g + n_ecs, 8, hipMemcpyDeviceToDevice, c->stream));
    HIPC(hipGetLastError());
    if (int rc = hy_streams(c)) return rc;
    const int sell_cus = c->n_cus;
    // the side that fits: the component-local plan over its rows (groups sized for the compute units it gets)
    CompStats cst{};
    c->labels_override = label;   // (the components of the side that fits are components of the whole matrix)
    int prc = n_s ? em_sell_plan_search(c, off_s, ids_s, cnt_s, wc_s, n_s, nnz_s, eff_lens, T, lds_budget, sell_cus, host_maps, K, &cst) : 1;
    c->labels_override = nullptr;
    if (prc < 0) return prc;
    if (prc == 1 && cst.max_nnz == 0) {
      // no row with two transcripts on that side: a plan without groups -- its transcripts keep their singleton counts
      K.P = kamd_em_sell::Plan{};
      K.P.T = T; K.P.row_base.assign(1, 0); K.P.tr_base.assign(1, 0);
      K.dev = EmSellDev{};
      Carver mv;
      const size_t m_ms = mv.take(T * 4 + 8), m_sa = mv.take(T * 8 + 8);
      if (int rc = c->hy_maps.ensure(mv.off, 0, c->stream)) return rc;
      char* mb = (char*)c->hy_maps.p;
      HIPC(hipMemsetAsync(mb + m_ms, 0xFF, T * 4, c->stream));
      HIPC(hipMemsetAsync(mb + m_sa, 0, T * 8, c->stream));
      if (n_s) hipLaunchKernelGGL(k_hy_single, dim3(grid_for(n_s, BLOCK)), dim3(BLOCK), 0, c->stream, off_s, ids_s, cnt_s, n_s, (double*)(mb + m_sa));
      HIPC(hipGetLastError());
      K.mslot = (u32*)(mb + m_ms); K.single_all = (double*)(mb + m_sa); K.row_final = nullptr; K.d_eff = nullptr;
      if (host_maps) {   // (several ranks scatter on the host: the singleton counts there too)
        K.P.single_all.resize(T);
        HIPC(hipMemcpyAsync(K.P.single_all.data(), K.single_all, T * 8, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
      }
      prc = 0;
    }
    if (prc == 1) continue;   // some component under the limit still does not fit its group: a lower limit
    // the oversized side: the streamed form's layout (kept rows, m-space, flagged entry streams) in arenas of its own
    for (DBuf* b : {&c->em_eff, &c->em_single}) if (int rc = b->ensure(T * sizeof(double), 0, c->stream)) return rc;
    if (int rc = c->em_colcnt.ensure(3 * (T + 1) * sizeof(u32), 0, c->stream)) return rc;
    if (int rc = c->em_coloff.ensure((T + 2) * sizeof(u64), 0, c->stream)) return rc;
    u32* col_cnt = c->em_colcnt.as<u32>();
    u32* col_fill = col_cnt + (T + 1);
    HIPC(hipMemcpyAsync(c->em_eff.p, eff_lens, T * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPC(hipMemsetAsync(col_cnt, 0, 2 * (T + 1) * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(c->em_single.p, 0, T * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_em_prepare, dim3(grid_for(n_g, BLOCK)), dim3(BLOCK), 0, c->stream, off_g, ids_g, cnt_g, n_g, col_cnt, c->em_single.as<double>());
    if (int rc = exclusive_scan(c, col_cnt, T, c->em_coloff.as<u64>(), c->em_coloff.as<u64>() + T)) return rc;
    G.plan = PmPlan{};
    const int src = em_streamed_setup(c, off_g, ids_g, cnt_g, wc_g, n_g, T, col_cnt, col_fill, &G.plan, &c->hy_a, &c->hy_b);
    if (src < 0) return src;
    if (src != 0) return 1;
    const PmArgs& A = G.plan.args;
    const u64 M1 = (u64)A.M + 1;
    Carver xv;
    size_t o_v[5];
    for (int j = 0; j < 5; j++) o_v[j] = xv.take(M1 * 8);
    const size_t o_desc = xv.take(sizeof(GiDesc));
    if (int rc = c->hy_x.ensure(xv.off, 0, c->stream)) return rc;
    char* xb = (char*)c->hy_x.p;
    G.G_al[0] = A.alpha0; G.G_al[1] = A.alpha1; G.G_a[0] = A.a0; G.G_a[1] = A.a1;   // (k_pm_minit: alpha0 = 1 / T, a0 = alpha0 / eff; the sentinels [M] = 0)
    G.S_al[0] = (double*)(xb + o_v[0]); G.S_al[1] = (double*)(xb + o_v[1]); G.S_a[0] = (double*)(xb + o_v[2]); G.S_a[1] = (double*)(xb + o_v[3]);
    G.ac = (double*)(xb + o_v[4]); G.desc = (GiDesc*)(xb + o_desc);
    hipLaunchKernelGGL(k_gi_zero_tail, dim3(1), dim3(64), 0, c->stream, G.S_al[0], G.S_al[1], G.S_a[0], G.S_a[1], G.ac, A.M);
    HIPC(hipGetLastError());
    G.gb.valid = false;
    if (c->tune.em_blocked != 2) { const int brc = gb_setup(c, G); if (brc < 0) return brc; }
    c->last_em_giant_pieces = G.gb.valid ? (u64)G.gb.args.rows.n_pieces + G.gb.args.cols.n_pieces : 0;
    G.stream = c->hy_giant_stream; G.ev = c->hy_ev_giant; G.use_graph = c->tune.em_graph != 2;
    G.nnz = nnz_g; G.rows = A.R;
    K.hybrid = true;
    K.hy.flag_s = flag_s; K.hy.flag_g = flag_g; K.hy.rpos_s = rpos_s; K.hy.rpos_g = rpos_g; K.hy.off_s = off_s; K.hy.off_g = off_g; K.hy.ids_s = ids_s; K.hy.ids_g = ids_g;
    K.hy.cnt_s = cnt_s; K.hy.cnt_g = cnt_g; K.hy.wc_s = wc_s; K.hy.wc_g = wc_g; K.hy.n_s = n_s; K.hy.n_g = n_g; K.hy.no_groups = K.row_final == nullptr;
    c->last_em_giant_nnz = c->last_em_nnz_multi; c->last_em_giant_rows = A.R; c->last_em_giant_tr = A.M; c->last_em_giant_chunks = G.plan.n_chunks;
    return 0;
  }
  return 1;
}
// ---- a cached hybrid plan takes new counts (bootstrap replicates: Bootstrap::run_em, src/Bootstrap.cpp:4-14 -- same matrix, resampled counts) ----
__global__ void k_hy_counts(const u32* __restrict__ counts, const u32* __restrict__ wcounts, u64 n_ecs, const u32* __restrict__ row_flag, const u64* __restrict__ row_pos,
                            u32* out_counts, u32* out_wcounts) {
  const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ecs || !row_flag[e]) return;
  out_counts[row_pos[e]] = counts[e]; out_wcounts[row_pos[e]] = wcounts[e];
}
// the oversized side: count words of its kept rows, singleton counts of its transcripts (transcript space)
__global__ void k_gi_refresh(const u64* __restrict__ off, const u32* __restrict__ ids, const u32* __restrict__ counts, const u32* __restrict__ wcounts, u64 n,
                             const u64* __restrict__ rpos, u64* cw, double* single_t) {
  const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const u64 a = off[e], len = off[e + 1] - a;
  if (len == 1) single_t[ids[a]] = (double)counts[e];
  else if (len >= 2) cw[rpos[e]] = (u64)counts[e] | ((u64)wcounts[e] << 32);
}
int em_hybrid_refresh(kamd_ctx* c, SellCache& K, const u32* d_counts, const u32* d_wcounts, u64 n_ecs, const double* eff_lens, u64 T) {
  const SellCache::HyKeep& H = K.hy;
  GiantPart& G = K.G;
  const PmArgs& A = G.plan.args;
  hipLaunchKernelGGL(k_hy_counts, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, d_counts, d_wcounts, n_ecs, H.flag_s, H.rpos_s, H.cnt_s, H.wc_s);
  hipLaunchKernelGGL(k_hy_counts, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, d_counts, d_wcounts, n_ecs, H.flag_g, H.rpos_g, H.cnt_g, H.wc_g);
  // the side that fits
  HIPC(hipMemsetAsync(K.single_all, 0, T * 8, c->stream));
  if (H.no_groups) { if (H.n_s) hipLaunchKernelGGL(k_hy_single, dim3(grid_for(H.n_s, BLOCK)), dim3(BLOCK), 0, c->stream, H.off_s, H.ids_s, (const u32*)H.cnt_s, H.n_s, K.single_all); }
  else {
    HIPC(hipMemcpyAsync(K.d_eff, eff_lens, T * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_sell_refresh, dim3(grid_for(std::max<u64>(H.n_s, T), BLOCK)), dim3(BLOCK), 0, c->stream, H.off_s, H.ids_s, (const u32*)H.cnt_s, (const u32*)H.wc_s,
                       H.n_s, K.d_eff, T, K.row_final, K.mslot, const_cast<u64*>(K.dev.cw), const_cast<double*>(K.dev.single), const_cast<double*>(K.dev.eff), K.single_all);
  }
  // the oversized side: count words, singleton counts and effective lengths in m-space, the vectors back at alpha = 1 / T
  for (DBuf* b : {&c->em_eff, &c->em_single}) if (int rc = b->ensure(T * sizeof(double), 0, c->stream)) return rc;
  HIPC(hipMemcpyAsync(c->em_eff.p, eff_lens, T * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemsetAsync(c->em_single.p, 0, T * sizeof(double), c->stream));
  hipLaunchKernelGGL(k_gi_refresh, dim3(grid_for(H.n_g, BLOCK)), dim3(BLOCK), 0, c->stream, H.off_g, H.ids_g, (const u32*)H.cnt_g, (const u32*)H.wc_g, H.n_g, G.plan.rpos,
                     const_cast<u64*>(A.cw), c->em_single.as<double>());
  hipLaunchKernelGGL(k_pm_minit, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, T, G.plan.mflag, G.plan.mpos, (const u64*)nullptr, (const double*)c->em_single.as<double>(),
                     (const double*)c->em_eff.as<double>(), (u64)A.M, (u64*)nullptr, const_cast<double*>(A.single), const_cast<double*>(A.eff), A.alpha0, A.alpha1, A.a0, A.a1, A.ac0, A.ac1);
  HIPC(hipGetLastError());
  return 0;
}
// part (several ranks, each with the rows of the components it owns): the per-round change counts of a chunk are summed over the
// ranks on the device before the host looks at them (the only coupling between components is the stop rule), and "this form does
// not apply" is agreed on by all ranks, so that every rank issues the same collectives.
int em_sell_run_device(kamd_ctx* c, const u64* d_ec_off, const u32* d_ec_ids, const u32* d_counts, const u32* d_wcounts, u64 n_ecs, u64 nnz,
                       const double* eff_lens, u64 T, int n_iter, int min_rounds, double* alpha, double* abz, int32_t* rounds,
                       const EmPartition* part = nullptr) {
  const bool multi = part && part->world > 1 && part->cb;
  if (c->n_cus == 0) { int v = 0; HIPC(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device)); c->n_cus = v > 0 ? v : 256; }
  if (!c->sell_cache) c->sell_cache = new SellCache;
  SellCache& K = *c->sell_cache;
  HIPC(hipEventRecord(c->ev0, c->stream));
  // the plan of the context's own finalized matrix is kept: a bootstrap replicate (same matrix, other counts) only refreshes
  // the count words, the singleton counts and the effective lengths
  const bool own = c->finalized && d_ec_off == (const u64*)c->result.d_ec_off && d_ec_ids == c->result.d_ec_ids;
  const bool hit = own && K.valid && K.d_ec_off == d_ec_off && K.d_ec_ids == d_ec_ids && K.n_ecs == n_ecs && K.nnz == nnz && K.T == T &&
                   K.generation == c->ec_generation && K.split_len == c->tune.em_split_len && K.group_div == c->tune.em_group_div &&
                   K.small_nnz == c->tune.em_small_nnz && (K.host_maps || !multi) &&
                   (!K.hybrid || (!multi && K.giant_nnz_t == c->tune.em_giant_nnz && K.blocked_t == c->tune.em_blocked && K.hybrid_t == c->tune.em_hybrid));
  if (hit && K.hybrid) {
    c->last_em_plan_ms = 0.f;
    if (int rc = em_hybrid_refresh(c, K, d_counts, d_wcounts, n_ecs, eff_lens, T)) return rc;
  } else if (hit) {
    c->last_em_plan_ms = 0.f;
    HIPC(hipMemcpyAsync(K.d_eff, eff_lens, T * 8, hipMemcpyHostToDevice, c->stream));
    HIPC(hipMemsetAsync(K.single_all, 0, T * 8, c->stream));
    hipLaunchKernelGGL(k_sell_refresh, dim3(grid_for(std::max<u64>(n_ecs, T), BLOCK)), dim3(BLOCK), 0, c->stream, d_ec_off, d_ec_ids, d_counts, d_wcounts,
                       n_ecs, K.d_eff, T, K.row_final, K.mslot, const_cast<u64*>(K.dev.cw), const_cast<double*>(K.dev.single), const_cast<double*>(K.dev.eff),
                       K.single_all);
    HIPC(hipGetLastError());
    if (multi) {
      K.P.single_all.resize(T);
      HIPC(hipMemcpyAsync(K.P.single_all.data(), K.single_all, T * 8, hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
    }
  } else {
    const auto plan_t0 = std::chrono::steady_clock::now();
    K.valid = false; K.hybrid = false;
    c->last_em_giant_nnz = 0; c->last_em_giant_rows = 0; c->last_em_giant_tr = 0; c->last_em_giant_chunks = 0; c->last_em_giant_pieces = 0;
    // groups of nnz / (CUs x div) entries; a group must fit a workgroup's LDS (components are not split: if one does not fit, the
    // cut is refined a few times; a single component beyond the CU's 160 KB sends the oversized components to the streamed kernels
    // beside the groups -- the hybrid, em_hybrid_setup -- or, with several ranks, the whole matrix to the streamed form)
    const u64 lds_budget = 160 * 1024 - 2048;
    int prc = 1;
    if (multi && n_ecs == 0) {   // a rank that owns no component still takes part in the collectives: an empty plan
      K.P = kamd_em_sell::Plan{};
      K.P.T = T; K.P.row_base.assign(1, 0); K.P.tr_base.assign(1, 0); K.P.single_all.assign(T, 0.0);
      K.dev = EmSellDev{};
      prc = 0;
    } else {
      // (several ranks: the rank that owns an oversized component runs the hybrid on its rows like one rank would; kamd_em_local::run drives it)
      const bool may_hybrid = c->tune.em_hybrid != 2 && n_ecs > 0;
      if (may_hybrid && c->em_prefer_hybrid) {   // the context's last matrix needed the hybrid (bootstrap replicates, the steps of a bench): start there
        prc = em_hybrid_setup(c, d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, nnz, eff_lens, T, lds_budget, K, true, multi);
        if (prc < 0) return prc;
        if (prc == 1) { K.hybrid = false; c->em_prefer_hybrid = false; }
      }
      if (prc == 1) {
        CompStats cst{};
        prc = em_sell_plan_search(c, d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, nnz, eff_lens, T, lds_budget, c->n_cus, multi, K, &cst);
        c->last_em_max_comp_nnz = cst.max_nnz;
        if (prc == 1 && may_hybrid && cst.max_nnz > 0) {
          prc = em_hybrid_setup(c, d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, nnz, eff_lens, T, lds_budget, K, false, multi);
          if (prc == 1) K.hybrid = false;
          else if (prc == 0) c->em_prefer_hybrid = true;
        }
      }
    }
    if (prc < 0) return prc;
    bool not_applicable = prc == 1 || (!multi && K.P.n_groups == 0 && !K.hybrid);
    if (multi) {
      int flag = not_applicable ? 1 : 0;
      if (int rc = c->pt_hist.ensure(64, 0, c->stream)) return rc;
      HIPC(hipMemcpyAsync(c->pt_hist.p, &flag, sizeof(int), hipMemcpyHostToDevice, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      if (int rc = part->cb(part->user, (int32_t*)c->pt_hist.p, 1)) return kamd::fail(-103, "kamd_em_run_partitioned: the sum callback failed (" + std::to_string(rc) + ")");
      HIPC(hipMemcpyAsync(&flag, c->pt_hist.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      not_applicable = flag != 0;
    }
    if (not_applicable) return 1;
    c->last_em_plan_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - plan_t0).count();
    if (own) {
      K.giant_nnz_t = c->tune.em_giant_nnz; K.blocked_t = c->tune.em_blocked; K.hybrid_t = c->tune.em_hybrid;
      K.valid = true; K.d_ec_off = d_ec_off; K.d_ec_ids = d_ec_ids; K.n_ecs = n_ecs; K.nnz = nnz; K.T = T; K.generation = c->ec_generation;
      K.split_len = c->tune.em_split_len; K.group_div = c->tune.em_group_div; K.small_nnz = c->tune.em_small_nnz; K.host_maps = multi;
    }
  }
  const kamd_em_sell::Plan& P = K.P;
  const int chunk = EML_MAX_ROUNDS;
  EmSellGpu B(c, P);
  B.dev = K.dev; B.M = P.tr_base[P.n_groups]; B.block = c->tune.em_local_block; B.part = multi ? part : nullptr;
  if (K.hybrid) { B.gi = &K.G; B.ev_sell = c->hy_ev_sell; }
  const int n_chunks = std::max(1, (n_iter + chunk - 1) / chunk);
  if (int rc = B.setup(multi ? chunk : n_chunks * chunk, K.dev.eff, multi ? 0 : T)) return rc;
  int r = 0;
  if (multi) {
    r = kamd_em_local::run(B, P, n_iter, min_rounds, chunk, alpha, abz);   // the change counts of a chunk are summed over the ranks before anyone reads them
    if (K.hybrid && !B.err) {
      // the oversized components' transcripts (kamd_em_local::run scattered the groups' and left these at their singleton counts)
      const PmPlan& GP = K.G.plan;
      std::vector<u32> mflag(T); std::vector<u64> mpos(T);
      HIPC(hipMemcpyAsync(mflag.data(), GP.mflag, T * 4, hipMemcpyDeviceToHost, c->stream));
      HIPC(hipMemcpyAsync(mpos.data(), GP.mpos, T * 8, hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      const bool have_final = B.host_reads >= 2;   // (the state before the final round was read, then the final one)
      for (u64 t = 0; t < T; t++) if (mflag[t]) {
        alpha[t] = B.h_gi_cur[mpos[t]];
        if (abz) abz[t] = have_final ? B.h_gi_prev[mpos[t]] : 0.0;
      }
    }
  }
  else if (int rc = em_sell_drive_async(c, B, K, T, n_iter, min_rounds, alpha, abz, &r)) return rc;
  if (B.err) return kamd::fail(B.err, "kamd_em_run: the component-local EM failed on the device");
  HIPC(hipEventRecord(c->ev1, c->stream));
  HIPC(hipEventSynchronize(c->ev1));
  HIPC(hipEventElapsedTime(&c->last_em_ms, c->ev0, c->ev1));
  c->last_em_iters = (uint64_t)r + (r < n_iter ? 1 : 0);
  c->last_em_nnz = nnz; c->last_em_k = -2; c->last_em_grid = P.n_groups; c->last_em_necs = n_ecs;
  c->last_em_lds = (uint32_t)P.max_group_bytes;
  c->last_em_plan_cached = hit ? 1 : 0;
  c->last_em_graph_fallback = (K.hybrid && c->tune.em_graph != 2 && !K.G.use_graph) ? 1 : 0;
  if (rounds) *rounds = r;
  return 0;
}
int em_run_impl(kamd_ctx* c, const uint64_t* d_ec_off, const uint32_t* d_ec_ids, const uint32_t* d_counts,
                const uint32_t* d_weight_counts, uint64_t n_ecs, const double* eff_lens, uint64_t n_targets, uint32_t n_iter,
                uint32_t min_rounds, double* alpha, double* alpha_before_zeroes, int32_t* rounds, const EmPartition& part) {
  if (!c || !eff_lens || !alpha) return kamd::fail(-1, "kamd_em_run: null argument");
  HIPC(hipSetDevice(c->device));
  if (!d_ec_off) {
    if (!c->finalized) return kamd::fail(-1, "kamd_em_run: no EC result (call kamd_ec_finalize or pass a CSR)");
    d_ec_off = c->result.d_ec_off; d_ec_ids = c->result.d_ec_ids; n_ecs = c->result.n_ecs;
    if (!d_counts) d_counts = c->result.d_counts;
    if (!d_weight_counts) d_weight_counts = c->result.d_counts;
  }
  const u32* d_wcounts = d_weight_counts ? d_weight_counts : d_counts;
  const u64 T = n_targets;
  if (T == 0) return kamd::fail(-1, "kamd_em_run: no targets");
  const bool spec = part.world > 1;
  if (spec && n_ecs) {
    // connected components of the transcript/EC graph, then the rows of the components this rank owns as a compact CSR
    u64 nnz_all = 0;
    HIPC(hipMemcpyAsync(&nnz_all, (const u64*)d_ec_off + n_ecs, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (int rc = c->pt_label.ensure((T + 1) * sizeof(u32), 0, c->stream)) return rc;
    if (int rc = c->pt_hist.ensure(64, 0, c->stream)) return rc;
    if (int rc = cc_labels(c, (const u64*)d_ec_off, d_ec_ids, (u64)n_ecs, T)) return rc;
    for (DBuf* b : {&c->pt_flag, &c->pt_len}) if (int rc = b->ensure((n_ecs + 1) * sizeof(u32), 0, c->stream)) return rc;
    for (DBuf* b : {&c->pt_rowpos, &c->pt_nnzpos}) if (int rc = b->ensure((n_ecs + 2) * sizeof(u64), 0, c->stream)) return rc;
    hipLaunchKernelGGL(k_part_sizes, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)d_ec_off, d_ec_ids, (u64)n_ecs,
                       c->pt_label.as<u32>(), part.rank, part.world, c->pt_flag.as<u32>(), c->pt_len.as<u32>());
    if (int rc = exclusive_scan(c, c->pt_flag.as<u32>(), n_ecs, c->pt_rowpos.as<u64>(), c->pt_rowpos.as<u64>() + n_ecs)) return rc;
    if (int rc = exclusive_scan(c, c->pt_len.as<u32>(), n_ecs, c->pt_nnzpos.as<u64>(), c->pt_nnzpos.as<u64>() + n_ecs)) return rc;
    u64 n_local = 0, nnz_local = 0;
    HIPC(hipMemcpyAsync(&n_local, c->pt_rowpos.as<u64>() + n_ecs, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(&nnz_local, c->pt_nnzpos.as<u64>() + n_ecs, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (int rc = c->pt_off.ensure((n_local + 2) * sizeof(u64), 0, c->stream)) return rc;
    if (int rc = c->pt_ids.ensure((nnz_local + 1) * sizeof(u32), 0, c->stream)) return rc;
    if (int rc = c->pt_counts.ensure((n_local + 1) * sizeof(u32), 0, c->stream)) return rc;
    if (int rc = c->pt_wcounts.ensure((n_local + 1) * sizeof(u32), 0, c->stream)) return rc;
    hipLaunchKernelGGL(k_part_copy, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)d_ec_off, d_ec_ids, d_counts,
                       d_wcounts, (u64)n_ecs, c->pt_flag.as<u32>(), c->pt_rowpos.as<u64>(), c->pt_nnzpos.as<u64>(), c->pt_off.as<u64>(),
                       c->pt_ids.as<u32>(), c->pt_counts.as<u32>(), c->pt_wcounts.as<u32>());
    HIPC(hipMemcpyAsync(c->pt_off.as<u64>() + n_local, &nnz_local, sizeof(u64), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));  // nnz_local is a stack variable
    HIPC(hipGetLastError());
    d_ec_off = c->pt_off.as<uint64_t>(); d_ec_ids = c->pt_ids.as<u32>(); d_counts = c->pt_counts.as<u32>(); d_wcounts = c->pt_wcounts.as<u32>();
    n_ecs = n_local;
    (void)nnz_all;
  }
  u64 nnz = 0;
  if (n_ecs) {
    if (c->finalized && d_ec_off == c->result.d_ec_off && n_ecs == c->result.n_ecs) nnz = c->result.nnz;   // (the context's own result: known since kamd_ec_finalize)
    else {
      HIPC(hipMemcpyAsync(&nnz, (const u64*)d_ec_off + n_ecs, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
    }
  }
  if (n_ecs || spec) {   // the component-local form (kamd_em_local.h); over several ranks only its sliced-ELLPACK kernel
    if (c->tune.em_form == 3) {
      const int rc = em_sell_run_device(c, (const u64*)d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, nnz, eff_lens, T, (int)n_iter, (int)min_rounds,
                                        alpha, alpha_before_zeroes, rounds, spec ? &part : nullptr);
      if (rc <= 0) return rc;   // 1 = not applicable (a component does not fit a workgroup): the streamed form takes over
    }
  }
  for (DBuf* b : {&c->em_alpha, &c->em_next, &c->em_eff, &c->em_a0, &c->em_a1, &c->em_single})
    if (int rc = b->ensure(T * sizeof(double), 0, c->stream)) return rc;
  if (int rc = c->em_state.ensure(2 * sizeof(EmState), 0, c->stream)) return rc;
  if (int rc = c->em_colrow.ensure((nnz + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->em_cn.ensure((n_ecs + 1) * sizeof(double), 0, c->stream)) return rc;
  if (int rc = c->em_colcnt.ensure(3 * (T + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->em_segoff.ensure((T + 2) * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->em_coloff.ensure((T + 2) * sizeof(u64), 0, c->stream)) return rc;
  u32* col_cnt = c->em_colcnt.as<u32>();
  u32* col_fill = col_cnt + (T + 1);
  HIPC(hipMemcpyAsync(c->em_eff.p, eff_lens, T * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemsetAsync(col_cnt, 0, 2 * (T + 1) * sizeof(u32), c->stream));
  HIPC(hipMemsetAsync(c->em_single.p, 0, T * sizeof(double), c->stream));
  // the record "before round 0" lives in slot 1 (round 0 has parity 0): iter -1 with a non-zero change count
  EmState st_init[2]; memset(st_init, 0, sizeof st_init);
  st_init[1].iter = -1; st_init[1].chcount = 1;
  HIPC(hipMemcpyAsync(c->em_state.p, st_init, sizeof st_init, hipMemcpyHostToDevice, c->stream));
  HIPC(hipStreamSynchronize(c->stream));  // st_init is a stack buffer
  HIPC(hipEventRecord(c->ev0, c->stream));
  // transposed (transcript-major) structure of the multi-transcript rows, built once per run
  if (n_ecs) hipLaunchKernelGGL(k_em_prepare, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)d_ec_off, d_ec_ids,
                                d_counts, (u64)n_ecs, col_cnt, c->em_single.as<double>());
  if (int rc = exclusive_scan(c, col_cnt, T, c->em_coloff.as<u64>(), c->em_coloff.as<u64>() + T)) return rc;
  EmState hs{};
  c->last_em_nnz = nnz; c->last_em_k = 0; c->last_em_grid = 0;
  // the streamed form (two launches per round over the re-laid-out matrix) unless KAMD_EM_STREAMED=0 asks for the CSR form
  PmPlan plan;
  bool streamed = false;
  {
    if (c->tune.em_form != 2) {
      const int rc = em_streamed_setup(c, (const u64*)d_ec_off, d_ec_ids, d_counts, d_wcounts, n_ecs, T, col_cnt, col_fill, &plan);
      if (rc < 0) return rc;
      streamed = rc == 0;
      if (!streamed) HIPC(hipMemsetAsync(col_fill, 0, (T + 1) * sizeof(u32), c->stream));
      plan.args.n_iter = (int)n_iter; plan.args.min_rounds = (int)min_rounds;
    }
  }
  u64 n_active = 0, n_seg = 0;
  if (!streamed) {
  if (n_ecs) hipLaunchKernelGGL(k_em_transpose, dim3(grid_for(n_ecs, BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)d_ec_off, d_ec_ids,
                                (u64)n_ecs, c->em_coloff.as<u64>(), col_fill, c->em_colrow.as<u32>());
  // work list of k_em_final: transcripts that occur in some EC
  if (int rc = c->em_actflag.ensure((T + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->em_actpos.ensure((T + 2) * sizeof(u64), 0, c->stream)) return rc;
  if (int rc = c->em_active.ensure((T + 1) * sizeof(u32), 0, c->stream)) return rc;
  hipLaunchKernelGGL(k_em_active, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, T, col_cnt, c->em_single.as<double>(),
                     c->em_actflag.as<u32>());
  if (int rc = exclusive_scan(c, c->em_actflag.as<u32>(), T, c->em_actpos.as<u64>(), c->em_actpos.as<u64>() + T)) return rc;
  hipLaunchKernelGGL(k_em_init, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, T, c->em_eff.as<double>(), c->em_actflag.as<u32>(),
                     c->em_actpos.as<u64>(), c->em_alpha.as<double>(), c->em_next.as<double>(), c->em_a0.as<double>(),
                     c->em_a1.as<double>(), c->em_active.as<u32>());
  HIPC(hipMemcpyAsync(&n_active, c->em_actpos.as<u64>() + T, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  // column segments
  u32* nseg = col_cnt + 2 * (T + 1);
  hipLaunchKernelGGL(k_em_nseg, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, col_cnt, T, nseg);
  if (int rc = exclusive_scan(c, nseg, T, c->em_segoff.as<u64>(), c->em_segoff.as<u64>() + T)) return rc;
  HIPC(hipMemcpyAsync(&n_seg, c->em_segoff.as<u64>() + T, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  u64 nnz_multi = 0;
  HIPC(hipMemcpy(&nnz_multi, c->em_coloff.as<u64>() + T, sizeof(u64), hipMemcpyDeviceToHost));
  c->last_em_nnz = nnz; c->last_em_nnz_multi = nnz_multi; c->last_em_nseg = n_seg; c->last_em_necs = n_ecs;
  if (int rc = c->em_segt.ensure((n_seg + 1) * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->em_partial.ensure((n_seg + 1) * sizeof(double), 0, c->stream)) return rc;
  hipLaunchKernelGGL(k_em_segsetup, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, c->em_coloff.as<u64>(), c->em_segoff.as<u64>(), T,
                     c->em_segt.as<u32>());
  HIPC(hipGetLastError());
  }  // !streamed
  const int chunk = 64;
  int row_lanes = 4;
  row_lanes = c->tune.em_row_lanes;
  const unsigned grid_rows = grid_for(std::max<u64>(n_ecs, 1) * row_lanes, BLOCK);
  const unsigned grid_seg = grid_for(std::max<u64>(n_seg, 1) * EM_SEG_LANES, BLOCK);
  unsigned fin_cap = 1024;
  fin_cap = (unsigned)c->tune.em_fin_blocks;
  const unsigned grid_fin = (unsigned)std::min<u64>(grid_for(std::max<u64>(n_active, 1) * EM_FIN_LANES, BLOCK), fin_cap);
  // The launch-bound inner loop is captured once as a hipGraph of `chunk` rounds (4 kernels each) on a private stream
  // and replayed until the device-side state says done; all loop state lives in device memory, so every replay is
  // the same graph.  KAMD_EM_GRAPH=0 falls back to plain launches.
  int* spec_hist = nullptr;
  if (spec) {
    if (int rc = c->pt_hist.ensure(((size_t)n_iter + 2 * chunk + 8) * sizeof(int), 0, c->stream)) return rc;
    HIPC(hipMemsetAsync(c->pt_hist.p, 0, ((size_t)n_iter + 2 * chunk + 8) * sizeof(int), c->stream));
    spec_hist = (int*)c->pt_hist.p;
  }
  plan.args.spec_hist = spec_hist;
  // state that a rewind of the partitioned run has to restore: {buffer of even rounds, buffer of odd rounds, doubles}
  struct CkSet { double* buf[2]; size_t n; };
  std::vector<CkSet> ck;
  if (streamed) {
    const size_t n = (size_t)plan.args.M + 1;
    ck = {{{plan.args.alpha0, plan.args.alpha1}, n}, {{plan.args.a0, plan.args.a1}, n}, {{plan.args.ac0, plan.args.ac1}, n}};
  } else {
    ck = {{{c->em_alpha.as<double>(), c->em_next.as<double>()}, (size_t)T}, {{c->em_a0.as<double>(), c->em_a1.as<double>()}, (size_t)T}};
  }
  if (spec) {
    size_t tot = 0;
    for (const CkSet& k : ck) tot += k.n;
    if (int rc = c->pt_ck_alpha.ensure(tot * sizeof(double), 0, c->stream)) return rc;
  }
  auto checkpoint = [&](int par, bool restore) -> int {
    double* store = c->pt_ck_alpha.as<double>();
    for (const CkSet& k : ck) {
      if (restore) HIPC(hipMemcpyAsync(k.buf[par], store, k.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      else HIPC(hipMemcpyAsync(store, k.buf[par], k.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      store += k.n;
    }
    return 0;
  };
  int parity = 0;  // parity of the next round to enqueue (round r uses record r & 1 and reads record (r & 1) ^ 1)
  auto enqueue_rounds = [&](hipStream_t s, int n_rounds) {
    for (int it = 0; it < n_rounds; it++, parity ^= 1) {
      if (streamed) { pm_enqueue_round(plan, s, parity); continue; }
#define KAMD_LAUNCH_ROWS(L)                                                                                                        \
  hipLaunchKernelGGL(k_em_rows<L>, dim3(grid_rows), dim3(BLOCK), 0, s, (const u64*)d_ec_off, d_ec_ids, d_counts, d_wcounts, (u64)n_ecs, \
                     c->em_alpha.as<double>(), c->em_next.as<double>(), c->em_a0.as<double>(), c->em_a1.as<double>(),              \
                     c->em_cn.as<double>(), (EmState*)c->em_state.p, parity, (int)n_iter, (int)min_rounds, spec_hist)
      if (row_lanes == 8) KAMD_LAUNCH_ROWS(8); else if (row_lanes == 2) KAMD_LAUNCH_ROWS(2); else KAMD_LAUNCH_ROWS(4);
#undef KAMD_LAUNCH_ROWS
      hipLaunchKernelGGL(k_em_seg, dim3(grid_seg), dim3(BLOCK), 0, s, c->em_coloff.as<u64>(), c->em_segoff.as<u64>(),
                         c->em_segt.as<u32>(), n_seg, c->em_colrow.as<u32>(), c->em_cn.as<double>(), c->em_partial.as<double>(),
                         (const EmState*)c->em_state.p, parity, (int)n_iter, (int)min_rounds, spec ? 1 : 0);
      hipLaunchKernelGGL(k_em_final, dim3(grid_fin), dim3(BLOCK), 0, s, c->em_segoff.as<u64>(), c->em_partial.as<double>(),
                         c->em_single.as<double>(), c->em_eff.as<double>(), c->em_active.as<u32>(), n_active, c->em_alpha.as<double>(),
                         c->em_next.as<double>(),
                         c->em_a0.as<double>(), c->em_a1.as<double>(), (EmState*)c->em_state.p, parity, (int)n_iter, (int)min_rounds,
                         spec ? 1 : 0);
    }
  };
  // state after the rounds enqueued so far: apply the loop rule once more to the last record (the round that notices
  // "the previous round was final" has not run yet)
  EmState recs[2];
  auto read_state = [&](hipStream_t s, EmNow* now) -> int {
    HIPC(hipMemcpyAsync(recs, c->em_state.p, sizeof recs, hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    *now = em_next_round(recs[parity ^ 1], (int)n_iter, (int)min_rounds, spec);
    return 0;
  };
  EmNow now{};
  if (!spec) {
    bool use_graph = c->tune.em_graph != 2;
    hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
    hipStream_t es = c->stream;
    if (use_graph) {
      if (!c->em_stream) HIPC(hipStreamCreateWithFlags(&c->em_stream, hipStreamNonBlocking));
      es = c->em_stream;
      HIPC(hipEventRecord(c->ev2, c->stream));           // the private stream starts after the preparation kernels
      HIPC(hipStreamWaitEvent(es, c->ev2, 0));
      HIPC(hipStreamBeginCapture(es, hipStreamCaptureModeThreadLocal));
      enqueue_rounds(es, chunk);                         // chunk is even: the captured parities 0,1,0,1... repeat on replay
      HIPC(hipStreamEndCapture(es, &graph));
      HIPC(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
    }
    for (;;) {
      if (use_graph) HIPC(hipGraphLaunch(gexec, es)); else enqueue_rounds(es, chunk);
      HIPC(hipGetLastError());
      if (int rc = read_state(es, &now)) return rc;
      if (now.done) break;
    }
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    hs.done = 1; hs.rounds = now.rounds; hs.final_round = now.fin;
  } else {
    // Partitioned EM: run a chunk of rounds speculatively, sum the per-round change counts over the ranks (the callback;
    // also the only synchronisation between ranks), find the first round s at which the reference's test
    // "chcount == 0 && s > min_rounds" (:202-205) holds globally, rewind to the chunk's checkpoint and replay up to s,
    // then the final round.  Every rank sees the same history, so every rank takes the same decision.
    std::vector<int> hist(chunk);
    int base = 0;
    for (;;) {
      if (int rc = checkpoint(base & 1, false)) return rc;
      const int n_run = (int)std::min<long>(chunk, (long)n_iter - base);
      enqueue_rounds(c->stream, n_run);
      HIPC(hipGetLastError());
      if (int rc = read_state(c->stream, &now)) return rc;
      // the history entry of a round is written by the NEXT round's k_em_rows; the chunk's last one is still in its record
      HIPC(hipMemcpy(spec_hist + base + n_run - 1, &recs[parity ^ 1].chcount, sizeof(int), hipMemcpyHostToDevice));
      if (int rc = part.cb(part.user, spec_hist + base, n_run)) return kamd::fail(-103, "kamd_em_run_partitioned: the sum callback failed (" + std::to_string(rc) + ")");
      HIPC(hipMemcpy(hist.data(), spec_hist + base, n_run * sizeof(int), hipMemcpyDeviceToHost));
      int stop = -1;
      for (int i = 0; i < n_run; i++) if (hist[i] == 0 && base + i > (int)min_rounds) { stop = base + i; break; }
      if (stop < 0) {
        base += n_run;
        if (base >= (int)n_iter) { hs.done = 1; hs.rounds = (int)n_iter; hs.final_round = 0; break; }  // the loop ran out
        continue;
      }
      // rewind + replay rounds base..stop, then the clamped final round
      if (int rc = checkpoint(base & 1, true)) return rc;
      EmState st0[2]; memset(st0, 0, sizeof st0);
      st0[1].iter = base - 1; st0[1].chcount = 1;                  // "the round before `base`" in slot 1, next parity 0
      HIPC(hipMemcpyAsync(c->em_state.p, st0, sizeof st0, hipMemcpyHostToDevice, c->stream));
      HIPC(hipStreamSynchronize(c->stream));                       // st0 is a stack buffer
      parity = 0;
      enqueue_rounds(c->stream, stop - base + 1);
      if (int rc = read_state(c->stream, &now)) return rc;
      const int one = 1;                                           // ask for the final round
      HIPC(hipMemcpy(&((EmState*)c->em_state.p)[parity ^ 1].force_final, &one, sizeof(int), hipMemcpyHostToDevice));
      enqueue_rounds(c->stream, 1);
      if (int rc = read_state(c->stream, &now)) return rc;
      if (!now.done || now.rounds != stop + 1 || !now.fin)
        return kamd::fail(-101, "kamd_em_run_partitioned: replay did not end on the final round");
      hs.done = 1; hs.rounds = now.rounds; hs.final_round = 1;
      break;
    }
  }
  if (streamed) {  // m-space -> transcript space, both buffers
    hipLaunchKernelGGL(k_pm_scatter, dim3(grid_for(T, BLOCK)), dim3(BLOCK), 0, c->stream, T, plan.mflag, plan.mpos, c->em_single.as<double>(),
                       plan.args.alpha0, plan.args.alpha1, c->em_alpha.as<double>(), c->em_next.as<double>());
    HIPC(hipGetLastError());
  }
  HIPC(hipEventRecord(c->ev1, c->stream));
  HIPC(hipEventSynchronize(c->ev1));
  HIPC(hipEventElapsedTime(&c->last_em_ms, c->ev0, c->ev1));
  c->last_em_iters = (uint64_t)hs.rounds + (hs.final_round ? 1 : 0);
  // result = the buffer the last executed round wrote; alpha_before_zeroes = the (unclamped) buffer it read
  double* bufs[2] = {c->em_alpha.as<double>(), c->em_next.as<double>()};
  const int last_read = hs.final_round ? (hs.rounds & 1) : ((hs.rounds - 1) & 1);
  HIPC(hipMemcpyAsync(alpha, bufs[last_read ^ 1], T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (alpha_before_zeroes) {
    if (hs.final_round) HIPC(hipMemcpyAsync(alpha_before_zeroes, bufs[last_read], T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    else memset(alpha_before_zeroes, 0, T * sizeof(double));
  }
  HIPC(hipStreamSynchronize(c->stream));
  if (rounds) *rounds = hs.rounds;
  return 0;
}
}  // namespace
// ---- bootstrap (Bootstrap::run_em, src/Bootstrap.cpp:4-14; Multinomial::sample, src/Multinomial.hpp:33-51) -----------
namespace {
constexpr u64 LCG_M = 2147483647ULL, LCG_A = 16807ULL;  // std::minstd_rand0 (libstdc++ default_random_engine)
__host__ __device__ inline u64 lcg_pow(u64 e) { u64 r = 1, b = LCG_A; while (e) { if (e & 1) r = r * b % LCG_M; b = b * b % LCG_M; e >>= 1; } return r; }
constexpr int DRAWS_PER_THREAD = 64;
// Draw i consumes engine outputs 2i+1 and 2i+2 (generate_canonical<double,53> makes two calls), so a thread can start
// anywhere by LCG skip-ahead: the sample is identical to the reference's N sequential draws.
// (blockIdx.y = replicate: x0s[b] is its seed, its sample goes to samp + b * n)
__global__ void k_multinomial(const double* __restrict__ cp, u64 n, u64 n_draws, const u64* __restrict__ x0s, double r2, u32* samp_all) {
  const u64 first = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * DRAWS_PER_THREAD;
  if (first >= n_draws) return;
  const u64 x0 = x0s[blockIdx.y];
  u32* samp = samp_all + (u64)blockIdx.y * n;
  u64 x = x0 * lcg_pow(2 * first) % LCG_M;
  const u64 last = min(n_draws, first + DRAWS_PER_THREAD);
  for (u64 d = first; d < last; d++) {
    x = x * LCG_A % LCG_M; const u64 g1 = x;
    x = x * LCG_A % LCG_M; const u64 g2 = x;
    // generate_canonical (bits/random.tcc): sum = (g1-min)*1 + (g2-min)*R, ret = sum / R^2 -- separate roundings, no FMA
    double sum = __dmul_rn((double)(g1 - 1), 1.0);
    sum = __dadd_rn(sum, __dmul_rn((double)(g2 - 1), 2147483646.0));
    double p = __ddiv_rn(sum, r2);
    if (p >= 1.0) p = 0.99999999999999989;  // nextafter(1.0, 0.0)
    u64 lo = 0, hi = n;                      // std::lower_bound(cp.begin(), cp.end(), p)
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (cp[mid] < p) lo = mid + 1; else hi = mid; }
    atomicAdd(&samp[lo], 1u);
  }
}
}  // namespace

namespace {
// Multinomial(counts, seed_b).sample() for replicates b = 0..n_rep-1 in ONE launch: samples at bs_samp + b * n_ecs
int resample_all(kamd_ctx* c, const uint32_t* d_counts, u64 n_ecs, const uint64_t* seeds, int n_rep) {
  // discrete_distribution<int>(counts): p = counts / sum, cp = partial_sum(p) (sequential FP64 adds), cp.back() = 1
  std::vector<u32> counts(n_ecs);
  HIPC(hipMemcpyAsync(counts.data(), d_counts, n_ecs * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  double sum = 0.0; u64 N = 0;
  for (u64 i = 0; i < n_ecs; i++) { sum += (double)counts[i]; N += counts[i]; }
  const int nsamp = (int)N;  // Multinomial::n_ is an int
  if (nsamp < 1) return kamd::fail(-1, "kamd_bootstrap: nothing to resample");
  std::vector<double> cp(n_ecs);
  double acc = 0.0;
  for (u64 i = 0; i < n_ecs; i++) { acc += (double)counts[i] / sum; cp[i] = acc; }
  cp[n_ecs - 1] = 1.0;
  std::vector<u64> x0s((size_t)n_rep);
  for (int b = 0; b < n_rep; b++) { u64 x0 = seeds[b] % LCG_M; if (x0 == 0) x0 = 1; x0s[b] = x0; }   // linear_congruential_engine::seed
  if (int rc = c->bs_cp.ensure(n_ecs * sizeof(double) + (size_t)n_rep * sizeof(u64) + 64, 0, c->stream)) return rc;
  if (int rc = c->bs_samp.ensure((size_t)n_rep * n_ecs * sizeof(u32), 0, c->stream)) return rc;
  u64* d_x0 = (u64*)((char*)c->bs_cp.p + ((n_ecs * sizeof(double) + 15) & ~(size_t)15));
  HIPC(hipMemcpyAsync(c->bs_cp.p, cp.data(), n_ecs * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemcpyAsync(d_x0, x0s.data(), (size_t)n_rep * sizeof(u64), hipMemcpyHostToDevice, c->stream));
  HIPC(hipMemsetAsync(c->bs_samp.p, 0, (size_t)n_rep * n_ecs * sizeof(u32), c->stream));
  if (n_ecs < 2) {  // _M_cp is empty: every draw returns 0
    std::vector<u32> all((size_t)n_rep, (u32)nsamp);
    HIPC(hipMemcpyAsync(c->bs_samp.p, all.data(), (size_t)n_rep * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  } else {
    const double r2 = (double)(2147483646.0L * 2147483646.0L);  // __tmp after two `__tmp *= __r` (long double) steps
    const u64 threads = ((u64)nsamp + DRAWS_PER_THREAD - 1) / DRAWS_PER_THREAD;
    hipLaunchKernelGGL(k_multinomial, dim3(grid_for(threads, BLOCK), (unsigned)n_rep), dim3(BLOCK), 0, c->stream, c->bs_cp.as<double>(), (u64)n_ecs,
                       (u64)nsamp, (const u64*)d_x0, r2, c->bs_samp.as<u32>());
    HIPC(hipGetLastError());
  }
  HIPC(hipStreamSynchronize(c->stream));  // cp / x0s are host staging buffers
  return 0;
}
}  // namespace

extern "C" int kamd_bootstrap(kamd_ctx* c, const uint64_t* d_ec_off, const uint32_t* d_ec_ids, const uint32_t* d_counts,
                              uint64_t n_ecs, uint64_t seed, const double* eff_lens, uint64_t n_targets, double* alpha,
                              int32_t* rounds, uint32_t* sample_out) {
  if (!c || !eff_lens || !alpha) return kamd::fail(-1, "kamd_bootstrap: null argument");
  HIPC(hipSetDevice(c->device));
  if (!d_ec_off) {
    if (!c->finalized) return kamd::fail(-1, "kamd_bootstrap: no EC result (call kamd_ec_finalize or pass a CSR)");
    d_ec_off = c->result.d_ec_off; d_ec_ids = c->result.d_ec_ids; d_counts = c->result.d_counts; n_ecs = c->result.n_ecs;
  }
  if (n_ecs == 0) return kamd::fail(-1, "kamd_bootstrap: no equivalence classes");
  if (int rc = resample_all(c, d_counts, n_ecs, &seed, 1)) return rc;
  if (sample_out) {
    HIPC(hipMemcpyAsync(sample_out, c->bs_samp.p, n_ecs * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  // fresh EMAlgorithm on the resampled counts; the weights still come from the original counts (EMAlgorithm.h:46);
  // run(10000, 50, false, false)
  return kamd_em_run(c, d_ec_off, d_ec_ids, c->bs_samp.as<u32>(), d_counts, n_ecs, eff_lens, n_targets, 10000, 50, alpha, nullptr, rounds);
}

// n_rep replicates of Bootstrap::run_em on the finalized EC result: every replicate's multinomial sample is drawn in one launch,
// the EMs run one after the other on the plan of the matrix (component-local form: built once -- by this call if need be --
// and only refreshed with each replicate's counts).  alpha: n_rep x n_targets, row b = replicate b.
extern "C" int kamd_bootstrap_batch(kamd_ctx* c, const uint64_t* seeds, int32_t n_rep, const double* eff_lens, uint64_t n_targets,
                                    double* alpha, int32_t* rounds) {
  if (!c || !seeds || !eff_lens || !alpha || n_rep < 0) return kamd::fail(-1, "kamd_bootstrap_batch: bad argument");
  if (!c->finalized) return kamd::fail(-1, "kamd_bootstrap_batch: no EC result (call kamd_ec_finalize first)");
  if (n_rep == 0) return 0;
  HIPC(hipSetDevice(c->device));
  const u64 n_ecs = c->result.n_ecs;
  if (n_ecs == 0) return kamd::fail(-1, "kamd_bootstrap_batch: no equivalence classes");
  if (int rc = resample_all(c, c->result.d_counts, n_ecs, seeds, n_rep)) return rc;
  for (int b = 0; b < n_rep; b++) {
    int32_t r = 0;
    if (int rc = kamd_em_run(c, c->result.d_ec_off, c->result.d_ec_ids, c->bs_samp.as<u32>() + (u64)b * n_ecs, c->result.d_counts, n_ecs, eff_lens,
                             n_targets, 10000, 50, alpha + (u64)b * n_targets, nullptr, &r)) return rc;
    if (rounds) rounds[b] = r;
  }
  return 0;
}

