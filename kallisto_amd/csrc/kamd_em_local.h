// kamd_em_local.h -- component-local EM (EMAlgorithm::run, src/EMAlgorithm.h:95-223).  EXPERIMENTAL: kamd_em_run only takes
// this form with KAMD_EM_LOCAL=1; the CPU side below is tested (tests/test_em_local.py), the kernel has not run on hardware yet.
//
// The EC x transcript matrix is block diagonal over the connected components of the transcript/EC graph (gene families)
// and the EM update never couples two components.  Components are therefore packed into GROUPS small enough for one
// workgroup's LDS (local 16-bit indices in both directions) and a group iterates on its own: row pass, barrier, column
// pass, barrier -- no kernel boundary and no global memory traffic per round (the streamed form of kamd_em.hip pays
// two kernel boundaries and two dependent global round trips per round: 25.5 us).
// Only the stop rule "chcount == 0 && i > min_rounds" (:202-205) is global.  It is handled like kamd_em_run_partitioned
// does across ranks: a chunk of rounds runs speculatively from a checkpoint while every group adds its per-round change
// count to a history, the first qualifying round is found, the chunk is replayed up to it, then the clamped final round.
//
// This header holds what can be checked without a GPU (tests/emu, tests/test_em_local.py):
//   * the plan (groups, local CSR in both directions) and a host reference builder for it,
//   * the per-group round as host/device functions written for thread-strided execution,
//   * the chunk / history / replay driver, templated on a backend (here: the serial CPU backend).
// The device side: k_em_local in kamd_em.hip calls the same round functions on LDS copies of a group; its plan is
// still built by build_plan_host() from a download of the CSR (slow: bring-up only).  Set-up kernels that build the same
// plan in HBM are the next step; what they produce can be validated against build_plan_host().
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#ifndef KAMD_HD
#if defined(__HIPCC__)
#define KAMD_HD __host__ __device__ __forceinline__
#else
#define KAMD_HD inline
#endif
#endif

namespace kamd_em_local {

// one group of components, as the round functions see it (pointers into the plan's arrays, or into LDS copies of them)
struct Group {
  uint32_t n_rows, n_tr;
  const uint32_t* row_ptr;   // [n_rows + 1] offsets into row_tr, relative to the group's first entry
  const uint16_t* row_tr;    // local transcript index of every entry, row by row
  const uint32_t* col_ptr;   // [n_tr + 1] offsets into col_row, relative to the group's first entry
  const uint16_t* col_row;   // local row index of every entry, transcript by transcript
  const uint64_t* cw;        // [n_rows] count | weight count << 32
  const double* single;      // [n_tr] count of the transcript's singleton set (0: none), :119-123
  const double* eff;         // [n_tr] effective length
};

// LDS bytes a group needs in the kernel: both index directions (u16), both offset arrays, g + cw per row,
// alpha / a (current and next) + single + eff per transcript
KAMD_HD uint64_t group_bytes(uint64_t nnz, uint64_t rows, uint64_t tr) {
  return nnz * 4 + (rows + 1) * 4 + (tr + 1) * 4 + rows * 16 + tr * 48;
}

// row pass of one round: g_e = count_e / S_e, S_e = sum of a_t over the row (a_t = alpha_t / eff_t; in the final round
// alpha < alpha_limit / 10 reads as 0, :217-219); rows the reference skips get 0 (count 0, :133-135; denom below
// denorm_min, :156-158).  Thread `tid` of `nthr` takes rows tid, tid + nthr, ...
KAMD_HD void rows_pass(const Group& G, uint32_t tid, uint32_t nthr, const double* alpha, const double* a, int clamp, double* g) {
  for (uint32_t r = tid; r < G.n_rows; r += nthr) {
    double S = 0.0;
    for (uint32_t j = G.row_ptr[r]; j < G.row_ptr[r + 1]; j++) {
      const uint32_t t = G.row_tr[j];
      S += (clamp && alpha[t] < 1e-7 / 10.0) ? 0.0 : a[t];
    }
    const uint64_t w = G.cw[r];
    const uint32_t cnt = (uint32_t)w, wc = (uint32_t)(w >> 32);
    g[r] = (cnt == 0 || (double)wc * S < 4.9406564584124654e-324) ? 0.0 : (double)cnt / S;
  }
}
// column pass: next_t = single_t + a_t * sum of g_e over the transcript's rows; returns this thread's share of the round's
// change count (:176-199)
KAMD_HD int cols_pass(const Group& G, uint32_t tid, uint32_t nthr, const double* alpha, const double* a, int clamp, const double* g,
                      double* alpha_nx, double* a_nx) {
  int ch = 0;
  for (uint32_t t = tid; t < G.n_tr; t += nthr) {
    double acc = 0.0;
    for (uint32_t j = G.col_ptr[t]; j < G.col_ptr[t + 1]; j++) acc += g[G.col_row[j]];
    const bool z = clamp && alpha[t] < 1e-7 / 10.0;
    const double al = z ? 0.0 : alpha[t], at = z ? 0.0 : a[t];
    const double nx = G.single[t] + at * acc;
    if (nx > 1e-2 && (fabs(nx - al) / nx) > 1e-2) ++ch;
    alpha_nx[t] = nx;
    a_nx[t] = nx / G.eff[t];
  }
  return ch;
}

// ---- the plan: groups in HBM (here: host vectors) --------------------------------------------------------------------
struct Plan {
  uint32_t n_groups = 0;
  uint32_t n_small = 0;                      // groups [0, n_small) hold only small components (two size classes: build_plan_steps_host)
  uint64_t T = 0;
  std::vector<uint32_t> row_base, tr_base;   // [n_groups + 1] first row / first (m-space) transcript of a group
  std::vector<uint64_t> nz_base;             // [n_groups + 1] first entry of a group (same in both directions)
  std::vector<uint32_t> row_ptr;             // per group n_rows + 1 entries, concatenated: at row_base[g] + g
  std::vector<uint16_t> row_tr;              // [NZ]
  std::vector<uint32_t> col_ptr;             // per group n_tr + 1 entries, concatenated: at tr_base[g] + g
  std::vector<uint16_t> col_row;             // [NZ]
  std::vector<uint64_t> cw;                  // [R]
  std::vector<double> single, eff;           // [M]
  std::vector<uint32_t> tr_id;               // [M] transcript id of an m-space slot
  std::vector<double> single_all;            // [T] singleton-set count of every transcript (what transcripts outside m-space keep)
  uint64_t max_group_bytes = 0;
  Group group(uint32_t g) const {
    Group G;
    G.n_rows = row_base[g + 1] - row_base[g]; G.n_tr = tr_base[g + 1] - tr_base[g];
    G.row_ptr = row_ptr.data() + row_base[g] + g; G.row_tr = row_tr.data() + nz_base[g];
    G.col_ptr = col_ptr.data() + tr_base[g] + g; G.col_row = col_row.data() + nz_base[g];
    G.cw = cw.data() + row_base[g]; G.single = single.data() + tr_base[g]; G.eff = eff.data() + tr_base[g];
    return G;
  }
};

// Host reference builder.  Rows with one transcript are folded into `single` (a transcript has at most one singleton set);
// rows with >= 2 transcripts and the transcripts that occur in them make up the groups.  Components are taken in the order
// of their smallest transcript id (genes are contiguous in transcript space) and packed greedily: a group is closed when
// the next component would push it over `budget_bytes` or over `target_nnz`.
// Returns 0 = ok, 1 = not applicable (a single component exceeds the budget or the 16-bit local index range).
inline int build_plan_host(const uint64_t* ec_off, const uint32_t* ec_ids, const uint32_t* counts, const uint32_t* wcounts, uint64_t n_ecs,
                           const double* eff, uint64_t T, uint64_t budget_bytes, uint64_t target_nnz, Plan* P) {
  std::vector<uint32_t> parent(T);
  std::iota(parent.begin(), parent.end(), 0u);
  auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
  P->T = T;
  P->single_all.assign(T, 0.0);
  std::vector<uint8_t> in_multi(T, 0);
  for (uint64_t e = 0; e < n_ecs; e++) {
    const uint64_t a = ec_off[e], b = ec_off[e + 1];
    if (b - a == 1) { P->single_all[ec_ids[a]] = (double)counts[e]; continue; }
    if (b - a < 2) continue;
    uint32_t r0 = find(ec_ids[a]);
    in_multi[ec_ids[a]] = 1;
    for (uint64_t j = a + 1; j < b; j++) {
      in_multi[ec_ids[j]] = 1;
      uint32_t r1 = find(ec_ids[j]);
      if (r1 != r0) { if (r1 < r0) std::swap(r0, r1); parent[r1] = r0; }   // the root is the smallest id of the component
    }
  }
  // per component: entries, rows, transcripts
  std::vector<uint64_t> c_nnz(T, 0);
  std::vector<uint32_t> c_rows(T, 0), c_tr(T, 0);
  for (uint64_t e = 0; e < n_ecs; e++) {
    const uint64_t a = ec_off[e], b = ec_off[e + 1];
    if (b - a < 2) continue;
    const uint32_t r = find(ec_ids[a]);
    c_nnz[r] += b - a; c_rows[r] += 1;
  }
  for (uint64_t t = 0; t < T; t++) if (in_multi[t]) c_tr[find((uint32_t)t)] += 1;
  // groups
  std::vector<uint32_t> group_of(T, 0xFFFFFFFFu);   // by root
  uint32_t ng = 0;
  uint64_t g_nnz = 0, g_rows = 0, g_tr = 0;
  bool open = false;
  P->max_group_bytes = 0;
  for (uint64_t r = 0; r < T; r++) {
    if (c_rows[r] == 0) continue;
    if (c_rows[r] > 65535 || c_tr[r] > 65535 || group_bytes(c_nnz[r], c_rows[r], c_tr[r]) > budget_bytes) return 1;
    if (open && (group_bytes(g_nnz + c_nnz[r], g_rows + c_rows[r], g_tr + c_tr[r]) > budget_bytes || g_nnz + c_nnz[r] > target_nnz ||
                 g_rows + c_rows[r] > 65535 || g_tr + c_tr[r] > 65535)) {
      P->max_group_bytes = std::max(P->max_group_bytes, group_bytes(g_nnz, g_rows, g_tr));
      ++ng; g_nnz = g_rows = g_tr = 0;
    }
    open = true;
    group_of[r] = ng;
    g_nnz += c_nnz[r]; g_rows += c_rows[r]; g_tr += c_tr[r];
  }
  if (open) { P->max_group_bytes = std::max(P->max_group_bytes, group_bytes(g_nnz, g_rows, g_tr)); ++ng; }
  P->n_groups = ng;
  // counting sort of rows and transcripts by group
  P->row_base.assign(ng + 1, 0); P->tr_base.assign(ng + 1, 0); P->nz_base.assign(ng + 1, 0);
  for (uint64_t e = 0; e < n_ecs; e++) {
    const uint64_t a = ec_off[e], b = ec_off[e + 1];
    if (b - a < 2) continue;
    const uint32_t g = group_of[find(ec_ids[a])];
    P->row_base[g + 1] += 1; P->nz_base[g + 1] += b - a;
  }
  for (uint64_t t = 0; t < T; t++) if (in_multi[t]) P->tr_base[group_of[find((uint32_t)t)] + 1] += 1;
  for (uint32_t g = 0; g < ng; g++) { P->row_base[g + 1] += P->row_base[g]; P->tr_base[g + 1] += P->tr_base[g]; P->nz_base[g + 1] += P->nz_base[g]; }
  const uint64_t R = P->row_base[ng], M = P->tr_base[ng], NZ = P->nz_base[ng];
  std::vector<uint32_t> local_of(T, 0);     // m-space slot of a transcript, relative to its group
  P->tr_id.assign(M, 0); P->single.assign(M, 0.0); P->eff.assign(M, 0.0);
  {
    std::vector<uint32_t> fill(ng, 0);
    for (uint64_t t = 0; t < T; t++) {
      if (!in_multi[t]) continue;
      const uint32_t g = group_of[find((uint32_t)t)];
      const uint32_t l = fill[g]++;
      local_of[t] = l;
      const uint64_t m = P->tr_base[g] + l;
      P->tr_id[m] = (uint32_t)t; P->single[m] = P->single_all[t]; P->eff[m] = eff[t];
    }
  }
  P->cw.assign(R, 0); P->row_ptr.assign(R + ng, 0); P->row_tr.assign(NZ, 0);
  P->col_ptr.assign(M + ng, 0); P->col_row.assign(NZ, 0);
  std::vector<uint32_t> row_local(n_ecs, 0);
  {
    std::vector<uint32_t> rfill(ng, 0);
    std::vector<uint64_t> zfill(ng, 0);
    for (uint64_t e = 0; e < n_ecs; e++) {
      const uint64_t a = ec_off[e], b = ec_off[e + 1];
      if (b - a < 2) continue;
      const uint32_t g = group_of[find(ec_ids[a])];
      const uint32_t r = rfill[g]++;
      row_local[e] = r;
      P->cw[P->row_base[g] + r] = (uint64_t)counts[e] | ((uint64_t)(wcounts ? wcounts[e] : counts[e]) << 32);
      uint32_t* rp = P->row_ptr.data() + P->row_base[g] + g;
      rp[r] = (uint32_t)zfill[g];
      for (uint64_t j = a; j < b; j++) P->row_tr[P->nz_base[g] + zfill[g]++] = (uint16_t)local_of[ec_ids[j]];
      rp[r + 1] = (uint32_t)zfill[g];
    }
  }
  // transposed direction: count, scan, fill (rows in group order, so a column's rows are ascending)
  for (uint64_t e = 0; e < n_ecs; e++) {
    const uint64_t a = ec_off[e], b = ec_off[e + 1];
    if (b - a < 2) continue;
    const uint32_t g = group_of[find(ec_ids[a])];
    uint32_t* cp = P->col_ptr.data() + P->tr_base[g] + g;
    for (uint64_t j = a; j < b; j++) cp[local_of[ec_ids[j]] + 1] += 1;
  }
  for (uint32_t g = 0; g < ng; g++) {
    uint32_t* cp = P->col_ptr.data() + P->tr_base[g] + g;
    const uint32_t nt = P->tr_base[g + 1] - P->tr_base[g];
    for (uint32_t t = 0; t < nt; t++) cp[t + 1] += cp[t];
  }
  {
    std::vector<uint32_t> cfill(M, 0);
    for (uint64_t e = 0; e < n_ecs; e++) {
      const uint64_t a = ec_off[e], b = ec_off[e + 1];
      if (b - a < 2) continue;
      const uint32_t g = group_of[find(ec_ids[a])];
      const uint32_t* cp = P->col_ptr.data() + P->tr_base[g] + g;
      for (uint64_t j = a; j < b; j++) {
        const uint32_t l = local_of[ec_ids[j]];
        P->col_row[P->nz_base[g] + cp[l] + cfill[P->tr_base[g] + l]++] = (uint16_t)row_local[e];
      }
    }
  }
  return 0;
}

// ---- the same plan built the way the device will build it: data-parallel steps over flat arrays ---------------------------
// Every step is a function of one index (a row, a transcript, a component root or a group) that only uses plain stores and
// atomic adds, so a kernel is `step(blockIdx.x * blockDim.x + threadIdx.x, A)`; between the steps sit exclusive scans.  On
// the host the steps run serially (build_plan_steps_host below), which is how tests/test_em_local.py checks them.  The order
// of rows / transcripts inside a group comes from atomic cursors, and is then made CANONICAL by ranking (steps F2, G2, K2: a member's
// final position is the number of members of its group with a smaller key -- one loop over the group per member, a few hundred each):
// transcripts by id, rows by (first transcript, hash of the transcript set), a column's entries by row.  Any order is a valid plan, but the order of
// the entries inside a column and the lanes a split segment lands on fix the association of the floating-point sums: with the
// canonical order two runs over the same matrix give bit-identical abundances.  It also puts the rows and transcripts of a gene
// family next to each other, so that the lanes of a slice often gather the same LDS word (a broadcast instead of a bank conflict).
#if defined(__HIP_DEVICE_COMPILE__)
#define KAMD_EML_ADD32(p, v) atomicAdd((p), (v))
#else
static inline uint32_t kamd_eml_add32(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
#define KAMD_EML_ADD32(p, v) kamd_eml_add32((p), (v))
#endif
struct BuildArgs {
  // input
  const uint64_t* ec_off; const uint32_t* ec_ids; const uint32_t* counts; const uint32_t* wcounts; uint64_t n_ecs;
  const double* eff; uint64_t T; const uint32_t* label;   // label[t] = smallest transcript id of t's component
  uint64_t target_nnz;
  // two size classes (cum_big != nullptr): components of at most small_limit entries are cut into groups of about target_nnz entries
  // by cum_nnz (which then only counts them) -- groups [0, ng_small), one wavefront each in the kernel; the larger components
  // follow in groups of about target_big entries by cum_big -- one workgroup each
  const uint64_t* cum_big; uint64_t target_big; uint32_t small_limit; uint32_t ng_small;
  uint32_t* c_small; uint32_t* c_big;   // scratch of the split: a component's entries in its own class, 0 in the other
  // per transcript / per component root, [T] (+1 where scanned)
  uint8_t* in_multi; double* single_all; uint32_t* c_nnz; uint32_t* c_rows; uint32_t* c_tr; const uint64_t* cum_nnz; uint32_t* local_of;
  // per group, [n_groups] (+1 where scanned)
  uint32_t n_groups; uint32_t* g_rows; uint32_t* g_tr; uint32_t* g_nnz; uint32_t* row_fill; uint32_t* tr_fill;
  const uint32_t* row_base; const uint32_t* tr_base; const uint64_t* nz_base;
  // per row
  uint32_t* row_new;      // [n_ecs] new (group-major) index of a kept row
  uint32_t* len_new;      // [R] length of the row at a new index
  const uint64_t* row_abs;   // [R + 1] scan of len_new: absolute entry offset of a row
  // per m-space transcript
  uint32_t* col_cnt;      // [M]
  const uint64_t* col_abs;   // [M + 1] scan of col_cnt
  uint32_t* col_fill;     // [M]
  // the plan's arrays
  uint32_t* row_ptr; uint16_t* row_tr; uint32_t* col_ptr; uint16_t* col_row; uint64_t* cw; double* single; double* eff_m; uint32_t* tr_id;
  // scratch of the canonical numbering (below): members of a group in arrival order, before they are ranked
  uint32_t* tmp_tr_id;    // [M] transcript of an m-space slot as the atomic cursor handed the slots out
  uint64_t* row_key;      // [R] (local index of the row's first transcript << 48 | 48 hash bits of the row's transcript set) of a row
                          //     slot, in arrival order: a key of the row's CONTENT -- EC indices differ from run to run (kamd_ec_finalize
                          //     emits the sets in the order its atomics hand out), the sets do not
  uint32_t* row_e;        // [R] EC index of a row slot (the last tie-break, and where G2 finds the row)
  uint32_t* row_e_final;  // [R] EC index of the row at a FINAL row index (what the device's per-group builder walks); may be null
  uint16_t* col_row_tmp;  // [NZ] the transposed entries in arrival order
  uint32_t* ent_col;      // [NZ] m-space slot of the column a transposed entry belongs to
};
KAMD_HD uint32_t eml_group_of(const BuildArgs& A, uint32_t root) {
  if (A.cum_big && A.c_nnz[root] > A.small_limit) return A.ng_small + (uint32_t)(A.cum_big[root] / A.target_big);
  return (uint32_t)(A.cum_nnz[root] / A.target_nnz);
}
// C (per root, two size classes only): the component's entries in its class
KAMD_HD void step_root_c(uint64_t r, const BuildArgs& A) {
  if (r >= A.T) return;
  const uint32_t n = A.c_nnz[r];
  const bool big = n > A.small_limit;
  A.c_small[r] = big ? 0u : n; A.c_big[r] = big ? n : 0u;
}
// A (per row): component sizes, which transcripts are in a kept row, singleton counts
KAMD_HD void step_rows_a(uint64_t e, const BuildArgs& A) {
  if (e >= A.n_ecs) return;
  const uint64_t a = A.ec_off[e], b = A.ec_off[e + 1];
  if (b - a == 1) { A.single_all[A.ec_ids[a]] = (double)A.counts[e]; return; }
  if (b - a < 2) return;
  const uint32_t root = A.label[A.ec_ids[a]];
  KAMD_EML_ADD32(&A.c_nnz[root], (uint32_t)(b - a));   // (the caller refuses matrices with 2^32 entries or more)
  KAMD_EML_ADD32(&A.c_rows[root], 1u);
  for (uint64_t j = a; j < b; j++) A.in_multi[A.ec_ids[j]] = 1;
}
// B (per transcript): transcripts per component
KAMD_HD void step_tr_b(uint64_t t, const BuildArgs& A) {
  if (t < A.T && A.in_multi[t]) KAMD_EML_ADD32(&A.c_tr[A.label[t]], 1u);
}
// (scan c_nnz -> cum_nnz; n_groups = (NZ - 1) / target + 1)
// D (per root): group sizes; a component lies wholly in the group its first entry falls into
KAMD_HD void step_root_d(uint64_t r, const BuildArgs& A) {
  if (r >= A.T || A.c_rows[r] == 0) return;
  const uint32_t g = eml_group_of(A, (uint32_t)r);
  KAMD_EML_ADD32(&A.g_rows[g], A.c_rows[r]);
  KAMD_EML_ADD32(&A.g_tr[g], A.c_tr[r]);
  KAMD_EML_ADD32(&A.g_nnz[g], A.c_nnz[r]);
}
// (scans of g_rows / g_tr / g_nnz -> row_base / tr_base / nz_base; the caller checks every group against the budget)
// F (per transcript): m-space slot
KAMD_HD void step_tr_f(uint64_t t, const BuildArgs& A) {
  if (t >= A.T || !A.in_multi[t]) return;
  const uint32_t g = eml_group_of(A, A.label[t]);
  const uint32_t l = KAMD_EML_ADD32(&A.tr_fill[g], 1u);
  A.tmp_tr_id[(uint64_t)A.tr_base[g] + l] = (uint32_t)t;
}
// F2 (per m-space slot in arrival order): the transcript's final slot = its rank by id among the group's transcripts
KAMD_HD void step_m_f2(uint64_t i, uint32_t g, const BuildArgs& A) {
  const uint32_t t = A.tmp_tr_id[i];
  const uint32_t lo = A.tr_base[g], hi = A.tr_base[g + 1];
  uint32_t rank = 0;
  for (uint32_t j = lo; j < hi; j++) rank += A.tmp_tr_id[j] < t ? 1u : 0u;
  A.local_of[t] = rank;
  const uint64_t m = (uint64_t)lo + rank;
  A.tr_id[m] = t; A.single[m] = A.single_all[t]; A.eff_m[m] = A.eff[t];
}
// G (per row): new row index, its length and count word
KAMD_HD void step_rows_g(uint64_t e, const BuildArgs& A) {
  if (e >= A.n_ecs) return;
  const uint64_t a = A.ec_off[e], b = A.ec_off[e + 1];
  if (b - a < 2) return;
  const uint32_t g = eml_group_of(A, A.label[A.ec_ids[a]]);
  const uint32_t rn = A.row_base[g] + KAMD_EML_ADD32(&A.row_fill[g], 1u);
  uint64_t h = 0x9E3779B97F4A7C15ULL;
  for (uint64_t j = a; j < b; j++) { h ^= A.ec_ids[j]; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 29; }
  A.row_key[rn] = ((uint64_t)A.local_of[A.ec_ids[a]] << 48) | (h >> 16);   // (after F2: local_of is final and < 65536)
  A.row_e[rn] = (uint32_t)e;
}
// group that owns row slot i (slots are group-major): last g with row_base[g] <= i
KAMD_HD uint32_t eml_group_of_row_slot(const BuildArgs& A, uint64_t i) {
  uint32_t lo = 0, hi = A.n_groups;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (A.row_base[mid] <= i) lo = mid; else hi = mid; }
  return lo;
}
// G2 (per row slot in arrival order): the row's final index = its rank by (first transcript, set hash; EC index on a tie) among the group's rows;
// its length and count word go there
KAMD_HD void step_slot_g2(uint64_t i, uint32_t g, const BuildArgs& A) {
  const uint64_t key = A.row_key[i];
  const uint64_t e = A.row_e[i];
  const uint32_t lo = A.row_base[g], hi = A.row_base[g + 1];
  uint32_t rank = 0;
  uint32_t j = lo;
  for (; j + 4 <= hi; j += 4) {   // (four independent loads per trip: the loop is bound by their latency)
    const uint64_t k0 = A.row_key[j], k1 = A.row_key[j + 1], k2 = A.row_key[j + 2], k3 = A.row_key[j + 3];
    rank += (k0 < key ? 1u : 0u) + (k1 < key ? 1u : 0u) + (k2 < key ? 1u : 0u) + (k3 < key ? 1u : 0u);
    if (k0 == key || k1 == key || k2 == key || k3 == key)   // equal keys (the row itself, or a 48-bit hash collision): the EC index decides
      rank += ((k0 == key && A.row_e[j] < e) ? 1u : 0u) + ((k1 == key && A.row_e[j + 1] < e) ? 1u : 0u) + ((k2 == key && A.row_e[j + 2] < e) ? 1u : 0u) +
              ((k3 == key && A.row_e[j + 3] < e) ? 1u : 0u);
  }
  for (; j < hi; j++) { const uint64_t kj = A.row_key[j]; rank += (kj < key || (kj == key && A.row_e[j] < e)) ? 1u : 0u; }
  const uint32_t rn = lo + rank;
  A.row_new[e] = rn;
  if (A.row_e_final) A.row_e_final[rn] = (uint32_t)e;
  A.len_new[rn] = (uint32_t)(A.ec_off[e + 1] - A.ec_off[e]);
  A.cw[rn] = (uint64_t)A.counts[e] | ((uint64_t)(A.wcounts ? A.wcounts[e] : A.counts[e]) << 32);
}
// (scan len_new -> row_abs)
// I (per row): the row's entries, its relative offset, column counts
KAMD_HD void step_rows_i(uint64_t e, const BuildArgs& A) {
  if (e >= A.n_ecs) return;
  const uint64_t a = A.ec_off[e], b = A.ec_off[e + 1];
  if (b - a < 2) return;
  const uint32_t g = eml_group_of(A, A.label[A.ec_ids[a]]);
  const uint32_t rn = A.row_new[e];
  const uint64_t at = A.row_abs[rn];
  A.row_ptr[(uint64_t)rn + g] = (uint32_t)(at - A.nz_base[g]);
  for (uint64_t j = a; j < b; j++) {
    const uint32_t l = A.local_of[A.ec_ids[j]];
    A.row_tr[at + (j - a)] = (uint16_t)l;
    KAMD_EML_ADD32(&A.col_cnt[(uint64_t)A.tr_base[g] + l], 1u);
  }
}
// (scan col_cnt -> col_abs)
// group that owns m-space slot m (slots are group-major): last g with tr_base[g] <= m
KAMD_HD uint32_t eml_group_of_slot(const BuildArgs& A, uint64_t m) {
  uint32_t lo = 0, hi = A.n_groups;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (A.tr_base[mid] <= m) lo = mid; else hi = mid; }
  return lo;
}
// J (per m-space transcript) + (per group): relative column offsets and the closing entry of both offset arrays
KAMD_HD void step_m_j(uint64_t m, uint32_t g, const BuildArgs& A) {   // g = group of slot m (the caller knows it: slots are group-major)
  A.col_ptr[m + g] = (uint32_t)(A.col_abs[m] - A.nz_base[g]);
}
KAMD_HD void step_group_j(uint64_t g, const BuildArgs& A) {
  if (g >= A.n_groups) return;
  const uint32_t nnz = (uint32_t)(A.nz_base[g + 1] - A.nz_base[g]);
  A.row_ptr[(uint64_t)A.row_base[g + 1] + g] = nnz;
  A.col_ptr[(uint64_t)A.tr_base[g + 1] + g] = nnz;
}
// K (per row): the transposed entries
KAMD_HD void step_rows_k(uint64_t e, const BuildArgs& A) {
  if (e >= A.n_ecs) return;
  const uint64_t a = A.ec_off[e], b = A.ec_off[e + 1];
  if (b - a < 2) return;
  const uint32_t g = eml_group_of(A, A.label[A.ec_ids[a]]);
  const uint32_t rl = A.row_new[e] - A.row_base[g];
  for (uint64_t j = a; j < b; j++) {
    const uint64_t m = (uint64_t)A.tr_base[g] + A.local_of[A.ec_ids[j]];
    const uint64_t p = A.col_abs[m] + KAMD_EML_ADD32(&A.col_fill[m], 1u);
    A.col_row_tmp[p] = (uint16_t)rl; A.ent_col[p] = (uint32_t)m;
  }
}
// K2 (per transposed entry in arrival order): its final place in the column = its rank by row (a row occurs once in a column)
KAMD_HD void step_ent_k2(uint64_t p, const BuildArgs& A) {
  const uint64_t m = A.ent_col[p];
  const uint32_t v = A.col_row_tmp[p];
  const uint64_t lo = A.col_abs[m], hi = A.col_abs[m + 1];
  uint64_t rank = 0;
  for (uint64_t q = lo; q < hi; q++) rank += A.col_row_tmp[q] < v ? 1u : 0u;
  A.col_row[lo + rank] = (uint16_t)v;
}

// labels the way the device computes them (k_cc_*: min-label propagation): smallest transcript id of the component
inline std::vector<uint32_t> component_labels_host(const uint64_t* ec_off, const uint32_t* ec_ids, uint64_t n_ecs, uint64_t T) {
  std::vector<uint32_t> parent(T);
  std::iota(parent.begin(), parent.end(), 0u);
  auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
  for (uint64_t e = 0; e < n_ecs; e++) {
    const uint64_t a = ec_off[e], b = ec_off[e + 1];
    if (b - a < 2) continue;
    uint32_t r0 = find(ec_ids[a]);
    for (uint64_t j = a + 1; j < b; j++) { uint32_t r1 = find(ec_ids[j]); if (r1 != r0) { if (r1 < r0) std::swap(r0, r1); parent[r1] = r0; } }
  }
  std::vector<uint32_t> lab(T);
  for (uint64_t t = 0; t < T; t++) lab[t] = find((uint32_t)t);
  return lab;
}
// the steps above, run serially, with std::exclusive_scan-like loops where the device has its scan kernel.
// Returns 0 = ok, 1 = not applicable (some group exceeds the budget or the 16-bit range: the caller may retry with a smaller
// target or take another EM form).
// small_limit != 0: two size classes -- components of at most small_limit entries in groups of about target_small entries (the first
// P->n_small groups), the others in groups of about target_nnz entries.
inline int build_plan_steps_host(const uint64_t* ec_off, const uint32_t* ec_ids, const uint32_t* counts, const uint32_t* wcounts, uint64_t n_ecs,
                                 const double* eff, uint64_t T, uint64_t budget_bytes, uint64_t target_nnz, Plan* P, uint32_t small_limit = 0,
                                 uint64_t target_small = 0) {
  const std::vector<uint32_t> label = component_labels_host(ec_off, ec_ids, n_ecs, T);
  auto scan32 = [](const std::vector<uint32_t>& in, uint64_t n, std::vector<uint64_t>& out) { out.assign(n + 1, 0); for (uint64_t i = 0; i < n; i++) out[i + 1] = out[i] + in[i]; };
  BuildArgs A{};
  A.ec_off = ec_off; A.ec_ids = ec_ids; A.counts = counts; A.wcounts = wcounts; A.n_ecs = n_ecs; A.eff = eff; A.T = T; A.label = label.data();
  A.target_nnz = std::max<uint64_t>(1, target_nnz);
  std::vector<uint8_t> in_multi(T, 0); std::vector<uint64_t> cum(T + 1, 0); std::vector<uint32_t> c_nnz(T, 0), c_rows(T, 0), c_tr(T, 0), local_of(T, 0);
  P->T = T; P->single_all.assign(T, 0.0);
  A.in_multi = in_multi.data(); A.single_all = P->single_all.data(); A.c_nnz = c_nnz.data(); A.c_rows = c_rows.data(); A.c_tr = c_tr.data(); A.local_of = local_of.data();
  for (uint64_t e = 0; e < n_ecs; e++) step_rows_a(e, A);
  for (uint64_t t = 0; t < T; t++) step_tr_b(t, A);
  std::vector<uint32_t> c_small, c_big; std::vector<uint64_t> cum_big;
  uint32_t ng = 0;
  uint64_t NZ = 0;
  if (small_limit) {
    c_small.assign(T, 0); c_big.assign(T, 0); cum_big.assign(T + 1, 0);
    A.small_limit = small_limit; A.c_small = c_small.data(); A.c_big = c_big.data();
    for (uint64_t r = 0; r < T; r++) step_root_c(r, A);
    for (uint64_t t = 0; t < T; t++) { cum[t + 1] = cum[t] + c_small[t]; cum_big[t + 1] = cum_big[t] + c_big[t]; }
    A.target_big = A.target_nnz; A.target_nnz = std::max<uint64_t>(1, target_small);
    A.cum_big = cum_big.data();
    A.ng_small = cum[T] ? (uint32_t)((cum[T] - 1) / A.target_nnz + 1) : 0;
    ng = A.ng_small + (cum_big[T] ? (uint32_t)((cum_big[T] - 1) / A.target_big + 1) : 0);
    NZ = cum[T] + cum_big[T];
    P->n_small = A.ng_small;
  } else {
    for (uint64_t t = 0; t < T; t++) cum[t + 1] = cum[t] + c_nnz[t];
    NZ = cum[T];
    ng = NZ ? (uint32_t)((NZ - 1) / A.target_nnz + 1) : 0;
  }
  A.cum_nnz = cum.data();
  P->n_groups = ng; A.n_groups = ng;
  std::vector<uint32_t> g_rows(ng, 0), g_tr(ng, 0), g_nnz(ng, 0), row_fill(ng, 0), tr_fill(ng, 0);
  A.g_rows = g_rows.data(); A.g_tr = g_tr.data(); A.g_nnz = g_nnz.data(); A.row_fill = row_fill.data(); A.tr_fill = tr_fill.data();
  for (uint64_t r = 0; r < T; r++) step_root_d(r, A);
  P->row_base.assign(ng + 1, 0); P->tr_base.assign(ng + 1, 0); P->nz_base.assign(ng + 1, 0);
  P->max_group_bytes = 0;
  for (uint32_t g = 0; g < ng; g++) {
    if (g_rows[g] > 65535 || g_tr[g] > 65535 || group_bytes(g_nnz[g], g_rows[g], g_tr[g]) > budget_bytes) return 1;
    P->max_group_bytes = std::max(P->max_group_bytes, group_bytes(g_nnz[g], g_rows[g], g_tr[g]));
    P->row_base[g + 1] = P->row_base[g] + g_rows[g]; P->tr_base[g + 1] = P->tr_base[g] + g_tr[g]; P->nz_base[g + 1] = P->nz_base[g] + g_nnz[g];
  }
  A.row_base = P->row_base.data(); A.tr_base = P->tr_base.data(); A.nz_base = P->nz_base.data();
  const uint64_t R = ng ? P->row_base[ng] : 0, M = ng ? P->tr_base[ng] : 0;
  P->tr_id.assign(M, 0); P->single.assign(M, 0.0); P->eff.assign(M, 0.0); P->cw.assign(R, 0);
  P->row_ptr.assign(R + ng, 0); P->row_tr.assign(NZ, 0); P->col_ptr.assign(M + ng, 0); P->col_row.assign(NZ, 0);
  std::vector<uint32_t> row_new(n_ecs, 0), len_new(R, 0), col_cnt(M, 0), col_fill(M, 0); std::vector<uint64_t> row_abs, col_abs;
  A.row_new = row_new.data(); A.len_new = len_new.data(); A.col_cnt = col_cnt.data(); A.col_fill = col_fill.data();
  A.row_ptr = P->row_ptr.data(); A.row_tr = P->row_tr.data(); A.col_ptr = P->col_ptr.data(); A.col_row = P->col_row.data();
  A.cw = P->cw.data(); A.single = P->single.data(); A.eff_m = P->eff.data(); A.tr_id = P->tr_id.data();
  std::vector<uint32_t> tmp_tr_id(M, 0), ent_col(NZ, 0); std::vector<uint64_t> row_key(R, 0); std::vector<uint16_t> col_row_tmp(NZ, 0); std::vector<uint32_t> row_e(R, 0);
  A.row_e = row_e.data(); A.tmp_tr_id = tmp_tr_id.data(); A.row_key = row_key.data(); A.col_row_tmp = col_row_tmp.data(); A.ent_col = ent_col.data();
  for (uint64_t t = 0; t < T; t++) step_tr_f(t, A);
  for (uint64_t m = 0; m < M; m++) step_m_f2(m, eml_group_of_slot(A, m), A);
  for (uint64_t e = 0; e < n_ecs; e++) step_rows_g(e, A);
  for (uint64_t i = 0; i < R; i++) step_slot_g2(i, eml_group_of_row_slot(A, i), A);
  scan32(len_new, R, row_abs); A.row_abs = row_abs.data();
  for (uint64_t e = 0; e < n_ecs; e++) step_rows_i(e, A);
  scan32(col_cnt, M, col_abs); A.col_abs = col_abs.data();
  for (uint64_t m = 0; m < M; m++) step_m_j(m, eml_group_of_slot(A, m), A);   // (as the kernel does: the group by binary search)
  for (uint64_t g = 0; g < ng; g++) step_group_j(g, A);
  for (uint64_t e = 0; e < n_ecs; e++) step_rows_k(e, A);
  for (uint64_t p = 0; p < NZ; p++) step_ent_k2(p, A);
  return 0;
}

// ---- serial CPU backend: the groups one after the other, "one thread" each ------------------------------------------
struct CpuBackend {
  const Plan& P;
  std::vector<double> alpha, a, alpha_nx, a_nx, g, ck_alpha, ck_a;
  explicit CpuBackend(const Plan& p) : P(p) {
    const uint64_t M = P.tr_base[P.n_groups], R = P.row_base[P.n_groups];
    alpha.assign(M, 1.0 / (double)P.T);                    // alpha_ = 1/T for every transcript (:38)
    a.resize(M); alpha_nx.resize(M); a_nx.resize(M); g.resize(R);
    for (uint64_t m = 0; m < M; m++) a[m] = alpha[m] / P.eff[m];
  }
  void checkpoint() { ck_alpha = alpha; ck_a = a; }
  void restore() { alpha = ck_alpha; a = ck_a; }
  const std::vector<double>& host_alpha() { return alpha; }   // m-space alpha of the current state
  // n rounds for every group; hist[i] += change count of round i (if hist)
  void run(int n, int clamp, int* hist) {
    for (uint32_t gi = 0; gi < P.n_groups; gi++) {
      const Group G = P.group(gi);
      double* al = alpha.data() + P.tr_base[gi]; double* av = a.data() + P.tr_base[gi];
      double* aln = alpha_nx.data() + P.tr_base[gi]; double* avn = a_nx.data() + P.tr_base[gi];
      double* gg = g.data() + P.row_base[gi];
      for (int i = 0; i < n; i++) {
        rows_pass(G, 0, 1, al, av, clamp, gg);
        const int ch = cols_pass(G, 0, 1, al, av, clamp, gg, aln, avn);
        if (hist) hist[i] += ch;
        std::swap(al, aln); std::swap(av, avn);
      }
      if (n & 1) {   // the result sits in the *_nx buffers of this group
        memcpy(alpha.data() + P.tr_base[gi], alpha_nx.data() + P.tr_base[gi], G.n_tr * sizeof(double));
        memcpy(a.data() + P.tr_base[gi], a_nx.data() + P.tr_base[gi], G.n_tr * sizeof(double));
      }
    }
  }
};

// ---- the driver: EMAlgorithm::run's loop control over a backend that can only run whole chunks ----------------------
// alpha_out / abz_out: [T].  Returns the number of rounds ("ran for i rounds").
template <class Backend, class PlanT>
int run(Backend& B, const PlanT& P, int n_iter, int min_rounds, int chunk, double* alpha_out, double* abz_out) {
  const uint64_t M = P.tr_base[P.n_groups];
  std::vector<int> hist((size_t)chunk);
  int base = 0, rounds = 0;
  bool have_final = false;
  std::vector<double> before;
  for (;;) {
    const int n = std::min(chunk, n_iter - base);
    B.checkpoint();
    std::fill(hist.begin(), hist.end(), 0);
    B.run(n, 0, hist.data());
    int stop = -1;
    for (int i = 0; i < n; i++) if (hist[i] == 0 && base + i > min_rounds) { stop = base + i; break; }   // :202-205
    if (stop < 0) {
      base += n;
      if (base >= n_iter) { rounds = n_iter; break; }            // the loop ran out: no final round
      continue;
    }
    B.restore();
    B.run(stop - base + 1, 0, nullptr);                          // replay rounds base..stop
    { const std::vector<double>& cur = B.host_alpha(); before.assign(cur.begin(), cur.begin() + M); }   // what the final round reads: alpha_before_zeroes_
    B.run(1, 1, nullptr);                                        // the final round (:212-221, clamp applied on read)
    have_final = true;
    rounds = stop + 1;
    break;
  }
  // back to transcript space: a transcript outside m-space keeps its singleton count from round 1 on (0 if in no set)
  for (uint64_t t = 0; t < P.T; t++) { alpha_out[t] = P.single_all[t]; if (abz_out) abz_out[t] = have_final ? P.single_all[t] : 0.0; }
  const std::vector<double>& fin = B.host_alpha();
  for (uint64_t m = 0; m < M; m++) {
    alpha_out[P.tr_id[m]] = fin[m];
    if (abz_out) abz_out[P.tr_id[m]] = have_final ? before[m] : 0.0;
  }
  return rounds;
}

}  // namespace kamd_em_local
