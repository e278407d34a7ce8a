// tests/emu/em_local_emu.cpp -- TEST INFRASTRUCTURE: drives kallisto_amd/csrc/kamd_em_local.h (plan builder, per-group
// rounds, chunk / history / replay driver) on the CPU so that it can be checked against the oracle's EMAlgorithm::run.
#include "../../kallisto_amd/csrc/kamd_em_local.h"
#include "../../kallisto_amd/csrc/kamd_em_sell.h"

#include <map>

extern "C" {
// returns 0 = ran, 1 = not applicable (a component exceeds the budget / the 16-bit local index range)
int emu_em_local(const uint64_t* ec_off, const uint32_t* ec_ids, const uint32_t* counts, uint64_t n_ecs, const double* eff, uint64_t T,
                 uint64_t budget_bytes, uint64_t target_nnz, int n_iter, int min_rounds, int chunk, double* alpha, double* abz,
                 int32_t* rounds, uint32_t* n_groups, uint64_t* max_group_bytes, int builder) {
  using namespace kamd_em_local;
  Plan P;   // builder 0: the host reference (greedy packing), 1: the data-parallel steps the device set-up is made of,
            // 2: 1 + conversion to the sliced-ELLPACK layout (kamd_em_sell.h) and its host model of the kernel's round
  // builder >= 10: the steps with two size classes (components of at most 40 entries in groups of about 60), then builder - 10
  const bool classes = builder >= 10;
  if (classes) builder -= 10;
  if (int rc = builder ? build_plan_steps_host(ec_off, ec_ids, counts, nullptr, n_ecs, eff, T, budget_bytes, target_nnz, &P, classes ? 40u : 0u, 60)
                       : build_plan_host(ec_off, ec_ids, counts, nullptr, n_ecs, eff, T, budget_bytes, target_nnz, &P)) return rc;
  *n_groups = P.n_groups; *max_group_bytes = P.max_group_bytes;
  if (builder >= 2) {   // 2 / 3 / 4: segments of more than 64 / 16 / 3 entries are split over lanes
    kamd_em_sell::Plan S;
    if (int rc = kamd_em_sell::from_csr_plan(P, budget_bytes, &S, builder == 2 ? 64u : builder == 3 ? 16u : 3u)) return rc;
    *max_group_bytes = S.max_group_bytes;
    kamd_em_sell::CpuBackend B(S);
    *rounds = run(B, S, n_iter, min_rounds, chunk, alpha, abz);
    return 0;
  }
  CpuBackend B(P);
  *rounds = run(B, P, n_iter, min_rounds, chunk, alpha, abz);
  return 0;
}
// the plan must hold exactly the rows with >= 2 transcripts (as sets, with their counts), both directions must describe
// the same matrix, every group must respect the budget; returns 0 = consistent, > 0 = which check failed
int emu_em_local_check_plan(const uint64_t* ec_off, const uint32_t* ec_ids, const uint32_t* counts, uint64_t n_ecs, const double* eff,
                            uint64_t T, uint64_t budget_bytes, uint64_t target_nnz, int builder) {
  using namespace kamd_em_local;
  Plan P;
  const bool classes = builder >= 10;
  if (classes) builder -= 10;
  if (builder ? build_plan_steps_host(ec_off, ec_ids, counts, nullptr, n_ecs, eff, T, budget_bytes, target_nnz, &P, classes ? 40u : 0u, 60)
              : build_plan_host(ec_off, ec_ids, counts, nullptr, n_ecs, eff, T, budget_bytes, target_nnz, &P)) return -1;
  if (classes) {   // small groups first, and only small components in them (a group of the small class stays below limit + target entries)
    for (uint32_t g = 0; g < P.n_small; g++) if (P.nz_base[g + 1] - P.nz_base[g] >= 40 + 60) return 10;
  }
  std::map<std::vector<uint32_t>, uint64_t> want, got;
  for (uint64_t e = 0; e < n_ecs; e++) {
    if (ec_off[e + 1] - ec_off[e] < 2) continue;
    want[std::vector<uint32_t>(ec_ids + ec_off[e], ec_ids + ec_off[e + 1])] += counts[e];
  }
  std::vector<uint8_t> seen(T, 0);
  for (uint32_t g = 0; g < P.n_groups; g++) {
    const Group G = P.group(g);
    const uint64_t nnz = P.nz_base[g + 1] - P.nz_base[g];
    if (group_bytes(nnz, G.n_rows, G.n_tr) > budget_bytes) return 1;
    if (G.row_ptr[0] != 0 || G.row_ptr[G.n_rows] != nnz || G.col_ptr[0] != 0 || G.col_ptr[G.n_tr] != nnz) return 2;
    for (uint32_t t = 0; t < G.n_tr; t++) {
      const uint32_t id = P.tr_id[P.tr_base[g] + t];
      if (id >= T || seen[id]) return 3;                       // a transcript belongs to exactly one group
      seen[id] = 1;
      if (G.eff[t] != eff[id]) return 4;
    }
    std::vector<std::vector<uint32_t>> from_cols(G.n_rows);
    for (uint32_t t = 0; t < G.n_tr; t++)
      for (uint32_t j = G.col_ptr[t]; j < G.col_ptr[t + 1]; j++) { if (G.col_row[j] >= G.n_rows) return 5; from_cols[G.col_row[j]].push_back(t); }
    for (uint32_t r = 0; r < G.n_rows; r++) {
      std::vector<uint32_t> loc(G.row_tr + G.row_ptr[r], G.row_tr + G.row_ptr[r + 1]), ids;
      for (uint32_t l : loc) { if (l >= G.n_tr) return 6; ids.push_back(P.tr_id[P.tr_base[g] + l]); }
      std::vector<uint32_t> a = loc, b = from_cols[r];
      std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
      if (a != b) return 7;                                    // the two directions disagree
      std::sort(ids.begin(), ids.end());
      got[ids] += (uint32_t)G.cw[r];
      if ((uint32_t)(G.cw[r] >> 32) != (uint32_t)G.cw[r]) return 8;   // weight counts default to the counts
    }
  }
  return want == got ? 0 : 9;
}
}
