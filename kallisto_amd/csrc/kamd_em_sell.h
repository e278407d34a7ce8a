// kamd_em_sell.h -- layout of the component-local EM (kamd_em_local.h) that the LDS kernel k_em_sell iterates over.
//
// kamd_em_local.h packs the connected components of the EC x transcript matrix into groups that fit one workgroup's LDS
// and gives every group a local CSR in both directions.  Iterating a CSR with one thread (or a few lanes) per row wastes
// most of the issue slots: rows have 2..60 entries, columns 1..thousands, and the work per group and round is small (a few
// thousand entries), so the fixed cost per row pass dominates (measured on config #3: ~7 000 wavefront instructions per
// group and round for 9 300 entry visits).  Here every direction of a group becomes a SLICED ELLPACK matrix instead:
//   * the segments (rows, resp. columns) are renumbered by decreasing length and cut into slices of 64 lanes; lane l of a
//     slice walks its segment's entries j < width of the slice (the longest segment in it; four trips to a 64-bit word, below); shorter
//     segments are padded with a sentinel index whose value is 0.0.  One trip of the inner loop = 64 entries, no row
//     pointers, no reductions, and the lengths inside a slice are nearly equal because of the sort;
//   * a segment with more than 64 entries is SPLIT over nv = min(64, ceil(len / 64)) adjacent lanes of one slice
//     (ceil(len / nv) entries each); the lanes' partial sums are combined by one segmented wavefront scan and the last lane
//     finishes the segment.  Slices that contain split segments carry a word of metadata per lane; all others are "plain":
//     lane l finishes segment first_seg + l.
// The local index of a segment is its position in this order, so the other direction's entries refer to it directly.
//
// Everything here is host/device code without atomics: one thread lays out one group (tests/emu runs it serially on the
// CPU and checks the resulting EM against the reference restatement, tests/test_em_local.py).
#pragma once
#include <cstdint>
#include <vector>

#include "kamd_em_local.h"

namespace kamd_em_sell {

static const uint32_t SELL_LANES = 64;
static const uint32_t SELL_PAD = 0xFFFFu;         // padding entry in the u16 streams (mapped to the zero slot when a group is loaded)
static const uint32_t SELL_META_WORDS = 64;       // u32 words of lane metadata in front of a slice that has split segments
// The entries of a slice are stored four trips to a 64-bit word: lane l's entries j = 4q .. 4q + 3 are the u16 quarters of word
// q * 64 + l, so the kernel fetches four indices with ONE conflict-free 8-byte LDS read per lane (an index read per entry was a
// quarter of the kernel's LDS instructions).  A slice therefore occupies quad_width(width) * 64 u16; the trips beyond `width` hold padding.
KAMD_HD uint32_t quad_width(uint32_t width) { return (width + 3u) & ~3u; }
// lane metadata: segment id | reach << 16 (lanes below that belong to the same segment) | LAST << 23 | ACTIVE << 24
static const uint32_t META_LAST = 1u << 23, META_ACTIVE = 1u << 24;

// how a segment of `len` entries is spread over lanes: more than `cap` entries (1 <= cap <= 64) -> ceil(len / cap) lanes, at
// most 64.  A small cap makes every slice short (balanced work for the wavefronts of a workgroup, a short dependent chain per
// lane) at the price of one segmented scan in the slices that hold the split segments.
KAMD_HD uint32_t seg_lanes(uint32_t len, uint32_t cap) { const uint32_t nv = (len + cap - 1) / cap; return len <= cap ? 1u : (nv > SELL_LANES ? SELL_LANES : nv); }
KAMD_HD uint32_t seg_vlen(uint32_t len, uint32_t cap) { const uint32_t nv = seg_lanes(len, cap); return (len + nv - 1) / nv; }

// slice descriptor (two u32 words)
//   d0 = offset of the slice in the group's u16 stream (in u16 units; the metadata words, if any, come first) | HAS_META << 31
//   d1 = width (trips of the inner loop) | first segment << 16 (plain slices)
static const uint32_t DESC_META = 1u << 31;

// Result of laying out one direction of one group.  The sink S receives
//   seg(old_index, new_id, first_lane_abs, nv, vlen)     for every segment (first_lane_abs = slice * 64 + lane)
//   slice(index, d0, d1)                                  for every slice
//   meta(slice_off_u16, lane, word)                       lane metadata of slices with split segments
// and the function returns {number of slices, u16 units of the stream}.
struct LayoutSize { uint32_t n_slices; uint32_t n_u16; };

struct NullSink {
  static const bool wants_segments = false;
  KAMD_HD void seg(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) const {}
  KAMD_HD void slice(uint32_t, uint32_t, uint32_t) const {}
  KAMD_HD void meta(uint32_t, uint32_t, uint32_t) const {}
};

// len[i * stride], i < n: segment lengths (>= 1) in the caller's ("old") order
// scratch: 3 * (SELL_LANES + 1) words with stride `ss` between consecutive words (the device keeps it in LDS, thread-transposed:
// dynamically indexed private arrays would live in scratch memory, one global round trip per access)
static const uint32_t LAYOUT_SCRATCH_WORDS = 3 * (SELL_LANES + 1);
template <class Sink>
KAMD_HD LayoutSize layout_group(const uint32_t* len, uint32_t n, uint32_t cap, Sink& sink, uint32_t* scratch, uint32_t ss) {
#define hist(b) scratch[(size_t)(b) * ss]
#define start(b) scratch[(size_t)(SELL_LANES + 1 + (b)) * ss]
#define cur(b) scratch[(size_t)(2 * (SELL_LANES + 1) + (b)) * ss]
  for (uint32_t b = 0; b <= SELL_LANES; b++) hist(b) = 0;
  uint32_t n_split = 0;
  for (uint32_t i = 0; i < n; i++) { const uint32_t l = len[i]; if (l > cap) ++n_split; else ++hist(l); }
  // 1. split segments, in the caller's order, packed into slices without straddling
  uint32_t lane = 0;          // absolute lane position (slice * 64 + lane in slice) of the next free lane
  uint32_t next_id = 0;
  for (uint32_t i = 0; i < n && n_split; i++) {
    const uint32_t l = len[i];
    if (l <= cap) continue;
    const uint32_t nv = seg_lanes(l, cap), vl = seg_vlen(l, cap);
    if ((lane % SELL_LANES) + nv > SELL_LANES) lane = (lane / SELL_LANES + 1) * SELL_LANES;   // does not fit: next slice
    sink.seg(i, next_id++, lane, nv, vl);
    lane += nv;
  }
  const uint32_t split_lanes = lane;
  // 2. the other segments by decreasing length: start lane of every length
  { uint32_t c = split_lanes; for (uint32_t b = SELL_LANES; b >= 1; b--) { start(b) = c; c += hist(b); } start(0) = c; }
  const uint32_t total_lanes = start(0);
  if (Sink::wants_segments) {
    for (uint32_t b = 0; b <= SELL_LANES; b++) cur(b) = 0;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t l = len[i];
      if (l > cap) continue;
      const uint32_t p = start(l) + cur(l);
      cur(l) = cur(l) + 1;
      sink.seg(i, n_split + (p - split_lanes), p, 1u, l);
    }
  }
  // 3. slices: width = longest lane of the slice; slices that hold split lanes carry metadata
  const uint32_t n_slices = (total_lanes + SELL_LANES - 1) / SELL_LANES;
  const uint32_t meta_slices = (split_lanes + SELL_LANES - 1) / SELL_LANES;   // slices 0 .. meta_slices - 1 contain split lanes
  // widths need the longest lane per slice: split lanes -> second walk over the split segments; plain part -> the length
  // whose start range covers the slice's first lane
  uint32_t off = 0;
  uint32_t si = 0;
  // walk the split segments again, slice by slice
  uint32_t i_split = 0, lane2 = 0;
  for (si = 0; si < n_slices; si++) {
    const uint32_t lo = si * SELL_LANES, hi = lo + SELL_LANES;
    uint32_t width = 0;
    const bool has_meta = si < meta_slices;
    if (has_meta) {
      // metadata defaults: inactive lanes
      for (uint32_t l = 0; l < SELL_LANES; l++) sink.meta(off, l, 0u);
      // split segments whose lanes fall into [lo, hi)
      uint32_t id = 0, ln = 0;
      // (re-walk from the beginning: the number of split segments per group is small)
      ln = 0; id = 0;
      for (uint32_t i = 0; i < n; i++) {
        const uint32_t l = len[i];
        if (l <= cap) continue;
        const uint32_t nv = seg_lanes(l, cap), vl = seg_vlen(l, cap);
        if ((ln % SELL_LANES) + nv > SELL_LANES) ln = (ln / SELL_LANES + 1) * SELL_LANES;
        if (ln >= lo && ln < hi) {
          if (vl > width) width = vl;
          for (uint32_t v = 0; v < nv; v++)
            sink.meta(off, ln - lo + v, id | (v << 16) | (v + 1 == nv ? META_LAST : 0u) | META_ACTIVE);
        }
        ln += nv; ++id;
      }
      (void)i_split; (void)lane2;
    }
    // plain lanes of this slice: lanes >= split_lanes
    const uint32_t plo = lo > split_lanes ? lo : split_lanes;
    if (plo < hi && plo < total_lanes) {
      // length of the lane at position plo: largest b with start[b] <= plo < start[b - 1 ... ] (start decreases with b)
      uint32_t b = SELL_LANES;
      while (b > 1 && !(plo >= start(b) && plo < start(b) + hist(b))) --b;
      if (b > width) width = b;
      if (has_meta) {
        const uint32_t phi = hi < total_lanes ? hi : total_lanes;
        for (uint32_t p = plo; p < phi; p++) sink.meta(off, p - lo, (n_split + (p - split_lanes)) | META_LAST | META_ACTIVE);
      }
    }
    const uint32_t first_seg = has_meta ? 0u : n_split + (lo - split_lanes);
    sink.slice(si, off | (has_meta ? DESC_META : 0u), width | (first_seg << 16));
    off += (has_meta ? 2 * SELL_META_WORDS : 0u) + quad_width(width) * SELL_LANES;
  }
#undef hist
#undef start
#undef cur
  return LayoutSize{n_slices, off};
}

// LDS bytes of a group in the kernel: 8-byte arrays (alpha, a: current and next, with a zero slot; single, eff; g with a zero
// slot; cw), slice descriptors, both u16 streams
KAMD_HD uint64_t group_bytes(uint64_t rows, uint64_t tr, uint64_t row_slices, uint64_t col_slices, uint64_t row_u16, uint64_t col_u16) {
  return (tr + 1) * 8 * 4 + tr * 8 * 2 + (rows + 1) * 8 + rows * 8 + (row_slices + col_slices) * 8 + ((row_u16 + col_u16 + 3) & ~3ULL) * 2 + 64;
}

// ---- the plan ------------------------------------------------------------------------------------------------------------
struct Plan {
  uint32_t n_groups = 0;
  uint32_t n_small = 0;                          // groups [0, n_small): small components only, one wavefront each (k_em_sell_wave)
  uint64_t T = 0;
  uint32_t cap = SELL_LANES;                     // entries per lane above which a segment is split
  std::vector<uint32_t> row_base, tr_base;       // [n_groups + 1]
  std::vector<uint32_t> rslice_base, cslice_base;   // [n_groups + 1] first slice descriptor of a group (rows / columns)
  std::vector<uint64_t> rell_base, cell_base;    // [n_groups + 1] first u16 of a group's stream
  std::vector<uint32_t> rdesc, cdesc;            // 2 words per slice
  std::vector<uint16_t> rell, cell;
  std::vector<uint64_t> cw;                      // [R] count | weight count << 32, new row order
  std::vector<double> single, eff;               // [M] new transcript order
  std::vector<uint32_t> tr_id;                   // [M] transcript id of an m-space slot
  std::vector<double> single_all;                // [T]
  uint64_t max_group_bytes = 0;                  // LDS bytes of the largest group (of the large class when there are two)
  uint64_t max_small_bytes = 0;                  // ... of the largest small group
};

// host reference: CSR plan (kamd_em_local::Plan) -> SELL plan.  Returns 0 = ok, 1 = not applicable (a group exceeds the budget)
struct VecSink {
  static const bool wants_segments = true;
  std::vector<uint32_t>* new_id; std::vector<uint32_t>* lane; std::vector<uint32_t>* nv; std::vector<uint32_t>* vlen;
  std::vector<uint32_t>* desc; std::vector<uint16_t>* ell; uint64_t ell0; uint32_t desc0; uint32_t seg0;
  void seg(uint32_t old, uint32_t id, uint32_t ln, uint32_t n, uint32_t vl) const { (*new_id)[seg0 + old] = id; (*lane)[seg0 + old] = ln; (*nv)[seg0 + old] = n; (*vlen)[seg0 + old] = vl; }
  void slice(uint32_t i, uint32_t d0, uint32_t d1) const { if (desc) { (*desc)[2 * (desc0 + i)] = d0; (*desc)[2 * (desc0 + i) + 1] = d1; } }
  void meta(uint32_t off, uint32_t l, uint32_t w) const { if (ell) { (*ell)[ell0 + off + 2 * l] = (uint16_t)w; (*ell)[ell0 + off + 2 * l + 1] = (uint16_t)(w >> 16); } }
};
// u16 position of entry q of a segment placed at absolute lane `lane` with (nv, vlen), given the slice descriptors of its group
KAMD_HD uint64_t entry_pos(const uint32_t* desc, uint32_t lane, uint32_t vlen, uint32_t q) {
  const uint32_t v = q / vlen, j = q % vlen;
  const uint32_t ln = lane + v, s = ln / SELL_LANES;
  const uint32_t d0 = desc[2 * s];
  const uint64_t base = (uint64_t)(d0 & ~DESC_META) + ((d0 & DESC_META) ? 2 * SELL_META_WORDS : 0u);
  return base + (uint64_t)(j / 4) * (4 * SELL_LANES) + (uint64_t)(ln % SELL_LANES) * 4 + (j % 4);
}

inline int from_csr_plan(const kamd_em_local::Plan& C, uint64_t budget_bytes, Plan* P, uint32_t cap = SELL_LANES) {
  const uint32_t ng = C.n_groups;
  P->cap = cap;
  P->n_groups = ng; P->n_small = C.n_small; P->T = C.T; P->row_base = C.row_base; P->tr_base = C.tr_base; P->single_all = C.single_all;
  const uint64_t R = C.row_base[ng], M = C.tr_base[ng];
  std::vector<uint32_t> rlen(R), clen(M), rnew(R), cnew(M), rlane(R), clane(M), rnv(R), cnv(M), rvl(R), cvl(M);
  P->rslice_base.assign(ng + 1, 0); P->cslice_base.assign(ng + 1, 0); P->rell_base.assign(ng + 1, 0); P->cell_base.assign(ng + 1, 0);
  P->max_group_bytes = 0;
  for (uint32_t g = 0; g < ng; g++) {
    const kamd_em_local::Group G = C.group(g);
    for (uint32_t r = 0; r < G.n_rows; r++) rlen[C.row_base[g] + r] = G.row_ptr[r + 1] - G.row_ptr[r];
    for (uint32_t t = 0; t < G.n_tr; t++) clen[C.tr_base[g] + t] = G.col_ptr[t + 1] - G.col_ptr[t];
    if (G.n_rows >= SELL_PAD || G.n_tr >= SELL_PAD) return 1;
    NullSink ns;
    uint32_t scratch[LAYOUT_SCRATCH_WORDS];
    const LayoutSize lr = layout_group(rlen.data() + C.row_base[g], G.n_rows, cap, ns, scratch, 1);
    const LayoutSize lc = layout_group(clen.data() + C.tr_base[g], G.n_tr, cap, ns, scratch, 1);
    const uint64_t gb = group_bytes(G.n_rows, G.n_tr, lr.n_slices, lc.n_slices, lr.n_u16, lc.n_u16);
    if (gb > budget_bytes) return 1;
    if (g < P->n_small) { if (gb > P->max_small_bytes) P->max_small_bytes = gb; }
    else if (gb > P->max_group_bytes) P->max_group_bytes = gb;
    P->rslice_base[g + 1] = P->rslice_base[g] + lr.n_slices; P->cslice_base[g + 1] = P->cslice_base[g] + lc.n_slices;
    P->rell_base[g + 1] = P->rell_base[g] + lr.n_u16; P->cell_base[g + 1] = P->cell_base[g] + lc.n_u16;
  }
  P->rdesc.assign(2 * (size_t)P->rslice_base[ng], 0); P->cdesc.assign(2 * (size_t)P->cslice_base[ng], 0);
  P->rell.assign(P->rell_base[ng], (uint16_t)SELL_PAD); P->cell.assign(P->cell_base[ng], (uint16_t)SELL_PAD);
  for (uint32_t g = 0; g < ng; g++) {
    const kamd_em_local::Group G = C.group(g);
    VecSink sr{&rnew, &rlane, &rnv, &rvl, &P->rdesc, &P->rell, P->rell_base[g], P->rslice_base[g], C.row_base[g]};
    VecSink sc{&cnew, &clane, &cnv, &cvl, &P->cdesc, &P->cell, P->cell_base[g], P->cslice_base[g], C.tr_base[g]};
    uint32_t scratch[LAYOUT_SCRATCH_WORDS];
    layout_group(rlen.data() + C.row_base[g], G.n_rows, cap, sr, scratch, 1);
    layout_group(clen.data() + C.tr_base[g], G.n_tr, cap, sc, scratch, 1);
  }
  // entries (indices renamed to the other direction's new ids) and the per-segment constants in the new order
  P->cw.assign(R, 0); P->single.assign(M, 0.0); P->eff.assign(M, 0.0); P->tr_id.assign(M, 0);
  for (uint32_t g = 0; g < ng; g++) {
    const kamd_em_local::Group G = C.group(g);
    const uint32_t r0 = C.row_base[g], t0 = C.tr_base[g];
    const uint32_t* rd = P->rdesc.data() + 2 * (size_t)P->rslice_base[g];
    const uint32_t* cd = P->cdesc.data() + 2 * (size_t)P->cslice_base[g];
    for (uint32_t r = 0; r < G.n_rows; r++) {
      P->cw[r0 + rnew[r0 + r]] = G.cw[r];
      for (uint32_t q = 0, j = G.row_ptr[r]; j < G.row_ptr[r + 1]; j++, q++)
        P->rell[P->rell_base[g] + entry_pos(rd, rlane[r0 + r], rvl[r0 + r], q)] = (uint16_t)cnew[t0 + G.row_tr[j]];
    }
    for (uint32_t t = 0; t < G.n_tr; t++) {
      const uint32_t m = t0 + cnew[t0 + t];
      P->single[m] = G.single[t]; P->eff[m] = G.eff[t]; P->tr_id[m] = C.tr_id[t0 + t];
      for (uint32_t q = 0, j = G.col_ptr[t]; j < G.col_ptr[t + 1]; j++, q++)
        P->cell[P->cell_base[g] + entry_pos(cd, clane[t0 + t], cvl[t0 + t], q)] = (uint16_t)rnew[r0 + G.col_row[j]];
    }
  }
  return 0;
}

// ---- host model of the kernel's round (one slice at a time, lanes serially): the semantics k_em_sell must reproduce ----------
// sums of one direction: out[seg] = sum over the segment's entries of src[index] (src has a zero slot at `zero`)
inline void host_pass(const uint32_t* desc, uint32_t n_slices, const uint16_t* ell, uint32_t n_segs, const double* src, uint32_t zero,
                      double* out) {
  for (uint32_t s = 0; s < n_slices; s++) {
    const uint32_t d0 = desc[2 * s], d1 = desc[2 * s + 1];
    const bool has_meta = (d0 & DESC_META) != 0;
    const uint32_t off = d0 & ~DESC_META, width = d1 & 0xFFFFu, first_seg = d1 >> 16;
    const uint16_t* e = ell + off + (has_meta ? 2 * SELL_META_WORDS : 0);
    double lane_sum[SELL_LANES];
    for (uint32_t l = 0; l < SELL_LANES; l++) {
      double S = 0.0;
      for (uint32_t j = 0; j < width; j++) { const uint32_t ix = e[(j / 4) * (4 * SELL_LANES) + l * 4 + (j % 4)]; S += src[ix == SELL_PAD ? zero : ix]; }
      lane_sum[l] = S;
    }
    if (!has_meta) {
      for (uint32_t l = 0; l < SELL_LANES; l++) if (first_seg + l < n_segs) out[first_seg + l] = lane_sum[l];
    } else {
      for (uint32_t l = 0; l < SELL_LANES; l++) {
        const uint32_t w = (uint32_t)ell[off + 2 * l] | ((uint32_t)ell[off + 2 * l + 1] << 16);
        if (!(w & META_ACTIVE) || !(w & META_LAST)) continue;
        const uint32_t reach = (w >> 16) & 0x7Fu;
        double S = 0.0;
        for (uint32_t v = 0; v <= reach; v++) S += lane_sum[l - reach + v];
        out[w & 0xFFFFu] = S;
      }
    }
  }
}

struct CpuBackend {
  const Plan& P;
  std::vector<double> alpha, a, ck_alpha, ck_a;
  explicit CpuBackend(const Plan& p) : P(p) {
    const uint64_t M = P.tr_base[P.n_groups];
    alpha.assign(M, 1.0 / (double)P.T);
    a.resize(M);
    for (uint64_t m = 0; m < M; m++) a[m] = alpha[m] / P.eff[m];
  }
  void checkpoint() { ck_alpha = alpha; ck_a = a; }
  void restore() { alpha = ck_alpha; a = ck_a; }
  const std::vector<double>& host_alpha() { return alpha; }
  void run(int n, int clamp, int* hist) {
    std::vector<double> av, g, S, acc;
    for (uint32_t gi = 0; gi < P.n_groups; gi++) {
      const uint32_t r0 = P.row_base[gi], nR = P.row_base[gi + 1] - r0, t0 = P.tr_base[gi], nT = P.tr_base[gi + 1] - t0;
      double* al = alpha.data() + t0; double* aa = a.data() + t0;
      if (clamp) for (uint32_t t = 0; t < nT; t++) if (al[t] < 1e-7 / 10.0) { al[t] = 0.0; aa[t] = 0.0; }
      av.assign(nT + 1, 0.0); g.assign(nR + 1, 0.0); S.assign(nR, 0.0); acc.assign(nT, 0.0);
      for (int i = 0; i < n; i++) {
        for (uint32_t t = 0; t < nT; t++) av[t] = aa[t];
        host_pass(P.rdesc.data() + 2 * (size_t)P.rslice_base[gi], P.rslice_base[gi + 1] - P.rslice_base[gi], P.rell.data() + P.rell_base[gi], nR,
                  av.data(), nT, S.data());
        for (uint32_t r = 0; r < nR; r++) {
          const uint64_t w = P.cw[r0 + r];
          const uint32_t cnt = (uint32_t)w, wc = (uint32_t)(w >> 32);
          g[r] = (cnt == 0 || (double)wc * S[r] < 4.9406564584124654e-324) ? 0.0 : (double)cnt / S[r];
        }
        host_pass(P.cdesc.data() + 2 * (size_t)P.cslice_base[gi], P.cslice_base[gi + 1] - P.cslice_base[gi], P.cell.data() + P.cell_base[gi], nT,
                  g.data(), nR, acc.data());
        int ch = 0;
        for (uint32_t t = 0; t < nT; t++) {
          const double nx = P.single[t0 + t] + aa[t] * acc[t];
          if (nx > 1e-2 && (fabs(nx - al[t]) / nx) > 1e-2) ++ch;
          al[t] = nx; aa[t] = nx / P.eff[t0 + t];
        }
        if (hist) hist[i] += ch;
      }
    }
  }
};

}  // namespace kamd_em_sell
