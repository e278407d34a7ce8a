// kamd_io.hip -- FASTQ text resident in HBM -> packed reads (FastqSequenceReader::fetchSequences)
#include "kamd_dev.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// FASTQ text resident in HBM -> packed reads (kamd_fq_core.h; the parsing half of FastqSequenceReader::fetchSequences,
// src/ProcessReads.cpp:3128-3267 + kseq_read src/kseq.h:174-215, for strict 4-line records).  A unit of text starts at a record
// and holds whole records; per file:
//   k_fq_count    newlines per 4 KB tile (one block per tile, 16 bytes per thread)
//   k_fq_scan     exclusive scan of the tile counts (one block: a unit has a few 10^4 tiles)
//   k_fq_fill     position of every newline, in order
// then k_fq_records checks the shape of every record (kamd_fq::fq_check_record) and emits {address, length} of its sequence
// line, mates interleaved, and k_fq_pack writes the 2-bit records kernel A reads.
// ------------------------------------------------------------------------------------------------------------------
struct FqResult { u32 max_len; u32 n_bad; u64 first_bad; u64 n_lines[2]; };
__device__ __forceinline__ u32 fq_thread_mask(const char* __restrict__ text, u64 n_bytes, u64 b0) {   // newline bits of bytes [b0, b0 + 16)
  u32 m = 0;
  if (b0 + 16 <= n_bytes) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + b0);   // (text is 16-byte aligned, b0 a multiple of 16)
    m = kamd_fq::nl_mask4(v.x) | (kamd_fq::nl_mask4(v.y) << 4) | (kamd_fq::nl_mask4(v.z) << 8) | (kamd_fq::nl_mask4(v.w) << 12);
  } else {
    for (u32 i = 0; i < 16 && b0 + i < n_bytes; i++) if (text[b0 + i] == '\n') m |= 1u << i;
  }
  return m;
}
__global__ __launch_bounds__(BLOCK) void k_fq_count(const char* __restrict__ text, u64 n_bytes, u32* tile_count) {
  __shared__ u32 wsum[BLOCK / 64];
  const u64 b0 = (u64)blockIdx.x * kamd_fq::FQ_TILE + (u64)threadIdx.x * 16;
  u32 c = b0 < n_bytes ? (u32)__popc(fq_thread_mask(text, n_bytes, b0)) : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int j = 0; j < BLOCK / 64; j++) t += wsum[j]; tile_count[blockIdx.x] = t; }
}
constexpr int FQ_SCAN_BLOCK = 1024;
__global__ __launch_bounds__(FQ_SCAN_BLOCK) void k_fq_scan(const u32* __restrict__ tile_count, u32 n_tiles, u32* tile_base, u64* total) {
  __shared__ u32 part[FQ_SCAN_BLOCK];
  const u32 per = (n_tiles + FQ_SCAN_BLOCK - 1) / FQ_SCAN_BLOCK;
  const u32 a = min(threadIdx.x * per, n_tiles), e = min(a + per, n_tiles);
  u32 sum = 0;
  for (u32 i = a; i < e; i++) sum += tile_count[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < FQ_SCAN_BLOCK; d <<= 1) {   // Hillis-Steele over the 1024 partial sums
    const u32 v = threadIdx.x >= (u32)d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - sum;
  for (u32 i = a; i < e; i++) { tile_base[i] = run; run += tile_count[i]; }
  if (threadIdx.x == FQ_SCAN_BLOCK - 1) *total = (u64)part[FQ_SCAN_BLOCK - 1];
}
__global__ __launch_bounds__(BLOCK) void k_fq_fill(const char* __restrict__ text, u64 n_bytes, const u32* __restrict__ tile_base, u32* nlpos, u64 cap) {
  __shared__ u32 wsum[BLOCK / 64];
  const u64 b0 = (u64)blockIdx.x * kamd_fq::FQ_TILE + (u64)threadIdx.x * 16;
  u32 m = b0 < n_bytes ? fq_thread_mask(text, n_bytes, b0) : 0u;
  const u32 c = (u32)__popc(m);
  const u32 incl = wave_incl_scan(c);
  if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  u32 before = 0;
  for (u32 j = 0; j < (threadIdx.x >> 6); j++) before += wsum[j];
  u64 o = (u64)tile_base[blockIdx.x] + before + incl - c;
  while (m) {
    const int i = __ffs((int)m) - 1;
    m &= m - 1;
    if (o < cap) nlpos[o] = (u32)(b0 + (u64)i);
    ++o;
  }
}
struct FqFiles { const char* text[2]; const u32* nlpos[2]; u64 n_bytes[2]; int n; };
__global__ __launch_bounds__(BLOCK) void k_fq_records(FqFiles F, u64 n_records, u64* recs, FqResult* res) {
  u32 mx = 0, bad = 0; u64 first = ~0ULL;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_records; j += (u64)gridDim.x * blockDim.x) {
    for (int f = 0; f < F.n; f++) {
      const u32* nl = F.nlpos[f] + 4 * j;
      const u64 l0 = j ? (u64)nl[-1] + 1 : 0ULL;
      kamd_fq::Record r; r.ok = false; r.seq_off = 0; r.seq_len = 0;
      // (the positions are only trusted after this test: a unit with fewer newlines than the caller promised leaves stale entries)
      if (l0 <= nl[0] && nl[0] < nl[1] && nl[1] < nl[2] && nl[2] < nl[3] && (u64)nl[3] < F.n_bytes[f])
        r = kamd_fq::fq_check_record(F.text[f], l0, nl[0], nl[1], nl[2], nl[3]);
      const bool ok = r.ok && r.seq_len <= kamd_fq::FQ_MAX_READ;
      recs[j * (u64)F.n + f] = kamd_fq::rec_word(F.text[f] + r.seq_off, ok ? r.seq_len : 0u);
      if (r.ok) mx = max(mx, r.seq_len);   // (a read beyond 65535 bases is reported through max_len, like the host reader does)
      else { ++bad; first = min(first, j); }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    mx = max(mx, (u32)__shfl_down(mx, d, 64)); bad += __shfl_down(bad, d, 64);
    const u64 o = __shfl_down(first, d, 64); first = min(first, o);
  }
  if (lane_id() == 0) {
    if (mx) atomicMax(&res->max_len, mx);
    if (bad) { atomicAdd(&res->n_bad, bad); atomicMin(&res->first_bad, first); }
  }
}
// 2-bit codes and "not ACGT" flags of four bases held in one dword (byte i -> bits 2i.. of *code, bit i of the result), branch-free:
// for A C G T (either case) the code is ((c >> 1) ^ (c >> 2)) & 3 = 0 1 2 3; anything else gets code 0 and its flag
__device__ __forceinline__ u32 pack4(u32 w, u32* code) {
  const u32 u = w & 0xDFDFDFDFu;   // (KmerIterator.cpp:12 masks with 0xDF)
  auto eq = [](u32 x, u32 c) { const u32 y = x ^ (c * 0x01010101u); const u32 t = (y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu; return ~(t | y | 0x7F7F7F7Fu); };   // 0x80 per equal byte
  const u32 ok = eq(u, 'A') | eq(u, 'C') | eq(u, 'G') | eq(u, 'T');
  const u32 c2 = ((w >> 1) ^ (w >> 2)) & 0x03030303u & ((ok >> 7) * 3u);   // codes of the valid bytes, 0 elsewhere
  *code = (c2 & 3u) | ((c2 >> 6) & 0xCu) | ((c2 >> 12) & 0x30u) | ((c2 >> 18) & 0xC0u);
  const u32 bad = ~ok & 0x80808080u;
  return ((bad >> 7) & 1u) | ((bad >> 14) & 2u) | ((bad >> 21) & 4u) | ((bad >> 28) & 8u);
}
// One thread per 32 bases of a read: eight dwords of text in, two sequence words and one mask word out (the packed buffer was zeroed:
// padding words stay 0, the has-N flag is OR-ed in by the rare thread that meets a base that is not ACGT).  The bytes behind a read's
// end that the last thread loads belong to the record's '+' and quality lines -- always inside the text.
__global__ __launch_bounds__(BLOCK) void k_fq_pack(const u64* __restrict__ recs, u64 n_reads, int groups, int seq_words, int rec_words, u32* out,
                                                   uint16_t* out_len) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 r = gid / (u64)groups;
  const int g = (int)(gid % (u64)groups);
  if (r >= n_reads) return;
  const u64 rw = recs[r];
  const int L = (int)(rw >> 48);
  if (g == 0) out_len[r] = (uint16_t)L;
  const int b0 = g * 32;
  if (b0 >= L) return;
  const char* p = reinterpret_cast<const char*>((uintptr_t)(rw & 0xFFFFFFFFFFFFULL)) + b0;
  u32 lo = 0, hi = 0, mask = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u32 w;
    __builtin_memcpy(&w, p + 4 * j, 4);   // (unaligned: the hardware takes it as one load in the default access mode)
    u32 code;
    const u32 bad = pack4(w, &code);
    if (j < 4) lo |= code << (8 * j); else hi |= code << (8 * (j - 4));
    mask |= bad << (4 * j);
  }
  const int left = L - b0;   // bases of this group that belong to the read
  if (left < 32) {
    const u64 keep = (1ULL << (2 * left)) - 1;
    lo &= (u32)keep; hi &= (u32)(keep >> 32);
    mask &= (1u << left) - 1u;
  }
  u32* o = out + r * (u64)rec_words;
  o[2 * g] = lo;
  if (left > 16) o[2 * g + 1] = hi;
  o[seq_words + g] = mask;
  if (mask) atomicOr(&o[seq_words - 1], kamd::REC_FLAG_HAS_N);
}

// kamd_pack_reads_device: reads given as {offset, length} into one buffer of ASCII bases -- the scheme of k_fq_pack (one thread per 32
// bases, dword loads, branch-free codes) instead of one thread per output word that walked its 16 / 32 bases byte by byte and, for the
// flag word, the whole read (9.7 ms per 2 M PE-100 pairs, 86 G scalar instructions: profiles/r03_pmc_sq_tcc_per_kernel.csv).  Nothing
// lies behind the last read of the buffer, so the dwords of a read's last, partial group are only loaded as far as the read goes (its
// final bytes one by one).  The output buffer was zeroed by the caller.
__global__ __launch_bounds__(BLOCK) void k_pack_reads(const char* __restrict__ seqs, const u64* __restrict__ off, const int32_t* __restrict__ len,
                                                      u64 n_reads, int groups, int seq_words, int rec_words, u32* out, uint16_t* out_len) {
  const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 r = gid / (u64)groups;
  const int g = (int)(gid % (u64)groups);
  if (r >= n_reads) return;
  const int L = min(len[r], (seq_words - 1) * 16);   // (a length beyond max_len is the caller's error: nothing behind the record's sequence words is read or written)
  if (g == 0) out_len[r] = (uint16_t)len[r];
  const int b0 = g * 32;
  if (b0 >= L) return;
  const char* p = seqs + off[r] + b0;
  const int left = L - b0;   // bases of this group that belong to the read
  u32 lo = 0, hi = 0, mask = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u32 w = 0x41414141u;   // 'A': code 0, no flag -- what the bytes behind the read's end count as
    const int have = left - 4 * j;
    if (have >= 4) __builtin_memcpy(&w, p + 4 * j, 4);   // (unaligned: the hardware takes it as one load in the default access mode)
    else if (have > 0) for (int b = 0; b < have; b++) w = (w & ~(0xFFu << (8 * b))) | ((u32)(unsigned char)p[4 * j + b] << (8 * b));
    u32 code;
    const u32 bad = pack4(w, &code);
    if (j < 4) lo |= code << (8 * j); else hi |= code << (8 * (j - 4));
    mask |= bad << (4 * j);
  }
  u32* o = out + r * (u64)rec_words;
  o[2 * g] = lo;
  if (left > 16) o[2 * g + 1] = hi;
  o[seq_words + g] = mask;
  if (mask) atomicOr(&o[seq_words - 1], kamd::REC_FLAG_HAS_N);
}

}  // namespace

extern "C" int kamd_pack_reads_device(kamd_ctx* c, const char* d_seqs, const uint64_t* d_off, const int32_t* d_len,
                                      uint64_t n_reads, int32_t max_len, uint32_t* d_out_words, uint16_t* d_out_len) {
  if (!c) return kamd::fail(-1, "kamd_pack_reads_device: null context");
  if (max_len <= 0 || max_len > 65535) return kamd::fail(-1, "kamd_pack_reads: max_len must be in [1, 65535]");
  HIPC(hipSetDevice(c->device));
  const int seq_words = (max_len + 15) / 16 + 1;
  const int rec_words = (int)kamd_packed_record_words(max_len);
  const u64 total = n_reads * (u64)rec_words;
  if (total == 0) return 0;
  const int groups = (max_len + 31) / 32;
  HIPC(hipMemsetAsync(d_out_words, 0, total * sizeof(u32), c->stream));   // padding words and the has-N flag start from 0
  hipLaunchKernelGGL(k_pack_reads, dim3(grid_for(n_reads * (u64)groups, BLOCK)), dim3(BLOCK), 0, c->stream, d_seqs, (const u64*)d_off, d_len,
                     (u64)n_reads, groups, seq_words, rec_words, d_out_words, d_out_len);
  HIPC(hipGetLastError());
  return 0;
}

// Strict 4-line FASTQ text already in HBM -> the packed batch kernel A reads (see k_fq_* above).  The caller cut the text into
// units at record boundaries (it counted the newlines while the bytes went by) and says how many records a unit holds; the device
// finds the lines, checks every record's shape and notes {address, length} of every sequence (kamd_fastq_unit_parse: mate 1 /
// mate 2 of record j become reads 2j / 2j + 1 of the unit); kamd_fastq_batch_pack then packs the reads of all units parsed since
// the last batch in one launch -- a batch should be millions of reads (kamd_pseudoalign has fixed costs per call), a unit is what
// one copy brings.  One host synchronisation per unit (the shape check decides whether the input may be used at all).
extern "C" int kamd_fastq_unit_parse(kamd_ctx* c, const char* const* d_text, const uint64_t* n_bytes, int32_t n_files, uint64_t n_records,
                                     kamd_fastq_unit* out) {
  if (!c || !d_text || !n_bytes || !out) return kamd::fail(-1, "kamd_fastq_unit_parse: null argument");
  if (n_files < 1 || n_files > 2) return kamd::fail(-1, "kamd_fastq_unit_parse: n_files must be 1 or 2");
  if (c->fq_batch_reads && c->fq_batch_files != n_files) return kamd::fail(-1, "kamd_fastq_unit_parse: the batch under construction has another number of files");
  memset(out, 0, sizeof *out);
  out->first_bad_record = ~0ULL;
  if (n_records == 0) return 0;
  for (int f = 0; f < n_files; f++) {
    if (!d_text[f] || n_bytes[f] == 0 || n_bytes[f] >= 0xFFFFFFFFULL) return kamd::fail(-1, "kamd_fastq_unit_parse: a unit holds 1 .. 2^32-2 bytes of text per file");
    if ((uintptr_t)d_text[f] & 15) return kamd::fail(-1, "kamd_fastq_unit_parse: the text must be 16-byte aligned");
    if (n_records > n_bytes[f] / 8) return kamd::fail(-1, "kamd_fastq_unit_parse: more records than the text can hold");
  }
  HIPC(hipSetDevice(c->device));
  if (!c->fq_host) { if (hipHostMalloc(&c->fq_host, sizeof(FqResult), hipHostMallocDefault) != hipSuccess) return kamd::fail(-100, "kamd_fastq_unit_parse: pinned allocation failed"); }
  u64 tiles[2] = {0, 0}, tile_off[2] = {0, 0}, all_tiles = 0;
  for (int f = 0; f < n_files; f++) { tiles[f] = (n_bytes[f] + kamd_fq::FQ_TILE - 1) / kamd_fq::FQ_TILE; tile_off[f] = all_tiles; all_tiles += tiles[f]; }
  if (int rc = c->fq_tiles.ensure(2 * all_tiles * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->fq_res.ensure(sizeof(FqResult), 0, c->stream)) return rc;
  const u64 nl_cap = 4 * n_records + 4;
  for (int f = 0; f < n_files; f++) if (int rc = c->fq_nlpos[f].ensure(nl_cap * sizeof(u32), 0, c->stream)) return rc;
  const u64 have = c->fq_batch_reads, n_reads = n_records * (u64)n_files;
  if (int rc = c->fq_recs.ensure((have + n_reads) * sizeof(u64), have * sizeof(u64), c->stream)) return rc;
  FqResult* res = c->fq_res.as<FqResult>();
  FqResult init{}; init.first_bad = ~0ULL;
  FqResult* h = (FqResult*)c->fq_host;
  *h = init;
  HIPC(hipMemcpyAsync(res, h, sizeof(FqResult), hipMemcpyHostToDevice, c->stream));
  FqFiles F{};
  F.n = n_files;
  for (int f = 0; f < n_files; f++) {
    u32* cnt = c->fq_tiles.as<u32>() + tile_off[f];
    u32* base = c->fq_tiles.as<u32>() + all_tiles + tile_off[f];
    hipLaunchKernelGGL(k_fq_count, dim3((unsigned)tiles[f]), dim3(BLOCK), 0, c->stream, d_text[f], (u64)n_bytes[f], cnt);
    hipLaunchKernelGGL(k_fq_scan, dim3(1), dim3(FQ_SCAN_BLOCK), 0, c->stream, (const u32*)cnt, (u32)tiles[f], base, &res->n_lines[f]);
    hipLaunchKernelGGL(k_fq_fill, dim3((unsigned)tiles[f]), dim3(BLOCK), 0, c->stream, d_text[f], (u64)n_bytes[f], (const u32*)base,
                       c->fq_nlpos[f].as<u32>(), nl_cap);
    F.text[f] = d_text[f]; F.nlpos[f] = c->fq_nlpos[f].as<u32>(); F.n_bytes[f] = n_bytes[f];
  }
  hipLaunchKernelGGL(k_fq_records, dim3((unsigned)std::min<u64>(grid_for(n_records, BLOCK), 4096)), dim3(BLOCK), 0, c->stream, F, (u64)n_records,
                     c->fq_recs.as<u64>() + have, res);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(h, res, sizeof(FqResult), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  out->n_items = n_records;
  for (int f = 0; f < n_files; f++)
    if (h->n_lines[f] < 4 * n_records) { out->status = 2; return 0; }   // fewer lines than promised (more: the caller's cut leaves the rest to the next unit -- not an error here)
  if (h->n_bad) { out->status = 1; out->first_bad_record = h->first_bad; return 0; }
  out->max_len = (int32_t)std::max<u32>(h->max_len, 1u);
  if (h->max_len > kamd_fq::FQ_MAX_READ) { out->status = 3; return 0; }   // reads beyond the packed layout's 16-bit lengths
  c->fq_batch_reads = have + n_reads; c->fq_batch_files = n_files;
  c->fq_batch_max_len = std::max(c->fq_batch_max_len, out->max_len);
  return 0;
}
extern "C" int kamd_fastq_batch_pack(kamd_ctx* c, kamd_fastq_unit* out) {
  if (!c || !out) return kamd::fail(-1, "kamd_fastq_batch_pack: null argument");
  memset(out, 0, sizeof *out);
  out->first_bad_record = ~0ULL;
  const u64 n_reads = c->fq_batch_reads;
  if (n_reads == 0) return 0;
  HIPC(hipSetDevice(c->device));
  out->max_len = c->fq_batch_max_len;
  out->n_items = n_reads / (u64)c->fq_batch_files;
  const int seq_words = (out->max_len + 15) / 16 + 1;
  const int rec_words = (int)kamd_packed_record_words(out->max_len);
  const u64 total = n_reads * (u64)rec_words;
  if (int rc = c->fq_words.ensure(total * sizeof(u32), 0, c->stream)) return rc;
  if (int rc = c->fq_len.ensure(n_reads * sizeof(uint16_t), 0, c->stream)) return rc;
  const int groups = (out->max_len + 31) / 32;
  HIPC(hipMemsetAsync(c->fq_words.p, 0, total * sizeof(u32), c->stream));
  hipLaunchKernelGGL(k_fq_pack, dim3(grid_for(n_reads * (u64)groups, BLOCK)), dim3(BLOCK), 0, c->stream, (const u64*)c->fq_recs.as<u64>(), n_reads, groups, seq_words,
                     rec_words, c->fq_words.as<u32>(), c->fq_len.as<uint16_t>());
  HIPC(hipGetLastError());
  out->d_words = c->fq_words.as<u32>(); out->d_len = c->fq_len.as<uint16_t>();
  c->fq_batch_reads = 0; c->fq_batch_max_len = 0; c->fq_batch_files = 0;
  return 0;
}
// one unit = one batch
extern "C" int kamd_fastq_unit_pack(kamd_ctx* c, const char* const* d_text, const uint64_t* n_bytes, int32_t n_files, uint64_t n_records,
                                    kamd_fastq_unit* out) {
  if (c && c->fq_batch_reads) return kamd::fail(-1, "kamd_fastq_unit_pack: a batch is under construction (kamd_fastq_batch_pack first)");
  if (int rc = kamd_fastq_unit_parse(c, d_text, n_bytes, n_files, n_records, out)) return rc;
  if (out->status != 0 || out->n_items == 0) return 0;
  return kamd_fastq_batch_pack(c, out);
}

