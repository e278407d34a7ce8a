/* include/kallisto_amd.h -- C ABI of libkallisto_amd.so, the MI355X (gfx950) implementation of the `kallisto quant`
 * hot path of pachterlab/kallisto v0.51.1.
 *
 * kallisto has no plugin/FFI interface; the hot path is reached by ordinary C++ calls inside one binary.  Each entry
 * point below replaces one of those seams; the reference-side binding a maintainer would add is
 * shown in INTEGRATION.md.  Conventions: plain pointers and sizes only (no C++/torch types), integer return codes
 * (0 = ok, <0 = error; kamd_last_error() holds the message), the caller owns host buffers it passes in, the library
 * owns everything it allocates (device and host) and frees it in the matching *_free / *_destroy call.  Pointers named
 * d_* are DEVICE pointers on the context's GPU; everything else is host memory.  All device work is enqueued on the
 * hipStream_t passed to kamd_ctx_create (0 = the null stream) so that a PyTorch caller can hand over its current stream.
 *
 * Reference interfaces replaced (file:line in the reference tree):
 *   S1  KmerIndex::load                          src/KmerIndex.cpp:1330-1559      -> kamd_index_load / kamd_index_upload
 *   S2  ReadProcessor::processBuffer             src/ProcessReads.cpp:968-1237    -> kamd_pseudoalign
 *       (KmerIndex::match src/KmerIndex.cpp:1698, MinCollector::intersectKmers src/MinCollector.cpp:160,
 *        KmerIndex::mapPair src/KmerIndex.cpp:1622, KmerIndex::findPosition src/KmerIndex.cpp:2188)
 *       MasterProcessor::update / processReads   src/ProcessReads.cpp:424-499,323-334 -> kamd_ec_finalize; over several GPUs kamd_ec_allreduce
 *                                                (kamd_comm: RCCL inside the library) + kamd_em_run_comm
 *       FastqSequenceReader::fetchSequences      src/ProcessReads.cpp:3128-3267   -> kamd_pack_reads (2-bit packing of a parsed batch)
 *   S3  EMAlgorithm ctor + run                   src/EMAlgorithm.h:26-48,95-223   -> kamd_em_run
 *   S4  Bootstrap::run_em / Multinomial::sample  src/Bootstrap.cpp:4-14, src/Multinomial.hpp:33-51 -> kamd_bootstrap, kamd_bootstrap_batch
 *       (the replicate pool of src/Bootstrap.cpp:15-92, src/main.cpp:2764-2782)
 */
#ifndef KALLISTO_AMD_H
#define KALLISTO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAMD_MAX_FRAG_LEN 1000   /* src/MinCollector.h:15 */
#define KAMD_SLOTS_PER_BUCKET 3  /* 3 x {key 8 B, payload 8 B, text position 4 B} + 4 spare bytes = one 64-byte line */

typedef struct kamd_index kamd_index; /* host-side flattened index (owned by the library) */
typedef struct kamd_ctx kamd_ctx;     /* one GPU: device copy of the index + work buffers */

/* Read-only view of the flattened index; arrays stay valid until kamd_index_free. */
typedef struct {
  int32_t k;
  uint64_t n_kmers, n_unitigs, n_blocks, n_uec, n_ecs, ec_nnz, n_targets, dlist_size;
  uint64_t n_buckets;             /* home buckets; table holds (n_buckets + pad_buckets) * KAMD_SLOTS_PER_BUCKET slots */
  uint64_t pad_buckets;
  const uint64_t* table;          /* bucket b = table[8b..8b+7] = {key0 | flags, key1, key2, payload0..2, text positions (3 x u32), spare};
                                     slot index = b * KAMD_SLOTS_PER_BUCKET + j (kallisto_amd/csrc/kamd_core.h) */
  const uint32_t* slot_block;     /* per slot: global block id          (aux table, FLD / positional paths only) */
  const uint32_t* slot_dist;      /* per slot: k-mer offset on its unitig (aux table) */
  const uint32_t* uec_ec;         /* (unitig, transcript-set) class -> de-duplicated transcript-set id */
  const uint64_t* ec_off;         /* [n_ecs + 1] */
  const uint32_t* ec_ids;         /* [ec_nnz] sorted transcript ids per set */
  const uint64_t* unitig_blk_off; /* [n_unitigs + 1] */
  const uint32_t* unitig_len;     /* bp */
  const uint32_t* blk_unitig;     /* per block */
  const uint32_t* blk_lb;
  const uint32_t* blk_ub;
  const uint32_t* blk_ec;
  const uint64_t* blk_pos_off;    /* [n_blocks + 1] into blk_posw / blk_sense (parallel to the block's transcript set) */
  const uint32_t* blk_posw;       /* first raw position word per (block, transcript); bit31 = reverse */
  const uint8_t* blk_sense;       /* 1 forward, 0 reverse, 2 ambiguous */
  const int32_t* target_lens;     /* [n_targets + dlist_size]: the D-list pseudo-targets (ids >= n_targets, off-list) have length 0 */
  const uint32_t* onlist_bits;    /* bit t set = transcript t is on-list */
  uint64_t onlist_words;
  /* D-list (src/KmerIndex.cpp:1386-1403; match(): :1818-1826, :1928-1939): the distinguishing flanking k-mers as a second
   * table of the k-mer table's layout (payload unused), and the hit pushed when a read contains one: um_dummy = the first
   * D-list k-mer looked up in the graph.  n_dbuckets == 0 when the index carries no D-list. */
  const uint64_t* dtable;
  uint64_t n_dbuckets, dpad_buckets;
  uint64_t dummy_slot;            /* slot of the dummy k-mer in `table` */
  uint32_t dummy_uec;             /* its (unitig, transcript-set) class */
  uint32_t dummy_strand;          /* const_UnitigMap::strand of um_dummy */
  /* 2-bit text of all unitigs laid end to end (unitig-forward; base i at bits 2*(i&15) of word i>>4, the reads' packing):
   * a k-mer that match() expects at a known offset of a unitig it already hit is compared with the text first */
  const uint32_t* utext;
  uint64_t utext_words;           /* incl. two words of padding */
  uint64_t text_bases;
  const uint64_t* unitig_gpos;    /* [n_unitigs + 1] first base of every unitig in utext */
  /* layout of `table` (kallisto_amd/csrc/kamd_core.h): 0 = wide, KAMD_SLOTS_PER_BUCKET slots of 20 bytes per line as described above;
   * 1 = compact, four 16-byte slots per line {tag | class, rem_f | rem_b | text position | flags}, exact by quotienting
   * (tag_q low bits of the k-mer's hash, its bits above 32 and the displacement from the home bucket; tag_w bits in all, the
   * displacement in bits [tag_dsh, tag_w): three or four of them, all ones = empty slot).  The D-list table is always wide.  The layout is chosen when the index is loaded:
   * KAMD_TABLE_LAYOUT=wide (default) | compact | auto (compact when its fields fit), KAMD_TABLE_LOAD = load factor of the compact
   * table (default 0.6; the wide one is built at 0.5).  slot index = b * slots_per_bucket + j. */
  uint32_t table_layout, slots_per_bucket, tag_q, tag_dsh, tag_w;
} kamd_index_view;

typedef struct {
  int32_t paired;           /* 0 = --single */
  double fld;               /* -l (0 = estimate from the data; paired only) */
  double sd;                /* -s */
  int32_t single_overhang;  /* --single-overhang */
  int32_t strand;           /* 0 unstranded, 1 --fr-stranded, 2 --rf-stranded */
  int32_t no_jump;          /* --no-jump: match() looks up every k-mer (src/KmerIndex.cpp:1776) */
  int32_t do_union;         /* --union: per mate the union of the hits' transcript sets instead of their intersection
                               (MinCollector::unionECs src/MinCollector.cpp:498, :163-169); match() runs with partial = false.
                               With a strand option --union and --no-jump filter per hit (src/ProcessReads.cpp:62-82) */
} kamd_quant_opts;

/* ---- errors, version ---- */
const char* kamd_last_error(void);
/* The structures of this header are passed by pointer and have grown from round to round (kamd_tuning, kamd_profile): a binding compiled against
 * another version of the header must refuse to run rather than read or write beyond what it allocated.  kamd_abi_version() returns the
 * KAMD_ABI_VERSION the library was built with; callers compare it with the one they were compiled against (kallisto_amd/api.py does at load). */
#define KAMD_ABI_VERSION 6
uint32_t kamd_abi_version(void);

/* ---- S1: index ---- */
int kamd_index_load(const char* path, int threads, kamd_index** out);
/* The same with the layout of the k-mer table given by the caller instead of the environment (kamd_index_load reads KAMD_TABLE_LAYOUT =
 * wide | compact | auto -- auto when unset -- and KAMD_TABLE_LOAD): KAMD_TABLE_WIDE = three 20-byte slots per 64-byte line at a load of 0.5,
 * KAMD_TABLE_COMPACT = four exact 16-byte slots per line (kamd_index_view::table_layout; an error when the index's class ids or text
 * positions do not fit the slot), KAMD_TABLE_AUTO = compact when it fits (the default).  load = load factor of the compact table in [0.2, 0.9]
 * (anything else: 0.4 or 0.5 while that table stays under 2.4 GB, 0.6 beyond).  A flattened file (kamd_index_save) carries its layout: asking for KAMD_TABLE_WIDE or _COMPACT (here, or through
 * KAMD_TABLE_LAYOUT) and naming a flattened file of the other layout is an error (-3); _AUTO takes what the file holds. */
#define KAMD_TABLE_WIDE 0
#define KAMD_TABLE_COMPACT 1
#define KAMD_TABLE_AUTO 2
int kamd_index_load_layout(const char* path, int threads, int layout, double load, kamd_index** out);
/* The flattened tables as a file: building them from a kallisto index takes seconds; kamd_index_save writes them once and
 * kamd_index_load recognises such a file by its magic and reads it back with plain reads (same kamd_index).  Native byte order,
 * format-versioned; not a replacement for the kallisto index, which stays the source of truth. */
int kamd_index_save(const kamd_index*, const char* path);
/* 1 when flat_path is a flattened file of this format version written from exactly the kallisto index at index_path (the file records the
 * index's size and a hash of its first and last 64 KiB), 0 when it is not, cannot be read, or does not say.  What a front-end asks before
 * it picks up `<index>.kamd` beside an index: modification times alone survive `cp -p` / `rsync -t` of a different index. */
int kamd_flat_index_matches(const char* flat_path, const char* index_path);
void kamd_index_free(kamd_index*);
int kamd_index_get_view(const kamd_index*, kamd_index_view* out);
const char* kamd_index_target_name(const kamd_index*, uint64_t i);

/* ---- context ---- */
/* Tuning knobs: which of the equivalent kernels / EM forms run and how they are shaped.  None of them changes a result.
 * kamd_ctx_create sets the defaults (one environment variable, KAMD_TUNE="field=value,field=value" over the field names below, is
 * read once there, for experiments: INTEGRATION.md section 8); kamd_ctx_tune overrides them for the following calls.
 * 0 in a field = keep the current value; on/off fields use 1 = on, 2 = off. */
typedef struct {
  int32_t text_verify;         /* kernel A: jump / middle / back-off windows are compared with the unitig text first (default on) */
  int32_t items_per_wave;      /* items per wavefront chunk of kernel A (default 1024, >= 64) */
  int32_t refill_min;          /* free lanes that trigger a refill (default 8, 1..64) */
  int32_t lds_pad;             /* diagnostic: unused LDS bytes added to every block of kernel A (lowers the occupancy); -1 = none */
  int32_t em_form;             /* EM: 1 = streamed two-launch form, 2 = CSR three-launch form (also what degenerate matrices -- no row with two
                                  transcripts -- run), 3 = component-local LDS form (falls back to 1 when a connected component does not
                                  fit a workgroup), default 3 */
  int32_t em_entries_per_lane; /* streamed form: K in {8,12,...,32}; -1 = automatic (default) */
  int32_t em_windowed;         /* streamed form: force the general windowed pass (test hook; default off) */
  int32_t em_graph;            /* rounds of the streamed / CSR forms replayed as a hipGraph (default on) */
  int32_t em_row_lanes;        /* CSR form: lanes per row, 2 / 4 (default) / 8 */
  int32_t em_fin_blocks;       /* CSR form: blocks of the final pass (default 1024) */
  int32_t em_local_block;      /* component-local form: threads per workgroup, 128 / 256 / 512 / 1024 (default) */
  int32_t em_group_div;        /* component-local form: groups hold about nnz / (CUs x this) entries; -1 (default) = the smallest divisor whose
                                  groups stay under ~10 000 entries (one 16-wavefront workgroup per CU, two rounds of workgroups on config #3) */
  int32_t em_split_len;        /* component-local form: a row / column with more entries is split over several lanes (1..64; default 16) */
  int32_t dedup_form;          /* record de-duplication: 1 = insert + verify launches, 2 = one launch (tag and owner in one CAS; default) */
  int32_t align_chunks;        /* kamd_pseudoalign: kernel A runs in this many launches, each classified and de-duplicated on a side stream while
                                  the next is matched (default 1 = one launch, everything in sequence: the stages are all bound by the rate of
                                  random memory requests and gain nothing from running side by side) */
  int32_t em_small_nnz;        /* component-local form: connected components of at most this many entries are packed into groups of about as
                                  many entries that ONE WAVEFRONT iterates (no block barrier inside a round); larger components keep
                                  workgroup-sized groups.  -1 = one size class only (every group a workgroup; the default: on the
                                  human-sized workload most entries sit in components of more than a thousand entries, the two kernels
                                  side by side were slower -- profiles/README.md) */
  int32_t em_reg_slices;       /* component-local form: 1 (default) = a wavefront keeps the index words and segment constants of its first slice
                                  per direction in registers for the rounds of a launch (split lengths up to 32), 2 = everything from LDS */
  int32_t em_hybrid;           /* component-local form on a matrix with connected components beyond a workgroup's LDS (repeat families, poly-A
                                  classes: one such component holds half of a real transcriptome's entries): 1 (default) = hybrid -- the
                                  components that fit keep the LDS form, the oversized ones are iterated BESIDE it by streamed kernels (two
                                  launches per round over flagged entry streams in HBM, on compute units of their own), one stop rule over
                                  both; 2 = off: such a matrix takes the streamed form as a whole (rounds 1-4) */
  int32_t overflow_second_pass; /* items whose class list overflowed kernel A's eight LDS entries (pairs inside repeat families, poly-A stretches, --no-jump
                                  runs): 1 (default) = they go through kernel A's data-flow loop once more with an append-only list of up to 192 classes in
                                  global memory (k_classify_long removes the duplicates), what overflows again takes the straight-line kernel -- the pass
                                  runs on a stream of its own beside the absorption of the batch's other tuple records (one launch of kernel A per batch, no
                                  positional filter, no --union; otherwise after it); 3 = the same pass, always after the absorption; 2 = all of
                                  them take the straight-line kernel k_pseudoalign_overflow (rounds 1-5) */
  int32_t em_giant_nnz;        /* hybrid: a component with more entries than this is "oversized"; -1 (default) = 6000, halved while the
                                  remaining components still do not fit their groups */
  int32_t em_blocked;          /* hybrid: 1 (default) = the oversized components are iterated in 2-D blocks -- the gathered vector's block in LDS, 16-bit in-block
                                  indices, partial sums per (segment, block) combined in a fixed order: three launches per round (k_gb_pass<0>, k_gb_pass<1>,
                                  k_gb_finish); 2 = the streamed kernels k_gi_rows / k_gi_cols gather through the vector memory pipeline (round 5) */
} kamd_tuning;
int kamd_ctx_tune(kamd_ctx*, const kamd_tuning*);
int kamd_ctx_get_tuning(const kamd_ctx*, kamd_tuning* out);
int kamd_ctx_create(int device, void* hip_stream, kamd_ctx** out);
void kamd_ctx_destroy(kamd_ctx*);
int kamd_index_upload(kamd_ctx*, const kamd_index*);
int kamd_ec_reset(kamd_ctx*);  /* forget all EC counts (a new MinCollector, src/MinCollector.h:22) */

/* ---- reads: 2-bit packing of one parsed batch (what fetchSequences hands to processBuffer) ----
 * Record layout per read: words_per_read u32 of 2-bit bases (base i at bits 2*(i&15) of word i>>4; A0 C1 G2 T3,
 * case-insensitive) followed by mask_words u32 (bit i&31 of word i>>5 set = base i is not ACGT).  A pair is two
 * consecutive records (mate 1, mate 2).  max_len fixes the stride: words_per_read = ceil(max_len/16),
 * mask_words = ceil(max_len/32). */
uint64_t kamd_packed_record_words(int32_t max_len);
/* host packer: seqs = concatenated sequences (no terminators needed), off[i]/len[i] locate read i */
int kamd_pack_reads_host(const char* seqs, const uint64_t* off, const int32_t* len, uint64_t n_reads, int32_t max_len,
                         uint32_t* out_words, uint16_t* out_len);
/* same, writing read r to record slot rec_first + r * rec_stride (mates of a pair from two files: stride 2, first 0 / 1);
 * re-entrant, so a front-end may pack disjoint ranges from several threads */
int kamd_pack_reads_host_strided(const char* seqs, const uint64_t* off, const int32_t* len, uint64_t n_reads, int32_t max_len,
                                 uint32_t* out_words, uint16_t* out_len, uint64_t rec_stride, uint64_t rec_first);
/* device packer: same, d_* are device pointers; runs on the context stream */
int kamd_pack_reads_device(kamd_ctx*, const char* d_seqs, const uint64_t* d_off, const int32_t* d_len, uint64_t n_reads,
                           int32_t max_len, uint32_t* d_out_words, uint16_t* d_out_len);

/* ---- reads: FASTQ text resident in HBM -> packed batch (the parsing half of FastqSequenceReader::fetchSequences,
 * src/ProcessReads.cpp:3128-3267, i.e. kseq_read, src/kseq.h:174-215, for strict 4-line records) ----
 * A *unit* is a piece of the (decompressed) text of one file -- or of the two mates' files -- that starts at the first byte of a
 * record and ends behind the last newline of a record; the caller cuts units by counting newlines while it moves the bytes and
 * tells how many records a unit holds (4 lines each).  The device finds the lines, verifies for every record that kseq_read would
 * return exactly its second line (header starts with '@', the third line with '+', quality as long as the sequence, no empty
 * sequence; a trailing '\r' is dropped as kseq does), and packs the sequences as kamd_pack_reads would: single-end -> item j =
 * record j, paired -> items 2j / 2j + 1 = record j of file 0 / file 1.  status != 0: the unit is NOT packed and the caller must
 * read this input with a general FASTA/FASTQ reader instead (multi-line records, FASTA, junk between records ...).
 * d_text[f]: device pointers, 16-byte aligned, readable up to 32 bytes behind n_bytes[f] (the packer loads whole dwords);
 * n_bytes[f] < 2^32.  Runs on the context stream and synchronises it once.
 * A unit is what one host-to-device copy brings (tens of MB); a *batch* -- what kamd_pseudoalign is called on -- should be millions of
 * reads, so parsing and packing are separate: kamd_fastq_unit_parse checks a unit and notes where its sequences are (the text must
 * stay in place until the batch is packed), kamd_fastq_batch_pack packs the reads of all units parsed since the last batch, in
 * order.  kamd_fastq_unit_pack = parse + pack of a single unit. */
typedef struct {
  const uint32_t* d_words;     /* packed batch, owned by the context, valid until the next kamd_fastq_batch_pack / _unit_pack on it */
  const uint16_t* d_len;
  uint64_t n_items;            /* unit_parse: = n_records; batch_pack: items of the batch */
  int32_t max_len;             /* longest read of the unit / batch: the max_len to hand to kamd_pseudoalign / kamd_fld_* */
  int32_t status;              /* 0 ok; 1 some record is not strict 4-line FASTQ (first_bad_record); 2 the text holds fewer than
                                  4 x n_records lines; 3 a read is longer than 65535 bases.  != 0: nothing was added to the batch */
  uint64_t first_bad_record;
} kamd_fastq_unit;
int kamd_fastq_unit_parse(kamd_ctx*, const char* const* d_text, const uint64_t* n_bytes, int32_t n_files, uint64_t n_records,
                          kamd_fastq_unit* out);
int kamd_fastq_batch_pack(kamd_ctx*, kamd_fastq_unit* out);
int kamd_fastq_unit_pack(kamd_ctx*, const char* const* d_text, const uint64_t* n_bytes, int32_t n_files, uint64_t n_records,
                         kamd_fastq_unit* out);

/* ---- S2: pseudoalignment of one batch resident in HBM ----
 * n_items = pairs (paired) or reads (single).  Accumulates into the context's EC state; call kamd_ec_finalize after
 * the last batch.  The EC state is bounded by the number of DISTINCT classes, like MinCollector's (src/MinCollector.cpp:251-269): a
 * dense count vector over the index's transcript sets plus a table of the distinct tuples of set ids seen so far; the per-item
 * records of a batch are recycled when the call returns. */
int kamd_pseudoalign(kamd_ctx*, const kamd_quant_opts*, const uint32_t* d_words, const uint16_t* d_len, uint64_t n_items,
                     int32_t max_len);
/* fragment-length histogram from the first 10000 qualifying pairs in input order (src/ProcessReads.cpp:981-1017,
 * 1174-1181 at -t 1).  flens: KAMD_MAX_FRAG_LEN u32 (host) and *n_used accumulate across calls: zero both, then call once
 * per batch in input order until *n_used reaches 10000 (or the input ends). */
int kamd_fld_from_batch(kamd_ctx*, const kamd_quant_opts*, const uint32_t* d_words, const uint16_t* d_len, uint64_t n_items,
                        int32_t max_len, uint32_t* flens, uint64_t* n_used);
/* optional: start the fragment-length kernel for the first prefix of this batch on a side stream and return at once.  Called
 * BEFORE kamd_pseudoalign on the same batch, the (latency-bound) kernel runs underneath kernel A; the following
 * kamd_fld_from_batch on the same pointers then only waits for it.  Purely a scheduling hint: results are identical. */
int kamd_fld_prefetch(kamd_ctx*, const kamd_quant_opts*, const uint32_t* d_words, const uint16_t* d_len, uint64_t n_items,
                      int32_t max_len);

/* statistics of the batches processed so far */
typedef struct {
  uint64_t n_processed;     /* items seen */
  uint64_t n_single;        /* items whose hits carried one transcript set */
  uint64_t n_multi;         /* items that needed an intersection */
  uint64_t n_probes;        /* k-mer table probes */
  uint64_t n_bucket_reads;  /* 64-byte bucket reads (>= n_probes) */
  uint64_t n_distinct_tuples;
  uint64_t n_stream_words;  /* u32 words of the last batch's record stream (one fixed slot per item + the records of overflow items; recycled
                               from batch to batch: its distinct tuple records move to the tuple store) */
  uint64_t n_raw_words;     /* u32 words kernel A wrote: per item 1 header + its distinct (unitig,set) classes */
  uint64_t n_text_hits;     /* probes answered from the unitig text instead of the table */
  uint64_t n_wave_iters;    /* kernel A: loop trips summed over the wavefronts ... */
  uint64_t n_lane_iters;    /* ... and lanes that issued a probe in them: n_lane_iters / (64 n_wave_iters) = lane utilisation */
} kamd_align_stats;
int kamd_align_stats_get(kamd_ctx*, kamd_align_stats* out);

/* durations measured with HIP events on the context stream (bench.py's roofline figures) */
typedef struct {
  float last_align_kernel_ms;  /* kernel A (k_match_v3) of the last kamd_pseudoalign call */
  float last_em_ms;            /* all EM launches of the last kamd_em_run call */
  uint64_t last_em_iters;      /* EM rounds executed by it */
  float last_classify_ms;      /* k_classify of the same call */
  int32_t kernel_a_version;    /* 3 = k_match_v3 + k_classify (the only form; versions 1 and 2 were removed in round 2) */
  uint64_t last_em_nnz;        /* shape of the EM problem of the last kamd_em_run: nnz of the EC x transcript matrix, */
  uint64_t last_em_nnz_multi;  /* nnz in multi-transcript rows, */
  uint64_t last_em_nseg;       /* column segments (CSR form) or chunks per direction (streamed form), */
  uint64_t last_em_necs;       /* rows */
  int32_t last_em_k;           /* streamed EM form: entries per lane; 0: the CSR form ran; -1 / -2: the component-local form (CSR / sliced ELLPACK) */
  uint32_t last_em_grid;       /* streamed form: grid of its two per-round launches (blocks of 256 threads, one chunk per wavefront);
                                  component-local form: number of groups (= workgroups per launch) */
  uint32_t last_em_lds;        /* component-local form: LDS bytes per workgroup */
  int32_t last_em_plan_cached; /* component-local form: 1 = the plan of an earlier run on the same matrix was reused */
  float last_finalize_ms;      /* kamd_ec_finalize: resolution of the distinct tuples, merge of equal sets, CSR (HIP events around it) */
  uint64_t last_fin_records;   /* ... distinct tuples it resolved, */
  uint64_t last_fin_stream_words;  /* u32 words of the tuple store it read, */
  uint64_t last_fin_cand_words;    /* u32 words of candidate transcript sets it wrote and merged */
  float absorb_ms;             /* since kamd_ec_reset: de-duplication of the batches' tuple records into the persistent tuple table
                                  (inside kamd_pseudoalign, after k_classify; HIP events around it) */
  uint64_t n_distinct_tuples;  /* entries of the tuple table; */
  uint64_t tuple_store_words;  /* u32 words of distinct tuple records kept; */
  uint64_t tuple_table_slots;  /* slots of the table (32 bytes each): what the EC state costs in HBM, whatever the number of reads */
  uint64_t last_em_max_comp_nnz;   /* component-local form: entries of the largest connected component of the last kamd_em_run's matrix */
  uint64_t last_em_giant_nnz;      /* hybrid: entries of the components iterated by the streamed kernels (0: everything fitted the LDS form), */
  uint64_t last_em_giant_rows;     /* their rows, */
  uint64_t last_em_giant_tr;       /* their transcripts, */
  uint32_t last_em_giant_chunks;   /* wavefront chunks per direction of those kernels, */
  int32_t last_em_graph_fallback;  /* hybrid: 1 = the hipGraph of a chunk's rounds could not be captured / instantiated and the chunks went out as plain launches */
  float last_em_plan_ms;           /* host time of the plan set-up of the last kamd_em_run (component labels, groups, layouts; 0 when cached) */
  uint64_t n_overflow_items;       /* since kamd_ec_reset: items kernel A handed to k_pseudoalign_overflow (more than 8 distinct (unitig, set)
                                      classes, or reads beyond the LDS budget), */
  float overflow_ms;               /* and that kernel's time (HIP events) */
  float last_merge_ms;             /* several ranks: the last kamd_ec_allreduce (one all-gather of sizes, the all-reduce of the dense counts, the all-gathers
                                      of the records, their de-duplication; HIP events) */
  float em_collective_ms;          /* several ranks: host wall time inside the collectives of the last kamd_em_run_comm (one all-reduce per chunk of 64
                                      rounds for the stop rule + the final sum of the abundance vectors; the host waits for each) */
  uint32_t em_collectives;         /* ... and their number */
  uint64_t n_overflow_second_pass; /* of n_overflow_items: those kernel A's second pass (an append-only list of up to 192 classes) took care of; the rest went to the
                                      straight-line kernel */
  uint64_t last_em_giant_pieces;   /* hybrid, blocked form: partial sums per round over both directions (segments cut at block and chunk boundaries); 0 = the
                                      streamed kernels ran */
} kamd_profile;
int kamd_profile_get(kamd_ctx*, kamd_profile* out);

/* diagnostic: rate of dependent random 64-byte bucket reads of the uploaded k-mer table at a given launch shape -- the
 * practical ceiling of kernel A's probe stream (profiles/README.md) */
int kamd_debug_random_lines(kamd_ctx*, uint32_t n_blocks, uint32_t block_threads, uint32_t iters, double* gbytes_per_s,
                            double* mlines_per_s);
/* the same with the reads confined to the first span_mb MiB of the table (0 = all) and 64- or 8-byte accesses: HBM vs MALL vs L2 */
int kamd_debug_random_lines_span(kamd_ctx*, uint32_t n_blocks, uint32_t block_threads, uint32_t iters, uint32_t span_mb,
                                 uint32_t access_bytes, double* gbytes_per_s, double* mlines_per_s);

/* ---- EC state exchange (multi-GPU; one process per GPU, the caller runs the collectives) ----
 * The EC state is (a) a dense count vector over index transcript sets and (b) a list of (tuple of index set ids,
 * count) for items whose hits carried several sets.  Both are keyed by index set ids, which are identical on every
 * rank, so ranks merge by all-reduce(a) + all-gather(b). */
int kamd_ec_dense_counts(kamd_ctx*, uint32_t** d_counts, uint64_t* n);            /* device pointer into the context */
int kamd_ec_tuples_export(kamd_ctx*, uint64_t* n_words, uint64_t* n_tuples);      /* de-duplicates; sizes for the gather */
/* [count, m, e0..e(m-1)] records + the word offset of each record */
int kamd_ec_tuples_copy(kamd_ctx*, uint32_t* d_out_words, uint64_t* d_out_rec_off);
/* install gathered records (offsets rebased by the caller to the concatenated buffer) */
int kamd_ec_tuples_replace(kamd_ctx*, const uint32_t* d_words, uint64_t n_words, const uint64_t* d_rec_off, uint64_t n_recs);

/* records [1, n, t0..t(n-1)] of items whose set was changed by a positional filter (findPosition / strand); content-keyed,
 * so ranks simply concatenate them */
int kamd_ec_explicit_export(kamd_ctx*, uint64_t* n_words, uint64_t* n_recs);
int kamd_ec_explicit_copy(kamd_ctx*, uint32_t* d_out_words, uint64_t* d_out_rec_off);
int kamd_ec_explicit_replace(kamd_ctx*, const uint32_t* d_words, uint64_t n_words, const uint64_t* d_rec_off, uint64_t n_recs);

/* ---- several GPUs: one process (or host thread) per GPU, collectives inside the library --------------------------------------
 * A communicator binds a context to its place among `world` ranks.  Two backends:
 *   RCCL       kamd_comm_create_rccl: ncclCommInitRank on the context's device (librccl is loaded at run time: the copy the
 *              process already holds -- PyTorch's -- else KAMD_RCCL_LIB, else the ROCm installation's); every collective is
 *              enqueued on the context stream, over xGMI between the GPUs of a node.  Rank 0 obtains the 128-byte id with
 *              kamd_comm_unique_id and the launcher hands it to the other ranks (torch.distributed broadcast, a file, shared
 *              memory between the threads of one process ...).
 *   callbacks  kamd_comm_create_callbacks: the caller supplies the collectives (device pointers, context stream already
 *              synchronised) -- how the tests drive two ranks on one GPU / on the CPU through gloo.
 * Replaces the reference's merge under a mutex, MasterProcessor::update (src/ProcessReads.cpp:424-481). */
typedef struct kamd_comm kamd_comm;
#define KAMD_COMM_ID_BYTES 128
typedef struct {
  /* element type: 0 = u32, 1 = i32, 2 = u64, 3 = f64.  All return 0 on success. */
  int (*allreduce_sum)(void* user, void* d_buf, uint64_t count, int32_t type);
  int (*allgather)(void* user, const void* d_send, void* d_recv, uint64_t bytes_per_rank);   /* d_recv: world x bytes_per_rank, rank order */
  int (*broadcast)(void* user, void* d_buf, uint64_t bytes, int32_t root);
} kamd_comm_callbacks;
int kamd_comm_unique_id(void* id128);
int kamd_comm_create_rccl(kamd_ctx*, int32_t rank, int32_t world, const void* id128, kamd_comm** out);
int kamd_comm_create_callbacks(kamd_ctx*, int32_t rank, int32_t world, const kamd_comm_callbacks*, void* user, kamd_comm** out);
void kamd_comm_destroy(kamd_comm*);
/* what the communicator is: its rank and world as created, the number of ranks the transport itself reports (ncclCommCount on the
 * RCCL backend; the world it was told for callbacks), backend 0 = none (a world of one without RCCL), 1 = RCCL, 2 = callbacks.
 * Any pointer may be null. */
int kamd_comm_info(const kamd_comm*, int32_t* rank, int32_t* world, int32_t* ranks_seen, int32_t* backend);
/* merge the EC state of all ranks: ONE all-reduce (sum) of the dense count vector + an all-gather of the de-duplicated tuple
 * records and of the explicit-set records; afterwards every rank holds the state of the whole input (call kamd_ec_finalize
 * next, on every rank).  Not with kamd_ec_track_order. */
int kamd_ec_allreduce(kamd_ctx*, kamd_comm*);
/* host helpers on the same communicator (small values of the quant driver): fragment-length sample of rank `root`, sums */
int kamd_comm_broadcast_host(kamd_ctx*, kamd_comm*, void* buf, uint64_t bytes, int32_t root);
int kamd_comm_sum_u64_host(kamd_ctx*, kamd_comm*, uint64_t* values, uint64_t count);
/* kamd_em_run_partitioned with the communicator's all-reduce as the sum callback, and the per-rank results summed: every
 * rank returns the same alpha / alpha_before_zeroes / rounds */
int kamd_em_run_comm(kamd_ctx*, kamd_comm*, const double* eff_lens, uint64_t n_targets, uint32_t n_iter, uint32_t min_rounds,
                     double* alpha, double* alpha_before_zeroes, int32_t* rounds);

/* ---- finalize: resolve intersections, apply the on-list mask, merge equal sets ----
 * Produces the EC multiset {sorted transcript set -> count} as CSR, on device and (optionally) host. */
typedef struct {
  uint64_t n_ecs, nnz, n_pseudoaligned;
  const uint64_t* d_ec_off;  /* [n_ecs + 1] device */
  const uint32_t* d_ec_ids;  /* [nnz] device */
  const uint32_t* d_counts;  /* [n_ecs] device */
} kamd_ec_result;
/* on != 0: kamd_ec_finalize emits the sets ordered by the first read (pair) that produced them -- the ids the reference
 * assigns at -t 1: new sets wait in MasterProcessor::newECcount (src/ProcessReads.h:314; an insertion-ordered
 * ankerl::unordered_dense map in C++17 builds, src/common.h:19-27) and are appended by MinCollector::increaseCount
 * (src/ProcessReads.cpp:452-464, src/MinCollector.cpp:251-269).  Only Bootstrap -- a multinomial over the count vector
 * in id order, src/Bootstrap.cpp:4-14 -- depends on the ids.  Call before the first batch; single-process only (merged
 * records of several ranks have no input order). */
int kamd_ec_track_order(kamd_ctx*, int on);
int kamd_ec_finalize(kamd_ctx*, kamd_ec_result* out);
int kamd_ec_finalize_result(kamd_ctx*, kamd_ec_result* out);   /* the result of the last kamd_ec_finalize (e.g. after kamd_quant_batches) */
int kamd_ec_download(kamd_ctx*, uint64_t* ec_off, uint32_t* ec_ids, uint32_t* counts);
/* `quant-tcc` (src/main.cpp:2802-3220): the equivalence classes come from a file (KmerIndex::loadECsFromFile,
 * src/KmerIndex.cpp:1561-1600) and every sample / cell brings its own count vector (EM_lambda: `collection.counts[ec] = count`,
 * src/main.cpp:2989-2996).  kamd_ec_upload makes a host CSR (sorted distinct transcript ids per class, no empty class; counts may be
 * NULL = zeros) the context's EC result, exactly as if kamd_ec_finalize had produced it: kamd_em_run(ctx, NULL, ...),
 * kamd_bootstrap(_batch) and kamd_ec_download work on it.  kamd_ec_set_counts replaces only the counts ([n_ecs], host): the next
 * kamd_em_run reuses the EM plan of the matrix, like a bootstrap replicate does. */
int kamd_ec_upload(kamd_ctx*, const uint64_t* ec_off, const uint32_t* ec_ids, const uint32_t* counts, uint64_t n_ecs);
int kamd_ec_set_counts(kamd_ctx*, const uint32_t* counts);

/* ---- S3: EM ---- */
/* Runs EMAlgorithm(counts, ...).run(n_iter, min_rounds) (src/EMAlgorithm.h:26-48,95-223) on the finalized EC result
 * (d_ec_off == NULL) or on a caller-provided device CSR in the form kamd_ec_finalize emits: the transcript ids of a class ascending and
 * distinct, no two classes with the same transcripts (kamd_ec_upload checks a host CSR for exactly that; a device CSR is not checked).
 * d_weight_counts: the counts the weights w = count/eff_len are
 * computed from (tc_.counts; NULL = d_counts -- they only differ in bootstraps).  eff_lens: host [n_targets].
 * Outputs (host): alpha, alpha_before_zeroes (nullable) [n_targets], rounds ("ran for i rounds"). */
int kamd_em_run(kamd_ctx*, const uint64_t* d_ec_off, const uint32_t* d_ec_ids, const uint32_t* d_counts,
                const uint32_t* d_weight_counts, uint64_t n_ecs, const double* eff_lens, uint64_t n_targets, uint32_t n_iter,
                uint32_t min_rounds, double* alpha, double* alpha_before_zeroes, int32_t* rounds);

/* The same EM over `world` GPUs (one process per GPU).  The EC x transcript matrix is block diagonal over the connected
 * components of the transcript/EC graph and EMAlgorithm::run never couples two components, so every rank runs the
 * unchanged EM on the components it owns (hash of the component label mod world) -- there is no per-round collective.
 * Only the stop rule is global: after every chunk of rounds the library calls sum_cb(user, d_counts, n), which must
 * replace the n device ints (per-round numbers of transcripts that still change, src/EMAlgorithm.h:176-199) by their sum
 * over all ranks, e.g. ncclAllReduce(ncclInt32, ncclSum); all ranks then stop at the round the reference would.
 * Requires every rank to hold the same finalized EC result.  alpha / alpha_before_zeroes receive this rank's transcripts
 * and zeros elsewhere: the caller sums them over the ranks. */
typedef int (*kamd_em_sum_cb)(void* user, int32_t* d_counts, int32_t n);
int kamd_em_run_partitioned(kamd_ctx*, uint32_t rank, uint32_t world, kamd_em_sum_cb sum_cb, void* user, const double* eff_lens,
                            uint64_t n_targets, uint32_t n_iter, uint32_t min_rounds, double* alpha, double* alpha_before_zeroes,
                            int32_t* rounds);

/* ---- S4: bootstrap ---- */
/* One replicate of Bootstrap::run_em (src/Bootstrap.cpp:4-14): Multinomial(counts, seed).sample() with libstdc++'s
 * minstd_rand0 + discrete_distribution semantics (identical sample for an identical EC order), then a fresh EM
 * run(10000, 50).  seed = seeds[b] of src/main.cpp:2746-2752 (kamd_bootstrap_seeds).  CSR: finalized result when
 * d_ec_off == NULL.  sample_out (nullable, host [n_ecs]) receives the resampled counts. */
int kamd_bootstrap(kamd_ctx*, const uint64_t* d_ec_off, const uint32_t* d_ec_ids, const uint32_t* d_counts, uint64_t n_ecs,
                   uint64_t seed, const double* eff_lens, uint64_t n_targets, double* alpha, int32_t* rounds,
                   uint32_t* sample_out);
/* n_rep replicates on the finalized EC result (the pool of src/Bootstrap.cpp:15-92, src/main.cpp:2764-2782): all multinomial
 * samples are drawn in one launch; the EMs reuse the plan of the matrix (component-local form) and only refresh its counts.
 * alpha: n_rep x n_targets (row b = replicate b = seeds[b]); rounds: n_rep (nullable). */
int kamd_bootstrap_batch(kamd_ctx*, const uint64_t* seeds, int32_t n_rep, const double* eff_lens, uint64_t n_targets, double* alpha,
                         int32_t* rounds);
/* seeds[b] = std::mt19937_64(seed)() for b = 0..n-1 (src/main.cpp:2746-2752) */
void kamd_bootstrap_seeds(uint64_t seed, int32_t n, uint64_t* seeds);

/* ---- the quant flow in one call (ProcessReads -> fragment-length model -> EMAlgorithm::run, src/main.cpp:2654-2730) ----
 * For a caller whose reads already sit in HBM in the packed layout: the batches are pseudoaligned in order (fragment-length sample:
 * the first 10 000 qualifying pairs of the input, src/ProcessReads.cpp:981-1017), with a communicator every rank passes its own
 * shard and the EC state is merged (kamd_ec_allreduce) and the EM partitioned (kamd_em_run_comm); then kamd_ec_finalize, effective
 * lengths, the EM with run(10000, 50), TPM.  Equivalent to calling those entry points one after the other; the EC result stays
 * in the context (kamd_ec_download, kamd_bootstrap_batch). */
typedef struct { const uint32_t* d_words; const uint16_t* d_len; uint64_t n_items; int32_t max_len; } kamd_batch;
typedef struct {
  uint64_t n_processed;          /* items of all ranks */
  int32_t em_rounds;
  uint32_t* flens;               /* [KAMD_MAX_FRAG_LEN], caller-provided: the fragment-length sample (zeros with -l) */
  double* eff_lens;              /* [n_targets], caller-provided */
  double* est_counts;            /* [n_targets], caller-provided */
  double* alpha_before_zeroes;   /* [n_targets], caller-provided; nullable */
  double* tpm;                   /* [n_targets], caller-provided; nullable */
} kamd_quant_out;
int kamd_quant_batches(kamd_ctx*, const kamd_quant_opts*, const kamd_batch* batches, uint64_t n_batches, const int32_t* target_lens,
                       uint64_t n_targets, kamd_comm* comm /* nullable */, kamd_quant_out* out);

/* ---- host-side helpers of the quant driver (FLD model, effective lengths; FP64 on the host, bit-exact) ---- */
void kamd_mean_frag_lens_trunc(const uint32_t* flens, double* mean_fl_trunc);
void kamd_trunc_gaussian_fld(int32_t start, int32_t stop, double mean, double sd, double* out);
void kamd_eff_lens(const int32_t* target_lens, uint64_t n, const double* mean_fl_trunc, double* eff_lens);
void kamd_counts_to_tpm(const double* est_counts, const double* eff_lens, uint64_t n, double* tpm);

#ifdef __cplusplus
}
#endif
#endif
