/* oracle/kallisto_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the `kallisto quant` hot path of pachterlab/kallisto v0.51.1 (index v13):
 * index parsing, k-mer lookup semantics, KmerIndex::match jump logic, MinCollector::intersectKmers,
 * KmerIndex::mapPair, KmerIndex::findPosition, the fragment-length tables, EMAlgorithm::run and the bootstrap
 * multinomial resampler.  Every function cites the reference file:line it restates (paths relative to the
 * reference tree).
 *
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the HIP path.
 * NOTHING on the product path (kallisto_amd/) may include, link or call it.
 *
 * Parity status: PINNED -- tests/test_oracle_vs_reference.py checks this restatement against the unmodified
 * reference compiled into oracle/_ref (dump_ec harness: EC multisets, flens, eff_lens, alpha) and against the
 * committed golden fixtures under tests/golden/ that were produced by that harness.
 */
#ifndef KALLISTO_ORACLE_H
#define KALLISTO_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_MAX_FRAG_LEN 1000 /* src/MinCollector.h:15 */

typedef struct ko_index ko_index;

/* One entry of the hit vector `v` filled by KmerIndex::match (src/KmerIndex.cpp:1698). */
typedef struct {
  uint32_t unitig;   /* global unitig id: long unitigs, then short, then abundant */
  uint32_t dist;     /* k-mer offset on the unitig (const_UnitigMap::dist) */
  uint32_t size;     /* unitig length in bp (const_UnitigMap::size) */
  uint32_t strand;   /* 1 = k-mer equals unitig forward text */
  uint32_t lb, ub;   /* Node::get_mc_contig(dist) */
  uint32_t block;    /* global block id */
  uint32_t ec;       /* de-duplicated transcript-set id of ec[dist].getIndices() */
  int32_t pos;       /* read position recorded with the hit (second member of the pair) */
} ko_hit;

typedef struct {
  int paired;           /* 0 = --single */
  double fld;           /* -l (0 = estimate from data; only valid when paired) */
  double sd;            /* -s */
  int single_overhang;  /* --single-overhang */
  int strand;           /* 0 = unstranded, 1 = --fr-stranded, 2 = --rf-stranded */
  int no_jump;          /* --no-jump: every k-mer of the read is looked up (src/KmerIndex.cpp:1776) */
  int do_union;         /* --union: per mate the union instead of the intersection of the hits' sets (src/MinCollector.cpp:163-169) */
} ko_opts;
#define KO_MATCH_NO_JUMP 2   /* ko_match: bit 1 of the `partial` argument */

/* Result of pseudoaligning a batch of reads: the EC multiset (in first-seen order) and the fragment
 * length histogram, i.e. what MasterProcessor::update accumulates (src/ProcessReads.cpp:424-499). */
typedef struct ko_result ko_result;

/* ---- index ---- */
ko_index* ko_index_load(const char* path, char* err, size_t errlen);
void ko_index_free(ko_index*);
int ko_index_k(const ko_index*);
uint64_t ko_index_num_unitigs(const ko_index*);
uint64_t ko_index_num_kmers(const ko_index*);
uint64_t ko_index_num_blocks(const ko_index*);
uint64_t ko_index_num_ecs(const ko_index*);
uint64_t ko_index_num_targets(const ko_index*);
uint64_t ko_index_dlist_size(const ko_index*);
const int32_t* ko_index_target_lens(const ko_index*);
const char* ko_index_target_name(const ko_index*, uint64_t i);
/* transcript set of de-duplicated EC `ec`; returns its size */
uint64_t ko_index_ec(const ko_index*, uint32_t ec, const uint32_t** ids);
/* k-mer lookup, CompactedDBG::find semantics; returns 1 if found */
int ko_index_find(const ko_index*, uint64_t kmer_fwd, ko_hit* out);

/* ---- per read ---- */
int ko_match(const ko_index*, const char* s, int l, int partial, ko_hit* out, int max_hits, int* n_probes);
/* match both mates + intersectKmers + onlist mask + (optional) findPosition / strand filters.
 * Returns |u| (0 = not pseudoaligned); writes the sorted transcript ids to out_set (capacity max_set). */
int ko_pseudoalign(const ko_index*, const ko_opts*, const char* s1, int l1, const char* s2, int l2,
                   double mean_fl, int has_mean_fl, uint32_t* out_set, int max_set, int* n_hits1, int* n_hits2);
int ko_map_pair(const ko_index*, const char* s1, int l1, const char* s2, int l2);
int ko_find_position(const ko_index*, uint32_t tr, const ko_hit* um, int p, int* sense_out);

/* ---- batch ---- */
ko_result* ko_result_new(void);
void ko_result_free(ko_result*);
/* reads: concatenated NUL-terminated sequences; off[i] = start of sequence i; paired input is interleaved
 * 1,2,1,2 (src/ProcessReads.cpp:1034-1041).  n_seqs counts sequences, not pairs. */
int64_t ko_process_reads(const ko_index*, const ko_opts*, const char* reads, const uint64_t* off,
                         const int32_t* len, uint64_t n_seqs, ko_result* res);
uint64_t ko_result_num_ecs(const ko_result*);
uint64_t ko_result_nnz(const ko_result*);
/* CSR of the EC multiset: ec_off[n_ecs+1], ec_ids[nnz], counts[n_ecs] */
void ko_result_export(const ko_result*, uint64_t* ec_off, uint32_t* ec_ids, uint32_t* counts);
const uint32_t* ko_result_flens(const ko_result*); /* KO_MAX_FRAG_LEN entries */
uint64_t ko_result_num_processed(const ko_result*);
uint64_t ko_result_num_probes(const ko_result*);
uint64_t ko_result_num_hits(const ko_result*);

/* ---- fragment length model / effective lengths ---- */
void ko_mean_frag_lens_trunc(const uint32_t* flens, double* mean_fl_trunc);              /* MinCollector.cpp:629 */
void ko_trunc_gaussian_fld(int start, int stop, double mean, double sd, double* out);    /* weights.cpp:248 */
void ko_trunc_gaussian_counts(int start, int stop, double mean, double sd, int total, uint32_t* out); /* :273 */
void ko_frag_len_means(const int32_t* lens, uint64_t n, const double* mean_fl_trunc, double* out); /* weights.cpp:7 */
void ko_calc_eff_lens(const int32_t* lens, uint64_t n, const double* means, double* out); /* weights.cpp:58 */

/* ---- EM ---- */
/* EMAlgorithm ctor + run (src/EMAlgorithm.h:26-48,95-223).  weight_counts are the counts the weights are
 * computed from (tc_.counts; equals counts except in bootstraps).  Returns the round index printed by the
 * reference ("ran for i rounds"). */
int ko_em_run(const uint64_t* ec_off, const uint32_t* ec_ids, const uint32_t* counts, const uint32_t* weight_counts,
              uint64_t n_ecs, const double* eff_lens, uint64_t n_tr, uint64_t n_iter, uint64_t min_rounds,
              double* alpha, double* alpha_before_zeroes);
void ko_counts_to_tpm(const double* est_counts, const double* eff_lens, uint64_t n, double* tpm); /* PlaintextWriter.cpp:5 */

/* ---- bootstrap ---- */
/* seeds[b] = mt19937_64(seed)() (src/main.cpp:2746-2752) */
void ko_bootstrap_seeds(uint64_t seed, int n, uint64_t* seeds);
/* Multinomial::sample (src/Multinomial.hpp:33-51) with libstdc++ minstd_rand0 + discrete_distribution semantics */
void ko_multinomial_sample(const uint32_t* counts, uint64_t n, uint64_t seed, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
