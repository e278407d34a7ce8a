// oracle/ref_harness/quant_bound.cpp -- TEST INFRASTRUCTURE, not product code.
//
// INTEGRATION.md, compiled: the reference's own option struct, FASTQ reader (FastqSequenceReader::fetchSequences,
// src/ProcessReads.cpp:3128-3267) and writers (plaintext_writer / plaintext_aux, src/PlaintextWriter.cpp:29-65,140-197), linked from
// oracle/_ref/libkallisto_ref.a, bound to libkallisto_amd.so at the four seams a maintainer would edit in src/main.cpp's quant branch:
//   S1  index.load(opt)                          -> kamd_index_load + kamd_index_upload          (src/main.cpp:2632-2633)
//   S2  ReadProcessor::processBuffer             -> kamd_pack_reads_host + kamd_pseudoalign      (src/ProcessReads.cpp:968-1237)
//   S3  MasterProcessor::update / FLD / eff lens -> kamd_fld_from_batch, kamd_ec_finalize, kamd_eff_lens (src/ProcessReads.cpp:424-499, src/main.cpp:2665-2681)
//   S4  EMAlgorithm::run                         -> kamd_em_run                                  (src/main.cpp:2683-2689)
// Everything between the seams -- reading the files, the mates' interleaving, the output files -- is the reference's code, so
// `quant_bound quant -i idx -o out [--single -l L -s S] r_1.fq r_2.fq` must write the abundance.tsv the reference writes.
//
//   build: make -C oracle ref_bound   (needs /root/reference; output oracle/_ref/quant_bound)
#include "common.h"
#include "ProcessReads.h"
#include "PlaintextWriter.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include <sys/stat.h>

#include "kallisto_amd.h"

#define CK(x) do { if ((x) != 0) { fprintf(stderr, "Error: %s\n", kamd_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "Error: %s\n", hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2 || std::string(argv[1]) != "quant") {
    fprintf(stderr, "usage: quant_bound quant -i index -o outdir [-t threads] [--single -l L -s S] [--fr-stranded|--rf-stranded] reads_1.fastq [reads_2.fastq]\n");
    return argc < 2 ? 0 : 1;
  }
  ProgramOptions opt;
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "-i" && i + 1 < argc) opt.index = argv[++i];
    else if (a == "-o" && i + 1 < argc) opt.output = argv[++i];
    else if (a == "-t" && i + 1 < argc) opt.threads = atoi(argv[++i]);
    else if (a == "-l" && i + 1 < argc) opt.fld = atof(argv[++i]);
    else if (a == "-s" && i + 1 < argc) opt.sd = atof(argv[++i]);
    else if (a == "--single") opt.single_end = true;
    else if (a == "--single-overhang") opt.single_overhang = true;
    else if (a == "--fr-stranded") { opt.strand_specific = true; opt.strand = ProgramOptions::StrandType::FR; }
    else if (a == "--rf-stranded") { opt.strand_specific = true; opt.strand = ProgramOptions::StrandType::RF; }
    else if (a == "--plaintext") opt.plaintext = true;
    else opt.files.push_back(a);
  }
  if (opt.index.empty() || opt.output.empty() || opt.files.empty() || (!opt.single_end && opt.files.size() % 2)) { fprintf(stderr, "Error: bad arguments\n"); return 1; }
  mkdir(opt.output.c_str(), 0777);
  char tbuf[64]; { time_t t = time(nullptr); strftime(tbuf, sizeof tbuf, "%a %b %e %H:%M:%S %Y", localtime(&t)); }
  const std::string start_time = tbuf;
  std::string call; for (int i = 0; i < argc; i++) { if (i) call += " "; call += argv[i]; }
  const bool paired = !opt.single_end;

  // ---- S1: the index (replaces index.load(opt)) ----
  kamd_index* gidx = nullptr; kamd_ctx* ctx = nullptr;
  CK(kamd_index_load(opt.index.c_str(), opt.threads, &gidx));
  CK(kamd_ctx_create(0, nullptr, &ctx));
  CK(kamd_index_upload(ctx, gidx));
  kamd_index_view v; CK(kamd_index_get_view(gidx, &v));
  std::vector<std::string> target_names; std::vector<uint32_t> target_lens(v.target_lens, v.target_lens + v.n_targets);
  for (uint64_t i = 0; i < v.n_targets; i++) target_names.push_back(kamd_index_target_name(gidx, i));

  // ---- the reference's reader, S2 in place of processBuffer ----
  const int strand = !opt.strand_specific ? 0 : (opt.strand == ProgramOptions::StrandType::FR ? 1 : 2);
  kamd_quant_opts o{paired ? 1 : 0, opt.fld, opt.sd, opt.single_overhang ? 1 : 0, strand, 0, 0};
  FastqSequenceReader SR(opt);
  const size_t bufsize = 1ULL << 23;   // MasterProcessor::bufsize (src/ProcessReads.h)
  std::vector<char> buffer(bufsize);
  std::vector<std::pair<const char*, int>> seqs, names, quals; std::vector<uint32_t> flags; std::vector<std::string> umis;
  uint32_t flens[KAMD_MAX_FRAG_LEN] = {0}; uint64_t fld_used = 0, num_processed = 0;
  std::vector<uint64_t> off; std::vector<int32_t> len; std::vector<uint32_t> words; std::vector<uint16_t> lens16;
  uint32_t* d_words = nullptr; uint16_t* d_len = nullptr; size_t dw_cap = 0, dl_cap = 0;
  while (!SR.empty()) {
    int readbatch_id = 0;
    seqs.clear(); names.clear(); quals.clear(); flags.clear(); umis.clear();
    SR.fetchSequences(buffer.data(), (int)bufsize, seqs, names, quals, flags, umis, readbatch_id, false);
    if (seqs.empty()) continue;
    off.resize(seqs.size()); len.resize(seqs.size());
    int32_t max_len = 1;
    for (size_t i = 0; i < seqs.size(); i++) { off[i] = (uint64_t)(seqs[i].first - buffer.data()); len[i] = seqs[i].second; max_len = std::max(max_len, len[i]); }
    const uint64_t rec = kamd_packed_record_words(max_len);
    words.resize(seqs.size() * rec); lens16.resize(seqs.size());
    CK(kamd_pack_reads_host(buffer.data(), off.data(), len.data(), seqs.size(), max_len, words.data(), lens16.data()));
    if (words.size() > dw_cap) { if (d_words) (void)hipFree(d_words); dw_cap = words.size() * 2; HK(hipMalloc((void**)&d_words, dw_cap * 4)); }
    if (lens16.size() > dl_cap) { if (d_len) (void)hipFree(d_len); dl_cap = lens16.size() * 2; HK(hipMalloc((void**)&d_len, dl_cap * 2)); }
    HK(hipMemcpy(d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_len, lens16.data(), lens16.size() * 2, hipMemcpyHostToDevice));
    const uint64_t n_items = paired ? seqs.size() / 2 : seqs.size();
    CK(kamd_pseudoalign(ctx, &o, d_words, d_len, n_items, max_len));
    // ---- S3, first half: the fragment-length sample (tlencount of processBuffer, :981-1008) ----
    if (paired && opt.fld == 0.0 && fld_used < 10000) CK(kamd_fld_from_batch(ctx, &o, d_words, d_len, n_items, max_len, flens, &fld_used));
    num_processed += n_items;
  }
  // ---- S3: the classes and the effective lengths ----
  kamd_ec_result ec; CK(kamd_ec_finalize(ctx, &ec));
  std::vector<uint64_t> ec_off(ec.n_ecs + 1); std::vector<uint32_t> ec_ids(ec.nnz), counts(ec.n_ecs);
  CK(kamd_ec_download(ctx, ec_off.data(), ec_ids.data(), counts.data()));
  uint64_t num_pseudoaligned = 0, num_unique = 0;
  for (uint64_t e = 0; e < ec.n_ecs; e++) { num_pseudoaligned += counts[e]; if (ec_off[e + 1] - ec_off[e] == 1) num_unique += counts[e]; }
  std::vector<double> mean_fl(KAMD_MAX_FRAG_LEN);
  if (opt.fld == 0.0) kamd_mean_frag_lens_trunc(flens, mean_fl.data()); else kamd_trunc_gaussian_fld(0, KAMD_MAX_FRAG_LEN, opt.fld, opt.sd, mean_fl.data());
  std::vector<double> eff(v.n_targets);
  kamd_eff_lens(v.target_lens, v.n_targets, mean_fl.data(), eff.data());
  // ---- S4: the EM ----
  std::vector<double> alpha(v.n_targets), abz(v.n_targets); int32_t rounds = 0;
  CK(kamd_em_run(ctx, nullptr, nullptr, nullptr, nullptr, 0, eff.data(), v.n_targets, 10000, 50, alpha.data(), abz.data(), &rounds));
  fprintf(stderr, "[   em] the Expectation-Maximization algorithm ran for %d rounds\n", rounds);
  // ---- the reference's writers ----
  plaintext_aux(opt.output + "/run_info.json", std::to_string(v.n_targets), std::to_string(0), std::to_string(num_processed), std::to_string(num_pseudoaligned),
                std::to_string(num_unique), KALLISTO_VERSION, std::to_string(13), std::to_string(v.k), start_time, call);
  plaintext_writer(opt.output + "/abundance.tsv", target_names, alpha, eff, target_lens);
  kamd_ctx_destroy(ctx); kamd_index_free(gidx);
  return num_pseudoaligned == 0 ? 1 : 0;
}
