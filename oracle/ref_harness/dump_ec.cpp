// oracle/ref_harness/dump_ec.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Thin driver around the UNMODIFIED reference (linked from oracle/_ref/libkallisto_ref.a, which oracle/Makefile
// compiles from /root/reference where it lies).  It exposes the reference's intermediate results, which the
// `kallisto` CLI never prints, so that the C restatement (oracle/kallisto_oracle.c) and the HIP path can be pinned:
//
//   dump_ec quant   <index> <threads> [--single -l L -s S] [--fr|--rf] [--boot B --seed S] [--no-em] [--flens FILE] <fastq> [<fastq2>]
//        runs exactly what main.cpp:2632-2689 runs (KmerIndex::load -> ProcessReads -> FLD -> EMAlgorithm::run)
//        and prints
//          NPROC <n>                               (ProcessReads return value)
//          EC <comma separated sorted tr ids> <count>   (index.ecmapinv x collection.counts, as a sorted multiset;
//                                                        EC ids are discovery order in the reference and carry no meaning)
//          FLEN <len> <count>                      (non-zero entries of collection.flens)
//          TR <i> <length> <eff_len %.17g> <alpha %.17g> <alpha_before_zeroes %.17g>
//          BS <b> <i> <alpha %.17g>                (bootstrap replicates, Bootstrap::run_em, main.cpp:2744-2782)
//        --no-em: stops behind ProcessReads (NPROC, EC and FLEN lines only) -- the EC multiset does not depend on the number of
//        threads, so a full-size input can be pseudoaligned on all cores without waiting for the single-threaded EM
//        --flens FILE ("<len> <count>" per line): the reference's EM runs on ITS OWN equivalence classes but with this fragment-length
//        sample in place of the one ProcessReads drew (the FLEN lines still print the reference's own).  With more than one thread the
//        reference's sample -- the first 10 000 qualifying pairs to reach the mutex -- depends on the schedule; handing it the -t 1
//        sample (which a prefix run pins) makes effective lengths and abundances of a multi-threaded full-size run comparable.
//        (EMAlgorithm::run's round count goes to stderr, as in the CLI: "... ran for 1,372 rounds".)
//
//   dump_ec perread <index> <reads.txt> [--single]
//        reads.txt: one read per line, or "read1<TAB>read2".  For every line calls KmerIndex::match on each mate and
//        MinCollector::intersectKmers (ProcessReads.cpp:1058-1072) and prints
//          R <line#> <n_hits1> <n_hits2> <set | *>
//        plus, with --hits,  H <line#> <mate> <read_pos> <unitig_bp> <dist> <strand> <lb> <ub> <|ec|>
//
// Nothing here re-implements reference logic; it only calls it.
#include "common.h"
#include "KmerIndex.h"
#include "MinCollector.h"
#include "ProcessReads.h"
#include "EMAlgorithm.h"
#include "Bootstrap.h"
#include "weights.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>
#include <fstream>

static void print_set(const Roaring& r) {
  bool first = true;
  for (auto t : r) { printf("%s%u", first ? "" : ",", t); first = false; }
}

static int run_quant(int argc, char** argv) {
  ProgramOptions opt;
  opt.index = argv[2];
  opt.threads = atoi(argv[3]);
  opt.output = "/tmp/dump_ec_out";
  int boot = 0;
  bool no_em = false;
  std::string flens_file;
  for (int i = 4; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--single") opt.single_end = true;
    else if (a == "-l") opt.fld = atof(argv[++i]);
    else if (a == "-s") opt.sd = atof(argv[++i]);
    else if (a == "--fr") { opt.strand_specific = true; opt.strand = ProgramOptions::StrandType::FR; }
    else if (a == "--rf") { opt.strand_specific = true; opt.strand = ProgramOptions::StrandType::RF; }
    else if (a == "--single-overhang") opt.single_overhang = true;
    else if (a == "--no-jump") opt.no_jump = true;
    else if (a == "--union") opt.do_union = true;
    else if (a == "--boot") boot = atoi(argv[++i]);
    else if (a == "--no-em") no_em = true;
    else if (a == "--flens") flens_file = argv[++i];
    else if (a == "--seed") opt.seed = strtoull(argv[++i], nullptr, 10);
    else opt.files.push_back(a);
  }
  opt.bootstrap = boot;
  KmerIndex index(opt);
  index.load(opt);
  Transcriptome model;
  MinCollector collection(index, opt);
  MasterProcessor MP(index, opt, collection, model);
  int64_t nproc = ProcessReads(MP, opt);

  printf("NPROC %lld\n", (long long)nproc);
  {
    std::map<std::vector<uint32_t>, uint64_t> m;
    for (auto& e : index.ecmapinv) {
      std::vector<uint32_t> v;
      for (auto t : e.first) v.push_back(t);
      m[v] += collection.counts[e.second];
    }
    for (auto& kv : m) {
      printf("EC ");
      for (size_t i = 0; i < kv.first.size(); i++) printf("%s%u", i ? "," : "", kv.first[i]);
      printf(" %llu\n", (unsigned long long)kv.second);
    }
  }
  for (size_t i = 0; i < collection.flens.size(); i++)
    if (collection.flens[i]) printf("FLEN %zu %u\n", i, collection.flens[i]);
  if (no_em) return 0;
  if (!flens_file.empty()) {
    std::ifstream ff(flens_file);
    if (!ff) { fprintf(stderr, "dump_ec: cannot read %s\n", flens_file.c_str()); return 2; }
    std::fill(collection.flens.begin(), collection.flens.end(), 0u);
    size_t len; uint32_t cnt;
    while (ff >> len >> cnt) if (len < collection.flens.size()) collection.flens[len] = cnt;
  }

  std::vector<uint32_t> fld;
  if (opt.fld == 0.0) {
    fld = collection.flens;
    collection.compute_mean_frag_lens_trunc();
  } else {
    collection.init_mean_fl_trunc(opt.fld, opt.sd);
    fld = trunc_gaussian_counts(0, MAX_FRAG_LEN, opt.fld, opt.sd, 10000);
  }
  auto fl_means = get_frag_len_means(index.target_lens_, collection.mean_fl_trunc);
  EMAlgorithm em(collection.counts, index, collection, fl_means, opt);
  em.run(10000, 50, true, false);

  for (size_t i = 0; i < em.alpha_.size(); i++)
    printf("TR %zu %u %.17g %.17g %.17g\n", i, index.target_lens_[i], em.eff_lens_[i], em.alpha_[i],
           i < em.alpha_before_zeroes_.size() ? em.alpha_before_zeroes_[i] : -1.0);

  if (boot > 0) {
    std::mt19937_64 rand;
    rand.seed(opt.seed);
    std::vector<size_t> seeds;
    for (int s = 0; s < boot; ++s) seeds.push_back(rand());
    for (int b = 0; b < boot; ++b) {
      Bootstrap bs(collection.counts, index, collection, em.eff_lens_, seeds[b], fl_means, opt);
      auto res = bs.run_em();
      for (size_t i = 0; i < res.alpha_.size(); i++) printf("BS %d %zu %.17g\n", b, i, res.alpha_[i]);
    }
  }
  return 0;
}

static int run_perread(int argc, char** argv) {
  ProgramOptions opt;
  opt.index = argv[2];
  opt.threads = 1;
  std::string reads = argv[3];
  bool single = false, hits = false;
  for (int i = 4; i < argc; i++) {
    if (!strcmp(argv[i], "--single")) single = true;
    if (!strcmp(argv[i], "--hits")) hits = true;
  }
  KmerIndex index(opt);
  index.load(opt);
  MinCollector tc(index, opt);
  std::ifstream in(reads);
  std::string line;
  std::vector<std::pair<const_UnitigMap<Node>, int32_t>> v1, v2;
  long ln = 0;
  while (std::getline(in, line)) {
    std::string s1 = line, s2;
    size_t tab = line.find('\t');
    bool paired = !single && tab != std::string::npos;
    if (tab != std::string::npos) { s1 = line.substr(0, tab); s2 = line.substr(tab + 1); }
    v1.clear(); v2.clear();
    index.match(s1.c_str(), (int)s1.size(), v1, !paired);
    if (paired) index.match(s2.c_str(), (int)s2.size(), v2, !paired);
    if (hits) {
      for (int mate = 0; mate < 2; mate++) {
        auto& v = mate ? v2 : v1;
        for (auto& h : v) {
          auto blk = h.first.getData()->get_mc_contig(h.first.dist);
          printf("H %ld %d %d %zu %zu %d %u %u %llu\n", ln, mate, h.second, (size_t)h.first.size, (size_t)h.first.dist,
                 (int)h.first.strand, blk.first, blk.second,
                 (unsigned long long)h.first.getData()->ec[h.first.dist].getIndices().cardinality());
        }
      }
    }
    size_t n1 = v1.size(), n2 = v2.size();
    Roaring u;
    tc.intersectKmers(v1, v2, !paired, u);
    u &= index.onlist_sequences;
    printf("R %ld %zu %zu ", ln, n1, n2);
    if (u.isEmpty()) printf("*"); else print_set(u);
    printf("\n");
    ln++;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: dump_ec quant <index> <threads> [opts] <fastq> [<fastq2>]\n"
                    "       dump_ec perread <index> <reads.txt> [--single] [--hits]\n");
    return 2;
  }
  if (!strcmp(argv[1], "quant")) return run_quant(argc, argv);
  if (!strcmp(argv[1], "perread")) return run_perread(argc, argv);
  return 2;
}
