// bus_main.cpp -- the two consumers next to `quant` that run the same kernels:
//
//   kallisto_amd_quant bus -x bulk ...   `kallisto bus -x bulk` (src/main.cpp:2336-2611 driver, option handling :541-766,:1048-1230;
//                                        BUSProcessor::processBuffer src/ProcessReads.cpp:1380-1832 in its bulk form: every file /
//                                        pair of files is a sample, barcode = sample, no UMI, the same match + intersectKmers as
//                                        quant, no position filter (`single_overhang = true`, :762), per-sample fragment lengths)
//   kallisto_amd_quant quant-tcc ...     `kallisto quant-tcc` (src/main.cpp:2802-3220): EMAlgorithm::run per row of a
//                                        transcript-compatibility-count matrix over the classes of an EC file
//
// The reference writes one BUS record per pseudoaligned read and leaves the counting to `bustools sort` / `bustools count`.
// In bulk mode all records of a sample with the same class are identical (barcode = sample, UMI = -1, count = 1), so this
// front-end writes what `bustools sort` would make of that file: one record per (sample, class) with the number of reads
// in `count`, ordered by barcode, then class -- the per-sample EC counts the quant path produces on the GPU anyway.
// Host code only; everything that computes runs through the C ABI.
#include <map>
#include <unordered_map>

#include "kamd_frontend.h"

namespace kamd_fe {
namespace {

const uint32_t FAKE_BARCODE_LEN = 16;   // BUSFORMAT_FAKE_BARCODE_LEN, src/BUSTools.h:9

struct BusRecord {                       // BUSData, src/BUSData.h:31-39
  uint64_t barcode, umi;
  int32_t ec;
  uint32_t count, flags, pad;
};
static_assert(sizeof(BusRecord) == 32, "BUS records are 32 bytes");

std::string barcode_string(uint64_t x, size_t len) {   // binaryToString, src/BUSData.cpp (2 bits per base, A C G T, most significant first)
  std::string s(len, 'A');
  for (size_t i = 0; i < len; i++) s[len - 1 - i] = "ACGT"[(x >> (2 * i)) & 3];
  return s;
}

void usage_bus() {
  std::cout << "kallisto_amd " << KALLISTO_COMPAT_VERSION << "-compatible (MI355X)\n"
            << "Generates the BUS file of bulk RNA-seq samples (`kallisto bus -x bulk`)\n\n"
            << "Usage: kallisto_amd_quant bus [arguments] FASTQ-files\n\n"
            << "Required arguments:\n"
            << "-i, --index=STRING            Filename for the kallisto index to be used for pseudoalignment\n"
            << "-o, --output-dir=STRING       Directory to write output to\n"
            << "-x, --technology=STRING       Must be `bulk`: every file (pair of files with --paired) is one sample\n\n"
            << "Optional arguments:\n"
            << "-B, --batch=FILE              Samples from a batch file (lines: id file1 [file2]) instead of the command line\n"
            << "    --paired                  Treat reads as paired\n"
            << "    --fr-stranded / --rf-stranded / --unstranded\n"
            << "    --union, --no-jump        As in quant\n"
            << "-t, --threads=INT             Host threads (default: 1)\n"
            << "    --bus-per-read            output.bus with one record per pseudoaligned read (count 1), as the reference writes it; by default\n"
            << "                              the records of a sample and class are collapsed into one with their number in `count` -- what\n"
            << "                              `bustools sort` makes of the reference's file.  Either way the records are sorted by barcode and\n"
            << "                              class, and matrix.ec lists the classes that occur, numbered in order of first appearance (the\n"
            << "                              reference numbers the index's classes first; class ids are arbitrary on both sides)\n"
            << "    --verbose                 Print out progress information\n";
}

struct VecHash {
  size_t operator()(const std::vector<uint32_t>& v) const {
    uint64_t h = 0x9e3779b97f4a7c15ULL ^ v.size();
    for (uint32_t x : v) { h ^= x + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2); h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; }
    return (size_t)h;
  }
};

}  // namespace

int bus_main(int argc, char** argv) {
  if (argc == 2) { usage_bus(); return 0; }
  std::string index, output, technology, batch_file, val;
  std::vector<std::string> files;
  bool paired = false, verbose = false, do_union = false, no_jump = false, per_read = false;
  int strand = 0, threads = 1;
  uint64_t batch = 4u << 20;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    if (take(a, "-i", "--index", i, argc, argv, val)) index = val;
    else if (take(a, "-o", "--output-dir", i, argc, argv, val)) output = val;
    else if (take(a, "-x", "--technology", i, argc, argv, val)) { technology = val; for (auto& ch : technology) ch = (char)toupper(ch); }
    else if (take(a, "-t", "--threads", i, argc, argv, val)) threads = atoi(val.c_str());
    else if (take(a, "-B", "--batch", i, argc, argv, val)) batch_file = val;
    else if (take(a, nullptr, "--batch-size", i, argc, argv, val)) batch = strtoull(val.c_str(), nullptr, 10);
    else if (a == "--paired") paired = true;
    else if (a == "--fr-stranded") strand = 1;
    else if (a == "--rf-stranded") strand = 2;
    else if (a == "--unstranded") strand = 0;
    else if (a == "--union") do_union = true;
    else if (a == "--no-jump") no_jump = true;
    else if (a == "--bus-per-read") per_read = true;
    else if (a == "--verbose") verbose = true;
    else if (a == "-l" || a == "--list" || a == "-b" || a == "--bam" || a == "-n" || a == "--num" || a == "--genomebam" || a == "-g" || a == "--gtf" ||
             a == "-c" || a == "--chromosomes" || a == "-T" || a == "--tag" || a == "--long" || a == "-P" || a == "--platform" || a == "-r" ||
             a == "--threshold" || a == "--unmapped" || a == "--aa" || a == "--inleaved" || a == "-N" || a == "--numReads" ||
             a == "--batch-barcodes" || a == "--dfk-onlist") {
      std::cerr << "Error: option " << a << " is outside the GPU bulk path; use the reference kallisto for it" << std::endl; return 1;
    } else if (!a.empty() && a[0] == '-') { std::cerr << "Error: unknown option " << a << std::endl; usage_bus(); return 1; }
    else files.push_back(a);
  }
  // CheckOptionsBus (src/main.cpp:1010-1230), the branch of `-x bulk` / batch files without a technology
  bool ok = true;
  struct stat st;
  std::cerr << std::endl;
  if (index.empty()) { std::cerr << "Error: kallisto index file missing" << std::endl; ok = false; }
  else if (stat(index.c_str(), &st) != 0) { std::cerr << "Error: kallisto index file not found " << index << std::endl; ok = false; }
  if (output.empty()) { std::cerr << "Error: need to specify output directory " << output << std::endl; ok = false; }
  else if (stat(output.c_str(), &st) == 0) {
    if (!S_ISDIR(st.st_mode)) { std::cerr << "Error: file " << output << " exists and is not a directory" << std::endl; ok = false; }
  } else if (mkdir(output.c_str(), 0777) == -1) { std::cerr << "Error: could not create directory " << output << std::endl; ok = false; }
  if (threads <= 0) { std::cerr << "Error: invalid number of threads " << threads << std::endl; ok = false; }
  if (!technology.empty() && technology != "BULK") {
    std::cerr << "Error: only `-x bulk` runs on the GPU path; single-cell technologies stay with the reference kallisto" << std::endl; ok = false;
  }
  std::vector<std::string> batch_ids;
  std::vector<std::vector<std::string>> batch_files;
  if (batch_file.empty()) {
    if (ok && technology.empty()) { std::cerr << "Error: the technology must be specified via -x, use \"bulk\" for regular RNA-seq reads" << std::endl; ok = false; }
    if (ok && files.empty()) { std::cerr << "Error: Missing read files" << std::endl; ok = false; }
    if (ok && paired && files.size() % 2 != 0) { std::cerr << "Error: paired-end mode requires an even number of input files" << std::endl; ok = false; }
    for (size_t i = 0, s = 0; ok && i < files.size(); s++) {
      batch_ids.push_back("batch" + std::to_string(s));
      std::vector<std::string> fs{files[i++]};
      if (paired) fs.push_back(files[i++]);
      for (const auto& f : fs) if (stat(f.c_str(), &st) != 0) { std::cerr << "Error: file not found " << f << std::endl; ok = false; }
      batch_files.push_back(fs);
    }
  } else {
    std::cerr << "[bus] will try running read files supplied in batch file" << std::endl;
    if (paired) std::cerr << "[bus] --paired ignored; single/paired-end is inferred from number of files supplied" << std::endl;
    if (!files.empty()) { std::cerr << "Error: cannot specify batch mode and supply read files" << std::endl; ok = false; }
    else {
      std::ifstream bf(batch_file);
      if (!bf.is_open()) { std::cerr << "Error: file not found " << batch_file << std::endl; ok = false; }
      std::string line;
      bool first = true;
      while (ok && std::getline(bf, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        std::string id, f1, f2;
        ss >> id;
        if (id.empty() || id[0] == '#') continue;
        ss >> f1 >> f2;
        if (first) { paired = !f2.empty(); first = false; }
        if (f1.empty() || (paired && f2.empty()) || (!paired && !f2.empty())) { std::cerr << "Error: batch file malformatted" << std::endl; ok = false; break; }
        std::vector<std::string> fs{f1};
        if (paired) fs.push_back(f2);
        for (const auto& f : fs) if (stat(f.c_str(), &st) != 0) { std::cerr << "Error: file not found " << f << std::endl; ok = false; }
        batch_ids.push_back(id);
        batch_files.push_back(fs);
      }
      if (ok && batch_ids.empty()) { std::cerr << "Error: Missing read files" << std::endl; ok = false; }
    }
  }
  if (!ok) { std::cerr << std::endl; usage_bus(); return 1; }
  const std::string start_time = now_string(), call = call_string(argc, argv);
  // batch_id_mapping (src/ProcessReads.h:211-223): lines with the same id share a barcode
  std::vector<uint64_t> barcode_of(batch_ids.size());
  {
    std::unordered_map<std::string, uint64_t> seen;
    for (size_t i = 0; i < batch_ids.size(); i++) {
      auto it = seen.find(batch_ids[i]);
      if (it == seen.end()) it = seen.emplace(batch_ids[i], (uint64_t)seen.size()).first;
      barcode_of[i] = it->second;
    }
  }

  (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);   // (see quant_main.cpp)
  threads = std::min(threads, effective_cpus());
  kamd_index* idx = nullptr;
  KX(kamd_index_load(index.c_str(), threads, &idx));
  kamd_index_view v; KX(kamd_index_get_view(idx, &v));
  std::cerr << "\n[index] k-mer length: " << v.k << "\n[index] number of targets: " << v.n_targets << "\n[index] number of k-mers: " << v.n_kmers << std::endl;
  kamd_ctx* ctx = nullptr;
  KX(kamd_ctx_create(0, nullptr, &ctx));
  KX(kamd_index_upload(ctx, idx));
  // single-end: match(partial) + intersectKmers without the position filter == quant --single --single-overhang (the mean / sd
  // only satisfy kamd_pseudoalign's argument check: no filter reads them)
  kamd_quant_opts qo{paired ? 1 : 0, paired ? 0.0 : 200.0, paired ? 0.0 : 20.0, 1, strand, no_jump ? 1 : 0, do_union ? 1 : 0};
  std::cerr << "[quant] running in " << (paired ? "paired-end" : "single-end") << " mode" << std::endl;
  for (const auto& fs : batch_files) {
    if (paired) std::cerr << "[quant] will process pair 1: " << fs[0] << std::endl << "                             " << fs[1] << std::endl;
    else std::cerr << "[quant] will process file 1: " << fs[0] << std::endl;
  }
  std::cerr << "[quant] finding pseudoalignments for all files ..."; std::cerr.flush();

  std::unordered_map<std::vector<uint32_t>, int32_t, VecHash> ec_of;   // transcript set -> id in matrix.ec (order of first appearance)
  std::vector<const std::vector<uint32_t>*> ec_sets;
  std::map<uint64_t, std::map<int32_t, uint64_t>> per_barcode;          // barcode -> class -> reads
  std::vector<std::vector<uint32_t>> sample_flens(batch_ids.size(), std::vector<uint32_t>(KAMD_MAX_FRAG_LEN, 0));
  uint64_t n_processed = 0, num_pseudoaligned = 0, num_unique = 0;
  double pack_s = 0.0;
  // one pipeline for all samples (text rings and buffers are allocated once); the sample under way is behind these two
  uint32_t* flens = nullptr;
  uint64_t fld_used = 0;
  auto run_batch = [&](int, PackedBatch& b, std::string& err) -> int {
    const bool want_fld = paired && fld_used < 10000;
    int rc = 0;
    if (want_fld) rc = kamd_fld_prefetch(ctx, &qo, b.d_words, b.d_len, b.n_items, b.max_len);
    if (!rc) rc = kamd_pseudoalign(ctx, &qo, b.d_words, b.d_len, b.n_items, b.max_len);
    if (!rc && want_fld) rc = kamd_fld_from_batch(ctx, &qo, b.d_words, b.d_len, b.n_items, b.max_len, flens, &fld_used);
    if (rc) err = kamd_last_error();
    return rc;
  };
  UnitFeeder feeder({ctx}, {0}, run_batch, [] { return false; });
  for (size_t sidx = 0; sidx < batch_files.size(); sidx++) {
    KX(kamd_ec_reset(ctx));                         // a sample starts from an empty collector
    flens = sample_flens[sidx].data();
    fld_used = 0;                                   // tlencounts[id]: the first 10 000 qualifying pairs of THIS sample (ProcessReads.cpp:1395-1399)
    {
      MultiPipe pipe(1, run_batch, [] { return false; });
      auto reset_sample = [&]() -> int { std::fill(sample_flens[sidx].begin(), sample_flens[sidx].end(), 0u); fld_used = 0; return kamd_ec_reset(ctx); };
      if (feed_files(batch_files[sidx], paired, batch, std::max(1, threads), threads, verbose, pipe, n_processed, pack_s, &feeder, reset_sample)) return 1;
      pipe.finish();
      if (pipe.failed()) { std::cerr << "Error: " << pipe.error() << std::endl; return 1; }
    }
    kamd_ec_result ec;
    KX(kamd_ec_finalize(ctx, &ec));
    std::vector<uint64_t> ec_off(ec.n_ecs + 1); std::vector<uint32_t> ec_ids(std::max<uint64_t>(ec.nnz, 1)), counts(std::max<uint64_t>(ec.n_ecs, 1));
    KX(kamd_ec_download(ctx, ec_off.data(), ec_ids.data(), counts.data()));
    auto& mine = per_barcode[barcode_of[sidx]];
    for (uint64_t e = 0; e < ec.n_ecs; e++) {
      if (counts[e] == 0) continue;
      std::vector<uint32_t> key(ec_ids.begin() + ec_off[e], ec_ids.begin() + ec_off[e + 1]);
      auto it = ec_of.find(key);
      if (it == ec_of.end()) { it = ec_of.emplace(std::move(key), (int32_t)ec_of.size()).first; ec_sets.push_back(&it->first); }
      mine[it->second] += counts[e];
      num_pseudoaligned += counts[e];
      if (ec_off[e + 1] - ec_off[e] == 1) num_unique += counts[e];
    }
  }
  std::cerr << " done" << std::endl;
  std::cerr << "[quant] processed " << n_processed << " reads, " << num_pseudoaligned << " reads pseudoaligned";
  if (num_pseudoaligned == 0) std::cerr << "[~warn] no reads pseudoaligned.";
  std::cerr << std::endl;

  {  // output.bus: header (writeBUSHeader, src/BUSTools.cpp:5-14) + sorted, collapsed records
    std::ofstream of(output + "/output.bus", std::ios::out | std::ios::binary);
    if (!of.is_open()) { std::cerr << "Error: Couldn't open file: " << output << "/output.bus" << std::endl; return 1; }
    const uint32_t version = 1, bclen = FAKE_BARCODE_LEN, umilen = 1;
    const std::string text = "BUS file produced by kallisto";
    const uint32_t tlen = (uint32_t)text.size();
    of.write("BUS\0", 4);
    of.write((const char*)&version, 4); of.write((const char*)&bclen, 4); of.write((const char*)&umilen, 4); of.write((const char*)&tlen, 4);
    of.write(text.data(), tlen);
    for (const auto& bcs : per_barcode)
      for (const auto& ec_n : bcs.second) {
        uint64_t left = ec_n.second;
        while (left) {   // `count` is 32 bits; --bus-per-read: one record per read, as BUSProcessor::processBuffer writes them (src/ProcessReads.cpp:1600-1640)
          BusRecord r{bcs.first, ~0ULL, ec_n.first, per_read ? 1u : (uint32_t)std::min<uint64_t>(left, 0xFFFFFFFFu), 0, 0};
          of.write((const char*)&r, sizeof r);
          left -= r.count;
        }
      }
  }
  {  // matrix.ec (writeECList, src/PlaintextWriter.cpp:235-266): the classes that occur, in order of first appearance
    std::ofstream of(output + "/matrix.ec");
    std::string line;
    for (size_t e = 0; e < ec_sets.size(); e++) {
      line = std::to_string(e) + "\t";
      const auto& s = *ec_sets[e];
      for (size_t j = 0; j < s.size(); j++) { if (j) line += ','; line += std::to_string(s[j]); }
      of << line << "\n";
    }
  }
  { std::ofstream of(output + "/matrix.cells"); for (const auto& id : batch_ids) of << id << "\n"; }             // writeCellIds
  { std::ofstream of(output + "/matrix.sample.barcodes"); for (uint64_t b : barcode_of) of << barcode_string(b, FAKE_BARCODE_LEN) << "\n"; }
  const uint64_t n_on = onlist_targets(v);
  { std::ofstream of(output + "/transcripts.txt"); for (uint64_t t = 0; t < n_on; t++) of << kamd_index_target_name(idx, t) << "\n"; }
  if (paired) {   // flens.txt: one line of MAX_FRAG_LEN counts per sample (src/main.cpp:2421-2452)
    std::ofstream of(output + "/flens.txt");
    for (const auto& fl : sample_flens) {
      for (size_t i = 0; i < fl.size(); i++) { if (i) of << " "; of << fl[i]; }
      of << "\n";
    }
  }
  write_run_info(output + "/run_info.json", n_on, 0, n_processed, num_pseudoaligned, num_unique, v.k, start_time, call);
  std::cerr << std::endl;
  kamd_ctx_destroy(ctx);
  kamd_index_free(idx);
  return num_pseudoaligned == 0 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------- quant-tcc
namespace {
void usage_tcc() {
  std::cout << "kallisto_amd " << KALLISTO_COMPAT_VERSION << "-compatible (MI355X)\n"
            << "Quantifies abundance from pre-computed transcript-compatibility counts (`kallisto quant-tcc`)\n\n"
            << "Usage: kallisto_amd_quant quant-tcc [arguments] transcript-compatibility-counts-file\n\n"
            << "Required arguments:\n"
            << "-o, --output-dir=STRING       Directory to write output to\n"
            << "-i, --index=STRING            Filename for the kallisto index (target names and lengths)\n"
            << "-e, --ec-file=FILE            File containing equivalence classes (matrix.ec)\n\n"
            << "Optional arguments:\n"
            << "-f, --fragment-file=FILE      File containing fragment length distribution(s) (flens.txt)\n"
            << "-l, --fragment-length=DOUBLE  Estimated average fragment length\n"
            << "-s, --sd=DOUBLE               Estimated standard deviation of fragment length\n"
            << "-g, --genemap=FILE            File for mapping transcripts to genes (transcript, gene[, common name] per line):\n"
            << "                              gene-level sums beside every output (matrix.abundance.gene[.tpm].mtx, genes.txt, abundance.gene*.tsv)\n"
            << "    --matrix-to-files         Also write abundance_N.tsv for every row of the matrix\n"
            << "    --matrix-to-directories   ... as abundance_N/abundance.tsv, a directory per row\n"
            << "-b, --bootstrap-samples=INT   Number of bootstrap samples (default: 0; plaintext, with --matrix-to-files)\n"
            << "    --seed=INT                Seed for the bootstrap sampling (default: 42)\n"
            << "    --plaintext               Accepted (all output is plaintext)\n"
            << "-t, --threads=INT             Accepted (the samples run one after the other on the GPU)\n";
}
// writeSparseBatchMatrix, src/PlaintextWriter.h:72-104
void write_sparse(const std::string& path, const std::vector<std::vector<std::pair<int32_t, double>>>& data, uint64_t cols) {
  uint64_t n = 0;
  for (const auto& r : data) for (const auto& p : r) if (p.second != 0.0) ++n;
  std::ofstream of(path);
  of << "%%MatrixMarket matrix coordinate real general\n" << data.size() << "\t" << cols << "\t" << n << "\n";
  for (size_t j = 0; j < data.size(); j++)
    for (const auto& p : data[j]) if (p.second != 0.0) of << (j + 1) << "\t" << (p.first + 1) << "\t" << p.second << "\n";
}
}  // namespace

int tcc_main(int argc, char** argv) {
  if (argc == 2) { usage_tcc(); return 0; }
  std::string index, output, ec_file, fld_file, tcc_file, genemap, val;
  double fld = 0.0, sd = 0.0;
  int bootstrap = 0, threads = 1;
  uint64_t seed = 42;
  bool matrix_to_files = false, matrix_to_dirs = false;
  std::vector<std::string> pos;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    if (take(a, "-i", "--index", i, argc, argv, val)) index = val;
    else if (take(a, "-o", "--output-dir", i, argc, argv, val)) output = val;
    else if (take(a, "-e", "--ec-file", i, argc, argv, val)) ec_file = val;
    else if (take(a, "-f", "--fragment-file", i, argc, argv, val)) fld_file = val;
    else if (take(a, "-l", "--fragment-length", i, argc, argv, val)) fld = atof(val.c_str());
    else if (take(a, "-s", "--sd", i, argc, argv, val)) sd = atof(val.c_str());
    else if (take(a, "-b", "--bootstrap-samples", i, argc, argv, val)) bootstrap = atoi(val.c_str());
    else if (take(a, "-d", "--seed", i, argc, argv, val)) seed = strtoull(val.c_str(), nullptr, 10);
    else if (take(a, "-t", "--threads", i, argc, argv, val)) threads = atoi(val.c_str());
    else if (a == "--matrix-to-files") matrix_to_files = true;
    else if (a == "--matrix-to-directories") matrix_to_files = matrix_to_dirs = true;   // (implies the files, src/main.cpp:499-502)
    else if (a == "--plaintext") {}
    else if (take(a, "-g", "--genemap", i, argc, argv, val)) genemap = val;
    else if (a == "-T" || a == "--txnames" || a == "--long" || a == "-P" || a == "--platform" ||
             a == "-G" || a == "--gtf" || a.rfind("--gtf=", 0) == 0 || a == "-p" || a == "--priors" || a.rfind("--priors=", 0) == 0) {
      std::cerr << "Error: option " << a << " is outside the GPU quant-tcc path; use the reference kallisto for it" << std::endl; return 1;
    } else if (!a.empty() && a[0] == '-') { std::cerr << "Error: unknown option " << a << std::endl; usage_tcc(); return 1; }
    else pos.push_back(a);
  }
  // CheckOptionsTCCQuant (src/main.cpp:1807-1960)
  bool ok = true;
  struct stat st;
  std::cerr << std::endl;
  if (index.empty()) { std::cerr << "Error: either a kallisto index file or a transcripts file need to be supplied" << std::endl; ok = false; }
  else if (stat(index.c_str(), &st) != 0) { std::cerr << "Error: kallisto index file not found " << index << std::endl; ok = false; }
  if (pos.size() != 1) { std::cerr << "Error: transcript-compatibility counts file missing" << std::endl; ok = false; }
  else { tcc_file = pos[0]; if (stat(tcc_file.c_str(), &st) != 0) { std::cerr << "Error: transcript-compatibility counts file not found " << tcc_file << std::endl; ok = false; } }
  if (ec_file.empty()) { std::cerr << "Error: equivalence class file must be supplied (-e)" << std::endl; ok = false; }
  else if (stat(ec_file.c_str(), &st) != 0) { std::cerr << "Error: equivalence class file not found " << ec_file << std::endl; ok = false; }
  if (!genemap.empty() && stat(genemap.c_str(), &st) != 0) { std::cerr << "Error: file for mapping transcripts to genes not found " << genemap << std::endl; ok = false; }
  if (!fld_file.empty() && stat(fld_file.c_str(), &st) != 0) { std::cerr << "Error: fragment length distribution file not found " << fld_file << std::endl; ok = false; }
  if ((fld != 0.0 || sd != 0.0) && !fld_file.empty()) { std::cerr << "Error: cannot supply mean or sd while also supplying a fragment length distribution file" << std::endl; ok = false; }
  if ((fld != 0.0 && sd == 0.0) || (sd != 0.0 && fld == 0.0)) { std::cerr << "Error: cannot supply mean/sd without supplying both -l and -s" << std::endl; ok = false; }
  if (fld < 0.0) { std::cerr << "Error: invalid value for mean fragment length " << fld << std::endl; ok = false; }
  if (sd < 0.0) { std::cerr << "Error: invalid value for fragment length standard deviation " << sd << std::endl; ok = false; }
  if (threads <= 0) { std::cerr << "Error: invalid number of threads " << threads << std::endl; ok = false; }
  if (bootstrap < 0) { std::cerr << "Error: number of bootstrap samples must be a non-negative integer." << std::endl; ok = false; }
  if (output.empty()) { std::cerr << "Error: need to specify output directory " << output << std::endl; ok = false; }
  else if (stat(output.c_str(), &st) == 0) {
    if (!S_ISDIR(st.st_mode)) { std::cerr << "Error: file " << output << " exists and is not a directory" << std::endl; ok = false; }
  } else if (mkdir(output.c_str(), 0777) == -1) { std::cerr << "Error: could not create directory " << output << std::endl; ok = false; }
  if (!ok) { std::cerr << std::endl; usage_tcc(); return 1; }

  kamd_index* idx = nullptr;
  KX(kamd_index_load(index.c_str(), threads, &idx));
  kamd_index_view v; KX(kamd_index_get_view(idx, &v));
  const uint64_t T = onlist_targets(v);   // index.load(opt, false, false): the D-list's pseudo-targets are dropped (KmerIndex.cpp:1551-1558)
  std::cerr << "\n[index] k-mer length: " << v.k << "\n[index] number of targets: " << T << std::endl;

  // KmerIndex::loadECsFromFile (src/KmerIndex.cpp:1561-1600)
  std::vector<uint64_t> ec_off{0};
  std::vector<uint32_t> ec_ids;
  {
    std::ifstream in(ec_file);
    if (!in.is_open()) { std::cerr << "Error: could not open file " << ec_file << std::endl; return 1; }
    std::string line;
    int64_t i = 0;
    while (std::getline(in, line)) {
      std::stringstream ss(line);
      long long ec = -1; std::string trs;
      ss >> ec >> trs;
      if (ec != i) { std::cerr << "Error: equivalence class file has a misplaced equivalence class. Found " << ec << ", expected " << i << std::endl; return 1; }
      std::vector<uint32_t> ids;
      std::stringstream s2(trs);
      std::string tok;
      while (std::getline(s2, tok, ',')) {
        const int x = atoi(tok.c_str());
        if (x < 0 || (uint64_t)x >= T) { std::cerr << "Error: equivalence class file has invalid value: " << tok << " in " << trs << std::endl; return 1; }
        ids.push_back((uint32_t)x);
      }
      std::sort(ids.begin(), ids.end());
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
      if (ids.empty()) { std::cerr << "Error: equivalence class file has an empty equivalence class: " << ec << std::endl; return 1; }
      ec_ids.insert(ec_ids.end(), ids.begin(), ids.end());
      ec_off.push_back(ec_ids.size());
      i++;
    }
  }
  const uint64_t n_ecs = ec_off.size() - 1;
  std::cerr << "[index] number of equivalence classes loaded from file: " << n_ecs << std::endl;

  // the TCC file (src/main.cpp:2815-2898): MatrixMarket rows = samples, columns = classes; or "class count" lines (one sample)
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> rows;
  bool is_matrix = false;
  {
    std::ifstream in(tcc_file);
    if (!in.is_open()) { std::cerr << "Error: could not open file " << tcc_file << std::endl; return 1; }
    std::string line;
    bool firstline = true;
    uint64_t nrow = 0, ncol = 0, nlines = 0, n_read = 0;
    long long prev_row = 0, prev_col = 0;
    while (std::getline(in, line)) {
      if (firstline) {
        firstline = false;
        if (line.rfind("%%MatrixMarket", 0) == 0) {
          std::cerr << "[tcc] Parsing transcript-compatibility counts (TCC) file as a matrix file" << std::endl;
          is_matrix = true;
          while (std::getline(in, line) && line.rfind("%", 0) == 0) {}
          std::stringstream ss(line);
          ss >> nrow >> ncol >> nlines;
          std::cerr << "[tcc] Matrix dimensions: " << nrow << " x " << ncol << std::endl;
          rows.assign(nrow, {});
          continue;
        }
        std::cerr << "[tcc] Transcript-compatibility counts (TCC) file is not in matrix format; it will not be parsed as a matrix file" << std::endl;
        rows.assign(1, {});
      }
      std::stringstream ss(line);
      long long row = 0, col = 0, valn = 0;
      if (is_matrix) {
        if (n_read >= nlines) { std::cerr << "[tcc] Warning: TCC matrix file contains additional lines which will not be read; only " << nlines
                                          << " entries, as specified on the first line, will be read." << std::endl; break; }
        ss >> row >> col >> valn;
        if (row > (long long)nrow || col > (long long)ncol) { std::cerr << "Error: TCC matrix file is malformed; row numbers or column numbers exceed the dimensions of the matrix." << std::endl; return 1; }
      } else { ss >> col >> valn; col += 1; row = 1; }
      if (row <= 0 || col <= 0) { std::cerr << "Error: Invalid indices in TCC file." << std::endl; return 1; }
      if (row < prev_row || (row == prev_row && col <= prev_col)) { std::cerr << "Error: TCC file is not sorted." << std::endl; return 1; }
      if ((uint64_t)col > n_ecs) { std::cerr << "Error: TCC file names equivalence class " << col - 1 << " but the EC file holds " << n_ecs << std::endl; return 1; }
      prev_row = row; prev_col = col;
      rows[row - 1].push_back({(uint32_t)(col - 1), (uint32_t)valn});
      n_read++;
    }
    if (is_matrix && n_read < nlines) { std::cerr << "Error: Found only " << n_read << " entries in TCC matrix file, expected " << nlines << std::endl; return 1; }
  }
  const size_t nrow = rows.size();

  // fragment length distributions (src/main.cpp:2929-2972)
  const bool calc_eff = !fld_file.empty() || fld != 0.0;
  std::vector<std::vector<uint32_t>> flds;
  if (!fld_file.empty()) {
    std::ifstream in(fld_file);
    if (!in.is_open()) { std::cerr << "Error: could not open file " << fld_file << std::endl; return 1; }
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty() || line[0] == '#') continue;
      std::vector<uint32_t> vals;
      std::stringstream ss(line);
      std::string tok;
      while (std::getline(ss, tok, ' ')) {
        const int x = atoi(tok.c_str());
        if (x < 0) { std::cerr << "Error: Fragment length distribution file contains invalid value: " << x << std::endl; return 1; }
        vals.push_back((uint32_t)x);
      }
      if (vals.size() != KAMD_MAX_FRAG_LEN) { std::cerr << "Error: Fragment length distribution file contains a line with " << vals.size() << " values; expected: " << KAMD_MAX_FRAG_LEN << std::endl; return 1; }
      flds.push_back(vals);
    }
    if (flds.size() != 1 && flds.size() != nrow) { std::cerr << "Error: Fragment length distribution file contains " << flds.size() << " valid lines; expected: " << nrow << std::endl; return 1; }
  }

  { std::ofstream of(output + "/transcripts.txt"); for (uint64_t t = 0; t < T; t++) of << kamd_index_target_name(idx, t) << "\n"; }
  // -g: transcripts -> genes (Transcriptome::parseGeneMap, src/GeneModel.cpp:580-632); gene-level sums beside every transcript-level output
  GeneMap genes;
  const bool gene_level = !genemap.empty();
  if (gene_level) {
    std::vector<std::string> names(T);
    for (uint64_t t = 0; t < T; t++) names[t] = kamd_index_target_name(idx, t);
    std::string err;
    if (!parse_genemap(genemap, names, &genes, &err)) { std::cerr << err << std::endl; return 1; }
  }
  std::vector<std::vector<std::pair<int32_t, double>>> gab_m(gene_level ? rows.size() : 0), gtpm_m(gene_level ? rows.size() : 0);
  std::vector<double> gc, gc_tpm;
  kamd_ctx* ctx = nullptr;
  KX(kamd_ctx_create(0, nullptr, &ctx));
  KX(kamd_ec_upload(ctx, ec_off.data(), ec_ids.data(), nullptr, n_ecs));   // the EC matrix goes to the device once
  std::cerr << "[quant] Running EM algorithm..." << std::endl;
  std::vector<std::vector<std::pair<int32_t, double>>> ab_m(nrow), tpm_m(nrow), eff_m(nrow);
  std::vector<std::pair<double, double>> fld_m(nrow);
  std::vector<uint32_t> counts(std::max<uint64_t>(n_ecs, 1));
  std::vector<double> mft(KAMD_MAX_FRAG_LEN), eff(T), alpha(T), abz(T), tpm(T);
  kamd_index_view vt = v; vt.n_targets = T;   // the writers see the on-list targets only
  for (size_t id = 0; id < nrow; id++) {
    std::cerr << "[quant] Processing sample/cell " << id << std::endl;
    std::fill(counts.begin(), counts.end(), 0u);
    uint64_t total = 0;
    for (const auto& p : rows[id]) { counts[p.first] = p.second; total += p.second; }
    if (calc_eff) {
      if (fld != 0.0) {
        kamd_trunc_gaussian_fld(0, KAMD_MAX_FRAG_LEN, fld, sd, mft.data());
        // MinCollector::get_mean_frag_len / get_sd_frag_len with the counts left at zero (src/MinCollector.cpp:583-627)
        volatile double zero_mass = 0.0; volatile size_t zero_counts = 0;   // evaluated at run time, like the reference's
        fld_m[id] = {mft[KAMD_MAX_FRAG_LEN - 1], std::sqrt(zero_mass / (double)zero_counts)};
      } else {
        const std::vector<uint32_t>& fl = flds.size() == 1 ? flds[0] : flds[id];
        kamd_mean_frag_lens_trunc(fl.data(), mft.data());
        int tc = 0; double mass = 0.0;
        for (size_t i = 0; i < fl.size(); i++) { tc += (int)fl[i]; mass += (double)((uint64_t)fl[i] * i); }
        const double m = tc ? mass / (double)tc : std::numeric_limits<double>::max();
        size_t tc2 = 0; double mass2 = 0.0;
        const double mm = tc ? m : 0.0;
        for (size_t i = 0; i < fl.size(); i++) { tc2 += fl[i]; mass2 += fl[i] * ((double)i - mm) * ((double)i - mm); }
        fld_m[id] = {m, std::sqrt(mass2 / (double)tc2)};
      }
      kamd_eff_lens(v.target_lens, T, mft.data(), eff.data());
    } else std::fill(eff.begin(), eff.end(), 1.0);   // fl_means = target lengths: every effective length is 1 (src/main.cpp:3016-3018)
    int32_t rounds = 0;
    if (total > 0) {
      KX(kamd_ec_set_counts(ctx, counts.data()));
      KX(kamd_em_run(ctx, nullptr, nullptr, nullptr, nullptr, 0, eff.data(), T, 10000, 50, alpha.data(), abz.data(), &rounds));
    } else std::fill(alpha.begin(), alpha.end(), 0.0);   // nothing to distribute: the EM's result is all zeros
    kamd_counts_to_tpm(alpha.data(), eff.data(), T, tpm.data());
    if (gene_level) gene_sums(genes, alpha, tpm, &gc, &gc_tpm);
    if (is_matrix) {
      for (uint64_t t = 0; t < T; t++)
        if (alpha[t] > 0.0) { ab_m[id].push_back({(int32_t)t, alpha[t]}); tpm_m[id].push_back({(int32_t)t, tpm[t]}); if (calc_eff) eff_m[id].push_back({(int32_t)t, eff[t]}); }
      if (gene_level)
        for (size_t g = 0; g < gc.size(); g++) if (gc[g] > 0.0) { gab_m[id].push_back({(int32_t)g, gc[g]}); gtpm_m[id].push_back({(int32_t)g, gc_tpm[g]}); }
      if (matrix_to_files) {
        // abundance_N.tsv beside the matrices, or (--matrix-to-directories) abundance_N/abundance.tsv (src/main.cpp:3061-3087)
        std::string dir = output + "/", suffix = "_" + std::to_string(id + 1);
        if (matrix_to_dirs) {
          dir = output + "/abundance_" + std::to_string(id + 1);
          struct stat sd;
          if (stat(dir.c_str(), &sd) == 0) { if (!S_ISDIR(sd.st_mode)) { std::cerr << "Error: file " << dir << " exists and is not a directory" << std::endl; return 1; } }
          else if (mkdir(dir.c_str(), 0777) == -1) { std::cerr << "Error: could not create directory " << dir << std::endl; return 1; }
          dir += "/"; suffix = "";
        }
        write_abundance(dir + "abundance" + suffix + ".tsv", idx, vt, alpha, eff);
        if (gene_level) write_abundance_gene(dir + "abundance.gene" + suffix + ".tsv", genes, gc, gc_tpm);
        if (bootstrap > 0) {
          std::vector<uint64_t> seeds(bootstrap);
          kamd_bootstrap_seeds(seed, bootstrap, seeds.data());
          std::vector<double> res((size_t)bootstrap * T, 0.0), a(T), at(T), bgc, bgt;
          if (total > 0) KX(kamd_bootstrap_batch(ctx, seeds.data(), bootstrap, eff.data(), T, res.data(), nullptr));
          for (int b = 0; b < bootstrap; b++) {
            a.assign(res.begin() + (size_t)b * T, res.begin() + (size_t)(b + 1) * T);
            write_abundance(dir + "bs_abundance" + suffix + "_" + std::to_string(b) + ".tsv", idx, vt, a, eff);
            if (gene_level) {   // the replicate's gene-level sums (src/main.cpp:3143-3146)
              kamd_counts_to_tpm(a.data(), eff.data(), T, at.data());
              gene_sums(genes, a, at, &bgc, &bgt);
              write_abundance_gene(dir + "bs_abundance.gene" + suffix + "_" + std::to_string(b) + ".tsv", genes, bgc, bgt);
            }
          }
        }
      }
    } else {
      write_abundance(output + "/abundance.tsv", idx, vt, alpha, eff);
      if (gene_level) write_abundance_gene(output + "/abundance.gene.tsv", genes, gc, gc_tpm);
      if (bootstrap > 0) {
        std::vector<uint64_t> seeds(bootstrap);
        kamd_bootstrap_seeds(seed, bootstrap, seeds.data());
        std::vector<double> res((size_t)bootstrap * T, 0.0), a(T);
        if (total > 0) KX(kamd_bootstrap_batch(ctx, seeds.data(), bootstrap, eff.data(), T, res.data(), nullptr));
        for (int b = 0; b < bootstrap; b++) {
          a.assign(res.begin() + (size_t)b * T, res.begin() + (size_t)(b + 1) * T);
          write_abundance(output + "/bs_abundance_" + std::to_string(b) + ".tsv", idx, vt, a, eff);
        }
      }
    }
  }
  std::cerr << " done" << std::endl << std::endl;
  if (is_matrix) {
    write_sparse(output + "/matrix.abundance.mtx", ab_m, T);
    write_sparse(output + "/matrix.abundance.tpm.mtx", tpm_m, T);
    if (calc_eff) write_sparse(output + "/matrix.efflens.mtx", eff_m, T);
    if (gene_level) {
      write_sparse(output + "/matrix.abundance.gene.mtx", gab_m, genes.name.size());
      write_sparse(output + "/matrix.abundance.gene.tpm.mtx", gtpm_m, genes.name.size());
      write_gene_names(output + "/genes.txt", genes);
    }
  }
  if (calc_eff) {
    { std::ofstream of(output + "/matrix.fld.tsv"); for (size_t j = 0; j < fld_m.size(); j++) of << j << "\t" << fld_m[j].first << "\t" << fld_m[j].second << "\n"; }   // writeFLD
    std::ofstream of(output + "/transcript_lengths.txt");
    for (uint64_t t = 0; t < T; t++) of << kamd_index_target_name(idx, t) << " " << v.target_lens[t] << "\n";
  }
  kamd_ctx_destroy(ctx);
  kamd_index_free(idx);
  return 0;
}

}  // namespace kamd_fe
