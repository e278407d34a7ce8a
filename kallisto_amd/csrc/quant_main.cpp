// quant_main.cpp -- `kallisto_amd_quant quant` (+ dispatch of `bus` / `quant-tcc`, bus_main.cpp): the command-line surface of `kallisto quant` (src/main.cpp:211-392 option
// parsing, :1600-1805 checks, :2620-2798 driver) on top of the C ABI of include/kallisto_amd.h.  Same flags, same
// index file, same abundance.tsv / run_info.json (PlaintextWriter.cpp:29-65,140-197), bs_abundance_N.tsv bootstraps.
// Host code only: FASTQ reading (FastqSequenceReader::fetchSequences, src/ProcessReads.cpp:3128-3267), batching,
// writers.  Everything that computes runs on the GPU through libkallisto_amd.so.
#include "kamd_frontend.h"

namespace {
using namespace kamd_fe;

struct Options {
  std::string index, output;
  std::vector<std::string> files;
  bool single = false, single_overhang = false, plaintext = false, verbose = false, no_jump = false, do_union = false, share_device = false;
  int gpus = 1;
  int table_layout = -1;   // --kmer-table: KAMD_TABLE_WIDE / _COMPACT / _AUTO; -1 = the library's choice (environment, default wide)
  int strand = 0, bootstrap = 0, threads = 1;
  double fld = 0.0, sd = 0.0;
  uint64_t seed = 42;
  uint64_t batch = 4u << 20;  // reads (or pairs) per device batch
};

bool parse_table_layout(const std::string& v, int* out) {
  if (v == "wide") *out = KAMD_TABLE_WIDE; else if (v == "compact") *out = KAMD_TABLE_COMPACT; else if (v == "auto") *out = KAMD_TABLE_AUTO; else return false;
  return true;
}

void usage() {
  std::cout << "kallisto_amd " << KALLISTO_COMPAT_VERSION << "-compatible (MI355X)\n"
            << "Computes equivalence classes for reads and quantifies abundances\n\n"
            << "Usage: kallisto_amd quant [arguments] FASTQ-files\n\n"
            << "Required arguments:\n"
            << "-i, --index=STRING            Filename for the kallisto index to be used for\n"
            << "                              quantification\n"
            << "-o, --output-dir=STRING       Directory to write output to\n\n"
            << "Optional arguments:\n"
            << "-b, --bootstrap-samples=INT   Number of bootstrap samples (default: 0)\n"
            << "    --seed=INT                Seed for the bootstrap sampling (default: 42)\n"
            << "    --plaintext               Output plaintext only (abundance.h5 is written otherwise, when libhdf5 can be loaded)\n"
            << "    --single                  Quantify single-end reads\n"
            << "    --single-overhang         Include reads where unobserved rest of fragment is\n"
            << "                              predicted to lie outside a transcript\n"
            << "    --fr-stranded             Strand specific reads, first read forward\n"
            << "    --rf-stranded             Strand specific reads, first read reverse\n"
            << "    --no-jump                 Look up every k-mer of a read (no jumping); not with a strand option\n"
            << "-l, --fragment-length=DOUBLE  Estimated average fragment length\n"
            << "-s, --sd=DOUBLE               Estimated standard deviation of fragment length\n"
            << "-t, --threads=INT             Host threads: the readers of the input files (FASTQ text into pinned memory, gzip / BGZF\n"
            << "                              inflate) and the index loader (default: 1, as the reference; give it the CPUs there are)\n"
            << "    --verbose                 Print out progress information\n"
            << "    --gpus=INT                GPUs of this node to use (default: 1): batches of reads go round the GPUs, the EC counts\n"
            << "                              are merged with one RCCL all-reduce + all-gathers, the EM runs partitioned over them\n"
            << "    --kmer-table=wide|compact|auto  layout of the k-mer table in HBM: three 20-byte slots per 64-byte line (wide, default) or\n"
            << "                              four exact 16-byte slots (compact: 27 instead of 43 bytes per k-mer; auto = compact when it fits)\n"
            << "    --share-device            with --gpus N: all N ranks on device 0, collectives staged through the host (runs the\n"
            << "                              several-GPU code path on a single-GPU box; for testing)\n";
}


// ---- abundance.h5 (H5Writer.cpp:4-69, h5utils.h:42-92): one chunk per dataset, deflate level 6, strings as fixed-size
// NUL-terminated C strings of the longest entry + 1.  libhdf5 is loaded at run time (dlopen), so the front-end neither
// needs it to build nor drags its dependencies into a process that already holds the HIP runtime.  Written without
// `--plaintext`, like a reference build with USE_HDF5; without the library the front-end behaves like `--plaintext`.
class H5Out {
 public:
  typedef int64_t hid_t; typedef unsigned long long hsize_t; typedef int herr_t;
  bool load() {
    const char* cands[] = {getenv("KAMD_HDF5_LIB"), "libhdf5.so", "libhdf5.so.103", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so"};
    for (const char* c : cands) { if (!c) continue; lib_ = dlopen(c, RTLD_NOW | RTLD_LOCAL); if (lib_) break; }
    if (!lib_) return false;
#define KAMD_H5SYM(name) do { *(void**)(&name) = dlsym(lib_, #name); if (!name) return false; } while (0)
    KAMD_H5SYM(H5open); KAMD_H5SYM(H5Fcreate); KAMD_H5SYM(H5Fclose); KAMD_H5SYM(H5Gcreate2); KAMD_H5SYM(H5Gopen2); KAMD_H5SYM(H5Gclose);
    KAMD_H5SYM(H5Pcreate); KAMD_H5SYM(H5Pset_chunk); KAMD_H5SYM(H5Pset_deflate); KAMD_H5SYM(H5Pclose);
    KAMD_H5SYM(H5Screate_simple); KAMD_H5SYM(H5Sclose); KAMD_H5SYM(H5Dcreate2); KAMD_H5SYM(H5Dwrite); KAMD_H5SYM(H5Dclose);
    KAMD_H5SYM(H5Tcopy); KAMD_H5SYM(H5Tset_size); KAMD_H5SYM(H5Tclose);
#undef KAMD_H5SYM
    if (H5open() < 0) return false;
    hid_t* g = nullptr;
    if (!(g = (hid_t*)dlsym(lib_, "H5T_NATIVE_INT_g"))) return false;
    t_int_ = *g;
    if (!(g = (hid_t*)dlsym(lib_, "H5T_NATIVE_DOUBLE_g"))) return false;
    t_double_ = *g;
    if (!(g = (hid_t*)dlsym(lib_, "H5T_C_S1_g"))) return false;
    t_c_s1_ = *g;
    if (!(g = (hid_t*)dlsym(lib_, "H5P_CLS_DATASET_CREATE_ID_g"))) return false;
    p_dcreate_ = *g;
    return true;
  }
  bool open(const std::string& path, bool with_bootstrap) {
    file_ = H5Fcreate(path.c_str(), 0x0002u /* H5F_ACC_TRUNC */, 0, 0);
    if (file_ < 0) return false;
    root_ = H5Gopen2(file_, "/", 0);
    aux_ = H5Gcreate2(file_, "/aux", 0, 0, 0);
    if (with_bootstrap) bs_ = H5Gcreate2(file_, "/bootstrap", 0, 0, 0);
    return root_ >= 0 && aux_ >= 0;
  }
  void close() {
    if (file_ < 0) return;
    if (bs_ >= 0) H5Gclose(bs_);
    H5Gclose(aux_); H5Gclose(root_); H5Fclose(file_);
    file_ = -1;
  }
  hid_t root() const { return root_; }
  hid_t aux() const { return aux_; }
  hid_t bs() const { return bs_; }
  void ints(hid_t g, const char* name, const std::vector<int32_t>& v) { write(g, name, t_int_, v.size(), v.data()); }
  void doubles(hid_t g, const char* name, const std::vector<double>& v) { write(g, name, t_double_, v.size(), v.data()); }
  void strings(hid_t g, const char* name, const std::vector<std::string>& v) {   // vec_to_ptr / get_datatype_id, h5utils.cpp:4-52
    size_t w = 0;
    for (auto& x : v) w = std::max(w, x.size());
    w += 1;
    std::vector<char> pool(w * v.size(), 0);
    for (size_t i = 0; i < v.size(); i++) memcpy(pool.data() + i * w, v[i].data(), v[i].size());
    const hid_t t = H5Tcopy(t_c_s1_);
    H5Tset_size(t, w);
    write(g, name, t, v.size(), pool.data());
    H5Tclose(t);
  }
 private:
  void write(hid_t g, const char* name, hid_t type, size_t n, const void* data) {   // vector_to_h5, h5utils.h:42-92
    hsize_t dims[1] = {(hsize_t)n};
    const hid_t prop = H5Pcreate(p_dcreate_);
    H5Pset_chunk(prop, 1, dims);          // chunk = the whole vector
    H5Pset_deflate(prop, 6);
    const hid_t space = H5Screate_simple(1, dims, nullptr);
    const hid_t ds = H5Dcreate2(g, name, type, space, 0, prop, 0);
    if (ds < 0 || H5Dwrite(ds, type, 0, 0, 0, data) < 0) { std::cerr << "Error: could not write dataset " << name << " of abundance.h5" << std::endl; exit(1); }
    H5Pclose(prop); H5Dclose(ds); H5Sclose(space);
  }
  void* lib_ = nullptr;
  hid_t file_ = -1, root_ = -1, aux_ = -1, bs_ = -1, t_int_ = -1, t_double_ = -1, t_c_s1_ = -1, p_dcreate_ = -1;
  herr_t (*H5open)() = nullptr;
  hid_t (*H5Fcreate)(const char*, unsigned, hid_t, hid_t) = nullptr;
  herr_t (*H5Fclose)(hid_t) = nullptr;
  hid_t (*H5Gcreate2)(hid_t, const char*, hid_t, hid_t, hid_t) = nullptr;
  hid_t (*H5Gopen2)(hid_t, const char*, hid_t) = nullptr;
  herr_t (*H5Gclose)(hid_t) = nullptr;
  hid_t (*H5Pcreate)(hid_t) = nullptr;
  herr_t (*H5Pset_chunk)(hid_t, int, const hsize_t*) = nullptr;
  herr_t (*H5Pset_deflate)(hid_t, unsigned) = nullptr;
  herr_t (*H5Pclose)(hid_t) = nullptr;
  hid_t (*H5Screate_simple)(int, const hsize_t*, const hsize_t*) = nullptr;
  herr_t (*H5Sclose)(hid_t) = nullptr;
  hid_t (*H5Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
  herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*) = nullptr;
  herr_t (*H5Dclose)(hid_t) = nullptr;
  hid_t (*H5Tcopy)(hid_t) = nullptr;
  herr_t (*H5Tset_size)(hid_t, size_t) = nullptr;
  herr_t (*H5Tclose)(hid_t) = nullptr;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 2 && std::string(argv[1]) == "version") { std::cout << "kallisto_amd, compatible with kallisto " << KALLISTO_COMPAT_VERSION << std::endl; return 0; }
  if (argc >= 2 && std::string(argv[1]) == "flatten") {   // kallisto_amd_quant flatten -i index.idx -o index.kamd [-t N]
    // writes the device tables of a kallisto index as a file (kamd_index_save); `-i index.kamd` then loads in a fraction of the time
    std::string in, out, val; int threads = 1, layout = -1;
    const char* use = "Usage: kallisto_amd_quant flatten -i index.idx -o index.kamd [-t threads] [--kmer-table wide|compact|auto]";
    for (int i = 2; i < argc; i++) {
      std::string a = argv[i];
      if (take(a, "-i", "--index", i, argc, argv, val)) in = val;
      else if (take(a, "-o", "--output", i, argc, argv, val)) out = val;
      else if (take(a, "-t", "--threads", i, argc, argv, val)) threads = atoi(val.c_str());
      else if (take(a, nullptr, "--kmer-table", i, argc, argv, val)) { if (!parse_table_layout(val, &layout)) { std::cerr << "Error: --kmer-table expects wide, compact or auto\n" << use << std::endl; return 1; } }
      else { std::cerr << "Error: unknown argument " << a << "\n" << use << std::endl; return 1; }
    }
    if (in.empty() || out.empty()) { std::cerr << use << std::endl; return 1; }
    kamd_index* idx = nullptr;
    const int lrc = layout < 0 ? kamd_index_load(in.c_str(), threads, &idx) : kamd_index_load_layout(in.c_str(), threads, layout, 0.0, &idx);
    if (lrc != 0 || kamd_index_save(idx, out.c_str()) != 0) { std::cerr << "Error: " << kamd_last_error() << std::endl; return 1; }
    kamd_index_free(idx);
    return 0;
  }
  if (argc >= 2 && std::string(argv[1]) == "bus") return bus_main(argc, argv);
  if (argc >= 2 && std::string(argv[1]) == "quant-tcc") return tcc_main(argc, argv);
  if (argc < 2 || std::string(argv[1]) != "quant") { usage(); return 1; }
  Options opt;
  std::string val;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    if (take(a, "-i", "--index", i, argc, argv, val)) opt.index = val;
    else if (take(a, "-o", "--output-dir", i, argc, argv, val)) opt.output = val;
    else if (take(a, "-b", "--bootstrap-samples", i, argc, argv, val)) opt.bootstrap = atoi(val.c_str());
    else if (take(a, nullptr, "--seed", i, argc, argv, val)) opt.seed = strtoull(val.c_str(), nullptr, 10);
    else if (take(a, "-l", "--fragment-length", i, argc, argv, val)) opt.fld = atof(val.c_str());
    else if (take(a, "-s", "--sd", i, argc, argv, val)) opt.sd = atof(val.c_str());
    else if (take(a, "-t", "--threads", i, argc, argv, val)) opt.threads = atoi(val.c_str());
    else if (take(a, nullptr, "--batch", i, argc, argv, val)) opt.batch = strtoull(val.c_str(), nullptr, 10);
    else if (take(a, nullptr, "--gpus", i, argc, argv, val)) opt.gpus = atoi(val.c_str());
    else if (a == "--single") opt.single = true;
    else if (a == "--single-overhang") opt.single_overhang = true;
    else if (a == "--fr-stranded") opt.strand = 1;
    else if (a == "--rf-stranded") opt.strand = 2;
    else if (a == "--no-jump") opt.no_jump = true;
    else if (a == "--union") opt.do_union = true;
    else if (a == "--plaintext") opt.plaintext = true;
    else if (a == "--verbose") opt.verbose = true;
    else if (a == "--share-device") opt.share_device = true;
    else if (take(a, nullptr, "--kmer-table", i, argc, argv, val)) {
      if (!parse_table_layout(val, &opt.table_layout)) { std::cerr << "Error: --kmer-table expects wide, compact or auto" << std::endl; return 1; }
    }
    else if (a == "--bias" || a == "--fusion" || a == "--pseudobam" || a == "--genomebam" || a == "--long" || a == "-p" || a == "--priors" ||
             a == "-g" || a == "--gtf" || a == "-c" || a == "--chromosomes" || a == "--dfk-onlist" ||
             a == "-P" || a == "--platform" || a == "-N" || a == "--numReads") {
      std::cerr << "Error: option " << a << " is outside the GPU quant path; use the reference kallisto for it" << std::endl; return 1;
    } else if (!a.empty() && a[0] == '-') { std::cerr << "Error: unknown option " << a << std::endl; usage(); return 1; }
    else opt.files.push_back(a);
  }
  // CheckOptionsEM (src/main.cpp:1600-1805): same tests, same order, same messages
  bool ok = true;
  struct stat st;
  std::cerr << std::endl;
  if (opt.index.empty()) { std::cerr << "Error: kallisto index file missing" << std::endl; ok = false; }
  else if (stat(opt.index.c_str(), &st) != 0) { std::cerr << "Error: kallisto index file not found " << opt.index << std::endl; ok = false; }
  if (opt.files.empty()) { std::cerr << "Error: Missing read files" << std::endl; ok = false; }
  else for (const auto& fn : opt.files) if (stat(fn.c_str(), &st) != 0) { std::cerr << "Error: file not found " << fn << std::endl; ok = false; }
  if (!opt.single && opt.files.size() % 2 != 0) { std::cerr << "Error: paired-end mode requires an even number of input files\n       (use --single for processing single-end reads)" << std::endl; ok = false; }
  if ((opt.fld != 0.0 && opt.sd == 0.0) || (opt.sd != 0.0 && opt.fld == 0.0)) { std::cerr << "Error: cannot supply mean/sd without supplying both -l and -s" << std::endl; ok = false; }
  if (opt.single && (opt.fld == 0.0 || opt.sd == 0.0)) { std::cerr << "Error: fragment length mean and sd must be supplied for single-end reads using -l and -s" << std::endl; ok = false; }
  else if (opt.fld == 0.0 && ok) std::cerr << "[quant] fragment length distribution will be estimated from the data" << std::endl;
  else if (ok && opt.fld > 0.0 && opt.sd > 0.0) std::cerr << "[quant] fragment length distribution is truncated gaussian with mean = " << opt.fld << ", sd = " << opt.sd << std::endl;
  if (!opt.single && opt.fld > 0.0 && opt.sd > 0.0) {
    std::cerr << "[~warn] you specified using a gaussian but have paired end data" << std::endl;
    std::cerr << "[~warn] we suggest omitting these parameters and let us estimate the distribution from data" << std::endl;
  }
  if (opt.fld < 0.0) { std::cerr << "Error: invalid value for mean fragment length " << opt.fld << std::endl; ok = false; }
  if (opt.sd < 0.0) { std::cerr << "Error: invalid value for fragment length standard deviation " << opt.sd << std::endl; ok = false; }
  if (opt.output.empty()) { std::cerr << "Error: need to specify output directory " << opt.output << std::endl; ok = false; }
  else if (stat(opt.output.c_str(), &st) == 0) {
    if (!S_ISDIR(st.st_mode)) { std::cerr << "Error: file " << opt.output << " exists and is not a directory" << std::endl; ok = false; }
  } else if (mkdir(opt.output.c_str(), 0777) == -1) { std::cerr << "Error: could not create directory " << opt.output << std::endl; ok = false; }
  if (opt.threads <= 0) { std::cerr << "Error: invalid number of threads " << opt.threads << std::endl; ok = false; }
  if (opt.bootstrap < 0) { std::cerr << "Error: number of bootstrap samples must be a non-negative integer." << std::endl; ok = false; }
  if (!ok) { std::cerr << std::endl; usage(); return 1; }
  const std::string start_time = now_string(), call = call_string(argc, argv);

  // device contexts first (they do not need the index): the pinned text rings of the device-parse pipeline are allocated by a
  // second thread while the index is read and flattened (KmerIndex::load)
  const auto t_start = std::chrono::steady_clock::now();
  // The index is read and flattened (pure host work, seconds) from the first moment on -- under the start of the HIP runtime, the
  // creation of the contexts and the pinning of the text rings.  The flattened tables written by `kallisto_amd_quant flatten` are used
  // when they lie beside the index (<index>.kamd) and say they were written from it; the kallisto index stays the source of truth.
  std::string index_path = opt.index;
  {
    struct stat si, sf;
    const std::string flat = opt.index + ".kamd";
    // ... and only when the file says it was written from exactly this index (size + hash of its head and tail: a replaced index keeps
    // neither, whatever its mtime); a stale or foreign .kamd is ignored and the index itself is flattened
    if (!getenv("KAMD_NO_FLAT_INDEX") && stat(opt.index.c_str(), &si) == 0 && stat(flat.c_str(), &sf) == 0) {
      if (kamd_flat_index_matches(flat.c_str(), opt.index.c_str()) == 1) index_path = flat;
      else std::cerr << "[index] " << flat << " was not written from " << opt.index << " (or by another version): ignored" << std::endl;
    }
  }
  kamd_index* idx = nullptr;
  kamd_index_view v{};
  double index_load_s = 0.0, index_ready_s = 0.0;
  int load_rc = 0; std::string load_err;
  const int load_threads = std::min(opt.threads, effective_cpus());
  std::thread load_early([&] {
    auto load = [&](const std::string& p) { return opt.table_layout < 0 ? kamd_index_load(p.c_str(), load_threads, &idx) : kamd_index_load_layout(p.c_str(), load_threads, opt.table_layout, 0.0, &idx); };
    load_rc = load(index_path);
    if (load_rc && index_path != opt.index) {   // a flattened file picked up beside the index that does not load (another format version, damaged): the index itself
      std::cerr << "[index] " << index_path << " ignored: " << kamd_last_error() << std::endl;
      index_path = opt.index;
      load_rc = load(index_path);
    }
    if (!load_rc) load_rc = kamd_index_get_view(idx, &v);
    if (load_rc) load_err = kamd_last_error();
    index_load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  });
  struct EarlyJoiner { std::thread& t; ~EarlyJoiner() { if (t.joinable()) t.join(); } } load_join{load_early};
  // threads that wait for the GPU sleep instead of spinning: the host's CPUs belong to the readers (the runtime's default burns one
  // CPU per waiting thread, and a container's CPU quota is easily exceeded -- see effective_cpus())
  (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  const int host_cpus = effective_cpus();
  if (opt.threads > host_cpus) { if (opt.verbose) std::cerr << "[quant] " << opt.threads << " threads asked for, " << host_cpus << " CPUs available to this process: using " << host_cpus << std::endl; opt.threads = host_cpus; }
  int n_dev = 0;
  HIPX(hipGetDeviceCount(&n_dev));
  const int n_gpus = std::max(1, opt.gpus);
  if (!opt.share_device && n_gpus > n_dev) { std::cerr << "Error: --gpus " << n_gpus << " but only " << n_dev << " HIP device(s) are visible" << std::endl; return 1; }
  std::vector<int> devices((size_t)n_gpus);
  for (int g = 0; g < n_gpus; g++) devices[g] = opt.share_device ? 0 : g;
  std::vector<kamd_ctx*> ctxs((size_t)n_gpus, nullptr);
  for (int g = 0; g < n_gpus; g++) KX(kamd_ctx_create(devices[g], nullptr, &ctxs[g]));
  kamd_ctx* ctx = ctxs[0];
  const bool paired = !opt.single;
  kamd_quant_opts qo{paired ? 1 : 0, opt.fld, opt.sd, opt.single_overhang ? 1 : 0, opt.strand, opt.no_jump ? 1 : 0, opt.do_union ? 1 : 0};
  uint32_t flens[KAMD_MAX_FRAG_LEN] = {0};
  uint64_t fld_used = 0, n_processed = 0;
  std::atomic<bool> fld_open{paired && opt.fld == 0.0};   // the fragment-length sample is still being collected (on GPU 0)
  // The index is loaded and uploaded on a thread of its own while the input is already read, copied to the device and parsed (none of
  // that needs the index): only the pseudoalignment of the first batch waits for it.  0 = loading, 1 = on every device, -1 = failed.
  std::mutex ix_m; std::condition_variable ix_cv; int ix_state = 0; std::string ix_err;
  auto index_ready = [&](std::string* err) -> bool {
    std::unique_lock<std::mutex> lk(ix_m);
    ix_cv.wait(lk, [&] { return ix_state != 0; });
    if (ix_state < 0 && err) *err = ix_err;
    return ix_state > 0;
  };
  auto run_batch = [&](int g, PackedBatch& b, std::string& err) -> int {
    if (!index_ready(&err)) return -1;
    const bool want_fld = g == 0 && paired && opt.fld == 0.0 && fld_used < 10000;
    int rc = 0;
    if (want_fld) rc = kamd_fld_prefetch(ctxs[g], &qo, b.d_words, b.d_len, b.n_items, b.max_len);   // runs underneath kernel A
    if (!rc) rc = kamd_pseudoalign(ctxs[g], &qo, b.d_words, b.d_len, b.n_items, b.max_len);
    if (!rc && want_fld) rc = kamd_fld_from_batch(ctxs[g], &qo, b.d_words, b.d_len, b.n_items, b.max_len, flens, &fld_used);
    if (g == 0 && fld_used >= 10000) fld_open = false;
    if (rc) err = kamd_last_error();
    return rc;
  };
  auto pin_first = [&] { return n_gpus > 1 && fld_open.load(); };
  UnitFeeder feeder(ctxs, devices, run_batch, pin_first);
  std::thread prep([&] { (void)feeder.prepare(paired ? 2 : 1); });
  std::thread ix_thread([&] {
    auto done = [&](int st, const std::string& e) { { std::lock_guard<std::mutex> lk(ix_m); ix_state = st; ix_err = e; } ix_cv.notify_all(); };
    load_early.join();
    if (load_rc) { done(-1, load_err); return; }
    std::cerr << "\n[index] k-mer length: " << v.k << "\n[index] number of targets: " << v.n_targets << "\n[index] number of k-mers: " << v.n_kmers << std::endl;
    std::vector<std::thread> th; std::vector<int> rcs((size_t)n_gpus, 0); std::vector<std::string> errs((size_t)n_gpus);
    for (int g = 0; g < n_gpus; g++) th.emplace_back([&, g] {   // the index is replicated in every GPU's HBM
      rcs[g] = kamd_index_upload(ctxs[g], idx);
      if (rcs[g]) errs[g] = kamd_last_error();
    });
    for (auto& x : th) x.join();
    for (int g = 0; g < n_gpus; g++) if (rcs[g]) { done(-1, errs[g]); return; }
    // bootstrap replicates are multinomials over the count vector in EC-id order: ask for the reference's (-t 1) ids
    // (one GPU only: records merged from several GPUs have no input order -- like the reference at -t > 1)
    if (opt.bootstrap > 0 && n_gpus == 1 && kamd_ec_track_order(ctx, 1) != 0) { done(-1, kamd_last_error()); return; }
    index_ready_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (index_path != opt.index) std::cerr << "[index] using the flattened tables of " << index_path << std::endl;
    if (opt.verbose) std::cerr << "[index] k-mer table: " << (v.table_layout ? "compact" : "wide") << " layout, " << v.slots_per_bucket << " slots per 64-byte line, "
                               << (v.n_buckets + v.pad_buckets) * 64 / 1000000 << " MB, load " << (double)v.n_kmers / (double)(v.n_buckets * v.slots_per_bucket) << std::endl;
    if (opt.verbose) std::cerr << "[timing] index file read + flattened in " << index_load_s << " s, on the device after " << index_ready_s << " s" << std::endl;
    done(1, "");
  });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } ix_join{ix_thread}, prep_join{prep};
  prep.join();
  // (measurements of the input path alone: KAMD_FQ_NO_OVERLAP makes the input wait for the index, as a very large input effectively does)
  if (getenv("KAMD_FQ_NO_OVERLAP")) (void)index_ready(nullptr);
  std::cerr << "[quant] running in " << (paired ? "paired-end" : "single-end") << " mode" << std::endl;
  double pack_s = 0.0;
  MultiPipe pipe(devices, run_batch, pin_first);
  // the device parser declined the input: the run starts over with the general reader
  auto reset_run = [&]() -> int {
    if (!index_ready(nullptr)) return -1;
    for (kamd_ctx* x : ctxs) if (int rc = kamd_ec_reset(x)) return rc;
    memset(flens, 0, sizeof flens); fld_used = 0; fld_open = paired && opt.fld == 0.0;
    return 0;
  };
  for (size_t fi = 0; fi < opt.files.size(); fi += paired ? 2 : 1) {
    std::cerr << "[quant] will process " << (paired ? "pair " : "file ") << (fi / (paired ? 2 : 1) + 1) << ": " << opt.files[fi] << std::endl;
    if (paired) std::cerr << "                             " << opt.files[fi + 1] << std::endl;
  }
  const int feed_rc = feed_files(opt.files, paired, opt.batch, std::max(1, opt.threads), opt.threads, opt.verbose, pipe, n_processed, pack_s, &feeder, reset_run);
  { std::string e; if (!index_ready(&e)) { std::cerr << "Error: " << e << std::endl; return 1; } }   // (also when there was no read to wait for it)
  if (feed_rc) return 1;
  pipe.finish();
  if (pipe.failed()) { std::cerr << "Error: " << pipe.error() << std::endl; return 1; }
  if (opt.verbose) std::cerr << "[timing] reads parsed, packed and pseudoaligned after " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << " s" << std::endl;
  if (opt.verbose)
    std::cerr << "[quant] device parser: " << feeder.units << " units, " << feeder.bytes << " bytes of text; consumers: waiting for copies " << feeder.copy_wait_s
              << " s, parse + pack " << feeder.parse_s << " s, pseudoalignment " << feeder.run_s << " s; dispatcher: waiting for the cutter " << feeder.cut_s
              << " s, for a text buffer " << feeder.wait_s << " s; general reader: host packing " << pack_s << " s, device "
              << pipe.device_s() << " s, host waited " << pipe.wait_s() << " s" << std::endl;
  std::cerr << "[quant] finding pseudoalignments for the reads ... done" << std::endl;

  // several GPUs: every GPU's EC state becomes the state of the whole input (kamd_ec_allreduce over RCCL), then each finalizes
  std::vector<kamd_comm*> comms((size_t)n_gpus, nullptr);
  auto on_all_gpus = [&](std::function<int(int)> f) -> bool {   // f(g) on one host thread per GPU (collectives need all of them)
    std::vector<std::thread> th; std::vector<int> rcs((size_t)n_gpus, 0); std::vector<std::string> errs((size_t)n_gpus);
    for (int g = 0; g < n_gpus; g++) th.emplace_back([&, g] { rcs[g] = f(g); if (rcs[g]) errs[g] = kamd_last_error(); });
    for (auto& x : th) x.join();
    for (int g = 0; g < n_gpus; g++) if (rcs[g]) { std::cerr << "Error: " << errs[g] << std::endl; return false; }
    return true;
  };
  SharedDeviceComm shared(n_gpus);
  const kamd_comm_callbacks shared_cb = SharedDeviceComm::callbacks();
  if (n_gpus > 1) {
    unsigned char uid[KAMD_COMM_ID_BYTES];
    if (!opt.share_device) KX(kamd_comm_unique_id(uid));
    if (opt.verbose) std::cerr << "[quant] merging the equivalence classes of " << n_gpus << " ranks: "
                               << (opt.share_device ? "host-staged callbacks, all ranks on device 0 (--share-device)" : "RCCL (kamd_comm, inside libkallisto_amd.so)") << std::endl;
    if (!on_all_gpus([&](int g) {
          int rc = opt.share_device ? kamd_comm_create_callbacks(ctxs[g], g, n_gpus, &shared_cb, shared.user(g), &comms[g])
                                    : kamd_comm_create_rccl(ctxs[g], g, n_gpus, uid, &comms[g]);
          if (!rc) rc = kamd_ec_allreduce(ctxs[g], comms[g]);
          if (!rc && g != 0) rc = kamd_ec_finalize(ctxs[g], nullptr);
          return rc;
        })) return 1;
  }
  kamd_ec_result ec;
  KX(kamd_ec_finalize(ctx, &ec));
  std::vector<uint64_t> ec_off(ec.n_ecs + 1); std::vector<uint32_t> ec_ids(std::max<uint64_t>(ec.nnz, 1)), counts(std::max<uint64_t>(ec.n_ecs, 1));
  KX(kamd_ec_download(ctx, ec_off.data(), ec_ids.data(), counts.data()));
  uint64_t num_pseudoaligned = 0, num_unique = 0;  // src/main.cpp:2704-2709
  for (uint64_t e = 0; e < ec.n_ecs; e++) { num_pseudoaligned += counts[e]; if (ec_off[e + 1] - ec_off[e] == 1) num_unique += counts[e]; }
  std::cerr << "[quant] processed " << n_processed << " reads, " << num_pseudoaligned << " reads pseudoaligned" << std::endl;
  if (num_pseudoaligned == 0) std::cerr << "[~warn] Warning, zero reads pseudoaligned check your input files and index" << std::endl;

  std::vector<double> mft(KAMD_MAX_FRAG_LEN);
  if (opt.fld == 0.0) {
    // no pair gave a fragment length: like the reference (compute_mean_frag_lens_trunc on all-zero counts, src/main.cpp:2665-2667)
    // the means stay 0 and eff_length = length + 1; MinCollector::get_mean_frag_len's error exit is not on this path
    kamd_mean_frag_lens_trunc(flens, mft.data());
    std::cerr << "[quant] estimated average fragment length: " << mft[KAMD_MAX_FRAG_LEN - 1] << std::endl;
  } else kamd_trunc_gaussian_fld(0, KAMD_MAX_FRAG_LEN, opt.fld, opt.sd, mft.data());
  std::vector<double> eff(v.n_targets), alpha(v.n_targets), abz(v.n_targets);
  kamd_eff_lens(v.target_lens, v.n_targets, mft.data(), eff.data());
  int32_t rounds = 0;
  if (num_pseudoaligned > 0) {
    if (n_gpus == 1) KX(kamd_em_run(ctx, nullptr, nullptr, nullptr, nullptr, 0, eff.data(), v.n_targets, 10000, 50, alpha.data(), abz.data(), &rounds));
    else {   // partitioned over the GPUs by connected component; every GPU returns the same vectors
      std::vector<std::vector<double>> al((size_t)n_gpus, std::vector<double>(v.n_targets)), az((size_t)n_gpus, std::vector<double>(v.n_targets));
      std::vector<int32_t> rr((size_t)n_gpus, 0);
      if (!on_all_gpus([&](int g) { return kamd_em_run_comm(ctxs[g], comms[g], eff.data(), v.n_targets, 10000, 50, al[g].data(), az[g].data(), &rr[g]); })) return 1;
      alpha = al[0]; abz = az[0]; rounds = rr[0];
    }
    std::cerr << "[   em] quantifying the abundances ... done\n[   em] the Expectation-Maximization algorithm ran for " << rounds << " rounds" << std::endl;
  }
  // run_info.json (plaintext_aux, PlaintextWriter.cpp:140-197; src/main.cpp:2715-2727)
  write_run_info(opt.output + "/run_info.json", onlist_targets(v), opt.bootstrap, n_processed, num_pseudoaligned, num_unique, v.k, start_time, call);
  // abundance.h5 (src/main.cpp:2693-2702): written unless --plaintext, when libhdf5 is there
  H5Out h5;
  bool use_h5 = false;
  if (!opt.plaintext) {
    use_h5 = h5.load() && h5.open(opt.output + "/abundance.h5", opt.bootstrap > 0);
    if (!use_h5) std::cerr << "Warning: libhdf5 could not be loaded (set KAMD_HDF5_LIB), abundance.h5 is not written; plaintext output only" << std::endl;
  }
  if (use_h5) {   // H5Writer::init + write_main
    std::vector<int32_t> fld_v(KAMD_MAX_FRAG_LEN, 0);
    if (opt.fld == 0.0) for (int i = 0; i < KAMD_MAX_FRAG_LEN; i++) fld_v[i] = (int32_t)flens[i];
    else {   // trunc_gaussian_counts(0, MAX_FRAG_LEN, mean, sd, 10000), src/weights.cpp:273-296
      double total_mass = 0.0;
      for (int i = 0; i < KAMD_MAX_FRAG_LEN; i++) { const double x = ((double)i - opt.fld) / opt.sd; total_mass += std::exp(-0.5 * x * x) / opt.sd; }
      for (int i = 0; i < KAMD_MAX_FRAG_LEN; i++) { const double x = ((double)i - opt.fld) / opt.sd; fld_v[i] = (int)std::round(std::exp(-0.5 * x * x) / opt.sd * 10000 / total_mass); }
    }
    h5.ints(h5.aux(), "num_bootstrap", {opt.bootstrap});
    h5.ints(h5.aux(), "num_processed", {(int32_t)n_processed});
    h5.ints(h5.aux(), "fld", fld_v);
    h5.ints(h5.aux(), "bias_observed", std::vector<int32_t>(4096, 1));      // preBias without --bias (src/main.cpp:2676)
    h5.doubles(h5.aux(), "bias_normalized", std::vector<double>(4096, 1.0));  // EMAlgorithm::post_bias_ (EMAlgorithm.h:37)
    h5.strings(h5.aux(), "kallisto_version", {KALLISTO_COMPAT_VERSION});
    h5.ints(h5.aux(), "index_version", {13});
    h5.strings(h5.aux(), "call", {call});
    h5.strings(h5.aux(), "start_time", {start_time});
    h5.doubles(h5.root(), "est_counts", alpha);
    std::vector<std::string> ids(v.n_targets);
    for (uint64_t t = 0; t < v.n_targets; t++) ids[t] = kamd_index_target_name(idx, t);
    h5.strings(h5.aux(), "ids", ids);
    h5.doubles(h5.aux(), "eff_lengths", eff);
    h5.ints(h5.aux(), "lengths", std::vector<int32_t>(v.target_lens, v.target_lens + v.n_targets));
  }
  write_abundance(opt.output + "/abundance.tsv", idx, v, alpha, eff);
  if (opt.bootstrap > 0 && num_pseudoaligned == 0 && use_h5) {   // nothing aligned: empty replicates (src/main.cpp:2733-2743)
    for (int b = 0; b < opt.bootstrap; b++) h5.doubles(h5.bs(), ("bs" + std::to_string(b)).c_str(), alpha);
  }
  if (opt.bootstrap > 0 && num_pseudoaligned > 0) {  // src/main.cpp:2744-2782
    std::vector<uint64_t> seeds(opt.bootstrap);
    kamd_bootstrap_seeds(opt.seed, opt.bootstrap, seeds.data());
    // replicates in batches: one launch draws the batch's multinomial samples, the EMs reuse the plan of the EC matrix
    // (several GPUs: every GPU holds the merged ECs; the replicates of a batch are dealt round the GPUs)
    const int batch = 32 * n_gpus;
    std::vector<double> ab((size_t)batch * v.n_targets), a(v.n_targets);
    for (int b0 = 0; b0 < opt.bootstrap; b0 += batch) {
      const int nb = std::min(batch, opt.bootstrap - b0);
      std::cerr << "[bstrp] running EM for the bootstrap: " << b0 + nb << "\r";
      if (n_gpus == 1) KX(kamd_bootstrap_batch(ctx, seeds.data() + b0, nb, eff.data(), v.n_targets, ab.data(), nullptr));
      else if (!on_all_gpus([&](int g) {   // GPU g: replicates [lo, hi) of this batch
             const int lo = (int)((int64_t)nb * g / n_gpus), hi = (int)((int64_t)nb * (g + 1) / n_gpus);
             return hi > lo ? kamd_bootstrap_batch(ctxs[g], seeds.data() + b0 + lo, hi - lo, eff.data(), v.n_targets, ab.data() + (size_t)lo * v.n_targets, nullptr) : 0;
           })) return 1;
      for (int b = b0; b < b0 + nb; b++) {
        a.assign(ab.begin() + (size_t)(b - b0) * v.n_targets, ab.begin() + (size_t)(b - b0 + 1) * v.n_targets);
        if (use_h5) h5.doubles(h5.bs(), ("bs" + std::to_string(b)).c_str(), a);   // H5Writer::write_bootstrap
        else write_abundance(opt.output + "/bs_abundance_" + std::to_string(b) + ".tsv", idx, v, a, eff);
      }
    }
    std::cerr << std::endl;
  }
  h5.close();
  if (opt.verbose) std::cerr << "[timing] total " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << " s" << std::endl;
  const int exit_code = num_pseudoaligned == 0 ? 1 : 0;  // src/main.cpp:2795-2797
  // (a profiler or tool attached to the process flushes its output from exit handlers: then the process leaves the ordinary way)
  const char* preload = getenv("LD_PRELOAD");
  const bool tool_attached = getenv("ROCP_TOOL_LIBRARIES") || getenv("HSA_TOOLS_LIB") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") ||
                             (preload && (strstr(preload, "rocprof") || strstr(preload, "roctracer") || strstr(preload, "asan")));
  if (n_gpus == 1 && !tool_attached && !getenv("KAMD_SLOW_EXIT")) {
    // every output file is written and closed: the process ends here instead of unmapping gigabytes of tables, destroying the
    // context and unloading the runtime piece by piece (a fraction of a second that produces nothing)
    std::cout.flush(); std::cerr.flush(); fflush(nullptr);
    _exit(exit_code);
  }
  for (kamd_comm* m : comms) kamd_comm_destroy(m);
  for (kamd_ctx* x : ctxs) kamd_ctx_destroy(x);
  kamd_index_free(idx);
  return exit_code;
}
