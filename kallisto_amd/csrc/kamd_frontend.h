// kamd_frontend.h -- pieces shared by the sub-commands of the stand-alone front-end (quant_main.cpp: `quant`; bus_main.cpp:
// `bus -x bulk`, `quant-tcc`): option helper, the host -> device batch pipeline, the FASTQ feeding loop, writers.
// Host code only; everything that computes goes through the C ABI of include/kallisto_amd.h.
#pragma once
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/kallisto_amd.h"
#include "kamd_fastq.h"

namespace kamd_fe {
using namespace kamd_io;

static const char* const KALLISTO_COMPAT_VERSION = "0.51.1";  // src/common.h:4

inline bool take(const std::string& a, const char* shortf, const char* longf, int& i, int argc, char** argv, std::string& val) {
  std::string lf = std::string(longf) + "=";
  if (a.rfind(lf, 0) == 0) { val = a.substr(lf.size()); return true; }
  if (a == longf || (shortf && a == shortf)) { if (i + 1 >= argc) { std::cerr << "Error: missing value for " << a << std::endl; exit(1); } val = argv[++i]; return true; }
  if (shortf && a.size() > 2 && a.compare(0, 2, shortf) == 0) { val = a.substr(2); return true; }   // getopt's attached form: -t4, -l200
  return false;
}

// FASTA/FASTQ input (SeqReader, ChunkReader, MappedFastq, BgzfSource): kamd_fastq.h
#define HIPX(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::cerr << "Error: " #x ": " << hipGetErrorString(e_) << std::endl; exit(1); } } while (0)
#define KX(x) do { if ((x) != 0) { std::cerr << "Error: " << kamd_last_error() << std::endl; exit(1); } } while (0)

// Two-slot pipeline between the host (FASTQ parsing + 2-bit packing into pinned memory, all host threads) and the device
// (H2D copy, kamd_pseudoalign, FLD sample): batch i+1 is packed while batch i runs.  One consumer thread, batches in input
// order (the FLD sample and the first-occurrence EC ids depend on it).
struct PackedBatch {
  uint32_t* h_words = nullptr; uint16_t* h_len = nullptr; size_t hw_cap = 0, hl_cap = 0;   // pinned host memory
  uint32_t* d_words = nullptr; uint16_t* d_len = nullptr; size_t dw_cap = 0, dl_cap = 0;
  uint64_t n_items = 0, n_reads = 0, n_words = 0;
  int32_t max_len = 1;
  bool filled = false;
};
class DevicePipe {
 public:
  // run: device work of one batch; returns 0 or an error code with the message in `err` (reported by the main thread: the
  // consumer never exits the process itself)
  explicit DevicePipe(std::function<int(PackedBatch&, std::string&)> run, int device = 0) : run_(std::move(run)), device_(device), th_([this] { loop(); }) {}
  bool failed() { std::lock_guard<std::mutex> lk(m_); return failed_; }
  std::string error() { std::lock_guard<std::mutex> lk(m_); return error_; }
  // a free slot whose pinned buffers hold n_words / n_reads entries (blocks while both slots are in flight)
  PackedBatch& acquire(uint64_t n_words, uint64_t n_reads) {
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !slot_[fill_].filled; });
    lk.unlock();
    wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    PackedBatch& b = slot_[fill_];
    if (n_words > b.hw_cap) { if (b.h_words) HIPX(hipHostFree(b.h_words)); b.hw_cap = n_words * 5 / 4; HIPX(hipHostMalloc((void**)&b.h_words, b.hw_cap * 4, hipHostMallocPortable)); }
    if (n_reads > b.hl_cap) { if (b.h_len) HIPX(hipHostFree(b.h_len)); b.hl_cap = n_reads * 5 / 4; HIPX(hipHostMalloc((void**)&b.h_len, b.hl_cap * 2, hipHostMallocPortable)); }
    b.n_words = n_words; b.n_reads = n_reads;
    return b;
  }
  void submit() {
    { std::lock_guard<std::mutex> lk(m_); slot_[fill_].filled = true; }
    cv_.notify_all();
    fill_ ^= 1;
  }
  ~DevicePipe() { finish(); }   // also on the error returns of main
  void finish() {
    if (!th_.joinable()) return;
    { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return !slot_[0].filled && !slot_[1].filled; }); stop_ = true; }
    cv_.notify_all();
    th_.join();
    (void)hipSetDevice(device_);
    for (PackedBatch& b : slot_) {
      if (b.h_words) (void)hipHostFree(b.h_words);
      if (b.h_len) (void)hipHostFree(b.h_len);
      if (b.d_words) (void)hipFree(b.d_words);
      if (b.d_len) (void)hipFree(b.d_len);
    }
  }
  double wait_s = 0.0, device_s = 0.0;   // producer blocked on a free slot / consumer busy
 private:
  void loop() {
    HIPX(hipSetDevice(device_));
    for (;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return slot_[run_i_].filled || stop_; });
      if (!slot_[run_i_].filled) return;
      lk.unlock();
      const auto t0 = std::chrono::steady_clock::now();
      PackedBatch& b = slot_[run_i_];
      if (b.n_words > b.dw_cap) { if (b.d_words) HIPX(hipFree(b.d_words)); b.dw_cap = b.n_words * 5 / 4; HIPX(hipMalloc((void**)&b.d_words, b.dw_cap * 4)); }
      if (b.n_reads > b.dl_cap) { if (b.d_len) HIPX(hipFree(b.d_len)); b.dl_cap = b.n_reads * 5 / 4; HIPX(hipMalloc((void**)&b.d_len, b.dl_cap * 2)); }
      HIPX(hipMemcpy(b.d_words, b.h_words, b.n_words * 4, hipMemcpyHostToDevice));
      HIPX(hipMemcpy(b.d_len, b.h_len, b.n_reads * 2, hipMemcpyHostToDevice));
      std::string err;
      const int rc = failed_ ? 0 : run_(b, err);   // after a failure the remaining batches are only drained
      device_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      { std::lock_guard<std::mutex> g(m_); b.filled = false; if (rc) { failed_ = true; error_ = err; } }
      cv_.notify_all();
      run_i_ ^= 1;
    }
  }
  std::function<int(PackedBatch&, std::string&)> run_;
  bool failed_ = false; std::string error_;
  int device_ = 0;
  PackedBatch slot_[2];
  std::mutex m_;
  std::condition_variable cv_;
  bool stop_ = false;
  int fill_ = 0, run_i_ = 0;
  std::thread th_;   // last member: started when everything above exists
};

// One pipeline per GPU; the producer's next batch goes to the GPU route() names (round robin -- except that batches go to GPU 0
// while the fragment-length sample, the first 10 000 qualifying pairs of the input IN ORDER, is still being collected there).
class MultiPipe {
 public:
  MultiPipe(int n, std::function<int(int, PackedBatch&, std::string&)> run, std::function<bool()> pin_to_first) : pin_(std::move(pin_to_first)) {
    for (int g = 0; g < n; g++) pipes_.emplace_back(new DevicePipe([run, g](PackedBatch& b, std::string& e) { return run(g, b, e); }, g));
  }
  PackedBatch& acquire(uint64_t n_words, uint64_t n_reads) {
    cur_ = pin_() ? 0 : (int)(rr_++ % pipes_.size());
    return pipes_[cur_]->acquire(n_words, n_reads);
  }
  void submit() { pipes_[cur_]->submit(); }
  bool failed() { for (auto& p : pipes_) if (p->failed()) return true; return false; }
  std::string error() { for (auto& p : pipes_) if (p->failed()) return p->error(); return ""; }
  void finish() { for (auto& p : pipes_) p->finish(); }
  double wait_s() const { double x = 0; for (auto& p : pipes_) x += p->wait_s; return x; }
  double device_s() const { double x = 0; for (auto& p : pipes_) x = std::max(x, p->device_s); return x; }
 private:
  std::vector<std::unique_ptr<DevicePipe>> pipes_;
  std::function<bool()> pin_;
  int cur_ = 0; uint64_t rr_ = 0;
};

// ---- abundance.h5 (H5Writer.cpp:4-69, h5utils.h:42-92): one chunk per dataset, deflate level 6, strings as fixed-size
// NUL-terminated C strings of the longest entry + 1.  libhdf5 is loaded at run time (dlopen), so the front-end neither
// needs it to build nor drags its dependencies into a process that already holds the HIP runtime.  Written without

inline std::string to_json(const std::string& id, const std::string& val, bool quote, bool comma = true) {  // PlaintextWriter.cpp:113-137
  std::string out = "\t\"" + id + "\": ";
  if (quote) out += '"';
  out += val;
  if (quote) out += '"';
  if (comma) out += ',';
  return out;
}

inline void write_abundance(const std::string& path, const kamd_index* idx, const kamd_index_view& v, const std::vector<double>& alpha,
                     const std::vector<double>& eff) {  // plaintext_writer, PlaintextWriter.cpp:29-65
  std::ofstream of(path);
  if (!of.is_open()) { std::cerr << "Error: Couldn't open file: " << path << std::endl; exit(1); }
  std::vector<double> tpm(alpha.size());
  kamd_counts_to_tpm(alpha.data(), eff.data(), alpha.size(), tpm.data());
  of << "target_id" << "\t" << "length" << "\t" << "eff_length" << "\t" << "est_counts" << "\t" << "tpm" << std::endl;
  for (size_t i = 0; i < alpha.size(); ++i)
    of << kamd_index_target_name(idx, i) << '\t' << (uint32_t)v.target_lens[i] << '\t' << eff[i] << '\t' << alpha[i] << '\t' << tpm[i] << std::endl;
}


// Reads the FASTA/FASTQ(.gz) files of one sample (FastqSequenceReader::fetchSequences, src/ProcessReads.cpp:3128-3267: mates in
// consecutive files), packs batches of `batch` reads / pairs with all host threads and hands them to the pipeline in input
// order.  Returns 0, or 1 after printing the error.
inline int feed_files(const std::vector<std::string>& files, bool paired, uint64_t batch, int host_threads, int io_threads, bool verbose,
                      MultiPipe& pipe, uint64_t& n_processed, double& pack_s) {
  auto pack_and_submit = [&](const char* d1, const uint64_t* off1, const int32_t* len1, const char* d2, const uint64_t* off2, const int32_t* len2,
                             uint64_t nb, int32_t max_len) -> int {
    if (max_len > 65535) { std::cerr << "Error: reads longer than 65535 bp are outside the short-read GPU path" << std::endl; return 1; }
    const uint64_t rec = kamd_packed_record_words(max_len), n_reads = nb * (paired ? 2 : 1);
    PackedBatch& pb = pipe.acquire(n_reads * rec, n_reads);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th; std::vector<int> rcs(host_threads, 0); std::vector<std::string> errs(host_threads);
    for (int t = 0; t < host_threads; t++) th.emplace_back([&, t] {
      const uint64_t a = nb * t / host_threads, e = nb * (t + 1) / host_threads;
      if (e == a) return;
      const uint64_t first = a * (paired ? 2 : 1);
      rcs[t] = kamd_pack_reads_host_strided(d1, off1 + a, len1 + a, e - a, max_len, pb.h_words, pb.h_len, paired ? 2 : 1, first);
      if (paired && rcs[t] == 0) rcs[t] = kamd_pack_reads_host_strided(d2, off2 + a, len2 + a, e - a, max_len, pb.h_words, pb.h_len, 2, first + 1);
      if (rcs[t]) errs[t] = kamd_last_error();   // (the library keeps its message per thread)
    });
    for (auto& x : th) x.join();
    for (int t = 0; t < host_threads; t++) if (rcs[t]) { std::cerr << "Error: " << errs[t] << std::endl; return 1; }
    pack_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pb.n_items = nb; pb.max_len = max_len;
    pipe.submit();
    if (pipe.failed()) { std::cerr << "Error: " << pipe.error() << std::endl; return 1; }
    n_processed += nb;
    if (verbose) std::cerr << "[quant] processed " << n_processed << (paired ? " pairs" : " reads") << std::endl;
    return 0;
  };
  for (size_t fi = 0; fi < files.size(); fi += paired ? 2 : 1) {
    // plain 4-line FASTQ: memory-map, index the records with all host threads, pack batches in parallel
    if (!MappedFastq::is_gzip(files[fi]) && (!paired || !MappedFastq::is_gzip(files[fi + 1]))) {
      MappedFastq m1, m2;
      bool fast = m1.open(files[fi]) && m1.index_records(host_threads);
      if (fast && paired) fast = m2.open(files[fi + 1]) && m2.index_records(host_threads);
      if (fast) {
        if (paired && m1.off.size() != m2.off.size()) { std::cerr << "Error: paired-end files have different numbers of reads" << std::endl; return 1; }
        const uint64_t n_items_total = m1.off.size();
        for (uint64_t b0 = 0; b0 < n_items_total; b0 += batch) {
          const uint64_t nb = std::min<uint64_t>(batch, n_items_total - b0);
          int32_t max_len = 1;
          for (uint64_t i = b0; i < b0 + nb; i++) { max_len = std::max(max_len, m1.len[i]); if (paired) max_len = std::max(max_len, m2.len[i]); }
          if (pack_and_submit(m1.data, m1.off.data() + b0, m1.len.data() + b0, paired ? m2.data : nullptr, paired ? m2.off.data() + b0 : nullptr,
                              paired ? m2.len.data() + b0 : nullptr, nb, max_len)) return 1;
        }
        m1.close(); m2.close();
        continue;
      }
      m1.close(); m2.close();
    }
    const int inflate_threads = std::max(1, io_threads / (paired ? 2 : 1) - 1);   // BGZF input: block-parallel inflate
    ChunkReader r1(files[fi], batch, inflate_threads);
    std::unique_ptr<ChunkReader> r2(paired ? new ChunkReader(files[fi + 1], batch, inflate_threads) : nullptr);
    SeqChunk c1, c2;
    while (r1.next(c1)) {
      if (paired && (!r2->next(c2) || c2.off.size() != c1.off.size())) { std::cerr << "Error: paired-end files have different numbers of reads" << std::endl; return 1; }
      const uint64_t nb = c1.off.size();
      int32_t max_len = 1;
      for (auto l : c1.len) max_len = std::max(max_len, l);
      if (paired) for (auto l : c2.len) max_len = std::max(max_len, l);
      if (pack_and_submit(c1.seqs.data(), c1.off.data(), c1.len.data(), paired ? c2.seqs.data() : nullptr, paired ? c2.off.data() : nullptr,
                          paired ? c2.len.data() : nullptr, nb, max_len)) return 1;
    }
    if (paired && r2->next(c2)) { std::cerr << "Error: paired-end files have different numbers of reads" << std::endl; return 1; }
  }
  return 0;
}

// run_info.json (plaintext_aux, PlaintextWriter.cpp:140-197)
inline void write_run_info(const std::string& path, uint64_t n_targets, int n_bootstraps, uint64_t n_processed, uint64_t n_pseudoaligned,
                           uint64_t n_unique, int k, const std::string& start_time, const std::string& call) {
  double p_uniq = 0.0, p_aln = 0.0;
  if (n_processed > 0) { p_uniq = 100.0 * (double)n_unique / (double)n_processed; p_aln = 100.0 * (double)n_pseudoaligned / (double)n_processed; }
  std::stringstream s1, s2; s1 << std::fixed << std::setprecision(1) << p_uniq; s2 << std::fixed << std::setprecision(1) << p_aln;
  std::ofstream of(path);
  of << "{" << std::endl
     << to_json("n_targets", std::to_string(n_targets), false) << std::endl
     << to_json("n_bootstraps", std::to_string(n_bootstraps), false) << std::endl
     << to_json("n_processed", std::to_string(n_processed), false) << std::endl
     << to_json("n_pseudoaligned", std::to_string(n_pseudoaligned), false) << std::endl
     << to_json("n_unique", std::to_string(n_unique), false) << std::endl
     << to_json("p_pseudoaligned", s2.str(), false) << std::endl
     << to_json("p_unique", s1.str(), false) << std::endl
     << to_json("kallisto_version", KALLISTO_COMPAT_VERSION, true) << std::endl
     << to_json("index_version", "13", false) << std::endl
     << to_json("k-mer length", std::to_string(k), false) << std::endl
     << to_json("start_time", start_time, true) << std::endl
     << to_json("call", call, true, false) << std::endl
     << "}" << std::endl;
}

inline std::string now_string() {
  std::time_t tt = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
  std::string s = std::ctime(&tt);
  if (!s.empty() && s.back() == '\n') s.pop_back();
  return s;
}
inline std::string call_string(int argc, char** argv) {
  std::string call;
  for (int i = 0; i < argc; i++) { if (i) call += ' '; call += argv[i]; }
  return call;
}
inline uint64_t onlist_targets(const kamd_index_view& v) {
  uint64_t n_on = 0;
  for (uint64_t t = 0; t < v.n_targets; t++) n_on += (v.onlist_bits[t >> 5] >> (t & 31)) & 1u;
  return n_on;
}

int bus_main(int argc, char** argv);   // bus_main.cpp: `bus -x bulk`
int tcc_main(int argc, char** argv);   // bus_main.cpp: `quant-tcc`

}  // namespace kamd_fe
