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
#include <sched.h>
#include <unistd.h>
#include <vector>

#include "../../include/kallisto_amd.h"
#include "kamd_fastq.h"
#include "kamd_textsource.h"
#include "kamd_genes.h"

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

// CPUs this process may actually keep busy: the CPU quota of its cgroup (containers routinely show all of the host's processors in
// nproc while cpu.max grants a fraction; running more busy threads than the quota gets the whole group throttled for the rest of every
// scheduling period, which stalls the thread that feeds the GPU), capped by the affinity mask and the processor count.
inline int effective_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n > 0 ? n : CPU_COUNT(&set), CPU_COUNT(&set));
  auto quota = [](const char* path, const char* path_period) -> double {
    std::ifstream f(path);
    if (!f.is_open()) return 0.0;
    std::string q; double period = 100000.0;
    f >> q;
    if (path_period) { std::ifstream g(path_period); if (g.is_open()) g >> period; } else f >> period;
    if (q.empty() || q == "max" || q[0] == '-') return 0.0;
    const double v = atof(q.c_str());
    return (v > 0 && period > 0) ? v / period : 0.0;
  };
  double q = quota("/sys/fs/cgroup/cpu.max", nullptr);
  if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  if (q > 0) n = std::min(n, std::max(1, (int)(q + 0.5)));
  if (const char* e = getenv("KAMD_CPUS")) n = std::max(1, atoi(e));
  return std::max(1, n);
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
  // pipeline g on HIP device devices[g] (several pipelines may share a device)
  MultiPipe(const std::vector<int>& devices, std::function<int(int, PackedBatch&, std::string&)> run, std::function<bool()> pin_to_first) : pin_(std::move(pin_to_first)) {
    for (size_t g = 0; g < devices.size(); g++) pipes_.emplace_back(new DevicePipe([run, g](PackedBatch& b, std::string& e) { return run((int)g, b, e); }, devices[g]));
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

// ---- the device-parse pipeline --------------------------------------------------------------------------------------------
// Strict 4-line FASTQ (plain, gzip or BGZF) never meets a host parser: host threads move the text of the file(s) into rings of
// pinned memory and count newlines (kamd_textsource.h), the calling thread cuts units of whole records and deals them to the
// GPUs (round robin; to GPU 0 while the fragment-length sample is open), a copy stream per GPU brings a unit's bytes into one of
// its text buffers, and the GPU's consumer thread runs kamd_fastq_unit_pack (lines, record check, 2-bit packing on the device) and
// then the same `run` callback the host-packed path uses.  Units are processed in input order on every GPU.  A unit the device
// parser declines (multi-line records, FASTA ...) stops the feeder with DECLINED: the caller starts over with the general reader.
class UnitFeeder {
 public:
  enum { OK = 0, FAILED = 1, DECLINED = 2 };
  // ring: pinned bytes per file; target: bytes of file 0 per unit; max_unit: capacity of a device text buffer; block: bytes a reader
  // thread moves (and counts) at a time
  struct Sizes { size_t ring = 192u << 20, target = 32u << 20, max_unit = 48u << 20, block = 1u << 20; int bufs = 20; uint64_t batch_items = 2u << 20; };
  static Sizes sizes_from_env() {
    Sizes z;
    size_t t = 0;
    if (const char* e = getenv("KAMD_FQ_UNIT_MB")) t = (size_t)std::max(1, atoi(e)) << 20;
    if (const char* e = getenv("KAMD_FQ_UNIT_BYTES")) t = (size_t)std::max(256, atoi(e));   // (tests: many units from small files)
    if (const char* e = getenv("KAMD_FQ_BATCH_ITEMS")) z.batch_items = (uint64_t)std::max(1, atoi(e));
    if (const char* e = getenv("KAMD_FQ_BUFS")) z.bufs = std::max(3, atoi(e));
    if (t) { z.target = t; z.max_unit = std::max<size_t>(t + t / 2, 1u << 20); z.block = std::min<size_t>(z.block, std::max<size_t>(t / 4, 64)); z.ring = std::max<size_t>(6 * t, z.max_unit + 4 * z.block + (1u << 16)); }
    if (const char* e = getenv("KAMD_FQ_RING_MB")) z.ring = std::max((size_t)std::max(1, atoi(e)) << 20, z.max_unit + 4 * z.block + (1u << 16));
    return z;
  }
  // ctxs[g] lives on HIP device devices[g] (several contexts may share a device)
  UnitFeeder(std::vector<kamd_ctx*> ctxs, std::vector<int> devices, std::function<int(int, PackedBatch&, std::string&)> run,
             std::function<bool()> pin_to_first, Sizes z = sizes_from_env())
      : ctxs_(std::move(ctxs)), devs_(std::move(devices)), run_(std::move(run)), pin_(std::move(pin_to_first)), z_(z) {
    for (size_t g = 0; g < ctxs_.size(); g++) gpus_.emplace_back(new Gpu);
  }
  ~UnitFeeder() {
    stop_consumers();
    for (size_t g = 0; g < gpus_.size(); g++) {
      (void)hipSetDevice(devs_[g]);
      Gpu& G = *gpus_[g];
      for (auto& b : G.bufs) { for (int f = 0; f < 2; f++) if (b.d[f]) (void)hipFree(b.d[f]); if (b.copied) (void)hipEventDestroy(b.copied); }
      if (G.copy) (void)hipStreamDestroy(G.copy);
    }
    for (int f = 0; f < 2; f++) if (ring_[f]) (void)hipHostFree(ring_[f]);
  }
  // pinned rings and device text buffers (slow: page pinning) -- callable from another thread while the index loads
  int prepare(int n_files) {
    std::lock_guard<std::mutex> lk(prep_m_);
    for (int f = 0; f < n_files; f++)
      if (!ring_[f] && hipHostMalloc((void**)&ring_[f], z_.ring, hipHostMallocPortable) != hipSuccess) { error_ = "pinned allocation of the text ring failed"; return FAILED; }
    for (size_t g = 0; g < gpus_.size(); g++) {
      Gpu& G = *gpus_[g];
      if (hipSetDevice(devs_[g]) != hipSuccess) { error_ = "hipSetDevice failed"; return FAILED; }
      if (!G.copy && hipStreamCreateWithFlags(&G.copy, hipStreamNonBlocking) != hipSuccess) { error_ = "hipStreamCreate failed"; return FAILED; }
      if (G.bufs.empty()) G.bufs.resize((size_t)z_.bufs);
      for (auto& b : G.bufs) {
        for (int f = 0; f < n_files; f++)
          if (!b.d[f] && hipMalloc((void**)&b.d[f], z_.max_unit + 64) != hipSuccess) { error_ = "device allocation of a text buffer failed"; return FAILED; }
        if (!b.copied && hipEventCreateWithFlags(&b.copied, hipEventDisableTiming) != hipSuccess) { error_ = "hipEventCreate failed"; return FAILED; }
      }
    }
    return OK;
  }
  const std::string& error() const { return error_; }
  double wait_s = 0.0, device_s = 0.0;   // dispatcher blocked on a free text buffer / busiest consumer
  double copy_wait_s = 0.0, parse_s = 0.0, run_s = 0.0, cut_s = 0.0;   // consumers (summed over GPUs): waiting for a unit's copy, kamd_fastq_unit_pack, the run callback; dispatcher: waiting for the cutter
  uint64_t units = 0, bytes = 0;

  // the reads of one file (single-end) or one pair of files; n_items receives the records handed to `run`
  int feed(const std::string& f0, const std::string* f1, int io_threads, uint64_t& n_items, bool verbose) {
    const int nf = f1 ? 2 : 1;
    if (prepare(nf) != OK) return FAILED;
    start_consumers();
    // readers: what the caller allows, within the CPUs this process may keep busy minus the dispatcher, the consumer and the runtime's own threads
    // (the dispatcher, the consumer and the runtime's threads mostly wait: one CPU is left to them)
    int spare = 1;
    if (const char* e = getenv("KAMD_FQ_SPARE_CPUS")) spare = std::max(0, atoi(e));
    const int per_file = std::max(1, (std::min(io_threads, effective_cpus()) - spare) / nf);
    std::unique_ptr<TextSource> src[2];
    for (int f = 0; f < nf; f++) {
      src[f].reset(new TextSource(f ? *f1 : f0, ring_[f], z_.ring, per_file, z_.block));
      if (src[f]->failed()) { error_ = src[f]->error(); return FAILED; }
    }
    UnitCutter cut(src[0].get(), nf == 2 ? src[1].get() : nullptr, z_.target, z_.max_unit);
    tracker_.reset(src, nf);
    UnitCut u;
    int rc = OK;
    for (;;) {
      const auto tc0 = std::chrono::steady_clock::now();
      const int c = cut.next(u);
      cut_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
      if (c == UnitCutter::DONE) break;
      if (c == UnitCutter::IO_ERROR) { for (int f = 0; f < nf; f++) if (src[f]->failed()) error_ = src[f]->error(); rc = FAILED; break; }
      if (c == UnitCutter::COUNT_MISMATCH) { error_ = "paired-end files have different numbers of reads"; rc = FAILED; break; }
      if (c < 0) { rc = DECLINED; break; }   // NOT_STRICT, TOO_LONG: the general reader decides what this input is
      if (state() != OK) break;
      const int g = pin_() ? 0 : (int)(rr_++ % gpus_.size());
      Gpu& G = *gpus_[g];
      int bi;
      {
        const auto t0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> lk(G.m);
        G.cv.wait(lk, [&] { for (auto& b : G.bufs) if (!b.busy) return true; return false; });
        wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bi = 0; while (G.bufs[bi].busy) ++bi;
        G.bufs[bi].busy = true;
      }
      Buf& b = G.bufs[bi];
      // a buffer that is marked busy but never queued has no consumer to clear it: drain() below would wait for it for ever (a lost
      // device would hang the process instead of ending it with the error) -- every failure path hands the buffer back first
      auto give_back = [&] { { std::lock_guard<std::mutex> lk(G.m); G.bufs[bi].busy = false; } G.cv.notify_all(); };
      if (hipSetDevice(devs_[g]) != hipSuccess) { error_ = "hipSetDevice failed"; rc = FAILED; give_back(); break; }
      bool ok = true;
      for (int f = 0; f < nf && ok; f++) {
        const char* p[2]; size_t n[2];
        const int np = src[f]->pieces(u.begin[f], u.end[f], p, n);
        size_t at = 0;
        for (int i = 0; i < np && ok; i++) { ok = hipMemcpyAsync(b.d[f] + at, p[i], n[i], hipMemcpyHostToDevice, G.copy) == hipSuccess; at += n[i]; }
        b.n_bytes[f] = u.end[f] - u.begin[f];
        bytes += b.n_bytes[f];
      }
      if (ok) ok = hipEventRecord(b.copied, G.copy) == hipSuccess;
      if (!ok) { error_ = "copy of a unit of text to the device failed"; rc = FAILED; give_back(); break; }
      b.n_records = u.n_records; b.n_files = nf;
      b.seq = tracker_.add(u.end);
      // (the releaser gets the unit's sequence number and event by value: by the time it looks, the consumer may have released the buffer
      // and this thread recorded the next unit in it -- it must complete THIS unit's number, or the ring is never released)
      { std::lock_guard<std::mutex> lk(G.m); G.queue.push_back(bi); G.copies.push_back(Copy{bi, b.seq, b.copied}); }
      G.cv.notify_all();
      n_items += u.n_records; ++units;
      if (verbose) std::cerr << "[quant] processed " << n_items << (nf == 2 ? " pairs" : " reads") << std::endl;
    }
    // every unit of this input leaves the rings before the sources go away
    drain();
    if (rc == OK) rc = state();
    if (rc == FAILED && error_.empty()) error_ = fail_msg();
    return rc;
  }
  void finish() { drain(); stop_consumers(); }
  // after DECLINED: the next input (another sample) may try the device parser again
  void clear_declined() { std::lock_guard<std::mutex> g(st_m_); if (state_ == DECLINED) state_ = OK; }

 private:
  struct Buf { char* d[2] = {nullptr, nullptr}; uint64_t n_bytes[2] = {0, 0}, n_records = 0, seq = 0; int n_files = 1; hipEvent_t copied = nullptr; bool busy = false; };
  struct Copy { int bi; uint64_t seq; hipEvent_t copied; };
  struct Gpu { std::vector<Buf> bufs; hipStream_t copy = nullptr; std::deque<int> queue; std::deque<Copy> copies; std::mutex m; std::condition_variable cv; std::thread th, rel;
               double busy_s = 0.0, copy_wait_s = 0.0, parse_s = 0.0, run_s = 0.0; };
  // the rings are released in input order, whatever order the GPUs' copies complete in
  struct Tracker {
    std::mutex m; TextSource* s[2] = {nullptr, nullptr}; int nf = 0;
    std::deque<std::pair<uint64_t, uint64_t>> ends; std::deque<char> done; uint64_t first = 0, next = 0;
    void reset(std::unique_ptr<TextSource> src[2], int n) { std::lock_guard<std::mutex> g(m); nf = n; for (int f = 0; f < 2; f++) s[f] = f < n ? src[f].get() : nullptr; ends.clear(); done.clear(); first = next; }
    uint64_t add(const uint64_t end[2]) { std::lock_guard<std::mutex> g(m); ends.emplace_back(end[0], end[1]); done.push_back(0); return next++; }
    void complete(uint64_t seq) {
      std::lock_guard<std::mutex> g(m);
      if (seq < first) return;
      done[(size_t)(seq - first)] = 1;
      while (!done.empty() && done.front()) {
        if (s[0]) s[0]->release(ends.front().first);
        if (s[1]) s[1]->release(ends.front().second);
        ends.pop_front(); done.pop_front(); ++first;
      }
    }
    void detach() { std::lock_guard<std::mutex> g(m); s[0] = s[1] = nullptr; }
  };
  int state() { std::lock_guard<std::mutex> g(st_m_); return state_; }
  std::string fail_msg() { std::lock_guard<std::mutex> g(st_m_); return fail_msg_; }
  void set_state(int s, const std::string& msg) { std::lock_guard<std::mutex> g(st_m_); if (state_ == OK) { state_ = s; fail_msg_ = msg; } }
  void start_consumers() {
    if (started_) return;
    started_ = true;
    for (size_t g = 0; g < gpus_.size(); g++) { gpus_[g]->th = std::thread([this, g] { consume((int)g); }); gpus_[g]->rel = std::thread([this, g] { release_loop((int)g); }); }
  }
  // the ring's bytes of a unit are free once its copy has completed -- long before the consumer, busy with a batch, gets to the unit
  void release_loop(int g) {
    Gpu& G = *gpus_[g];
    (void)hipSetDevice(devs_[g]);
    for (;;) {
      Copy cp;
      {
        std::unique_lock<std::mutex> lk(G.m);
        G.cv.wait(lk, [&] { return !G.copies.empty(); });
        cp = G.copies.front(); G.copies.pop_front();
      }
      if (cp.bi < 0) return;
      // (an event recorded again for a later unit of the same copy stream completes no earlier than this unit's copy did)
      if (hipEventSynchronize(cp.copied) != hipSuccess) set_state(FAILED, "copy of a unit of text to the device failed");
      tracker_.complete(cp.seq);
    }
  }
  void stop_consumers() {
    if (!started_) return;
    for (auto& G : gpus_) { { std::lock_guard<std::mutex> lk(G->m); G->queue.push_back(-1); G->copies.push_back(Copy{-1, 0, nullptr}); } G->cv.notify_all(); }
    for (auto& G : gpus_) { if (G->th.joinable()) G->th.join(); if (G->rel.joinable()) G->rel.join(); }
    started_ = false;
    device_s = copy_wait_s = parse_s = run_s = 0.0;
    for (auto& G : gpus_) { device_s = std::max(device_s, G->busy_s); copy_wait_s += G->copy_wait_s; parse_s += G->parse_s; run_s += G->run_s; }
  }
  void drain() {   // until no text buffer is in flight
    for (auto& G : gpus_) {
      std::unique_lock<std::mutex> lk(G->m);
      G->cv.wait(lk, [&] { for (auto& b : G->bufs) if (b.busy) return false; return true; });
    }
    tracker_.detach();
  }
  // Units are parsed as they arrive; their reads are packed and pseudoaligned as ONE batch once z_.batch_items records have come
  // together -- or the queue runs dry, when there is nothing better to do (kamd_pseudoalign has fixed costs of a few ms per call).
  void consume(int g) {
    Gpu& G = *gpus_[g];
    if (hipSetDevice(devs_[g]) != hipSuccess) { set_state(FAILED, "hipSetDevice failed"); }
    std::vector<int> pending;   // text buffers of the batch under construction
    uint64_t pending_records = 0;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&t0](double& acc) { const auto t1 = std::chrono::steady_clock::now(); acc += std::chrono::duration<double>(t1 - t0).count(); t0 = t1; };
    auto flush = [&] {
      if (pending.empty()) return;
      t0 = std::chrono::steady_clock::now();
      if (state() == OK) {
        kamd_fastq_unit fu;
        if (kamd_fastq_batch_pack(ctxs_[g], &fu) != 0) set_state(FAILED, kamd_last_error());
        else if (fu.n_items) {
          PackedBatch pb;
          pb.d_words = const_cast<uint32_t*>(fu.d_words); pb.d_len = const_cast<uint16_t*>(fu.d_len);
          pb.n_items = fu.n_items; pb.n_reads = fu.n_items * (uint64_t)G.bufs[pending[0]].n_files; pb.max_len = fu.max_len; pb.filled = true;
          std::string err;
          if (run_(g, pb, err)) set_state(FAILED, err);
        }
      }
      lap(G.run_s);
      { std::lock_guard<std::mutex> lk(G.m); for (int bi : pending) G.bufs[bi].busy = false; }
      G.cv.notify_all();
      pending.clear(); pending_records = 0;
    };
    for (;;) {
      int bi;
      {
        std::unique_lock<std::mutex> lk(G.m);
        if (G.queue.empty() && !pending.empty()) { lk.unlock(); flush(); lk.lock(); }
        G.cv.wait(lk, [&] { return !G.queue.empty(); });
        bi = G.queue.front(); G.queue.pop_front();
      }
      if (bi < 0) { flush(); G.busy_s = G.copy_wait_s + G.parse_s + G.run_s; return; }
      Buf& b = G.bufs[bi];
      t0 = std::chrono::steady_clock::now();
      const bool copied = hipEventSynchronize(b.copied) == hipSuccess;
      lap(G.copy_wait_s);
      if (!copied) set_state(FAILED, "copy of a unit of text to the device failed");
      pending.push_back(bi);
      if (state() == OK) {   // (after a failure or a declined unit the remaining units are only drained)
        kamd_fastq_unit fu;
        const char* txt[2] = {b.d[0], b.d[1]};
        const int prc = kamd_fastq_unit_parse(ctxs_[g], txt, b.n_bytes, b.n_files, b.n_records, &fu);
        lap(G.parse_s);
        if (prc != 0) set_state(FAILED, kamd_last_error());
        else if (fu.status == 3) set_state(FAILED, "reads longer than 65535 bp are outside the short-read GPU path");
        else if (fu.status != 0) set_state(DECLINED, "");
        else pending_records += b.n_records;
      }
      if (state() != OK || pending_records >= z_.batch_items || pending.size() + 2 >= G.bufs.size()) flush();
    }
  }
  std::vector<kamd_ctx*> ctxs_; std::vector<int> devs_;
  std::function<int(int, PackedBatch&, std::string&)> run_;
  std::function<bool()> pin_;
  Sizes z_;
  char* ring_[2] = {nullptr, nullptr};
  std::vector<std::unique_ptr<Gpu>> gpus_;
  Tracker tracker_;
  std::mutex st_m_, prep_m_; int state_ = OK; std::string fail_msg_, error_;
  bool started_ = false;
  uint64_t rr_ = 0;
};

// ---- --share-device: the several-GPU flow of the front-end on ONE device --------------------------------------------------------
// RCCL refuses two ranks on one device, so a single-GPU box could never execute `--gpus N` (one context, pipeline and host thread
// per rank, EC merge, partitioned EM, replicates dealt round the ranks).  With --share-device every rank's context lives on
// device 0 and the collectives are these callbacks (kamd_comm_create_callbacks): a barrier among the ranks' host threads, the data
// staged through the host.  A test vehicle for the N > 1 code path, not a way to go faster.
class SharedDeviceComm {
 public:
  explicit SharedDeviceComm(int world) : world_(world), stage_((size_t)world), ranks_((size_t)world) { for (int r = 0; r < world; r++) ranks_[r] = Rank{this, r}; }
  struct Rank { SharedDeviceComm* self; int rank; };
  void* user(int r) { return &ranks_[(size_t)r]; }
  static kamd_comm_callbacks callbacks() { kamd_comm_callbacks cb; cb.allreduce_sum = &allreduce; cb.allgather = &allgather; cb.broadcast = &broadcast; return cb; }
 private:
  void barrier() {
    std::unique_lock<std::mutex> lk(m_);
    const uint64_t gen = gen_;
    if (++arrived_ == world_) { arrived_ = 0; ++gen_; cv_.notify_all(); }
    else cv_.wait(lk, [&] { return gen_ != gen; });
  }
  static int allreduce(void* user, void* d_buf, uint64_t count, int32_t type) {
    Rank* R = (Rank*)user; SharedDeviceComm* S = R->self;
    const size_t esz = type <= 1 ? 4 : 8, bytes = (size_t)count * esz;
    std::vector<unsigned char>& mine = S->stage_[(size_t)R->rank];
    mine.resize(bytes);
    if (hipMemcpy(mine.data(), d_buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    S->barrier();
    std::vector<unsigned char> sum(bytes, 0);
    for (int r = 0; r < S->world_; r++) {
      const unsigned char* p = S->stage_[(size_t)r].data();
      for (uint64_t i = 0; i < count; i++) {
        if (type == 0) ((uint32_t*)sum.data())[i] += ((const uint32_t*)p)[i];
        else if (type == 1) ((int32_t*)sum.data())[i] += ((const int32_t*)p)[i];
        else if (type == 2) ((uint64_t*)sum.data())[i] += ((const uint64_t*)p)[i];
        else ((double*)sum.data())[i] += ((const double*)p)[i];
      }
    }
    S->barrier();   // everybody has read the staging buffers
    return hipMemcpy(d_buf, sum.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
  }
  static int allgather(void* user, const void* d_send, void* d_recv, uint64_t bytes) {
    Rank* R = (Rank*)user; SharedDeviceComm* S = R->self;
    std::vector<unsigned char>& mine = S->stage_[(size_t)R->rank];
    mine.resize((size_t)bytes);
    if (hipMemcpy(mine.data(), d_send, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    S->barrier();
    int rc = 0;
    for (int r = 0; r < S->world_ && !rc; r++)
      rc = hipMemcpy((char*)d_recv + (size_t)r * bytes, S->stage_[(size_t)r].data(), (size_t)bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
    S->barrier();
    return rc;
  }
  static int broadcast(void* user, void* d_buf, uint64_t bytes, int32_t root) {
    Rank* R = (Rank*)user; SharedDeviceComm* S = R->self;
    int rc = 0;
    if (R->rank == root) { S->stage_[(size_t)root].resize((size_t)bytes); rc = hipMemcpy(S->stage_[(size_t)root].data(), d_buf, (size_t)bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1; }
    S->barrier();
    if (R->rank != root) rc = hipMemcpy(d_buf, S->stage_[(size_t)root].data(), (size_t)bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
    S->barrier();
    return rc;
  }
  int world_;
  std::vector<std::vector<unsigned char>> stage_;
  std::vector<Rank> ranks_;
  std::mutex m_; std::condition_variable cv_; int arrived_ = 0; uint64_t gen_ = 0;
};

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
  // (the same bytes as the reference's `<< std::endl` per line, without its flush -- one write call per target, 200 000 of them per file)
  of << "target_id" << "\t" << "length" << "\t" << "eff_length" << "\t" << "est_counts" << "\t" << "tpm" << '\n';
  for (size_t i = 0; i < alpha.size(); ++i)
    of << kamd_index_target_name(idx, i) << '\t' << (uint32_t)v.target_lens[i] << '\t' << eff[i] << '\t' << alpha[i] << '\t' << tpm[i] << '\n';
  of.flush();
  if (!of) { std::cerr << "Error: could not write " << path << std::endl; exit(1); }
}


// Reads the FASTA/FASTQ(.gz) files of one sample (FastqSequenceReader::fetchSequences, src/ProcessReads.cpp:3128-3267: mates in
// consecutive files), packs batches of `batch` reads / pairs with all host threads and hands them to the pipeline in input
// order.  Returns 0, or 1 after printing the error.
// a file the device parser may take: a regular file that is gzip, or plain text that starts like a FASTQ record
inline bool device_parse_candidate(const std::string& path) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) return false;
  unsigned char m[2] = {0, 0};
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const size_t n = fread(m, 1, 2, f); fclose(f);
  return (n == 2 && m[0] == 0x1f && m[1] == 0x8b) || (n >= 1 && m[0] == '@');
}

// `units` (optional): the device-parse pipeline; strict 4-line FASTQ goes through it, and when it declines an input `reset` must put
// the run back to its start (EC counts, fragment-length sample, whatever `run` accumulates): all files are then read again with
// the general reader below.
inline int feed_files(const std::vector<std::string>& files, bool paired, uint64_t batch, int host_threads, int io_threads, bool verbose,
                      MultiPipe& pipe, uint64_t& n_processed, double& pack_s, UnitFeeder* units = nullptr,
                      const std::function<int()>& reset = nullptr) {
  const uint64_t n_before = n_processed;
  if (units && !getenv("KAMD_HOST_PARSE")) {
    bool all = true;
    for (const auto& f : files) all = all && device_parse_candidate(f);
    int rc = all ? UnitFeeder::OK : UnitFeeder::DECLINED;
    for (size_t fi = 0; all && fi < files.size() && rc == UnitFeeder::OK; fi += paired ? 2 : 1)
      rc = units->feed(files[fi], paired ? &files[fi + 1] : nullptr, io_threads, n_processed, verbose);
    units->finish();
    if (rc == UnitFeeder::OK) return 0;
    if (rc == UnitFeeder::FAILED) { std::cerr << "Error: " << units->error() << std::endl; return 1; }
    // declined: not (only) strict 4-line FASTQ -- start over with the general reader
    if (all && verbose) std::cerr << "[quant] input is not strict 4-line FASTQ: reading it with the general FASTA/FASTQ reader" << std::endl;
    if (all && reset && reset()) { std::cerr << "Error: " << kamd_last_error() << std::endl; return 1; }
    units->clear_declined();
    n_processed = n_before;
  }
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
