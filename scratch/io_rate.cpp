// scratch/io_rate.cpp -- the host side of the input path alone: TextSource (plain / gzip / BGZF readers into a ring) + UnitCutter, the units
// released unread.  No GPU: what the reader threads of a box can deliver.
//   g++ -O3 -std=c++17 -march=native -pthread scratch/io_rate.cpp -o /tmp/io_rate -lz -ldl
//   /tmp/io_rate threads_per_file ring_mb unit_mb r_1.fq[.gz] [r_2.fq[.gz]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include "../kallisto_amd/csrc/kamd_textsource.h"
int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: io_rate threads_per_file ring_mb unit_mb file [file]\n"); return 1; }
  const int threads = atoi(argv[1]); const size_t ring = (size_t)atoi(argv[2]) << 20, unit = (size_t)atoi(argv[3]) << 20;
  const int nf = argc - 4;
  char* rings[2] = {nullptr, nullptr};
  std::unique_ptr<kamd_io::TextSource> src[2];
  for (int f = 0; f < nf; f++) { if (posix_memalign((void**)&rings[f], 4096, ring)) return 1; memset(rings[f], 1, ring); }
  const auto t0 = std::chrono::steady_clock::now();
  for (int f = 0; f < nf; f++) { src[f].reset(new kamd_io::TextSource(argv[4 + f], rings[f], ring, threads, 1 << 20)); if (src[f]->failed()) { fprintf(stderr, "%s\n", src[f]->error().c_str()); return 1; } }
  kamd_io::UnitCutter cut(src[0].get(), nf == 2 ? src[1].get() : nullptr, unit, unit + unit / 2);
  kamd_io::UnitCut u; uint64_t bytes = 0, recs = 0, units = 0; double first = 0;
  for (;;) {
    const int rc = cut.next(u);
    if (rc == kamd_io::UnitCutter::DONE) break;
    if (rc < 0) { fprintf(stderr, "cutter: %d %s\n", rc, src[0]->error().c_str()); return 1; }
    if (!units) first = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int f = 0; f < nf; f++) { bytes += u.end[f] - u.begin[f]; src[f]->release(u.end[f]); }
    recs += u.n_records; ++units;
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("%lu records, %lu units, %.2f GB of text in %.3f s = %.2f GB/s, %.2f M records/s (first unit after %.1f ms)\n", recs, units, bytes / 1e9, s, bytes / 1e9 / s, recs / 1e6 / s, first * 1e3);
}
