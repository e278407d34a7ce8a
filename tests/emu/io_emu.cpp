// tests/emu/io_emu.cpp -- TEST INFRASTRUCTURE: drives the front-end's sequence readers (kallisto_amd/csrc/kamd_fastq.h)
// on a box without a GPU.  Never linked into the product.
#include "../../kallisto_amd/csrc/kamd_fastq.h"

extern "C" {
// ChunkReader path (gzip / BGZF / FASTA): every sequence followed by '\n' into out; returns bytes written, -1 if cap is
// too small; *n_seqs = number of sequences
int64_t io_read_chunks(const char* path, int inflate_threads, uint64_t chunk, char* out, uint64_t cap, uint64_t* n_seqs) {
  kamd_io::ChunkReader r(path, chunk, inflate_threads);
  kamd_io::SeqChunk c;
  uint64_t o = 0, n = 0;
  while (r.next(c)) {
    for (size_t i = 0; i < c.off.size(); i++) {
      if (o + (uint64_t)c.len[i] + 1 > cap) return -1;
      memcpy(out + o, c.seqs.data() + c.off[i], (size_t)c.len[i]); o += (uint64_t)c.len[i]; out[o++] = '\n';
      ++n;
    }
  }
  *n_seqs = n;
  return (int64_t)o;
}
// MappedFastq path (plain 4-line FASTQ, `threads` parser threads); returns -2 when the file is not plain 4-line FASTQ
int64_t io_index_fastq(const char* path, int threads, char* out, uint64_t cap, uint64_t* n_seqs) {
  kamd_io::MappedFastq m;
  if (!m.open(path) || !m.index_records(threads)) { m.close(); return -2; }
  uint64_t o = 0;
  for (size_t i = 0; i < m.off.size(); i++) {
    if (o + (uint64_t)m.len[i] + 1 > cap) { m.close(); return -1; }
    memcpy(out + o, m.data + m.off[i], (size_t)m.len[i]); o += (uint64_t)m.len[i]; out[o++] = '\n';
  }
  *n_seqs = m.off.size();
  m.close();
  return (int64_t)o;
}
// read everything, keep nothing: number of sequences (throughput measurements)
uint64_t io_count(const char* path, int inflate_threads, uint64_t chunk, uint64_t* bases) {
  kamd_io::ChunkReader r(path, chunk, inflate_threads);
  kamd_io::SeqChunk c;
  uint64_t n = 0, b = 0;
  while (r.next(c)) { n += c.off.size(); b += c.seqs.size(); }
  *bases = b;
  return n;
}
int io_is_bgzf(const char* path) { return kamd_io::BgzfSource::is_bgzf(path) ? 1 : 0; }
}
