// kamd_fq_core.h -- strict 4-line FASTQ text -> reads, the part of FastqSequenceReader::fetchSequences
// (src/ProcessReads.cpp:3128-3267) that is kseq_read (src/kseq.h:174-215), written once as host/device inline functions:
// the kernels of kamd_io.hip (k_fq_*) run them per tile / per record on text that already sits in HBM, tests/emu runs
// them on the CPU.
//
// Scope: a *unit* of text that starts at the first byte of a record and consists of whole 4-line records
//     @name [comment] \n  sequence \n  +[anything] \n  quality \n
// For such text kseq_read returns the second line of every record, minus a trailing '\r' when the line has more than one
// character (ks_getuntil2, src/kseq.h:137).  kseq itself accepts much more (FASTA, sequences and qualities over several lines,
// junk between records); fq_check_record says whether a record has the strict shape, i.e. whether kseq_read would return
// exactly its second line and leave the stream at the start of the next record:
//   * line 0 starts with '@'            (kseq_read scans for the next '@' or '>' when last_char == 0, :178-181)
//   * line 1 is not empty and does not start with '>', '+' or '@'   (:189-194: an empty line is skipped, those three end the sequence)
//   * line 2 starts with '+'            (otherwise the line is more sequence, :189-194)
//   * line 3, the quality, has the length of the sequence           (:210-213: shorter -> more lines are read, longer -> -2)
// Anything else is reported to the caller, which re-reads the input with the general reader (kamd_fastq.h SeqReader).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KAMD_FQ_HD __host__ __device__ __forceinline__
#else
#define KAMD_FQ_HD inline
#endif

namespace kamd_fq {

static const uint32_t FQ_TILE = 4096;        // bytes of text per tile (one block of 256 threads x 16 bytes)
static const uint32_t FQ_MAX_READ = 65535;   // read lengths are 16-bit in the packed layout

// 0x80 in every byte of w that equals '\n' (exact per byte: no borrow between bytes)
KAMD_FQ_HD uint32_t nl_flags(uint32_t w) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}
// the four flag bits of a word as bits 0..3
KAMD_FQ_HD uint32_t nl_mask4(uint32_t w) {
  const uint32_t f = nl_flags(w);
  return ((f >> 7) & 1u) | ((f >> 14) & 2u) | ((f >> 21) & 4u) | ((f >> 28) & 8u);
}

// length of a line as kseq hands it out: the trailing '\r' of a line of more than one character is dropped (src/kseq.h:137)
KAMD_FQ_HD uint32_t line_len(const char* text, uint64_t begin, uint64_t end) {
  uint64_t n = end - begin;
  if (n > 1 && text[end - 1] == '\r') --n;
  return (uint32_t)n;
}

struct Record { uint64_t seq_off; uint32_t seq_len; bool ok; };
// l0: first byte of the record; nl[0..3]: positions of the four '\n' that end its lines
KAMD_FQ_HD Record fq_check_record(const char* text, uint64_t l0, uint64_t nl0, uint64_t nl1, uint64_t nl2, uint64_t nl3) {
  Record r;
  const uint64_t s1 = nl0 + 1, s2 = nl1 + 1, s3 = nl2 + 1;
  r.seq_off = s1;
  r.seq_len = line_len(text, s1, nl1);
  bool ok = nl0 > l0 && text[l0] == '@';
  ok = ok && nl1 > s1;
  if (ok) { const char c = text[s1]; ok = c != '>' && c != '+' && c != '@'; }
  ok = ok && nl2 > s2 && text[s2] == '+';
  // the quality is appended line-wise with the same '\r' rule; one line must give exactly the sequence's length
  ok = ok && line_len(text, s3, nl3) == r.seq_len;
  r.ok = ok;
  return r;
}

// a record handed from the parse kernel to the packer: absolute address of the sequence (48 bits) | length << 48
KAMD_FQ_HD uint64_t rec_word(const char* seq, uint32_t len) { return (uint64_t)(uintptr_t)seq | ((uint64_t)len << 48); }

}  // namespace kamd_fq
