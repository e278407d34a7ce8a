// kamd_genes.h -- gene-level abundances of `quant-tcc -g`: the transcript -> gene mapping file, the sums, the two writers.  Host code of
// the front-end only (no GPU in here); a header of its own so that the no-GPU tests can drive it (tests/emu/fq_emu.cpp).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace kamd_fe {

// ---- gene-level abundances of `quant-tcc -g` (src/main.cpp:2975-2980, 3040-3068; GeneModel.cpp:580-632; PlaintextWriter.cpp:67-112, 300-315) ----
// The mapping file: lines "transcript gene [common name]" (white space); genes are numbered in order of first appearance; a transcript
// that no line names belongs to no gene (-1).  Errors are the reference's: a line without a gene, a transcript the index does not hold.
struct GeneMap {
  std::vector<int32_t> tr_gene;                 // per transcript
  std::vector<std::string> name, common;        // per gene
};
inline bool parse_genemap(const std::string& path, const std::vector<std::string>& target_names, GeneMap* gm, std::string* err) {
  std::unordered_map<std::string, int32_t> tr_of, gene_of;
  for (size_t i = 0; i < target_names.size(); i++) tr_of.insert({target_names[i], (int32_t)i});   // (insert: the first of equal names wins)
  gm->tr_gene.assign(target_names.size(), -1);
  gm->name.clear(); gm->common.clear();
  std::ifstream in(path);
  if (!in.is_open()) { *err = "Error: could not open file " + path; return false; }
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    std::string txp, gene, common;
    ss >> txp >> gene >> common;
    if (gene.empty()) { *err = "Error: No gene associated with transcript " + txp + " in " + path; return false; }
    const auto it = tr_of.find(txp);
    if (it == tr_of.end()) { *err = "Error: Invalid transcript: " + txp + " in " + path; return false; }
    auto g = gene_of.find(gene);
    if (g == gene_of.end()) { g = gene_of.insert({gene, (int32_t)gm->name.size()}).first; gm->name.push_back(gene); gm->common.push_back(common); }
    gm->tr_gene[(size_t)it->second] = g->second;
  }
  return true;
}
// sums over the transcripts with a positive count, in transcript order (the order of the reference's additions)
inline void gene_sums(const GeneMap& gm, const std::vector<double>& alpha, const std::vector<double>& tpm, std::vector<double>* gc, std::vector<double>* gc_tpm) {
  gc->assign(gm.name.size(), 0.0); gc_tpm->assign(gm.name.size(), 0.0);
  for (size_t i = 0; i < alpha.size() && i < gm.tr_gene.size(); i++)
    if (alpha[i] > 0.0 && gm.tr_gene[i] != -1) { (*gc)[(size_t)gm.tr_gene[i]] += alpha[i]; (*gc_tpm)[(size_t)gm.tr_gene[i]] += tpm[i]; }
}
inline void write_abundance_gene(const std::string& path, const GeneMap& gm, const std::vector<double>& gc, const std::vector<double>& gc_tpm) {   // plaintext_writer_gene
  std::ofstream of(path);
  if (!of.is_open()) { std::cerr << "Error: Couldn't open file: " << path << std::endl; exit(1); }
  of << "gene_id" << "\t" << "gene_name" << "\t" << "est_counts" << "\t" << "tpm" << "\n";
  for (size_t i = 0; i < gc.size(); i++) of << gm.name[i] << '\t' << gm.common[i] << '\t' << gc[i] << '\t' << gc_tpm[i] << "\n";
}
inline void write_gene_names(const std::string& path, const GeneMap& gm) {   // writeGeneList(..., writeNamesOnly = true)
  std::ofstream of(path);
  if (!of.is_open()) { std::cerr << "Error: Couldn't open file: " << path << std::endl; exit(1); }
  for (const std::string& n : gm.name) of << n << "\n";
}


}  // namespace kamd_fe
