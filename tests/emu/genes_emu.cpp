// tests/emu/genes_emu.cpp -- TEST INFRASTRUCTURE: the front-end's gene-level helpers of `quant-tcc -g` (kallisto_amd/csrc/kamd_genes.h) on a
// box without a GPU.  Never linked into the product.
#include "../../kallisto_amd/csrc/kamd_genes.h"

#include <cstring>

extern "C" {
// names: T target names separated by '\n'.  Parses the mapping file, sums alpha / tpm per gene, writes the per-sample gene file and the
// gene name list.  Returns the number of genes, or -1 with the parser's message in err (cap bytes).
int64_t fe_gene_outputs(const char* genemap_path, const char* names, uint64_t T, const double* alpha, const double* tpm, const char* tsv_path,
                        const char* names_path, int32_t* tr_gene_out, char* err, uint64_t cap) {
  std::vector<std::string> nm;
  {
    const char* p = names;
    for (uint64_t i = 0; i < T; i++) { const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p); nm.emplace_back(p, e); p = *e ? e + 1 : e; }
  }
  kamd_fe::GeneMap gm; std::string msg;
  if (!kamd_fe::parse_genemap(genemap_path, nm, &gm, &msg)) { strncpy(err, msg.c_str(), cap - 1); err[cap - 1] = 0; return -1; }
  std::vector<double> a(alpha, alpha + T), t(tpm, tpm + T), gc, gct;
  kamd_fe::gene_sums(gm, a, t, &gc, &gct);
  kamd_fe::write_abundance_gene(tsv_path, gm, gc, gct);
  kamd_fe::write_gene_names(names_path, gm);
  if (tr_gene_out) memcpy(tr_gene_out, gm.tr_gene.data(), T * sizeof(int32_t));
  return (int64_t)gm.name.size();
}
}
