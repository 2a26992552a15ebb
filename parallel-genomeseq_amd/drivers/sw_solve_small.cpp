// sw_solve_small — drop-in for the reference driver src/sw_solve_small.cpp on the MI355X engine.
// Same inputs (FASTA reference with one header line; CSV index,QNAME,SEQ,POS), same output rows
// `<input_line>, <pos_pred>, <score>` under a header `<input header>,pos_pred,score`, same closing
// "[...] GCUP" line (GCUPS = cells / DP-fill device time, sw_solve_small.cpp:88-89,102).
//
//   sw_solve_small [fa] [csv_in] [csv_out] [--engine=u8|f32] [--npiece=N --overlap=R] [--one-by-one]
//
// Defaults are the reference's hard-coded paths and its serial configuration (SWAligner<Skewed>);
// --npiece=17 --overlap=2.0 is its USEOMP configuration (sw_solve_small.cpp:82).
// Default mode batches all reads into one device call; --one-by-one runs the reference's loop verbatim
// through the mirrored classes (include/parseq).
#include <cstdio>
#include <iostream>
#include <memory>

#include "driver_common.h"
#include "parseq/localaligner.h"
#include "parseq/plocalaligner.h"
#include "parseq/smithwaterman.h"

template <class SMT>
int run(const drv::Args &a, const std::string &fa_string, const std::string &in_path, const std::string &out_path) {
  std::ifstream align_input(in_path);
  if (!align_input) { std::cerr << "cannot open " << in_path << std::endl; return 2; }
  std::ofstream align_output(out_path);
  const int npiece = std::stoi(a.get("npiece", "0"));
  const float overlap = std::stof(a.get("overlap", "2.0"));
  std::vector<std::string> lines, seqs;
  std::string input_line, header;
  int i = 0;
  while (std::getline(align_input, input_line)) {
    if (i == 0) header = input_line;
    else { lines.push_back(input_line); seqs.push_back(drv::split_row(input_line).at(2)); }
    i++;
  }
  align_output << header << ",pos_pred,score\n";
  double time_us = 0.0;
  unsigned long long num_cells = 0;
  std::vector<unsigned int> pos(seqs.size());
  std::vector<float> score(seqs.size());
  if (a.has("one-by-one") || npiece > 0) {
    for (size_t k = 0; k < seqs.size(); ++k) {
      if (npiece > 0) {
        auto la = std::make_unique<OMPParallelLocalAligner<SMT, SWAligner<SMT>>>(seqs[k], fa_string, npiece, overlap);
        score[k] = la->calculateScore(); pos[k] = la->getPos(); time_us += la->getTimings()[0];
      } else {
        auto la = std::make_unique<SWAligner<SMT>>(seqs[k], fa_string);
        score[k] = la->calculateScore(); pos[k] = la->getPos(); time_us += la->getTimings()[0];
      }
      num_cells += (unsigned long long)seqs[k].size() * fa_string.size();
      if ((k + 1) % 50 == 0) std::cout << "progress: " << k + 1 << std::endl;
    }
  } else {
    mi355_sw_ctx *ctx = parseq::context();
    parseq::check(mi355_sw_set_reference(ctx, fa_string.data(), fa_string.size()), "set_reference");
    std::vector<const char *> xs(seqs.size());
    std::vector<size_t> nxs(seqs.size());
    for (size_t k = 0; k < seqs.size(); ++k) { xs[k] = seqs[k].data(); nxs[k] = seqs[k].size(); num_cells += (unsigned long long)nxs[k] * fa_string.size(); }
    std::vector<mi355_sw_result> res(seqs.size());
    mi355_sw_params p;
    mi355_sw_default_params(&p);
    p.semantics = SMT::semantics;
    parseq::check(mi355_sw_align_batch(ctx, seqs.size(), xs.data(), nxs.data(), &p, 0, res.data()), "align_batch");
    double t[6];
    mi355_sw_last_timings(ctx, t);
    time_us = t[0] > 0 ? t[0] : t[3];
    for (size_t k = 0; k < seqs.size(); ++k) { pos[k] = res[k].pos; score[k] = res[k].score; }
    mi355_sw_free_results(res.data(), res.size());
  }
  for (size_t k = 0; k < seqs.size(); ++k) align_output << lines[k] << ", " << pos[k] << ", " << score[k] << "\n";
  const double GCUPs = num_cells / time_us * 1e-3;
  std::cout << "Average SW iter_ad_read times: " << time_us / (double)(seqs.size() + 1) << "us, GCUP:" << GCUPs << std::endl;
  std::cout << "Done, output file see: " << out_path << std::endl;
  return 0;
}

int main(int argc, char **argv) {
  const drv::Args a = drv::parse(argc, argv);
  const std::string fa_file_path = a.pos.size() > 0 ? a.pos[0] : "data/data_small/genome.chr22.5K.fa";
  const std::string input_file_path = a.pos.size() > 1 ? a.pos[1] : "data/data_small_ground_truth.csv";
  const std::string output_file_path = a.pos.size() > 2 ? a.pos[2] : "data/align_output.csv";
  std::cout << "Hello sw_solve_small" << std::endl;
  std::string fa_string;
  if (!drv::read_fasta_skip_header(fa_file_path, fa_string)) { std::cerr << "cannot open " << fa_file_path << std::endl; return 2; }
  if (a.get("engine", "u8") == "f32") return run<Similarity_Matrix>(a, fa_string, input_file_path, output_file_path);
  return run<Similarity_Matrix_Skewed>(a, fa_string, input_file_path, output_file_path);
}
