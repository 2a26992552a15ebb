// sw_solve_big — drop-in for the reference benchmark driver src/sw_solve_big.cpp on the MI355X engine.
//   sw_solve_big <npiece> <nrepeat> [ref.fa] [reads.csv] [--engine=u8|f32]
// Reference = single headerless line (sw_solve_big.cpp:30-37), reads = CSV column 2 (:69); per read the
// aligner is built once and calculateScore() repeated nrepeat times, keeping the minimum getTimings()[0]
// (:82-92); prints the same [INFO] lines (GCUPS overall, mean and std over reads, :99-106).
// npiece == 0 runs the serial configuration SWAligner<Skewed>; otherwise, as the reference's USEOMP build,
// OMPParallelLocalAligner<Skewed,SWAligner<Skewed>> with npiece*2 pieces and overlap 2.0 (:53,:78).
#include <cmath>
#include <iostream>
#include <memory>

#include "driver_common.h"
#include "parseq/localaligner.h"
#include "parseq/plocalaligner.h"
#include "parseq/smithwaterman.h"

template <class SMT>
int run(int npiece, int nrepeat, const std::string &fa_string, const std::string &in_path) {
  std::ifstream align_input(in_path);
  if (!align_input) { std::cerr << "cannot open " << in_path << std::endl; return 2; }
  std::string input_line;
  double time_avg = 0.0, time_iter_avg = 0.0;
  unsigned long long num_cells = 0;
  int i = 0;
  std::vector<double> GCUPS_vec;
  const double overlaprate = 2.0;
  while (std::getline(align_input, input_line)) {
    if (i > 0) {
      const std::vector<std::string> row = drv::split_row(input_line);
      const auto matsize = row.at(2).size() * fa_string.size();
      if (i == 1) {
        std::cout << "[INFO] Estimated Memory consumption of the reference's matrix " << (double)matsize * 1e-9
                  << "GB (not allocated here)" << std::endl;
      }
      double time_min = 9e20, time_iter_min = 9e20;
      if (npiece > 0) {
        auto la = std::make_unique<OMPParallelLocalAligner<SMT, SWAligner<SMT>>>(row[2], fa_string, npiece * 2, overlaprate);
        for (int j = 0; j < nrepeat; j++) {
          la->calculateScore();
          time_min = std::min(time_min, (double)la->getTimings()[0]);
          time_iter_min = std::min(time_iter_min, (double)la->getTimings()[1]);
        }
      } else {
        auto la = std::make_unique<SWAligner<SMT>>(row[2], fa_string);
        for (int j = 0; j < nrepeat; j++) {
          la->calculateScore();
          time_min = std::min(time_min, (double)la->getTimings()[0]);
          time_iter_min = time_min;
        }
      }
      time_avg += time_min;
      time_iter_avg += time_iter_min;
      GCUPS_vec.emplace_back(matsize / time_min * 1e-3);
      num_cells += matsize;
    }
    i++;
  }
  if (GCUPS_vec.empty()) { std::cerr << "no reads" << std::endl; return 2; }
  const double GCUPS = num_cells / time_avg * 1e-3, GCUPS_iter = num_cells / time_iter_avg * 1e-3;
  time_avg /= (i - 1);
  double mean = 0, var = 0;
  for (double g : GCUPS_vec) mean += g;
  mean /= GCUPS_vec.size();
  for (double g : GCUPS_vec) var += (g - mean) * (g - mean);
  std::cout << "[INFO] Average SW iter_ad_read times: " << time_avg * 1e-6 << "s, GCUPS:" << GCUPS
            << ", GCPUS per iteration: " << GCUPS_iter << std::endl;
  std::cout << "[INFO] GCUPS avg:" << mean << ", GCUPS std:" << std::sqrt(var / GCUPS_vec.size()) << std::endl;
  for (double g : GCUPS_vec) std::cout << g << " ";
  std::cout << std::endl;
  return 0;
}

int main(int argc, char **argv) {
  const drv::Args a = drv::parse(argc, argv);
  if (a.pos.size() < 2) {
    std::cout << "Please specify number of pieces to break e.g: `sw_solve_big <npiece> <nrepeat>`" << std::endl;
    return -1;
  }
  const int npiece = std::stoi(a.pos[0]), nrepeat = std::stoi(a.pos[1]);
  std::cout << "[INFO] npiece: " << npiece << ", nrepeat:" << nrepeat << std::endl;
  const std::string fa_file_path = a.pos.size() > 2 ? a.pos[2] : "data/custom_ref_1.fa";
  const std::string input_file_path = a.pos.size() > 3 ? a.pos[3] : "data/custom_reads_1.csv";
  std::string fa_string;
  if (!drv::read_single_line(fa_file_path, fa_string)) { std::cerr << "cannot open " << fa_file_path << std::endl; return 2; }
  if (a.get("engine", "u8") == "f32") return run<Similarity_Matrix>(npiece, nrepeat, fa_string, input_file_path);
  return run<Similarity_Matrix_Skewed>(npiece, nrepeat, fa_string, input_file_path);
}
