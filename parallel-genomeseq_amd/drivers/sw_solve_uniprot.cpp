// sw_solve_uniprot — the many-alignment batch of the reference driver src/mpi_sw_solve_uniprot.cpp
// (each database sequence as FIRST argument, the query protein as SECOND, SWAligner<Similarity_Matrix>,
// default scoring, :120-122) as ONE device batch instead of an MPI task farm with a writer rank.
//   sw_solve_uniprot [query.fasta] [db] [out.csv] [--count=N] [--devices=all|0,1,...] [--rccl]
// --devices (or MI355_SW_DEVICES): the database sequences are dealt to several GPUs of the node (query replicated, no
// exchange during compute) in place of the reference's MPI worker ranks (:95-138); --rccl (or MI355_SW_MULTI_RCCL=1)
// merges the per-device best (score, index) with ncclAllReduce(ncclMax, ncclUint64) instead of on the host.
// db = directory with <k>.fasta files (the reference's layout, data/uniprot/<k>.fasta, count from
// stats.txt or --count) or a single multi-FASTA file.  Output as the writer rank's (:143-170):
// header `read,pos_pred,score`, rows `<first 126 chars of the sequence>, <pos>, <score>`.
#include <cstdio>
#include <iostream>
#include <sys/stat.h>

#include "driver_common.h"
#include "parseq/similaritymatrix.h"

int main(int argc, char **argv) {
  const drv::Args a = drv::parse(argc, argv);
  const std::string fa_file_path = a.pos.size() > 0 ? a.pos[0] : "data/query/P02232.fasta";
  const std::string db_path = a.pos.size() > 1 ? a.pos[1] : "data/uniprot/";
  const std::string output_file_path = a.pos.size() > 2 ? a.pos[2] : "data/align_output.csv";
  std::string fa_string;
  if (!drv::read_fasta_skip_header(fa_file_path, fa_string)) { std::cerr << "cannot open " << fa_file_path << std::endl; return 2; }
  std::vector<std::string> seqs;
  struct stat st;
  if (stat(db_path.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) {
    int filecnt = 0;
    if (a.has("count")) filecnt = std::stoi(a.get("count", "0"));
    else { std::ifstream stats(db_path + "/stats.txt"); stats >> filecnt; }
    for (int k = 0; k < filecnt; ++k) {
      std::string s;
      if (!drv::read_fasta_skip_header(db_path + "/" + std::to_string(k) + ".fasta", s)) break;
      seqs.push_back(s);
    }
  } else {
    std::ifstream f(db_path);
    if (!f) { std::cerr << "cannot open " << db_path << std::endl; return 2; }
    std::string line, cur;
    bool any = false;
    while (std::getline(f, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (!line.empty() && line[0] == '>') { if (any) seqs.push_back(cur); cur.clear(); any = true; }
      else cur += line;
    }
    if (any) seqs.push_back(cur);
  }
  std::cout << seqs.size() << " sequences against a " << fa_string.size() << "-residue query" << std::endl;
  if (a.has("devices")) setenv("MI355_SW_DEVICES", a.get("devices", "all").c_str(), 1);
  if (a.has("rccl")) setenv("MI355_SW_MULTI_RCCL", "1", 1);
  mi355_sw_multi *multi = parseq::multi_context();
  mi355_sw_ctx *ctx = multi ? nullptr : parseq::context();
  if (multi) parseq::check_multi(mi355_sw_multi_set_reference(multi, fa_string.data(), fa_string.size()), "set_reference");
  else parseq::check(mi355_sw_set_reference(ctx, fa_string.data(), fa_string.size()), "set_reference");
  double cells = 0;
  for (const std::string &sq : seqs) cells += (double)sq.size() * fa_string.size();
  mi355_sw_params p;
  mi355_sw_default_params(&p);
  int64_t best_index = -1;
  double t[6];
  std::vector<float> score(seqs.size());
  std::vector<uint32_t> pos(seqs.size());
  if (multi) {
    std::vector<const char *> xs(seqs.size());
    std::vector<size_t> nxs(seqs.size());
    for (size_t k = 0; k < seqs.size(); ++k) { xs[k] = seqs[k].data(); nxs[k] = seqs[k].size(); }
    std::vector<mi355_sw_result> res(seqs.size());
    parseq::check_multi(mi355_sw_multi_align_batch(multi, seqs.size(), xs.data(), nxs.data(), &p, 0, res.data(), &best_index), "align_batch");
    mi355_sw_multi_last_timings(multi, t);
    std::cout << mi355_sw_multi_device_count(multi) << " devices";
    if (mi355_sw_multi_rccl_version(multi)) std::cout << ", RCCL " << mi355_sw_multi_rccl_version(multi);
    std::cout << std::endl;
    for (size_t k = 0; k < seqs.size(); ++k) { score[k] = res[k].score; pos[k] = res[k].pos; }
    mi355_sw_free_results(res.data(), res.size());
  } else {
    // one device: the database as ONE buffer + offsets (what the concatenated FASTA lines are), results as a struct of arrays
    // in library memory — no pointer, no allocation and no string copy per sequence (the CSV needs pos and score only)
    std::string all;
    std::vector<int64_t> offs(seqs.size() + 1, 0);
    size_t total = 0;
    for (const std::string &sq : seqs) total += sq.size();
    all.reserve(total);
    for (size_t k = 0; k < seqs.size(); ++k) { all += seqs[k]; offs[k + 1] = (int64_t)all.size(); }
    parseq::check(mi355_sw_batch_upload_packed(ctx, seqs.size(), all.data(), offs.data()), "batch_upload_packed");
    mi355_sw_batch_view v;
    parseq::check(mi355_sw_batch_run_view(ctx, &p, 0, &v), "batch_run_view");
    mi355_sw_last_timings(ctx, t);
    for (size_t k = 0; k < seqs.size(); ++k) { score[k] = v.score[k]; pos[k] = v.pos[k]; }
  }
  std::ofstream out(output_file_path);
  out << "read,pos_pred,score\n";
  size_t best = 0;
  for (size_t k = 0; k < seqs.size(); ++k) {
    char buff[127];
    std::snprintf(buff, sizeof buff, "%.126s", seqs[k].c_str());
    out << buff << ", " << pos[k] << ", " << score[k] << "\n";
    if (score[k] > score[best]) best = k;
  }
  if (multi && best_index >= 0 && (size_t)best_index != best) { std::cerr << "internal: best index mismatch" << std::endl; return 3; }
  if (!seqs.empty())
    std::cout << "best: sequence " << best << " score " << score[best] << " pos " << pos[best] << std::endl;
  std::cout << "[INFO] device time " << t[3] * 1e-6 << "s, GCUPS:" << cells / t[3] * 1e-3 << std::endl;
  return 0;
}
