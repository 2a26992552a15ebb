// driver_common.h — file helpers shared by the drivers (formats of the reference's own drivers).
#pragma once
#include <fstream>
#include <string>
#include <vector>

namespace drv {

// FASTA as the reference reads it (src/sw_solve_small.cpp:20-31): drop line 0, concatenate the rest.
inline bool read_fasta_skip_header(const std::string &path, std::string &out) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  int i = 0;
  out.clear();
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (i > 0) out += line;
    i++;
  }
  return true;
}

// Single-line headerless reference (src/sw_solve_big.cpp:30-37).
inline bool read_single_line(const std::string &path, std::string &out) {
  std::ifstream f(path);
  if (!f) return false;
  std::getline(f, out);
  if (!out.empty() && out.back() == '\r') out.pop_back();
  return true;
}

// CSV row split exactly as src/sw_solve_small.cpp:56-67 does it (append ',', cut at every ',').
inline std::vector<std::string> split_row(const std::string &line) {
  std::vector<std::string> row;
  std::string tmp = line + ",";
  size_t pos = 0;
  while ((pos = tmp.find(',')) != std::string::npos) {
    row.push_back(tmp.substr(0, pos));
    tmp.erase(0, pos + 1);
  }
  return row;
}

struct Args {
  std::vector<std::string> pos;
  std::vector<std::pair<std::string, std::string>> opt;
  bool has(const std::string &k) const { for (auto &o : opt) if (o.first == k) return true; return false; }
  std::string get(const std::string &k, const std::string &d) const { for (auto &o : opt) if (o.first == k) return o.second; return d; }
};

inline Args parse(int argc, char **argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    if (s.rfind("--", 0) == 0) {
      const size_t eq = s.find('=');
      if (eq == std::string::npos) a.opt.emplace_back(s.substr(2), "1");
      else a.opt.emplace_back(s.substr(2, eq - 2), s.substr(eq + 1));
    } else a.pos.push_back(s);
  }
  return a;
}

}  // namespace drv
