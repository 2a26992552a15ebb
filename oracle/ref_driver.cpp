// ref_driver.cpp — build-owned command driver around the UNMODIFIED reference aligner.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it #includes the reference's public headers
// (src/aligner/{localaligner,smithwaterman,plocalaligner,similaritymatrix}.h) and is linked
// by oracle/build_ref.sh against the reference's own src/aligner/*.cpp compiled where they
// lie under /root/reference.  Outputs go to oracle/_ref/ only.  It exists to (1) validate the
// C restatement in sw_oracle.c, (2) generate tests/golden/*.json, (3) time the reference's
// own OpenMP path as bench.py's cpu_baseline (kind "reference").
//
// Protocol: one command per stdin line, space-separated tokens, sequences last.
//   align  <f32|u8> <match> <mismatch> <gap> <x> <y>
//   alignlut <f32|u8> <seed> <scale> <gap> <x> <y>      (LUT scoring, see make_lut)
//   split  <f32|u8> <f32|u8> <match> <mismatch> <gap> <npiece> <ratio> <x> <y>
//   matrix <f32|u8> <match> <mismatch> <gap> <x> <y>
//   range  <npiece> <shortlen> <longlen> <ratio>
//   true2raw <m> <n> <ti> <tj>      raw2true <m> <n> <ri> <rj>     (m=|x|, n=|y|)
//   bench  <file> <npiece> <nrepeat>                     (file: line 1 ref, then reads)
//   loadref <file>                                       (line 1 of the file = y for the alignref commands that follow)
//   alignref <f32|u8> <match> <mismatch> <gap> <x>       (align against the loaded y: full-size fixtures, 50 Mbp)
//   manyfirst <file> <y>                                 (every line of the file as FIRST argument against y, float engine,
//                                                         default scoring: the loop of src/mpi_sw_solve_uniprot.cpp:95-138;
//                                                         replies one line "score pos" per sequence, then "done <count>")
// Replies: one line per command.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "localaligner.h"
#include "plocalaligner.h"
#include "similaritymatrix.h"
#include "smithwaterman.h"

// defined (non-static) in the reference's plocalaligner.cpp:44
std::vector<std::pair<Eigen::Index, Eigen::Index>> _make_string_range(int, Eigen::Index, Eigen::Index, float);

namespace {

uint64_t splitmix64(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Deterministic 256x256 table shared (by construction, not by code) with tests/sw_testlib.py:
// off-diagonal in [-5,5]*scale, diagonal in [1,6]*scale.
std::shared_ptr<std::vector<float>> make_lut(uint64_t seed, float scale) {
  auto lut = std::make_shared<std::vector<float>>(65536);
  uint64_t s = seed;
  for (int a = 0; a < 256; ++a)
    for (int b = 0; b < 256; ++b) {
      uint64_t r = splitmix64(s);
      float v = (a == b) ? (float)(1 + (int)(r % 6)) : (float)((int)(r % 11) - 5);
      (*lut)[a * 256 + b] = v * scale;
    }
  return lut;
}

template <class LA>
void print_result(LA &la, long ex, long ey) {
  std::string cx(la.getConsensus_x()), cy(la.getConsensus_y());
  if (cx.empty()) cx = "*";
  if (cy.empty()) cy = "*";
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.9g", (double)la.getScore());
  std::cout << buf << " " << la.getPos() << " " << ex << " " << ey << " " << cx << " " << cy << "\n";
}

template <class SMT>
void do_align(const std::string &x, const std::string &y,
              std::function<float(const char &, const char &)> f, float gap) {
  SWAligner<SMT> la(x, y, std::move(f), gap);
  la.calculateScore();
  auto [ix, iy, mx] = la.getSimilarity_matrix().find_index_of_maximum();
  (void)mx;
  print_result(la, (long)ix, (long)iy);
}

template <class SMT, class LAT>
void do_split(const std::string &x, const std::string &y, float ma, float mi, float gap, int npiece,
              float ratio) {
  OMPParallelLocalAligner<SMT, LAT> la(x, y, npiece, ratio,
                                       [ma, mi](const char &a, const char &b) { return a == b ? ma : mi; }, gap);
  la.calculateScore();
  print_result(la, -1, -1);
}

template <class SMT>
void do_matrix(const std::string &x, const std::string &y, float ma, float mi, float gap) {
  SMT sm(x, y);
  sm.iterate([ma, mi](const char &a, const char &b) { return a == b ? ma : mi; }, gap);
  std::ostringstream os;
  for (size_t i = 0; i <= x.size(); ++i)
    for (size_t j = 0; j <= y.size(); ++j) os << sm((Eigen::Index)i, (Eigen::Index)j) << " ";
  std::cout << os.str() << "\n";
}

void do_bench(const std::string &path, int npiece, int nrepeat) {
  std::ifstream f(path);
  std::string ref, line;
  std::getline(f, ref);
  std::vector<std::string> reads;
  while (std::getline(f, line))
    if (!line.empty()) reads.push_back(line);
  double t_sum_us = 0, cells = 0, wall_us = 0;
  for (auto &r : reads) {
    // same construction and min-of-nrepeat timing as src/sw_solve_big.cpp:78-92
    auto t0 = std::chrono::high_resolution_clock::now();
    double tmin = 9e20;
#ifdef USEOMP
    OMPParallelLocalAligner<Similarity_Matrix_Skewed, SWAligner<Similarity_Matrix_Skewed>> la(r, ref, npiece, 2.0);
#else
    (void)npiece;
    SWAligner<Similarity_Matrix_Skewed> la(r, ref);
#endif
    for (int k = 0; k < nrepeat; ++k) {
      la.calculateScore();
      double t = (double)la.getTimings()[0];
      if (t < tmin) tmin = t;
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    wall_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
    t_sum_us += tmin;
    cells += (double)r.size() * (double)ref.size();
  }
  char buf[256];
  std::snprintf(buf, sizeof buf, "gcups_iterate %.6f gcups_wall %.6f reads %zu cells %.0f iterate_us %.1f wall_us %.1f",
                cells / t_sum_us * 1e-3, cells * nrepeat / wall_us * 1e-3, reads.size(), cells, t_sum_us, wall_us);
  std::cout << buf << "\n";
}

}  // namespace

int main() {
  std::ios::sync_with_stdio(false);
  std::string line;
  std::string loaded;                        // y of the alignref commands
  while (std::getline(std::cin, line)) {
    std::istringstream is(line);
    std::string cmd;
    is >> cmd;
    if (cmd == "align") {
      std::string sem, x, y; float ma, mi, gap;
      is >> sem >> ma >> mi >> gap >> x >> y;
      auto f = [ma, mi](const char &a, const char &b) { return a == b ? ma : mi; };
      if (sem == "f32") do_align<Similarity_Matrix>(x, y, f, gap); else do_align<Similarity_Matrix_Skewed>(x, y, f, gap);
    } else if (cmd == "alignlut") {
      std::string sem, x, y; uint64_t seed; float scale, gap;
      is >> sem >> seed >> scale >> gap >> x >> y;
      auto lut = make_lut(seed, scale);
      auto f = [lut](const char &a, const char &b) { return (*lut)[(size_t)(uint8_t)a * 256 + (uint8_t)b]; };
      if (sem == "f32") do_align<Similarity_Matrix>(x, y, f, gap); else do_align<Similarity_Matrix_Skewed>(x, y, f, gap);
    } else if (cmd == "split") {
      std::string s1, s2, x, y; float ma, mi, gap, ratio; int npiece;
      is >> s1 >> s2 >> ma >> mi >> gap >> npiece >> ratio >> x >> y;
      if (s1 == "f32" && s2 == "f32") do_split<Similarity_Matrix, SWAligner<Similarity_Matrix>>(x, y, ma, mi, gap, npiece, ratio);
      else if (s1 == "u8" && s2 == "f32") do_split<Similarity_Matrix_Skewed, SWAligner<Similarity_Matrix>>(x, y, ma, mi, gap, npiece, ratio);
      else if (s1 == "f32" && s2 == "u8") do_split<Similarity_Matrix, SWAligner<Similarity_Matrix_Skewed>>(x, y, ma, mi, gap, npiece, ratio);
      else do_split<Similarity_Matrix_Skewed, SWAligner<Similarity_Matrix_Skewed>>(x, y, ma, mi, gap, npiece, ratio);
    } else if (cmd == "matrix") {
      std::string sem, x, y; float ma, mi, gap;
      is >> sem >> ma >> mi >> gap >> x >> y;
      if (sem == "f32") do_matrix<Similarity_Matrix>(x, y, ma, mi, gap); else do_matrix<Similarity_Matrix_Skewed>(x, y, ma, mi, gap);
    } else if (cmd == "range") {
      int npiece; long s, l; float ratio;
      is >> npiece >> s >> l >> ratio;
      auto v = _make_string_range(npiece, s, l, ratio);
      for (auto &p : v) std::cout << p.first << " " << p.second << " ";
      std::cout << "\n";
    } else if (cmd == "true2raw" || cmd == "raw2true") {
      size_t m, n; long a, b;
      is >> m >> n >> a >> b;
      Similarity_Matrix_Skewed sm(std::string(m, 'A'), std::string(n, 'A'));
      auto r = cmd == "true2raw" ? sm.trueindex2rawindex(index_tuple(a, b)) : sm.rawindex2trueindex(index_tuple(a, b));
      std::cout << r.first << " " << r.second << "\n";
    } else if (cmd == "loadref") {
      std::string path;
      is >> path;
      std::ifstream f(path);
      std::getline(f, loaded);
      std::cout << "loaded " << loaded.size() << "\n";
    } else if (cmd == "alignref") {
      std::string sem, x; float ma, mi, gap;
      is >> sem >> ma >> mi >> gap >> x;
      auto f = [ma, mi](const char &a, const char &b) { return a == b ? ma : mi; };
      if (sem == "f32") do_align<Similarity_Matrix>(x, loaded, f, gap); else do_align<Similarity_Matrix_Skewed>(x, loaded, f, gap);
    } else if (cmd == "manyfirst") {
      std::string path, y, seq;
      is >> path >> y;
      std::ifstream f(path);
      size_t count = 0;
      char buf[64];
      while (std::getline(f, seq)) {
        SWAligner<Similarity_Matrix> la(seq, y);          // (db sequence, query): mpi_sw_solve_uniprot.cpp:120
        la.calculateScore();
        std::snprintf(buf, sizeof buf, "%.9g %u\n", (double)la.getScore(), la.getPos());
        std::cout << buf;
        ++count;
      }
      std::cout << "done " << count << "\n";
    } else if (cmd == "bench") {
      std::string path; int npiece, nrepeat;
      is >> path >> npiece >> nrepeat;
      do_bench(path, npiece, nrepeat);
    } else if (!cmd.empty()) {
      std::cout << "ERR unknown command\n";
    }
    std::cout.flush();
  }
  return 0;
}
