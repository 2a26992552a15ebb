// test_dropin.cpp — what the reference's own gtest files do not cover of include/parseq/*.h: getters before and after
// repeated calculateScore() calls, get_matrix() / print_matrix_raw() / getTimings() as the reference types them, custom
// scoring, OMPParallelLocalAligner as src/sw_solve_small.cpp:82 constructs it, aligners on concurrent host threads.
// (The reference's gtest files themselves are compiled in place: tests/cpp/build_dropin.sh, test_reference_gtests.bin.)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "parseq/localaligner.h"
#include "parseq/plocalaligner.h"
#include "parseq/similaritymatrix.h"
#include "parseq/smithwaterman.h"

#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

int main() {
  {  // SWAligner_Test fixture + Example_small_sequence_alignment + Verify_consensus_strings
    std::string sequence_x = "GGTTGACTA";
    std::string sequence_y = "TGTTACGG";
    auto la = std::make_unique<SWAligner<Similarity_Matrix_Skewed>>(sequence_x, sequence_y);
    EXPECT(la->getScore() == -1 && la->getPos() == 0);
    la->calculateScore();
    EXPECT(la->getScore() == 13);
    EXPECT(la->getPos() == 2);
    EXPECT(la->getConsensus_x() == std::string_view("CAGTTG"));
    EXPECT(la->getConsensus_y() == std::string_view("CA-TTG"));
    auto [ix, iy, mx] = la->getSimilarity_matrix().find_index_of_maximum();
    EXPECT(ix == 7 && iy == 6 && mx == 13);
    EXPECT(la->getSimilarity_matrix()(7, 6) == 13);
    EXPECT(la->getTimings()[0] > 0);
    // repeated calculateScore() (sw_solve_big.cpp:84-88): pos / score unchanged; the consensus is replaced — or, with
    // PARSEQ_APPEND_CONSENSUS in the environment, appended to as the reference does (smithwaterman.cpp:40-78 never clears)
    la->calculateScore();
    EXPECT(la->getScore() == 13 && la->getPos() == 2);
    if (std::getenv("PARSEQ_APPEND_CONSENSUS")) {
      EXPECT(la->getConsensus_x() == std::string_view("CAGTTGCAGTTG"));
      EXPECT(la->getConsensus_y() == std::string_view("CA-TTGCA-TTG"));
      std::printf("consensus appended on repeat\n");
    } else {
      EXPECT(la->getConsensus_x() == std::string_view("CAGTTG") && la->getConsensus_y() == std::string_view("CA-TTG"));
    }
  }
  // (SimilarityMatrix.SkewedMatrixDP / SkewedMatrixIndex and the SWAligner_Test fixture: the reference's own test files are
  //  compiled where they lie against these headers — tests/cpp/build_dropin.sh, test_reference_gtests.bin)
  {  // get_matrix() (similaritymatrix.h:44,73), print_matrix_raw() (:72), getTimings() as the reference types them
    std::string sequence_x = "GGTTGACTA";
    std::string sequence_y = "TGTTACG";
    auto skewed = Similarity_Matrix_Skewed(sequence_x, sequence_y);
    auto normal = Similarity_Matrix(sequence_x, sequence_y);
    auto scoring_function = [](const char &a, const char &b) { return a == b ? 3.0 : -3.0; };
    skewed.iterate(scoring_function, 2.0);
    normal.iterate(scoring_function, 2.0);
    const auto &M = normal.get_matrix();
    const MatrixX8u &R = skewed.get_matrix();
    EXPECT(M.rows() == 10 && M.cols() == 8);
    EXPECT(R.rows() == 8 + 32 && R.cols() == 10);            // nrows + N_PACK pad rows, ncols (similaritymatrix.cpp:287)
    long used = 0;
    for (int j = 0; j < 8; j++)
      for (int i = 0; i < 10; i++) {
        EXPECT(M(i, j) == normal(i, j));
        auto [ri, rj] = skewed.trueindex2rawindex(index_tuple(j, i));
        EXPECT(R(ri, rj) == (uint8_t)M(i, j));
        used += R(ri, rj) != 0;
      }
    long nonzero = 0;
    for (int j = 0; j < R.cols(); j++)
      for (int i = 0; i < R.rows(); i++) nonzero += R(i, j) != 0;
    EXPECT(nonzero == used && used > 10);                     // nothing outside the mapped cells, pad rows are zero
    auto t = skewed.getTimings();
    EXPECT(t.size() == 2 && t(0) > 0);
#ifdef PARSEQ_HAVE_EIGEN
    const Eigen::MatrixXf &E = normal.get_matrix();
    Eigen::VectorXf tv = normal.getTimings();
    Eigen::Index ex, ey;
    EXPECT(E.maxCoeff(&ex, &ey) == std::get<2>(normal.find_index_of_maximum()));   // similaritymatrix.cpp:21-28
    EXPECT(ex == std::get<0>(normal.find_index_of_maximum()) && ey == std::get<1>(normal.find_index_of_maximum()));
    EXPECT(tv.size() == 2);
    std::printf("Eigen signatures: ok\n");
#endif
    skewed.print_matrix_raw();
    normal.print_matrix();
  }
  {  // custom scoring through std::function, float engine (SURVEY App. B probe: 9 / pos 2)
    SWAligner<Similarity_Matrix> la("GGTTGACTA", "TGTTACGG", [](const char &a, const char &b) { return a == b ? 2.0f : -1.0f; }, 1.0f);
    EXPECT(la.calculateScore() == 9 && la.getPos() == 2);
    EXPECT(la.getConsensus_x() == std::string_view("CAGTTG") && la.getConsensus_y() == std::string_view("CA-TTG"));
  }
  {  // OMPParallelLocalAligner as sw_solve_small.cpp:82 constructs it (two equal hits: first piece wins)
    std::string q = "ACGTACGTTG";
    std::string ref = "TTTT" + q + "CCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCC" + q + "GGGGGGGGGGGGGGGGGGGGGGGGGGGGG";
    auto la = std::make_unique<OMPParallelLocalAligner<Similarity_Matrix_Skewed, SWAligner<Similarity_Matrix_Skewed>>>(q, ref, 3, 2.0);
    EXPECT(la->getScore() == -1);
    float s = la->calculateScore();
    EXPECT(s == 30);
    EXPECT(la->getPos() == 5);
    EXPECT(la->getConsensus_x() == std::string_view("GTTGCATGCA"));
    auto r = _make_string_range(4, 10, 100, 2.0f);
    EXPECT(r.size() == 4 && r[1].first == 20 && r[1].second == 60 && r[3].second == 100);
  }
  {  // independent aligner objects on concurrent host threads (how plocalaligner.cpp:110-115 runs its pieces):
     // every thread gets its own engine context; results must equal the serial ones
    std::string ref;
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int k = 0; k < 60000; ++k) ref.push_back("ACGT"[rnd() & 3]);
    std::vector<std::string> reads;
    for (int k = 0; k < 24; ++k) {
      std::string r = ref.substr(rnd() % (ref.size() - 200), 100 + rnd() % 100);
      r[r.size() / 2] = 'A';
      reads.push_back(r);
    }
    struct Out { float score; unsigned pos; std::string cx, cy; };
    auto run = [&](size_t k) {
      SWAligner<Similarity_Matrix> la(reads[k], ref);
      la.calculateScore();
      return Out{la.getScore(), la.getPos(), std::string(la.getConsensus_x()), std::string(la.getConsensus_y())};
    };
    std::vector<Out> serial;
    for (size_t k = 0; k < reads.size(); ++k) serial.push_back(run(k));
    std::vector<Out> par(reads.size());
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
      th.emplace_back([&, t]() { for (size_t k = t; k < reads.size(); k += 4) par[k] = run(k); });
    for (auto &x : th) x.join();
    for (size_t k = 0; k < reads.size(); ++k) {
      EXPECT(serial[k].score > 200);
      EXPECT(par[k].score == serial[k].score && par[k].pos == serial[k].pos);
      EXPECT(par[k].cx == serial[k].cx && par[k].cy == serial[k].cy);
    }
  }
  std::printf("ALL OK\n");
  return 0;
}
