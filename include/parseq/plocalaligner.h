// parseq/plocalaligner.h — drop-in for OMPParallelLocalAligner<SMT,LAT> (reference
// src/aligner/plocalaligner.h:6-33, plocalaligner.cpp:44-143), SERIAL semantics (the reference's OpenMP
// build is racy): pieces from _make_string_range, first strictly greater piece maximum wins,
// the winner is re-aligned by LAT with DEFAULT scoring (plocalaligner.cpp:135), pos += left.
// On the GPU all pieces are swept by one score-kernel launch (grid.y = pieces); with MI355_SW_DEVICES set, piece p is
// swept by device p mod ndev (mi355_sw_multi_align_split) — the devices take the place of the OpenMP threads of
// plocalaligner.cpp:110-115.
#ifndef PARSEQ_PLOCALALIGNER_H_
#define PARSEQ_PLOCALALIGNER_H_

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "localaligner.h"
#include "smithwaterman.h"

inline std::vector<std::pair<parseq::Index, parseq::Index>> _make_string_range(int npiece, parseq::Index shortstringlength,
                                                                            parseq::Index longstringlength,
                                                                            float overlap_ratio) {
  std::vector<int64_t> l((size_t)(npiece > 0 ? npiece : 1)), r(l.size());
  if (mi355_sw_make_string_range(npiece, shortstringlength, longstringlength, overlap_ratio, l.data(), r.data()) != 0) {
    // the reference asserts here (plocalaligner.cpp:52,63,65; asserts are live in its release build)
    std::fprintf(stderr, "_make_string_range: Assertion `overlaplength <= piecelength && right < longstringlength' failed.\n");
    std::abort();
  }
  std::vector<std::pair<parseq::Index, parseq::Index>> out;
  for (int k = 0; k < npiece; ++k) out.emplace_back((parseq::Index)l[k], (parseq::Index)r[k]);
  return out;
}

template <class Similarity_Matrix_Type, class LocalAligner_Type>
class OMPParallelLocalAligner : public ParallelLocalAligner<Similarity_Matrix_Type, LocalAligner_Type> {
 public:
  OMPParallelLocalAligner(std::string_view first, std::string_view second, int npiece, float overlap_ratio)
      : OMPParallelLocalAligner(first, second, npiece, overlap_ratio, 2.0f) {}
  OMPParallelLocalAligner(std::string_view first, std::string_view second, int npiece, float overlap_ratio, float gap_penalty)
      : pos(0), max_score(-1), gap_penalty(gap_penalty), overlap_ratio(overlap_ratio), npiece(npiece), sequence_x(first),
        sequence_y(second), string_ranges(_make_string_range(npiece, first.size(), second.size(), overlap_ratio)) {}
  OMPParallelLocalAligner(std::string_view first, std::string_view second, int npiece, float overlap_ratio,
                          std::function<float(const char &, const char &)> &&scoring_function)
      : OMPParallelLocalAligner(first, second, npiece, overlap_ratio, std::move(scoring_function), 2.0f) {}
  OMPParallelLocalAligner(std::string_view first, std::string_view second, int npiece, float overlap_ratio,
                          std::function<float(const char &, const char &)> &&scoring_function, float gap_penalty)
      : pos(0), max_score(-1), gap_penalty(gap_penalty), overlap_ratio(overlap_ratio), npiece(npiece), sequence_x(first),
        sequence_y(second), lut(parseq::tabulate(scoring_function)),
        string_ranges(_make_string_range(npiece, first.size(), second.size(), overlap_ratio)) {}

  float calculateScore() override {
    mi355_sw_params p{lut ? lut->data() : nullptr, 3.0f, -3.0f, gap_penalty, Similarity_Matrix_Type::semantics};
    mi355_sw_result r;
    int piece = 0;
    if (mi355_sw_multi *multi = parseq::multi_context())     // MI355_SW_DEVICES: piece p on device p mod ndev
      parseq::check_multi(mi355_sw_multi_align_split(multi, sequence_x.data(), sequence_x.size(), sequence_y.data(),
                                                     sequence_y.size(), &p, Similarity_Matrix_Type::semantics,
                                                     LocalAligner_Type::matrix_type::semantics, npiece, overlap_ratio, &r, &piece),
                          "OMPParallelLocalAligner::calculateScore");
    else
      parseq::check(mi355_sw_align_split(parseq::context(), sequence_x.data(), sequence_x.size(), sequence_y.data(),
                                         sequence_y.size(), &p, Similarity_Matrix_Type::semantics,
                                         LocalAligner_Type::matrix_type::semantics, npiece, overlap_ratio, &r, &piece),
                    "OMPParallelLocalAligner::calculateScore");
    max_score = r.score;
    pos = r.pos;
    consensus_x.assign(r.cons_x, r.cons_len);
    consensus_y.assign(r.cons_y, r.cons_len);
    sm_timings.v[0] = r.timings_us[0];
    sm_timings.v[1] = r.timings_us[1];
    winning_piece = piece;
    mi355_sw_free_result(&r);
    return max_score;
  }
  float getScore() const override { return max_score; }
  unsigned int getPos() const override { return pos; }
  std::string_view getConsensus_x() const override { return consensus_x; }
  std::string_view getConsensus_y() const override { return consensus_y; }
  parseq::TimingsVec getTimings() const override { return sm_timings; }
  int getWinningPiece() const { return winning_piece; }

 private:
  parseq::Timings sm_timings;
  unsigned int pos;
  float max_score;
  float gap_penalty;
  float overlap_ratio;
  int npiece;
  int winning_piece = 0;
  std::string consensus_x;
  std::string consensus_y;
  std::string_view sequence_x;
  std::string_view sequence_y;
  std::shared_ptr<std::vector<float>> lut;
  std::vector<std::pair<parseq::Index, parseq::Index>> string_ranges;
};

#endif
