// parseq/smithwaterman.h — drop-in for SWAligner<SMT> (reference src/aligner/smithwaterman.h:11-58,
// smithwaterman.cpp:6-108) on the MI355X engine.  Same four constructors, defaults (match 3 / mismatch -3 /
// gap 2, smithwaterman.cpp:8), getters and ownership: the aligner keeps string_views, the CALLER keeps both
// sequences alive; consensus strings are owned by the aligner, reversed, '-' for gaps.
// calculateScore() = one mi355_sw_align call (score pass -> argmax -> greedy traceback on the GPU).
// Repeated calls REPLACE the consensus.  The reference APPENDS to it (traceback pushes onto strings that calculateScore
// never clears, smithwaterman.cpp:40-78, :80-108 — an artefact its drivers never read back: sw_solve_big.cpp:84-88 only reads
// pos / score / timings after repeats); a caller that depends on it gets the reference's behaviour with
// PARSEQ_APPEND_CONSENSUS set in the environment (read once per process).
#ifndef PARSEQ_SMITHWATERMAN_H_
#define PARSEQ_SMITHWATERMAN_H_

#include <cstdlib>
#include <functional>
#include <string>
#include <string_view>

#include "localaligner.h"

template <class Similarity_Matrix_Type>
class SWAligner : public LocalAligner<Similarity_Matrix_Type> {
 public:
  typedef Similarity_Matrix_Type matrix_type;
  SWAligner(std::string_view first_sequence, std::string_view second_sequence)
      : SWAligner(first_sequence, second_sequence, 2.0f) {}
  SWAligner(std::string_view first_sequence, std::string_view second_sequence, float gap_penalty)
      : pos(0), max_score(-1), gap_penalty(gap_penalty), sequence_x(first_sequence), sequence_y(second_sequence),
        similarity_matrix(first_sequence, second_sequence) {}
  SWAligner(std::string_view first_sequence, std::string_view second_sequence,
            std::function<float(const char &, const char &)> &&scoring_function)
      : SWAligner(first_sequence, second_sequence, std::move(scoring_function), 2.0f) {}
  SWAligner(std::string_view first_sequence, std::string_view second_sequence,
            std::function<float(const char &, const char &)> &&scoring_function, float gap_penalty)
      : pos(0), max_score(-1), gap_penalty(gap_penalty), sequence_x(first_sequence), sequence_y(second_sequence),
        similarity_matrix(first_sequence, second_sequence), lut(parseq::tabulate(scoring_function)) {}

  float calculateScore() override {
    mi355_sw_params p{lut ? lut->data() : nullptr, 3.0f, -3.0f, gap_penalty, Similarity_Matrix_Type::semantics};
    mi355_sw_result r;
    parseq::check(mi355_sw_align(parseq::context(), sequence_x.data(), sequence_x.size(), sequence_y.data(),
                                 sequence_y.size(), &p, &r), "SWAligner::calculateScore");
    max_score = r.score;
    pos = r.pos;
    static const bool append = std::getenv("PARSEQ_APPEND_CONSENSUS") != nullptr;
    if (!append) { consensus_x.clear(); consensus_y.clear(); }
    consensus_x.append(r.cons_x, r.cons_len);
    consensus_y.append(r.cons_y, r.cons_len);
    sm_timings.v[0] = r.timings_us[0];
    sm_timings.v[1] = r.timings_us[1];
    auto l = lut;
    if (!l) l = parseq::tabulate([](const char &a, const char &b) { return a == b ? 3.0f : -3.0f; });
    similarity_matrix.set_result((parseq::Index)r.end_x, (parseq::Index)r.end_y, r.score, r.timings_us[0], l, gap_penalty);
    mi355_sw_free_result(&r);
    return max_score;
  }
  float getScore() const override { return max_score; }
  unsigned int getPos() const override { return pos; }
  std::string_view getConsensus_x() const override { return consensus_x; }
  std::string_view getConsensus_y() const override { return consensus_y; }
  const Similarity_Matrix_Type &getSimilarity_matrix() const override { return similarity_matrix; }
  parseq::TimingsVec getTimings() const override { return sm_timings; }

 private:
  parseq::Timings sm_timings;
  unsigned int pos;
  float max_score;
  float gap_penalty;
  std::string_view sequence_x;
  std::string_view sequence_y;
  std::string consensus_x;
  std::string consensus_y;
  Similarity_Matrix_Type similarity_matrix;
  std::shared_ptr<std::vector<float>> lut;
};

#endif
