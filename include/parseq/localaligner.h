// parseq/localaligner.h — the two abstract aligner interfaces of the reference
// (src/aligner/localaligner.h:7-17 `LocalAligner`, :19-28 `ParallelLocalAligner`), expressed once:
// both interfaces share the same five accessors plus calculateScore(); LocalAligner adds access to its
// similarity matrix.  Drivers written against the reference's header compile unchanged.
#ifndef PARSEQ_LOCAL_ALIGNER_H_
#define PARSEQ_LOCAL_ALIGNER_H_

#include <string>
#include <string_view>

#include "similaritymatrix.h"

namespace parseq {

// What every aligner reports after calculateScore(): the maximum cell value, the 1-based position in the
// second sequence where the greedy traceback stopped, both consensus strings (reversed, '-' for gaps, owned
// by the aligner) and the device timings.
struct AlignerInterface {
  virtual ~AlignerInterface() = default;
  virtual float calculateScore() = 0;
  virtual float getScore() const = 0;
  virtual unsigned int getPos() const = 0;
  virtual std::string_view getConsensus_x() const = 0;
  virtual std::string_view getConsensus_y() const = 0;
  virtual TimingsVec getTimings() const = 0;
};

}  // namespace parseq

template <class Similarity_Matrix_Type>
class LocalAligner : public parseq::AlignerInterface {
 public:
  virtual const Similarity_Matrix_Type &getSimilarity_matrix() const = 0;
};

template <class Similarity_Matrix_Type, class LocalAligner_Type>
class ParallelLocalAligner : public parseq::AlignerInterface {};

#endif
