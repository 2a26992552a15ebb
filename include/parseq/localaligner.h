// parseq/localaligner.h — the reference's pure-virtual aligner interfaces (src/aligner/localaligner.h:7-28).
#ifndef PARSEQ_LOCAL_ALIGNER_H_
#define PARSEQ_LOCAL_ALIGNER_H_

#include <string>
#include <string_view>

#include "similaritymatrix.h"

template <class Similarity_Matrix_Type>
class LocalAligner {
 public:
  virtual ~LocalAligner() = default;
  virtual float calculateScore() = 0;
  virtual float getScore() const = 0;
  virtual unsigned int getPos() const = 0;
  virtual std::string_view getConsensus_x() const = 0;
  virtual std::string_view getConsensus_y() const = 0;
  virtual const Similarity_Matrix_Type &getSimilarity_matrix() const = 0;
  virtual parseq::Timings getTimings() const = 0;
};

template <class Similarity_Matrix_Type, class LocalAligner_Type>
class ParallelLocalAligner {
 public:
  virtual ~ParallelLocalAligner() = default;
  virtual float calculateScore() = 0;
  virtual float getScore() const = 0;
  virtual unsigned int getPos() const = 0;
  virtual std::string_view getConsensus_x() const = 0;
  virtual std::string_view getConsensus_y() const = 0;
  virtual parseq::Timings getTimings() const = 0;
};
#endif
