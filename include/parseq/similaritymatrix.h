// parseq/similaritymatrix.h — drop-in for the reference's src/aligner/similaritymatrix.h:13-100 on top of
// the MI355X engine (include/mi355_sw.h).  Same class names and member signatures:
//   Abstract_Similarity_Matrix, Similarity_Matrix (float32 engine), Similarity_Matrix_Skewed (uint8
//   saturating engine), iterate(), find_index_of_maximum(), operator()(row, col), getTimings().
// Differences, by design:
//   * the matrix is never materialised by iterate(): it runs the GPU score pass + argmax.  operator()
//     fills the full matrix on first use through mi355_sw_fill_matrix and is meant for small problems
//     (throws std::length_error above 2^28 cells);
//   * getTimings() returns parseq::Timings (two floats, indexable with [] and ()) instead of
//     Eigen::VectorXf; when <Eigen/Dense> was included first it converts implicitly to VectorXf;
//   * Index types are std::ptrdiff_t (what Eigen::Index is).
#ifndef PARSEQ_SIMILARITY_MATRIX_H_
#define PARSEQ_SIMILARITY_MATRIX_H_

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <tuple>
#include <utility>
#include <vector>

#include "../mi355_sw.h"

namespace parseq {

typedef std::ptrdiff_t Index;

struct Timings {
  float v[2] = {0.0f, 0.0f};
  float operator[](int i) const { return v[i]; }
  float &operator[](int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
  float sum() const { return v[0] + v[1]; }
  int size() const { return 2; }
#ifdef EIGEN_CORE_H
  operator Eigen::VectorXf() const { Eigen::VectorXf r(2); r(0) = v[0]; r(1) = v[1]; return r; }
#endif
};

// One engine context per host thread (contexts are not thread-safe; independent aligner objects may
// run concurrently from different threads, as in the reference).  Device from MI355_SW_DEVICE (default 0).
inline mi355_sw_ctx *context() {
  struct Holder {
    mi355_sw_ctx *c = nullptr;
    Holder() {
      const char *e = std::getenv("MI355_SW_DEVICE");
      const int dev = e ? std::atoi(e) : 0;
      if (mi355_sw_create(&c, dev) != 0) {
        std::fprintf(stderr, "parseq: no usable MI355X device %d (mi355_sw_create failed); there is no CPU fallback\n", dev);
        std::abort();
      }
    }
    ~Holder() { mi355_sw_destroy(c); }
  };
  static thread_local Holder h;
  return h.c;
}

inline void check(int rc, const char *what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mi355_sw_last_error(context()));
}

typedef std::function<float(const char &, const char &)> scoring_fn;

// Tabulate a std::function scoring into the 256x256 table the device consumes.
inline std::shared_ptr<std::vector<float>> tabulate(const scoring_fn &f) {
  auto lut = std::make_shared<std::vector<float>>(65536);
  for (int a = 0; a < 256; ++a)
    for (int b = 0; b < 256; ++b) (*lut)[(size_t)a * 256 + b] = f((char)a, (char)b);
  return lut;
}

}  // namespace parseq

typedef std::pair<parseq::Index, parseq::Index> index_tuple;

class Abstract_Similarity_Matrix {
 public:
  virtual ~Abstract_Similarity_Matrix() = default;
  virtual void iterate(const std::function<float(const char &, const char &)> &scoring_function, float gap_penalty) = 0;
  virtual std::tuple<parseq::Index, parseq::Index, float> find_index_of_maximum() const = 0;
  virtual void print_matrix() const = 0;
  virtual float operator()(parseq::Index row, parseq::Index col) const = 0;
  virtual parseq::Timings getTimings() const = 0;
};

namespace parseq {

template <int SEMANTICS>
class Similarity_Matrix_HIP : public Abstract_Similarity_Matrix {
 public:
  static constexpr int semantics = SEMANTICS;
  Similarity_Matrix_HIP(std::string_view sequence_x, std::string_view sequence_y)
      : sequence_x(sequence_x), sequence_y(sequence_y) {}

  void iterate(const std::function<float(const char &, const char &)> &scoring_function, float gap_penalty) override {
    lut = tabulate(scoring_function);
    gap = gap_penalty;
    mi355_sw_params p{lut->data(), 3.0f, -3.0f, gap, SEMANTICS};
    int64_t ix = 0, iy = 0;
    float mx = 0;
    check(mi355_sw_argmax(context(), sequence_x.data(), sequence_x.size(), sequence_y.data(), sequence_y.size(), &p, &ix, &iy, &mx),
          "Similarity_Matrix::iterate");
    double t[6];
    mi355_sw_last_timings(context(), t);
    timings.v[0] = (float)t[3];
    max_x = ix; max_y = iy; max_v = mx;
    cells.clear();
    iterated = true;
  }
  std::tuple<Index, Index, float> find_index_of_maximum() const override { return {max_x, max_y, max_v}; }
  float operator()(Index row, Index col) const override {
    fill();
    return cells[(size_t)col * (sequence_x.size() + 1) + (size_t)row];
  }
  void print_matrix() const override {
    fill();
    for (size_t i = 0; i <= sequence_x.size(); ++i) {
      for (size_t j = 0; j <= sequence_y.size(); ++j) std::cout << (*this)((Index)i, (Index)j) << " ";
      std::cout << "\n";
    }
  }
  Timings getTimings() const override { return timings; }
  // skewed index maps (similaritymatrix.cpp:330-369); meaningful for the uint8 engine's storage order
  index_tuple rawindex2trueindex(index_tuple raw_index) const {
    size_t a, b;
    mi355_sw_raw2true(sequence_x.size(), sequence_y.size(), (size_t)raw_index.first, (size_t)raw_index.second, &a, &b);
    return index_tuple((Index)a, (Index)b);
  }
  index_tuple trueindex2rawindex(index_tuple true_index) const {
    size_t a, b;
    mi355_sw_true2raw(sequence_x.size(), sequence_y.size(), (size_t)true_index.first, (size_t)true_index.second, &a, &b);
    return index_tuple((Index)a, (Index)b);
  }
  // used by SWAligner to publish what its own device call already computed
  void set_result(Index ix, Index iy, float mx, float iterate_us, std::shared_ptr<std::vector<float>> l, float g) {
    max_x = ix; max_y = iy; max_v = mx; timings.v[0] = iterate_us; lut = std::move(l); gap = g; cells.clear(); iterated = true;
  }

 private:
  void fill() const {
    if (!cells.empty()) return;
    const size_t n = (sequence_x.size() + 1) * (sequence_y.size() + 1);
    if (n > ((size_t)1 << 28)) throw std::length_error("Similarity_Matrix::operator(): matrix too large to materialise");
    cells.assign(n, 0.0f);
    if (!iterated) return;   // zero-initialised matrix before iterate(), as the reference
    mi355_sw_params p{lut ? lut->data() : nullptr, 3.0f, -3.0f, gap, SEMANTICS};
    check(mi355_sw_fill_matrix(context(), sequence_x.data(), sequence_x.size(), sequence_y.data(), sequence_y.size(), &p, cells.data()),
          "Similarity_Matrix::operator()");
  }
  std::string_view sequence_x, sequence_y;
  std::shared_ptr<std::vector<float>> lut;
  float gap = 2.0f;
  bool iterated = false;
  Index max_x = 0, max_y = 0;
  float max_v = 0.0f;
  Timings timings;
  mutable std::vector<float> cells;
};

}  // namespace parseq

typedef parseq::Similarity_Matrix_HIP<MI355_SW_F32> Similarity_Matrix;
typedef parseq::Similarity_Matrix_HIP<MI355_SW_U8SAT> Similarity_Matrix_Skewed;

#endif
