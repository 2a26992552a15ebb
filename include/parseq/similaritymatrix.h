// parseq/similaritymatrix.h — drop-in for the reference's src/aligner/similaritymatrix.h:13-100 on top of
// the MI355X engine (include/mi355_sw.h).  Same class names and member signatures:
//   Abstract_Similarity_Matrix, Similarity_Matrix (float32 engine), Similarity_Matrix_Skewed (uint8
//   saturating engine), iterate(), find_index_of_maximum(), operator()(row, col), getTimings().
// Differences, by design:
//   * the matrix is never materialised by iterate(): it runs the GPU score pass + argmax.  operator(),
//     get_matrix(), print_matrix() and print_matrix_raw() fill the full matrix on first use through
//     mi355_sw_fill_matrix and are meant for small problems (std::length_error above 2^28 cells);
//   * Eigen is optional.  When <Eigen/Dense> is on the include path (the reference vendors 3.3.7) the signatures are
//     the reference's: getTimings() returns Eigen::VectorXf, Similarity_Matrix::get_matrix() a
//     const Eigen::MatrixXf &, Similarity_Matrix_Skewed::get_matrix() a const MatrixX8u & in the skewed RAW layout
//     (nrows + 32 pad rows, similaritymatrix.cpp:287).  Without Eigen the same members return parseq::Timings /
//     parseq::DenseMatrix<T> (column-major, (i, j), rows(), cols(), data(), operator<<).  -DPARSEQ_NO_EIGEN forces that;
//   * Index types are std::ptrdiff_t (what Eigen::Index is).
#ifndef PARSEQ_SIMILARITY_MATRIX_H_
#define PARSEQ_SIMILARITY_MATRIX_H_

#if !defined(PARSEQ_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define PARSEQ_HAVE_EIGEN 1
#endif
#endif

#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <tuple>
#include <type_traits>
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
#ifdef PARSEQ_HAVE_EIGEN
  operator Eigen::VectorXf() const { Eigen::VectorXf r(2); r(0) = v[0]; r(1) = v[1]; return r; }
#endif
};

// Column-major dense matrix: what get_matrix() returns when Eigen is not available.
template <class T>
class DenseMatrix {
 public:
  DenseMatrix() = default;
  DenseMatrix(Index rows, Index cols) : nr(rows), nc(cols), buf((size_t)rows * (size_t)cols, T(0)) {}
  void resize(Index rows, Index cols) { nr = rows; nc = cols; buf.assign((size_t)rows * (size_t)cols, T(0)); }
  void setZero() { std::fill(buf.begin(), buf.end(), T(0)); }
  Index rows() const { return nr; }
  Index cols() const { return nc; }
  Index size() const { return nr * nc; }
  T *data() { return buf.data(); }
  const T *data() const { return buf.data(); }
  T &operator()(Index i, Index j) { return buf[(size_t)j * (size_t)nr + (size_t)i]; }
  const T &operator()(Index i, Index j) const { return buf[(size_t)j * (size_t)nr + (size_t)i]; }
  friend std::ostream &operator<<(std::ostream &os, const DenseMatrix &m) {
    for (Index i = 0; i < m.nr; ++i) {
      for (Index j = 0; j < m.nc; ++j) os << (j ? " " : "") << +m(i, j);
      if (i + 1 < m.nr) os << "\n";
    }
    return os;
  }

 private:
  Index nr = 0, nc = 0;
  std::vector<T> buf;
};

#ifdef PARSEQ_HAVE_EIGEN
typedef Eigen::VectorXf TimingsVec;      // localaligner.h:16,27 / similaritymatrix.h:23
typedef Eigen::MatrixXf MatrixF;
typedef Eigen::Matrix<uint8_t, Eigen::Dynamic, Eigen::Dynamic, Eigen::ColMajor> Matrix8u;
#else
typedef Timings TimingsVec;
typedef DenseMatrix<float> MatrixF;
typedef DenseMatrix<uint8_t> Matrix8u;
#endif

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

// Several GPUs for the aligners that split their work (OMPParallelLocalAligner): MI355_SW_DEVICES = "all" or a comma
// list ("0,1,2,3") selects them, MI355_SW_MULTI_RCCL=1 merges the per-device best keys with an RCCL all-reduce.
// Unset: nullptr, and everything runs on context()'s single device.  One handle per host thread.
inline mi355_sw_multi *multi_context() {
  struct Holder {
    mi355_sw_multi *m = nullptr;
    Holder() {
      const char *e = std::getenv("MI355_SW_DEVICES");
      if (!e || !*e) return;
      std::vector<int> devs;
      const std::string s(e);
      if (s != "all") {
        size_t at = 0;
        while (at < s.size()) {
          size_t end = s.find(',', at);
          if (end == std::string::npos) end = s.size();
          if (end > at) devs.push_back(std::atoi(s.substr(at, end - at).c_str()));
          at = end + 1;
        }
      }
      const char *r = std::getenv("MI355_SW_MULTI_RCCL");
      const int flags = (r && *r && *r != '0') ? MI355_SW_MULTI_RCCL : 0;
      if (mi355_sw_multi_create(&m, (int)devs.size(), devs.empty() ? nullptr : devs.data(), flags) != 0) {
        std::fprintf(stderr, "parseq: mi355_sw_multi_create failed for MI355_SW_DEVICES=%s; there is no CPU fallback\n", e);
        std::abort();
      }
    }
    ~Holder() { mi355_sw_multi_destroy(m); }
  };
  static thread_local Holder h;
  return h.m;
}

inline void check(int rc, const char *what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mi355_sw_last_error(context()));
}
inline void check_multi(int rc, const char *what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + mi355_sw_multi_last_error(multi_context()));
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
typedef parseq::Matrix8u MatrixX8u;      // similaritymatrix.h:9

class Abstract_Similarity_Matrix {
 public:
  virtual ~Abstract_Similarity_Matrix() = default;
  virtual void iterate(const std::function<float(const char &, const char &)> &scoring_function, float gap_penalty) = 0;
  virtual std::tuple<parseq::Index, parseq::Index, float> find_index_of_maximum() const = 0;
  virtual void print_matrix() const = 0;
  virtual float operator()(parseq::Index row, parseq::Index col) const = 0;
  virtual parseq::TimingsVec getTimings() const = 0;
};

namespace parseq {

constexpr Index kSkewedPadRows = 32;     // N_PACK pad rows of the skewed storage (similaritymatrix.cpp:287)


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
    filled = false;
    iterated = true;
  }
  std::tuple<Index, Index, float> find_index_of_maximum() const override { return {max_x, max_y, max_v}; }
  float operator()(Index row, Index col) const override {
    fill();
    return cells(row, col);
  }
  // similaritymatrix.cpp:37-39 / :301-311: the (|x|+1) x (|y|+1) matrix, rows = first sequence
  void print_matrix() const override {
    fill();
    if (SEMANTICS == MI355_SW_U8SAT) std::cout << "\n";
    std::cout << cells << std::endl;
  }
  // similaritymatrix.h:44 (Similarity_Matrix: the float matrix itself) / :73 (Similarity_Matrix_Skewed: the RAW skewed
  // uint8 storage, raw(ri, rj) with (ri, rj) = trueindex2rawindex(column of y, row of x), 32 zero pad rows)
  const std::conditional_t<SEMANTICS == MI355_SW_U8SAT, Matrix8u, MatrixF> &get_matrix() const {
    fill();
    if constexpr (SEMANTICS == MI355_SW_U8SAT) { fill_raw(); return raw; }
    else return cells;
  }
  // similaritymatrix.cpp:313-321 (skewed engine only in the reference; here the float engine prints its matrix)
  virtual void print_matrix_raw() const {
    if constexpr (SEMANTICS == MI355_SW_U8SAT) {
      fill(); fill_raw();
      MatrixF m(raw.rows(), raw.cols());
      for (Index j = 0; j < raw.cols(); ++j)
        for (Index i = 0; i < raw.rows(); ++i) m(i, j) = (float)raw(i, j);
      std::cout << "\n" << m << std::endl;
    } else {
      print_matrix();
    }
  }
  TimingsVec getTimings() const override { return timings; }
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
    max_x = ix; max_y = iy; max_v = mx; timings.v[0] = iterate_us; lut = std::move(l); gap = g; filled = false; iterated = true;
  }

 private:
  void fill() const {
    if (filled) return;
    const size_t n = (sequence_x.size() + 1) * (sequence_y.size() + 1);
    if (n > ((size_t)1 << 28)) throw std::length_error("Similarity_Matrix::operator(): matrix too large to materialise");
    cells.resize((Index)sequence_x.size() + 1, (Index)sequence_y.size() + 1);
    cells.setZero();
    raw_filled = false;
    filled = true;
    if (!iterated) return;   // zero-initialised matrix before iterate(), as the reference
    mi355_sw_params p{lut ? lut->data() : nullptr, 3.0f, -3.0f, gap, SEMANTICS};
    // column-major over y with |x|+1 rows: exactly the layout mi355_sw_fill_matrix writes
    check(mi355_sw_fill_matrix(context(), sequence_x.data(), sequence_x.size(), sequence_y.data(), sequence_y.size(), &p, cells.data()),
          "Similarity_Matrix::operator()");
  }
  // the skewed class's raw storage rebuilt from the true cells (similaritymatrix.cpp:274-289, :353-364)
  void fill_raw() const {
    if (raw_filled) return;
    const Index len_x = (Index)sequence_y.size() + 1, len_y = (Index)sequence_x.size() + 1;   // constructor swap
    const Index nrows = len_x < len_y ? len_x : len_y, ncols = len_x < len_y ? len_y : len_x;
    raw.resize(nrows + kSkewedPadRows, ncols);
    raw.setZero();
    for (Index ti = 0; ti < len_x; ++ti)
      for (Index tj = 0; tj < len_y; ++tj) {
        const index_tuple r = trueindex2rawindex(index_tuple(ti, tj));
        raw(r.first, r.second) = (uint8_t)cells(tj, ti);
      }
    raw_filled = true;
  }
  std::string_view sequence_x, sequence_y;
  std::shared_ptr<std::vector<float>> lut;
  float gap = 2.0f;
  bool iterated = false;
  Index max_x = 0, max_y = 0;
  float max_v = 0.0f;
  Timings timings;
  mutable MatrixF cells;                   // (|x|+1) x (|y|+1), column-major
  mutable Matrix8u raw;
  mutable bool filled = false, raw_filled = false;
};

}  // namespace parseq

typedef parseq::Similarity_Matrix_HIP<MI355_SW_F32> Similarity_Matrix;
typedef parseq::Similarity_Matrix_HIP<MI355_SW_U8SAT> Similarity_Matrix_Skewed;

#endif
