// sw_exact_kernel.h — "exact" anti-diagonal engine: one wavefront per (sub-)problem, full
// reference semantics including the argmax order and the traceback decisions.
//
// Used for:
//   * the tile(s) that hold the maximum after the score pass  -> argmax cell in the reference's
//     storage order (similaritymatrix.cpp:21-28 F32, :291-299 + :353-364 U8SAT);
//   * the traceback window around the argmax -> per-cell decision of the greedy walk
//     (smithwaterman.cpp:40-78), followed by sw_walk_kernel;
//   * small problems end-to-end, mi355_sw_fill_matrix (operator()), and the |x| == |y| quirk of
//     the uint8 engine (oracle/sw_oracle.c, SQUARE-CASE QUIRK).
//
// The three most recent anti-diagonals live in LDS, indexed by the row (or by the column when the
// window is narrower than the query).  Lanes stride over the cells of a diagonal; a wavefront
// executes its LDS operations in order, so no barrier is needed between diagonals.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355sw {

enum : int { kDirStop = 0, kDirNW = 1, kDirW = 2, kDirN = 3 };

struct ExactProblem {
  const uint8_t *x;       // query bytes (device)
  const uint8_t *y;       // first byte of the window (device): local column jl uses y[jl-1]
  int32_t m;              // rows
  int32_t nw;             // window columns
  int64_t col_offset;     // true column = col_offset + jl
  int64_t full_n;         // |y| of the full problem (argmax order of the uint8 engine)
  int32_t own_lo;         // only local columns >= own_lo compete for the argmax
  int32_t square_quirk;   // uint8 engine with |x| == |y|, window == whole problem
  float target;           // >= 0: only cells equal to target compete; < 0: track the maximum
  uint8_t *dirs;          // traceback decisions, diagonal-major: dirs[d * dstride + (i - max(1, d - nw))] for
                          // cell (i, jl), d = i + jl, dstride = min(m, nw) — lanes of a diagonal store
                          // consecutive bytes; or null
  float *hout;            // [(nw+1)][(m+1)] matrix values, or null (borders pre-zeroed)
  // outputs
  float *best;            // maximum (or target) found, -1 when nothing competed
  int64_t *cell;          // [2] row, TRUE column of the first maximum in storage order
};

struct ExactScoring {
  const float *lut;       // device 256x256 or null
  float match, mismatch, gap;
  int32_t u8M, u8X, u8G;  // uint8 engine parameters (similaritymatrix.cpp:389-392)
};

// storage-order key of cell (row i, true column j); smaller = earlier in maxCoeff's scan
template <int SEM>
__device__ __forceinline__ unsigned long long order_key(int64_t i, int64_t j, int64_t m, int64_t n) {
  if (SEM == 0) return ((unsigned long long)j << 32) | (unsigned long long)i;   // column-major
  // Similarity_Matrix_Skewed: internal ti = j, tj = i, len_x = n+1, len_y = m+1
  const int64_t len_x = n + 1, len_y = m + 1;
  const int64_t nrows = len_x < len_y ? len_x : len_y;
  const int64_t ncols = len_x < len_y ? len_y : len_x;
  const int64_t ti = j, tj = i;
  int64_t ri, rj;
  if (ti + tj < nrows - 1) { ri = ti; rj = ti + tj; }
  else if (ti + tj > ncols - 1) { ri = ti - ncols + len_y; rj = ti + tj - (ncols - 1) - 1; }
  else { ri = (len_x <= len_y) ? ti : len_y - 1 - tj; rj = ti + tj; }
  return ((unsigned long long)rj << 32) | (unsigned long long)ri;
}

// NT = 64: one wavefront per problem, no barriers (its LDS operations complete in order).
// NT = 1024: sixteen wavefronts share one problem's diagonals (long queries / wide windows), one
// workgroup barrier per diagonal.
template <int SEM, int NT>
__global__ __launch_bounds__(NT) void sw_exact_kernel(const ExactProblem *probs, const ExactScoring sc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
  const ExactProblem P = probs[blockIdx.x];
  const int lane = threadIdx.x;
  const int m = P.m, nw = P.nw;
  const bool rows_short = m <= nw;
  const int plen = (rows_short ? m : nw) + 2;
  float *D0 = reinterpret_cast<float *>(smem_raw);
  float *D1 = D0 + plen;
  float *D2 = D1 + plen;
  uint8_t *xs = reinterpret_cast<uint8_t *>(D2 + plen);
  for (int k = lane; k < 3 * plen; k += NT) D0[k] = 0.0f;
  for (int k = lane; k < m; k += NT) xs[k] = P.x[k];
  if (NT > 64) __syncthreads();

  float best = -1.0f;
  unsigned long long bkey = ~0ull;
  int64_t bi = 0, bj = 0;
  const float g = sc.gap;

  const int dstride = m < nw ? m : nw;
  float *Dc = D0, *Dp = D1, *Dpp = D2;   // current, d-1, d-2
  // diagonals d = i + jl; d = 0 and 1 are all border (zero, already cleared)
  for (int d = 2; d <= m + nw; ++d) {
    // rotate: the buffer of d-2 becomes the new current after use, so order is (Dc <- old Dpp)
    float *t = Dpp; Dpp = Dp; Dp = Dc; Dc = t;
    // border cells of this diagonal
    if (lane == 0) {
      if (rows_short) { Dc[0] = 0.0f; if (d <= m) Dc[d] = 0.0f; }
      else { Dc[0] = 0.0f; if (d <= nw) Dc[d] = 0.0f; }
    }
    const int ilo = d - nw > 1 ? d - nw : 1;
    const int ihi = d - 1 < m ? d - 1 : m;
    for (int i = ilo + lane; i <= ihi; i += NT) {
      const int jl = d - i;
      const int ic = rows_short ? i : jl;          // index of (i, jl)
      const int iw = rows_short ? i : jl - 1;      // (i, jl-1)   on d-1
      const int in_ = rows_short ? i - 1 : jl;     // (i-1, jl)   on d-1
      const int inw = rows_short ? i - 1 : jl - 1; // (i-1, jl-1) on d-2
      const float n1 = Dpp[inw], n2 = Dp[iw], n3 = Dp[in_];
      const uint8_t a = xs[i - 1], b = P.y[jl - 1];
      float h;
      if (SEM == 0) {
        const float s = sc.lut ? sc.lut[(int)a * 256 + b] : (a == b ? sc.match : sc.mismatch);
        // similaritymatrix.cpp:49-54, same operation order
        const float xx = n1 + s, yy = n2 - g, zz = n3 - g;
        h = fmaxf(fmaxf(xx, yy), fmaxf(zz, 0.0f));
      } else {
        float nwv = n1;
        if (P.square_quirk && i + jl == nw + 1) nwv = (i >= 2) ? Dpp[rows_short ? i - 2 : jl] : 0.0f;
        // similaritymatrix.cpp:75-81 on one lane, values are integers 0..255 held in float
        float xx = (a == b) ? fminf(nwv + (float)sc.u8M, 255.0f) : fmaxf(nwv - (float)sc.u8X, 0.0f);
        const float yy = fmaxf(n2 - (float)sc.u8G, 0.0f), zz = fmaxf(n3 - (float)sc.u8G, 0.0f);
        h = fmaxf(fmaxf(xx, yy), zz);
      }
      Dc[ic] = h;
      if (P.hout) P.hout[(size_t)jl * (size_t)(m + 1) + (size_t)i] = h;
      if (P.dirs) {
        // smithwaterman.cpp:51,59,66,72 evaluated at cell (i, jl)
        int dir;
        if (n1 == 0.0f || n2 == 0.0f || n3 == 0.0f) dir = kDirStop;
        else if (n1 >= n2 && n1 >= n3) dir = kDirNW;
        else if (n2 >= n1 && n2 >= n3) dir = kDirW;
        else dir = kDirN;
        P.dirs[(size_t)d * (size_t)dstride + (size_t)(i - ilo)] = (uint8_t)dir;
      }
      if (jl >= P.own_lo && h > 0.0f) {
        const bool cand = (P.target >= 0.0f) ? (h == P.target) : (h >= best);
        if (cand) {
          const int64_t jt = P.col_offset + jl;
          const unsigned long long key = order_key<SEM>(i, jt, m, P.full_n);
          if (h > best || key < bkey) { best = h; bkey = key; bi = i; bj = jt; }
        }
      }
    }
    if (NT > 64) __syncthreads();
  }
  // wave reduction: larger value first, then smaller key
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const unsigned long long ok = __shfl_xor(bkey, off);
    const long long oi = __shfl_xor((long long)bi, off);
    const long long oj = __shfl_xor((long long)bj, off);
    if (ob > best || (ob == best && ok < bkey)) { best = ob; bkey = ok; bi = oi; bj = oj; }
  }
  if (NT > 64) {
    // combine the wavefronts' winners through LDS (the diagonal buffers are free now)
    __shared__ float sb[NT / 64];
    __shared__ unsigned long long sk[NT / 64];
    __shared__ long long si[NT / 64], sj[NT / 64];
    const int w = lane >> 6;
    if ((lane & 63) == 0) { sb[w] = best; sk[w] = bkey; si[w] = bi; sj[w] = bj; }
    __syncthreads();
    if (lane == 0)
      for (int k = 1; k < NT / 64; ++k)
        if (sb[k] > best || (sb[k] == best && sk[k] < bkey)) { best = sb[k]; bkey = sk[k]; bi = si[k]; bj = sj[k]; }
  }
  if (lane == 0) {
    if (P.best) *P.best = best;
    if (P.cell) { P.cell[0] = best > 0.0f ? bi : 0; P.cell[1] = best > 0.0f ? bj : 0; }
  }
}

// Greedy traceback walk over the decisions written by sw_exact_kernel (smithwaterman.cpp:40-78).
struct WalkProblem {
  const uint8_t *x;
  const uint8_t *y;       // window base, as ExactProblem::y
  const uint8_t *dirs;    // diagonal-major, as ExactProblem::dirs
  int32_t m, nw;
  int32_t start_i, start_jl;
  int32_t exact_lo;       // decisions at local columns < exact_lo read inexact cells (0 = all exact)
  float need_slope;       // > 0: row i is already exact from local column i + ceil(i * need_slope) + 2 on (a path
                          // ending in row i has at most i diagonal steps: DESIGN.md §3.3 with m := i)
  int64_t col_offset;
  char *cons_x;           // capacity cap
  char *cons_y;
  int32_t cap;
  // outputs: [0] consensus length, [1] pos (true column), [2] status (0 ok, 1 window too small,
  // 2 capacity exceeded)
  int64_t *out;
};

__global__ void sw_walk_kernel(const WalkProblem *probs, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const WalkProblem W = probs[p];
  int ix = W.start_i, jl = W.start_jl;
  int len = 0;
  int64_t status = 0, pos = 0;
  if (ix <= 0 || jl <= 0) { W.out[0] = 0; W.out[1] = 0; W.out[2] = 0; return; }
  for (;;) {
    // the decision at (ix, jl) looks at columns jl-1 and jl: both must be exact
    if (W.exact_lo > 0) {
      int need = W.exact_lo;
      if (W.need_slope > 0.0f) {
        const int rn = ix + (int)ceilf((float)ix * W.need_slope) + 2;
        need = rn < need ? rn : need;
      }
      if (jl - 1 < need) { status = 1; break; }
    }
    if (len >= W.cap) { status = 2; break; }
    const int d = ix + jl;
    const int ilo = d - W.nw > 1 ? d - W.nw : 1;
    const int dstride = W.m < W.nw ? W.m : W.nw;
    const int dir = W.dirs[(size_t)d * (size_t)dstride + (size_t)(ix - ilo)];
    if (dir == kDirStop) {
      W.cons_x[len] = (char)W.x[ix - 1]; W.cons_y[len] = (char)W.y[jl - 1]; ++len;
      pos = W.col_offset + jl;
      break;
    } else if (dir == kDirNW) {
      W.cons_x[len] = (char)W.x[ix - 1]; W.cons_y[len] = (char)W.y[jl - 1]; ++len; --ix; --jl;
    } else if (dir == kDirW) {
      W.cons_x[len] = '-'; W.cons_y[len] = (char)W.y[jl - 1]; ++len; --jl;
    } else {
      W.cons_x[len] = (char)W.x[ix - 1]; W.cons_y[len] = '-'; ++len; --ix;
    }
  }
  W.out[0] = len; W.out[1] = pos; W.out[2] = status;
}

}  // namespace mi355sw
