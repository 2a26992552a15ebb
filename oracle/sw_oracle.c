/*
 * sw_oracle.c — CPU restatement of the reference's Smith-Waterman hot path.
 * TEST INFRASTRUCTURE ONLY (see sw_oracle.h): never linked into the product.
 *
 * Every function cites the reference file:line whose behaviour it restates
 * (paths relative to /root/reference).  Written from the semantics in
 * SURVEY.md Appendix A; no reference source text is reproduced.
 */
#include "sw_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline float score_of(const sw_oracle_scoring *sc, char a, char b) {
  if (sc->lut) return sc->lut[(size_t)(uint8_t)a * 256u + (uint8_t)b];
  return a == b ? sc->match : sc->mismatch;
}

static inline float fmax2(float a, float b) { return a > b ? a : b; }

/* src/aligner/similaritymatrix.cpp:49-54 — scalar dp_func, same operation order. */
static inline float dp_f32(float north, float west, float north_west, float s, float g) {
  float a = north_west + s;
  float b = west - g;
  float c = north - g;
  return fmax2(fmax2(a, b), fmax2(c, 0.0f));
}

/* src/aligner/similaritymatrix.cpp:376-384 — _saturate: clamp to [0,255], truncate. */
static inline uint8_t sat8(float a) {
  if (a < 0) return 0;
  if (a > 255) return 255;
  return (uint8_t)a;
}

static inline uint8_t adds8(uint8_t a, uint8_t b) { unsigned s = (unsigned)a + b; return s > 255 ? 255 : (uint8_t)s; }
static inline uint8_t subs8(uint8_t a, uint8_t b) { return a > b ? (uint8_t)(a - b) : 0; }
static inline uint8_t max8(uint8_t a, uint8_t b) { return a > b ? a : b; }

typedef struct { uint8_t M, X, G; } u8_params;

/* src/aligner/similaritymatrix.cpp:389-392 — only two probes of the scoring function are used. */
static u8_params u8_params_of(const sw_oracle_scoring *sc) {
  u8_params p;
  p.M = sat8(score_of(sc, 'A', 'A'));
  p.X = sat8(-score_of(sc, 'A', 'T'));
  p.G = sat8(sc->gap);
  return p;
}

/* src/aligner/similaritymatrix.cpp:75-81 — 8-bit unsigned saturating dp_func, one lane. */
static inline uint8_t dp_u8(uint8_t north, uint8_t west, uint8_t north_west, int match, u8_params p) {
  uint8_t a = adds8(north_west, match ? p.M : 0);
  a = subs8(a, match ? 0 : p.X);
  uint8_t b = subs8(west, p.G);
  uint8_t c = subs8(north, p.G);
  return max8(max8(a, b), c);
}

/* src/aligner/similaritymatrix.cpp:16-19 (zeroed (m+1)x(n+1) matrix) and :99-264 (fill).
 * The reference sweeps anti-diagonals; every cell depends only on its N/W/NW
 * neighbours, so a column sweep yields the same values. */
void sw_oracle_fill_f32(const char *x, size_t m, const char *y, size_t n,
                        const sw_oracle_scoring *sc, float *H) {
  const size_t ld = m + 1;
  memset(H, 0, sizeof(float) * ld * (n + 1));
  for (size_t j = 1; j <= n; ++j) {
    float *cur = H + j * ld;
    const float *prv = cur - ld;
    for (size_t i = 1; i <= m; ++i)
      cur[i] = dp_f32(cur[i - 1], prv[i], prv[i - 1], score_of(sc, x[i - 1], y[j - 1]), sc->gap);
  }
}

/* src/aligner/similaritymatrix.cpp:274-289 (zeroed storage) and :386-561 (three-phase skewed
 * fill).  Cell rule of :75-81 with byte equality (:415-417); stored here in true coordinates. */
void sw_oracle_fill_u8(const char *x, size_t m, const char *y, size_t n,
                       const sw_oracle_scoring *sc, uint8_t *H) {
  const size_t ld = m + 1;
  const u8_params p = u8_params_of(sc);
  memset(H, 0, ld * (n + 1));
  /* SQUARE-CASE QUIRK (|x| == |y|), reproduced for bit parity.  In Phase 3 the first
   * lower-triangle anti-diagonal (raw column 0, similaritymatrix.cpp:520-532) reads its NW
   * operand at row offset di_nw = 1 (:523), which assumes raw column ncols-2 is stored in the
   * "vertical band" layout.  When len_x == len_y there is no band (nrows == ncols), column
   * ncols-2 is still in upper-triangle layout, and the load lands one cell off: a cell (i,j)
   * with i + j == n + 1 takes H(i-2, j) in place of H(i-1, j-1) (0 when i < 2; for i == 1 the
   * load hits a lower-triangle slot not yet written, also 0).  Verified cell-by-cell against
   * oracle/_ref (tests/test_oracle_vs_reference.py). */
  const int square = (m == n);
  for (size_t j = 1; j <= n; ++j) {
    uint8_t *cur = H + j * ld;
    const uint8_t *prv = cur - ld;
    for (size_t i = 1; i <= m; ++i) {
      uint8_t nw = prv[i - 1];
      if (square && i + j == n + 1) nw = (i >= 2) ? cur[i - 2] : 0;
      cur[i] = dp_u8(cur[i - 1], prv[i], nw, x[i - 1] == y[j - 1], p);
    }
  }
}

/* src/aligner/similaritymatrix.cpp:21-28 — Eigen maxCoeff on a column-major matrix: first
 * maximum with columns outer, rows inner, strict '>' (Eigen 3.3.7 Visitor.h:49-54). */
void sw_oracle_argmax_f32(const float *H, size_t m, size_t n, int64_t *ix, int64_t *iy, float *mx) {
  const size_t ld = m + 1;
  float best = H[0];
  size_t bi = 0, bj = 0;
  for (size_t j = 0; j <= n; ++j)
    for (size_t i = 0; i <= m; ++i)
      if (H[j * ld + i] > best) { best = H[j * ld + i]; bi = i; bj = j; }
  *ix = (int64_t)bi; *iy = (int64_t)bj; *mx = best;
}

/* src/aligner/similaritymatrix.cpp:330-346 */
void sw_oracle_raw2true(size_t ri, size_t rj, size_t nrows, size_t ncols, size_t len_x,
                        size_t len_y, size_t *ti, size_t *tj) {
  (void)ncols;
  if (rj < nrows - 1) {
    if (ri <= rj) { *ti = ri; *tj = rj - ri; }                          /* upper triangle */
    else { *ti = len_x - nrows + ri; *tj = len_y - ri + rj; }            /* lower triangle */
  } else if (len_x <= len_y) { *ti = ri; *tj = rj - ri; }                /* band, +y */
  else { *ti = rj - (nrows - 1) + ri; *tj = nrows - 1 - ri; }            /* band, +x */
}

/* src/aligner/similaritymatrix.cpp:353-364 */
void sw_oracle_true2raw(size_t ti, size_t tj, size_t nrows, size_t ncols, size_t len_x,
                        size_t len_y, size_t *ri, size_t *rj) {
  if (ti + tj < nrows - 1) { *ri = ti; *rj = ti + tj; }
  else if (ti + tj > ncols - 1) { *ri = ti - ncols + len_y; *rj = ti + tj - (ncols - 1) - 1; }
  else { *ri = (len_x <= len_y) ? ti : len_y - 1 - tj; *rj = ti + tj; }
}

/* src/aligner/similaritymatrix.cpp:291-299 — maxCoeff over the RAW skewed storage (the 32 zero
 * pad rows of :287 can only win when everything is zero, in which case raw (0,0) wins anyway),
 * mapped back through :330-346 and swapped (:298).  Internal coordinates: ti = column of y,
 * tj = row of x, len_x = n+1, len_y = m+1 (constructor swap, :274-285). */
void sw_oracle_argmax_u8(const uint8_t *H, size_t m, size_t n, int64_t *ix, int64_t *iy, float *mx) {
  const size_t ld = m + 1;
  const size_t len_x = n + 1, len_y = m + 1;
  const size_t nrows = len_x < len_y ? len_x : len_y;
  const size_t ncols = len_x < len_y ? len_y : len_x;
  uint8_t best = 0;
  size_t bti = 0, btj = 0;
  int first = 1;
  for (size_t rj = 0; rj < ncols; ++rj)
    for (size_t ri = 0; ri < nrows; ++ri) {
      size_t ti, tj;
      sw_oracle_raw2true(ri, rj, nrows, ncols, len_x, len_y, &ti, &tj);
      uint8_t v = H[ti * ld + tj];
      if (first || v > best) { best = v; bti = ti; btj = tj; first = 0; }
    }
  *ix = (int64_t)btj; *iy = (int64_t)bti; *mx = (float)best;
}

typedef struct { char *p; size_t len, cap; } strbuf;
static void sb_push(strbuf *s, char c) {
  if (s->len + 2 > s->cap) { s->cap = s->cap ? s->cap * 2 : 64; s->p = (char *)realloc(s->p, s->cap); }
  s->p[s->len++] = c; s->p[s->len] = 0;
}

/* src/aligner/smithwaterman.cpp:40-78 — greedy-by-neighbour-value traceback. `cell` reads
 * H(row, col) as a float, as both operator() implementations do (similaritymatrix.h:45,:76-79). */
#define TRACEBACK_BODY(CELL)                                                              \
  int64_t ix = *pix, iy = *piy;                                                           \
  for (;;) {                                                                              \
    float n1 = CELL(ix - 1, iy - 1), n2 = CELL(ix, iy - 1), n3 = CELL(ix - 1, iy);        \
    if (n1 == 0 || n2 == 0 || n3 == 0) {                                                  \
      sb_push(cx, x[ix - 1]); sb_push(cy, y[iy - 1]); *pos = (uint32_t)iy; break;         \
    }                                                                                     \
    if (n1 >= n2 && n1 >= n3) { sb_push(cx, x[ix - 1]); sb_push(cy, y[iy - 1]); --ix; --iy; } \
    else if (n2 >= n1 && n2 >= n3) { sb_push(cx, '-'); sb_push(cy, y[iy - 1]); --iy; }    \
    else { sb_push(cx, x[ix - 1]); sb_push(cy, '-'); --ix; }                              \
  }                                                                                       \
  *pix = ix; *piy = iy;

static void traceback_f32(const float *H, size_t ld, const char *x, const char *y,
                          int64_t *pix, int64_t *piy, strbuf *cx, strbuf *cy, uint32_t *pos) {
#define CELLF(r, c) (H[(size_t)(c) * ld + (size_t)(r)])
  TRACEBACK_BODY(CELLF)
#undef CELLF
}
static void traceback_u8(const uint8_t *H, size_t ld, const char *x, const char *y,
                         int64_t *pix, int64_t *piy, strbuf *cx, strbuf *cy, uint32_t *pos) {
#define CELLU(r, c) ((float)H[(size_t)(c) * ld + (size_t)(r)])
  TRACEBACK_BODY(CELLU)
#undef CELLU
}

static void result_init(sw_oracle_result *out) {
  memset(out, 0, sizeof(*out));
  out->score = -1.0f;               /* smithwaterman.cpp:27-33 / plocalaligner.cpp:78-84 */
  out->cons_x = (char *)calloc(1, 1);
  out->cons_y = (char *)calloc(1, 1);
}

/* src/aligner/smithwaterman.cpp:80-108 — iterate -> find_index_of_maximum -> traceback.
 * DELIBERATE DIVERGENCE: an all-zero matrix is undefined behaviour in the
 * reference (traceback reads index -1); here it yields score 0, pos 0, empty consensus. */
int sw_oracle_align(const char *x, size_t m, const char *y, size_t n,
                    const sw_oracle_scoring *sc, int semantics, sw_oracle_result *out) {
  result_init(out);
  const size_t ld = m + 1, cells = ld * (n + 1);
  int64_t ix = 0, iy = 0;
  float mx = 0;
  strbuf cx = {0, 0, 0}, cy = {0, 0, 0};
  if (semantics == SW_ORACLE_F32) {
    float *H = (float *)malloc(sizeof(float) * cells);
    if (!H) return -1;
    sw_oracle_fill_f32(x, m, y, n, sc, H);
    sw_oracle_argmax_f32(H, m, n, &ix, &iy, &mx);
    out->score = mx; out->end_x = ix; out->end_y = iy;
    if (mx > 0) traceback_f32(H, ld, x, y, &ix, &iy, &cx, &cy, &out->pos);
    free(H);
  } else {
    uint8_t *H = (uint8_t *)malloc(cells);
    if (!H) return -1;
    sw_oracle_fill_u8(x, m, y, n, sc, H);
    sw_oracle_argmax_u8(H, m, n, &ix, &iy, &mx);
    out->score = mx; out->end_x = ix; out->end_y = iy;
    if (mx > 0) traceback_u8(H, ld, x, y, &ix, &iy, &cx, &cy, &out->pos);
    free(H);
  }
  if (mx <= 0) { out->end_x = 0; out->end_y = 0; }
  if (cx.p) { free(out->cons_x); free(out->cons_y); out->cons_x = cx.p; out->cons_y = cy.p; out->cons_len = cx.len; }
  return 0;
}

/* Traceback (smithwaterman.cpp:40-78) from a GIVEN start cell over the matrix of (x, y): the companion of
 * sw_oracle_locate for problems whose full matrix does not fit.  y is a window of the full reference that ENDS at the
 * argmax column; the argmax itself comes from the full-size sw_oracle_locate (the uint8 engine's storage order depends
 * on the full problem's ncols, SURVEY.md App. A.5: it must never be re-derived on a window), the window supplies
 * only the H values the walk reads.  out->score = H(start) so the caller can check the window was long enough for
 * the start cell to be exact; end_x / end_y = the start cell; pos is window-local. */
int sw_oracle_trace_from(const char *x, size_t m, const char *y, size_t n, const sw_oracle_scoring *sc,
                         int semantics, int64_t start_x, int64_t start_y, sw_oracle_result *out) {
  result_init(out);
  if (start_x < 1 || start_y < 1 || (size_t)start_x > m || (size_t)start_y > n) return -1;
  const size_t ld = m + 1, cells = ld * (n + 1);
  int64_t ix = start_x, iy = start_y;
  strbuf cx = {0, 0, 0}, cy = {0, 0, 0};
  if (semantics == SW_ORACLE_F32) {
    float *H = (float *)malloc(sizeof(float) * cells);
    if (!H) return -1;
    sw_oracle_fill_f32(x, m, y, n, sc, H);
    out->score = H[(size_t)iy * ld + (size_t)ix];
    if (out->score > 0) traceback_f32(H, ld, x, y, &ix, &iy, &cx, &cy, &out->pos);
    free(H);
  } else {
    uint8_t *H = (uint8_t *)malloc(cells);
    if (!H) return -1;
    sw_oracle_fill_u8(x, m, y, n, sc, H);
    out->score = (float)H[(size_t)iy * ld + (size_t)ix];
    if (out->score > 0) traceback_u8(H, ld, x, y, &ix, &iy, &cx, &cy, &out->pos);
    free(H);
  }
  out->end_x = start_x; out->end_y = start_y;
  if (cx.p) { free(out->cons_x); free(out->cons_y); out->cons_x = cx.p; out->cons_y = cy.p; out->cons_len = cx.len; }
  return 0;
}

/* src/aligner/plocalaligner.cpp:44-67 */
int sw_oracle_make_string_range(int npiece, int64_t shortlen, int64_t longlen, float overlap_ratio,
                                int64_t *lefts, int64_t *rights) {
  int64_t overlap = (int64_t)((float)shortlen * overlap_ratio);
  if (npiece < 1) return -1;
  if (npiece == 1) { lefts[0] = 0; rights[0] = longlen; return 0; }
  int64_t piecelen = (longlen + (int64_t)(npiece - 1) * overlap) / npiece;
  if (!(overlap <= piecelen)) return -1;                 /* assert :52 */
  int64_t left = 0, right = piecelen;
  int k = 0;
  lefts[k] = left; rights[k] = right; ++k;
  while (k < npiece - 1) {
    left = right - overlap; if (left < 0) left = 0;
    right = left + piecelen; if (right > longlen) right = longlen;
    lefts[k] = left; rights[k] = right; ++k;
  }
  if (!(right < longlen)) return -1;                     /* assert :63 */
  left = right - overlap; if (left < 0) left = 0;
  lefts[k] = left; rights[k] = longlen; ++k;
  return 0;
}

static float piece_max(const char *x, size_t m, const char *y, size_t n,
                       const sw_oracle_scoring *sc, int semantics) {
  /* find_index_of_maximum()'s value only (plocalaligner.cpp:123); value is order independent. */
  return sw_oracle_score_only(x, m, y, n, sc, semantics);
}

/* src/aligner/plocalaligner.cpp:105-143, serial build. */
int sw_oracle_align_split(const char *x, size_t m, const char *y, size_t n,
                          const sw_oracle_scoring *sc, int sm_semantics, int la_semantics,
                          int npiece, float overlap_ratio, sw_oracle_result *out,
                          int *winning_piece) {
  int64_t *lefts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(npiece > 0 ? npiece : 1) * 2);
  int64_t *rights = lefts + (npiece > 0 ? npiece : 1);
  if (sw_oracle_make_string_range(npiece, (int64_t)m, (int64_t)n, overlap_ratio, lefts, rights)) {
    free(lefts); result_init(out); return -1;
  }
  float best = -1.0f;
  int bp = 0;
  for (int p = 0; p < npiece; ++p) {
    float v = piece_max(x, m, y + lefts[p], (size_t)(rights[p] - lefts[p]), sc, sm_semantics);
    if (v > best) { best = v; bp = p; }                   /* strict '>' (:125) */
  }
  sw_oracle_scoring def = {NULL, 3.0f, -3.0f, 2.0f};      /* default ctor of LAT (:135) */
  int rc = sw_oracle_align(x, m, y + lefts[bp], (size_t)(rights[bp] - lefts[bp]), &def, la_semantics, out);
  if (rc == 0 && out->score > 0) { out->pos += (uint32_t)lefts[bp]; out->end_y += lefts[bp]; }
  else if (rc == 0) { out->pos = (uint32_t)lefts[bp]; }   /* pos = 0 + left (:137) */
  if (winning_piece) *winning_piece = bp;
  free(lefts);
  return rc;
}

void sw_oracle_free_result(sw_oracle_result *r) {
  free(r->cons_x); free(r->cons_y); r->cons_x = r->cons_y = NULL; r->cons_len = 0;
}

float sw_oracle_score_only(const char *x, size_t m, const char *y, size_t n,
                           const sw_oracle_scoring *sc, int semantics) {
  if (semantics == SW_ORACLE_F32) {
    float *col = (float *)calloc(m + 1, sizeof(float));
    float best = 0;
    for (size_t j = 1; j <= n; ++j) {
      float nw = 0, north = 0;               /* H(0,j-1), H(0,j) */
      const char b = y[j - 1];
      for (size_t i = 1; i <= m; ++i) {
        float w = col[i];
        float h = dp_f32(north, w, nw, score_of(sc, x[i - 1], b), sc->gap);
        nw = w; col[i] = h; north = h;
        if (h > best) best = h;
      }
    }
    free(col);
    return best;
  } else {
    const u8_params p = u8_params_of(sc);
    uint8_t *col = (uint8_t *)calloc(m + 1, 1);
    uint8_t best = 0;
    for (size_t j = 1; j <= n; ++j) {
      uint8_t nw = 0, north = 0;
      const char b = y[j - 1];
      uint8_t north2 = 0;                    /* H(i-2, j) for the square-case quirk */
      for (size_t i = 1; i <= m; ++i) {
        uint8_t w = col[i];
        uint8_t d = (m == n && i + j == n + 1) ? north2 : nw;
        uint8_t h = dp_u8(north, w, d, x[i - 1] == b, p);
        nw = w; col[i] = h; north2 = north; north = h;
        if (h > best) best = h;
      }
    }
    free(col);
    return (float)best;
  }
}

/* src/aligner/similaritymatrix.cpp:21-28 / :291-299 without storing the matrix (see sw_oracle.h). */
void sw_oracle_locate(const char *x, size_t m, const char *y, size_t n, const sw_oracle_scoring *sc,
                      int semantics, float *mx, int64_t *ix, int64_t *iy) {
  *mx = 0; *ix = 0; *iy = 0;
  if (semantics == SW_ORACLE_F32) {
    float *col = (float *)calloc(m + 1, sizeof(float));
    float best = 0;
    for (size_t j = 1; j <= n; ++j) {
      float nw = 0, north = 0;
      const char b = y[j - 1];
      for (size_t i = 1; i <= m; ++i) {
        float w = col[i];
        float h = dp_f32(north, w, nw, score_of(sc, x[i - 1], b), sc->gap);
        nw = w; col[i] = h; north = h;
        if (h > best) { best = h; *ix = (int64_t)i; *iy = (int64_t)j; }   /* columns outer, rows inner, strict */
      }
    }
    free(col);
    *mx = best;
  } else {
    const u8_params p = u8_params_of(sc);
    const size_t len_x = n + 1, len_y = m + 1;
    const size_t nrows = len_x < len_y ? len_x : len_y;
    const size_t ncols = len_x < len_y ? len_y : len_x;
    uint8_t *col = (uint8_t *)calloc(m + 1, 1);
    uint8_t best = 0;
    size_t bri = 0, brj = 0;
    for (size_t j = 1; j <= n; ++j) {
      uint8_t nw = 0, north = 0, north2 = 0;
      const char b = y[j - 1];
      for (size_t i = 1; i <= m; ++i) {
        uint8_t w = col[i];
        uint8_t d = (m == n && i + j == n + 1) ? north2 : nw;
        uint8_t h = dp_u8(north, w, d, x[i - 1] == b, p);
        nw = w; col[i] = h; north2 = north; north = h;
        if (h >= best && h > 0) {
          size_t ri, rj;
          sw_oracle_true2raw(j, i, nrows, ncols, len_x, len_y, &ri, &rj);
          if (h > best || rj < brj || (rj == brj && ri < bri)) { best = h; bri = ri; brj = rj; *ix = (int64_t)i; *iy = (int64_t)j; }
        }
      }
    }
    free(col);
    *mx = (float)best;
  }
}
