/*
 * sw_oracle.h — CPU restatement of the reference's Smith-Waterman hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.  The product path is the
 * HIP library behind include/mi355_sw.h and never links or calls this file.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle_*.py) against
 *   - the reference's own gtest known answers (test/test_localaligner.cpp:24-27,
 *     :53-58; test/test_skewedmatrix.cpp:5-66),
 *   - fixtures in tests/golden/ produced by the real reference compiled from
 *     /root/reference (oracle/build_ref.sh -> oracle/_ref/ref_driver), and
 *   - the data_small digests of SURVEY.md Appendix B.
 *
 * Notation: x = first ctor argument = rows 1..m (query),
 * y = second = columns 1..n (reference); H(0,.) = H(.,0) = 0.
 */
#ifndef SW_ORACLE_H_
#define SW_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SW_ORACLE_F32 = 0, SW_ORACLE_U8SAT = 1 };

typedef struct {
  const float *lut; /* 256x256, lut[(uint8)a*256+(uint8)b] = f(a,b); NULL => match/mismatch */
  float match;      /* f(a,a) when lut == NULL (reference default 3)  */
  float mismatch;   /* f(a,b), a!=b, when lut == NULL (default -3)    */
  float gap;        /* gap_penalty (default 2)                         */
} sw_oracle_scoring;

typedef struct {
  float score;      /* max cell value; -1 before any computation (smithwaterman.cpp:27-33) */
  uint32_t pos;     /* 1-based column in y where the greedy traceback stopped */
  int64_t end_x;    /* argmax row (1-based into x), 0 when score == 0 */
  int64_t end_y;    /* argmax column (1-based into y)                  */
  char *cons_x;     /* reversed consensus, '-' gaps, NUL terminated, malloc'd */
  char *cons_y;
  size_t cons_len;
} sw_oracle_result;

/* Full matrices, column-major over the reference: H[j*(m+1)+i], i=0..m, j=0..n.
 * similaritymatrix.cpp:49-54,99-264 (F32) and :75-81,386-561 (U8SAT). */
void sw_oracle_fill_f32(const char *x, size_t m, const char *y, size_t n,
                        const sw_oracle_scoring *sc, float *H);
void sw_oracle_fill_u8(const char *x, size_t m, const char *y, size_t n,
                       const sw_oracle_scoring *sc, uint8_t *H);

/* First maximum in the reference's storage order.
 * similaritymatrix.cpp:21-28 (F32) and :291-299 with :330-364 (U8SAT). */
void sw_oracle_argmax_f32(const float *H, size_t m, size_t n, int64_t *ix, int64_t *iy, float *mx);
void sw_oracle_argmax_u8(const uint8_t *H, size_t m, size_t n, int64_t *ix, int64_t *iy, float *mx);

/* Skewed index maps (similaritymatrix.cpp:330-346, :353-364); internal coordinates. */
void sw_oracle_raw2true(size_t ri, size_t rj, size_t nrows, size_t ncols, size_t len_x,
                        size_t len_y, size_t *ti, size_t *tj);
void sw_oracle_true2raw(size_t ti, size_t tj, size_t nrows, size_t ncols, size_t len_x,
                        size_t len_y, size_t *ri, size_t *rj);

/* SWAligner<SMT>::calculateScore (smithwaterman.cpp:80-108 incl. traceback :40-78). */
int sw_oracle_align(const char *x, size_t m, const char *y, size_t n,
                    const sw_oracle_scoring *sc, int semantics, sw_oracle_result *out);

/* _make_string_range (plocalaligner.cpp:44-67). Returns 0, or -1 where the
 * reference's asserts (:52,:63,:65) would fire. */
int sw_oracle_make_string_range(int npiece, int64_t shortlen, int64_t longlen, float overlap_ratio,
                                int64_t *lefts, int64_t *rights);

/* OMPParallelLocalAligner<SMT,LAT>::calculateScore, SERIAL semantics
 * (plocalaligner.cpp:105-143): per-piece max with the user's scoring under
 * sm_semantics, first strictly greater piece wins, winner re-aligned under
 * la_semantics with DEFAULT scoring (:135), pos += left (:137). */
int sw_oracle_align_split(const char *x, size_t m, const char *y, size_t n,
                          const sw_oracle_scoring *sc, int sm_semantics, int la_semantics,
                          int npiece, float overlap_ratio, sw_oracle_result *out,
                          int *winning_piece);

/* Memory-lean iterate + find_index_of_maximum: rolling column, no matrix.  Same first-maximum rule as
 * sw_oracle_argmax_f32 / _u8 (column-major strict '>' for F32; smallest raw (rj, ri) key for U8SAT).
 * For sizes where the full matrix does not fit (150 bp x 50 Mbp = 30 GB in float). */
void sw_oracle_locate(const char *x, size_t m, const char *y, size_t n, const sw_oracle_scoring *sc,
                      int semantics, float *mx, int64_t *ix, int64_t *iy);

/* Greedy traceback (smithwaterman.cpp:40-78) from a given start cell of the matrix of (x, y) — y a window of the
 * full reference ending at the argmax column, the start cell from the full-size sw_oracle_locate.  out->score is
 * H(start) in the window (equal to the full-size maximum when the window is long enough, App. A.5). */
int sw_oracle_trace_from(const char *x, size_t m, const char *y, size_t n, const sw_oracle_scoring *sc,
                         int semantics, int64_t start_x, int64_t start_y, sw_oracle_result *out);

void sw_oracle_free_result(sw_oracle_result *r);

/* Score-only rolling-column pass (no matrix): max cell value.  Used for the
 * bench "port" CPU baseline and for large-size value checks. */
float sw_oracle_score_only(const char *x, size_t m, const char *y, size_t n,
                           const sw_oracle_scoring *sc, int semantics);

#ifdef __cplusplus
}
#endif
#endif
