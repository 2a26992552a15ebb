/*
 * mi355_sw.h — C-ABI of the MI355X (gfx950) Smith-Waterman engine.
 *
 * This is the drop-in boundary for the reference's hot path (kosta777/parallel-genomeseq):
 * the entry points are what a binding of `SWAligner<SMT>::calculateScore()` and
 * `OMPParallelLocalAligner<SMT,LAT>::calculateScore()` needs, with plain pointers and sizes.
 * Reference interfaces replaced (paths relative to the reference root):
 *
 *   mi355_sw_align        SWAligner<SMT>::calculateScore + getScore/getPos/getConsensus_x/_y
 *                         (src/aligner/smithwaterman.h:11-58, smithwaterman.cpp:80-108)
 *   mi355_sw_align_batch  the loop of independent alignments in src/sw_solve_small.cpp:82-93,
 *                         src/sw_solve_big.cpp:78-92, src/mpi_sw_solve_uniprot.cpp:95-138
 *   mi355_sw_align_split  OMPParallelLocalAligner<SMT,LAT>::calculateScore
 *                         (src/aligner/plocalaligner.h:6-33, plocalaligner.cpp:105-143)
 *   mi355_sw_make_string_range   _make_string_range (plocalaligner.cpp:44-67)
 *   mi355_sw_fill_matrix  Abstract_Similarity_Matrix::iterate + operator()(row,col)
 *                         (src/aligner/similaritymatrix.h:13-24,45,76-79)
 *   mi355_sw_argmax       Abstract_Similarity_Matrix::find_index_of_maximum
 *                         (similaritymatrix.cpp:21-28, :291-299)
 *
 * Semantics: MI355_SW_F32 follows Similarity_Matrix (float cells, linear gap), MI355_SW_U8SAT
 * follows Similarity_Matrix_Skewed (uint8 saturating cells; only f('A','A'), f('A','T') and the
 * gap are used, similaritymatrix.cpp:389-392).  Argmax tie-breaks, the greedy traceback and the
 * reversed consensus strings are those of the reference.
 *
 * Deliberate divergence: an all-zero matrix (no positive cell) is undefined behaviour in the
 * reference (smithwaterman.cpp:46-48); here it yields score 0, pos 0, empty consensus.
 *
 * Ownership: inputs are caller-owned and only read during the call.  Output strings are
 * library-owned (malloc) and released with mi355_sw_free_result(s).  One context per host
 * thread / per GPU; a context is not thread-safe.  All functions return 0 on success or a
 * negative MI355_SW_E* code; mi355_sw_last_error() gives the message.  There is NO CPU
 * fallback: without a usable HIP device every compute entry point fails with MI355_SW_ENODEV.
 */
#ifndef MI355_SW_H_
#define MI355_SW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MI355_SW_F32 = 0, MI355_SW_U8SAT = 1 };

enum {
  MI355_SW_OK = 0,
  MI355_SW_EINVAL = -22,   /* bad argument */
  MI355_SW_ENOMEM = -12,   /* host or device allocation failed */
  MI355_SW_ENODEV = -19,   /* no usable HIP device / HIP call failed */
  MI355_SW_ENOTSUP = -95,  /* input outside what this build's kernels cover (message says what) */
  MI355_SW_ERANGE = -34    /* the reference's own asserts would fire (plocalaligner.cpp:52,63,65) */
};

/* flags for mi355_sw_align_batch / mi355_sw_batch_run */
enum {
  MI355_SW_SCORE_ONLY = 1  /* score + argmax cell only: pos = 0, empty consensus */
};

typedef struct mi355_sw_ctx mi355_sw_ctx;

typedef struct {
  const float *lut;  /* 256x256 table, lut[(uint8)a*256 + (uint8)b] = f(a,b) with a from x, b from y;
                        NULL => f(a,b) = (a == b ? match : mismatch)  (smithwaterman.cpp:8) */
  float match;       /* reference default  3 */
  float mismatch;    /* reference default -3 */
  float gap;         /* gap_penalty, reference default 2 */
  int semantics;     /* MI355_SW_F32 | MI355_SW_U8SAT */
} mi355_sw_params;

typedef struct {
  float score;          /* getScore(): maximum cell value */
  uint32_t pos;         /* getPos(): 1-based column of y where the traceback stopped */
  int64_t end_x;        /* argmax row    (1-based index into x), 0 when score == 0 */
  int64_t end_y;        /* argmax column (1-based index into y), 0 when score == 0 */
  char *cons_x;         /* getConsensus_x(): reversed, '-' for gaps, NUL terminated */
  char *cons_y;         /* getConsensus_y(); lives in the same allocation as cons_x: release both with
                           mi355_sw_free_result, never with free() */
  size_t cons_len;
  float timings_us[2];  /* getTimings(): [0] DP-fill device time of the call that produced this
                           result (shared by all results of one batch), [1] sum over pieces */
} mi355_sw_result;

int mi355_sw_create(mi355_sw_ctx **ctx, int device);
void mi355_sw_destroy(mi355_sw_ctx *ctx);
const char *mi355_sw_last_error(const mi355_sw_ctx *ctx);
void mi355_sw_default_params(mi355_sw_params *p); /* 3 / -3 / 2, F32, no table */

/* A/B and diagnostic switches of one context (DESIGN.md §8.1 lists them; NONE changes a result — each selects another
 * kernel instance or pipeline for the same answer, which is what the parity tests use them for).  Defaults come from the
 * environment, MI355_SW_<NAME IN CAPITALS>, read once by mi355_sw_create.  key: e.g. "no_f16" (or "MI355_SW_NO_F16");
 * value: NULL, "", "0", "off", "false" = off, anything else = on (integer options: the number).
 * mi355_sw_option_names(): comma-separated list of the keys this build knows.  MI355_SW_EINVAL for an unknown key. */
int mi355_sw_set_option(mi355_sw_ctx *ctx, const char *key, const char *value);
const char *mi355_sw_option_names(void);

/* One alignment of x (rows) against y (columns).  The last y of such calls stays resident on the device and is
 * used again when a later call passes the same bytes: same length and same 128-bit content hash (two independent
 * 64-bit hashes; re-hashed on every call, on helper threads, while the call already runs on the resident copy; a
 * changed buffer costs one extra upload).  MI355_SW_NO_REF_CACHE=1 in the environment switches the reuse off (every
 * call uploads y).  A context serves one host thread at a time; use one context per thread. */
int mi355_sw_align(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                   const mi355_sw_params *params, mi355_sw_result *out);

/* Make y resident in HBM for subsequent batch calls (copied; caller may free y). */
int mi355_sw_set_reference(mi355_sw_ctx *ctx, const char *y, size_t ny);

/* n independent alignments of xs[k] (length nxs[k]) against the resident reference. */
int mi355_sw_align_batch(mi355_sw_ctx *ctx, size_t n, const char *const *xs, const size_t *nxs,
                         const mi355_sw_params *params, int flags, mi355_sw_result *outs);

/* Same, with the queries made resident first so that a timed region can start with all inputs
 * in HBM (bench.py).  mi355_sw_batch_run may be called repeatedly. */
int mi355_sw_batch_upload(mi355_sw_ctx *ctx, size_t n, const char *const *xs, const size_t *nxs);
int mi355_sw_batch_run(mi355_sw_ctx *ctx, const mi355_sw_params *params, int flags,
                       mi355_sw_result *outs);
/* mi355_sw_batch_upload for a batch that already lies in ONE buffer: sequence k = buf[offsets[k], offsets[k + 1]), n + 1
 * ascending offsets — what a multi-FASTA reader has (the reference concatenates the lines of each of its 561 356 files into
 * one string, src/mpi_sw_solve_uniprot.cpp:97-110).  No pointer per sequence and no staging copy: the caller's bytes go to
 * the device as they are while the index is built. */
int mi355_sw_batch_upload_packed(mi355_sw_ctx *ctx, size_t n, const char *buf, const int64_t *offsets);

/* mi355_sw_batch_run with the results as a struct of arrays in LIBRARY-OWNED memory (valid until the next call on this
 * context; nothing to free): for batches of very many small alignments (the 561 356 sequences of
 * src/mpi_sw_solve_uniprot.cpp's run), where one malloc'ed string pair per alignment costs more than the alignments.
 * cons_x[k] / cons_y[k]: cons_len[k] bytes each, NOT NUL-terminated, NULL when cons_len[k] == 0. */
typedef struct {
  size_t n;
  const float *score;
  const uint32_t *pos;
  const int64_t *end_x;
  const int64_t *end_y;
  const char *const *cons_x;
  const char *const *cons_y;
  const uint32_t *cons_len;
  float timings_us[2];
} mi355_sw_batch_view;
int mi355_sw_batch_run_view(mi355_sw_ctx *ctx, const mi355_sw_params *params, int flags, mi355_sw_batch_view *out);

/* OMPParallelLocalAligner: split y into npiece overlapping pieces, pick the first piece with the
 * strictly greatest maximum under (params, sm_semantics), re-align it under la_semantics with
 * DEFAULT scoring (plocalaligner.cpp:135), pos += left.  winning_piece may be NULL. */
int mi355_sw_align_split(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                         const mi355_sw_params *params, int sm_semantics, int la_semantics,
                         int npiece, float overlap_ratio, mi355_sw_result *out, int *winning_piece);

/* Maximum cell value of every resident query over each sub-range [lefts[k], rights[k]) of the resident
 * reference, each range an independent problem with a zero left border (what
 * OMPParallelLocalAligner's per-piece iterate + find_index_of_maximum produce, plocalaligner.cpp:110-129).
 * maxima[k * n_queries + q].  Used by multi-GPU reference sharding: every rank sweeps its own pieces. */
int mi355_sw_score_ranges(mi355_sw_ctx *ctx, size_t nranges, const int64_t *lefts, const int64_t *rights,
                          const mi355_sw_params *params, float *maxima);

/* mi355_sw_score_ranges for callers that only need the WINNER: per resident query the first range with the strictly
 * greatest maximum (best[q], best_range[q]; -1 / -1 without ranges) — all OMPParallelLocalAligner does with the per-piece
 * maxima (plocalaligner.cpp:106,122-129).  That freedom lets a lone long query take the cheaper sampled sweep: ranges that
 * cannot hold the greatest maximum are not re-evaluated, so their entries of `maxima` (optional, may be NULL) are LOWER BOUNDS
 * of unspecified slack (the sampled running maximum and the optimistic warm-up margin below both only ever lower a value; never
 * above the true maximum of the range).  Exact: best, best_range, and the entry of every range whose maximum equals `best`.
 * A lone long query is also swept with an OPTIMISTIC warm-up margin in front of its tiles: enough for maxima above 11/12 of
 * the best possible score, not for every cell.  exact_above == NULL: the call checks its own best against
 * what that margin certifies and sweeps again with the margin its best needs when it falls short — results as above.
 * exact_above != NULL (multi-GPU reference sharding: every rank calls this on its own pieces and the packed (best, ~piece)
 * keys are merged by one 8-byte all-reduce(MAX)): no second sweep here; *exact_above = the value above which this call was
 * exact (the same on every rank; -1: everything).  When the MERGED best does not exceed it, every rank calls again with
 * known_best = the merged best (a lower bound of the true one): the margin is then the one that value needs, and the second
 * merge is exact.  known_best <= 0: nothing known. */
int mi355_sw_best_range(mi355_sw_ctx *ctx, size_t nranges, const int64_t *lefts, const int64_t *rights,
                        const mi355_sw_params *params, float known_best, float *maxima, float *best, int64_t *best_range,
                        float *exact_above);

/* Finishes range `range_index` of the LAST mi355_sw_score_ranges / mi355_sw_best_range call on this context as a stand-alone problem: argmax
 * cell and traceback of every resident query within [lefts[k], rights[k]) — what LAT(sequence_x, winning piece) computes
 * (plocalaligner.cpp:132-137); pos / end_y are relative to the range start (the caller adds `left`, :137).  When `params`
 * equals the scoring and engine of that sweep, its per-range keys are used and only the argmax window and the traceback
 * run (the reference sweeps the winning piece a second time); otherwise the range is swept again under `params`.
 * outs[n_queries].  MI355_SW_EINVAL when reference or batch changed since the sweep. */
int mi355_sw_align_scored_range(mi355_sw_ctx *ctx, size_t range_index, const mi355_sw_params *params, int flags,
                                mi355_sw_result *outs);

/* Host-only helper: piece ranges [left,right). Returns MI355_SW_ERANGE where the reference asserts. */
int mi355_sw_make_string_range(int npiece, int64_t shortlen, int64_t longlen, float overlap_ratio,
                               int64_t *lefts, int64_t *rights);

/* Host-only helpers: Similarity_Matrix_Skewed::trueindex2rawindex / rawindex2trueindex
 * (similaritymatrix.cpp:330-346, :353-364) for a matrix built from (x of length nx, y of length ny);
 * indices are in the skewed class's INTERNAL coordinates (ti = column of y, tj = row of x). */
void mi355_sw_true2raw(size_t nx, size_t ny, size_t ti, size_t tj, size_t *ri, size_t *rj);
void mi355_sw_raw2true(size_t nx, size_t ny, size_t ri, size_t rj, size_t *ti, size_t *tj);

/* Full matrix on the device, copied out as float, column-major over y:
 * H[j*(nx+1) + i] = matrix(i, j), i = 0..nx, j = 0..ny.  For small problems (tests, operator()). */
int mi355_sw_fill_matrix(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                         const mi355_sw_params *params, float *H);

/* find_index_of_maximum(): first maximum in the reference's storage order. */
int mi355_sw_argmax(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                    const mi355_sw_params *params, int64_t *index_x, int64_t *index_y, float *max);

/* Device timings of the last batch/align call, microseconds (HIP events on the library's stream):
 * [0] score kernel(s), [1] argmax rescan, [2] traceback window + walk, [3] whole call (device),
 * [4] number of score-kernel launches, [5] cells swept by the score kernel(s). */
int mi355_sw_last_timings(const mi355_sw_ctx *ctx, double out[6]);

/* What the candidate filters of the last batch / align call did: [0] queries that exceeded their
 * candidate cap and were swept a second time on the exact instances (per-query fallback), [1] times the WHOLE batch was swept
 * again (more than half of it exceeded the cap), [2] candidate sub-chunks re-evaluated exactly, [3] walks of the
 * many-small-alignments batch that left the decision window in front of their argmax and were redone on whole-problem decisions
 * (DESIGN.md §4.3). */
int mi355_sw_last_counters(const mi355_sw_ctx *ctx, uint64_t out[4]);
/* One counter of the last call by name: "requeried", "whole_batch_again", "candidates", "left_window" (= [0..3] above) and
 * "first_settled" — uint8 engine: queries over their candidate cap whose first candidates, evaluated in order, settled them
 * without a second sweep; "saved_locates" / "saved_traces" — finish steps of a lone long query that started from the columns and
 * strip rows its sweep saved, "saved_fallbacks" — those that took the zero-border windows instead; "wait_retries" — launches
 * repeated on a non-waiting instance after a wait between workgroups expired; "early_settled" — uint8 engine: queries settled
 * from the first and last sub-chunks without a sweep; "beyond_f16" — sequences of a many-small-alignments batch whose maximum lay
 * beyond the packed float16 pass's key range and were redone on float32 cells.  MI355_SW_EINVAL for an unknown name. */
int mi355_sw_last_counter(const mi355_sw_ctx *ctx, const char *name, uint64_t *out);

/* Which kernels and pipeline decisions the last call used: space-separated tags, each at most once, e.g.
 * "score[cell=f16,SL=8,R=19,...,sampled=1,...] strip[R=3,mode=max,...] wave[orient=0,...,dirs=1,...] walk_wave" — what the parity
 * tests assert a switch of mi355_sw_set_option ENGAGED with.  Valid until the next call on the context; never NULL. */
const char *mi355_sw_last_path(const mi355_sw_ctx *ctx);

/* Which sw_score_kernel instance swept the most cells in the last call (what the `iterate` of
 * similaritymatrix.cpp:99-264 / :386-561 became for this input): reporting aid for drivers and bench.py, so
 * that nobody re-derives the library's choice.  cells == 0 when the call did not use the score kernel. */
enum {
  MI355_SW_CELL_I16 = 0,   /* two queries per register, packed 16-bit integers */
  MI355_SW_CELL_U8 = 1,    /* uint8 engine, packed 16-bit integers with explicit saturation */
  MI355_SW_CELL_F32 = 2,   /* one query per register, float32 cells scaled by 2^-k */
  MI355_SW_CELL_F32U8 = 3, /* uint8 engine rule on integer-valued float32 cells, one query per register */
  MI355_SW_CELL_F16 = 4,   /* two queries per register, packed float16 cells holding H / 2048 */
  MI355_SW_CELL_U8H = 5    /* uint8 engine, packed float16 cells holding (H + 1) / 256 */
};
typedef struct {
  int cell;                 /* MI355_SW_CELL_* */
  int lanes;                /* lanes per tile: 8, 16 or 64 */
  int rows_per_lane;        /* R */
  int strips;               /* query swept in strips of lanes * rows_per_lane rows */
  int twin;                 /* two tiles of one query per packed register */
  int64_t chunk_len;        /* own columns per tile */
  int64_t sub_len;          /* columns per reported sub-chunk maximum */
  int64_t warm;             /* warm-up columns in front of a tile */
  double cells;             /* cells this instance swept in the call */
  double valu_ops_per_cell; /* VALU instructions per cell and lane of the inner loop (cost model) */
  char name[160];           /* e.g. "sw_score_kernel<R=19, f16x2, SL=8>" */
} mi355_sw_kernel_info;
int mi355_sw_last_kernel(const mi355_sw_ctx *ctx, mi355_sw_kernel_info *out);

/* ---- several GPUs of one node behind one handle (one process, one engine context + host thread per device) ----
 * The reference spreads the same work over OpenMP threads (pieces, src/aligner/plocalaligner.cpp:110-129) and MPI ranks
 * (independent alignments, src/mpi_sw_solve_uniprot.cpp:95-138).  Results are identical to the single-device calls.
 * devices == NULL or ndev <= 0: all visible devices.  A device may be listed more than once (the contexts are
 * independent), except with MI355_SW_MULTI_RCCL. */
typedef struct mi355_sw_multi mi355_sw_multi;
enum {
  MI355_SW_MULTI_RCCL = 1  /* merge the per-device best keys with ncclAllReduce(ncclMax, ncclUint64) over xGMI (RCCL is
                              loaded at run time) instead of on the host; needs distinct devices */
};
int mi355_sw_multi_create(mi355_sw_multi **m, int ndev, const int *devices, int flags);
void mi355_sw_multi_destroy(mi355_sw_multi *m);
const char *mi355_sw_multi_last_error(const mi355_sw_multi *m);
int mi355_sw_multi_set_option(mi355_sw_multi *m, const char *key, const char *value);   /* mi355_sw_set_option on every device's context */
int mi355_sw_multi_device_count(const mi355_sw_multi *m);
int mi355_sw_multi_rccl_version(const mi355_sw_multi *m);   /* ncclGetVersion code, 0 without MI355_SW_MULTI_RCCL */

/* mi355_sw_align_split with piece p swept by device p mod ndev; the per-device best (score, lowest piece) keys are
 * merged by MAX (serial rule plocalaligner.cpp:122-129), the owner of the winning piece re-aligns it (:132-141).
 * Every device keeps a copy of y resident between calls (same reuse rule as mi355_sw_align). */
int mi355_sw_multi_align_split(mi355_sw_multi *m, const char *x, size_t nx, const char *y, size_t ny,
                               const mi355_sw_params *params, int sm_semantics, int la_semantics,
                               int npiece, float overlap_ratio, mi355_sw_result *out, int *winning_piece);

/* mi355_sw_set_reference / mi355_sw_align_batch with the reference replicated and the alignments dealt to the
 * devices by length; outs[k] belongs to xs[k].  best_index (may be NULL): the alignment with the highest score,
 * lowest index on ties, -1 for an empty batch — the "best over the database" of the UniProt-shaped run. */
int mi355_sw_multi_set_reference(mi355_sw_multi *m, const char *y, size_t ny);
int mi355_sw_multi_align_batch(mi355_sw_multi *m, size_t n, const char *const *xs, const size_t *nxs,
                               const mi355_sw_params *params, int flags, mi355_sw_result *outs, int64_t *best_index);
/* as mi355_sw_last_timings: [0..3] maximum over the devices (they run side by side), [4..5] sums */
int mi355_sw_multi_last_timings(const mi355_sw_multi *m, double out[6]);

void mi355_sw_free_result(mi355_sw_result *r);
void mi355_sw_free_results(mi355_sw_result *r, size_t n);

/* Build information: "gfx950;..." */
const char *mi355_sw_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355_SW_H_ */
