// mi355_sw.hip — host side of the C-ABI in include/mi355_sw.h (gfx950 only, no CPU fallback).
//
// Pipeline for every alignment:
//   1. score pass      sw_score_kernel  — packed 16-bit wavefront sweep over (query pair x chunk) tiles,
//                                         per-query (max, first sub-chunk) by 64-bit atomicMax      host_score.h
//   2. locate          sw_wave_kernel / sw_strip_kernel (track) — the sub-chunk(s) that can hold the first
//                                         maximum in the reference's storage order -> argmax cell   host_pipeline.h
//   3. traceback       sw_wave_kernel / sw_strip_kernel (dirs) — window left of the argmax -> greedy decisions,
//                      sw_wave_walk_kernel — the walk itself (smithwaterman.cpp:40-78)            host_wave.h
//   (sw_exact_kernel + sw_walk_kernel, host_exact.h: table scoring on short queries, whole uint8 problems)
// Problems the score kernel does not cover (see bucket_fast_ok) run 2+3 on the whole matrix.
// The host code is one translation unit; the fragments below are included in order.
#include "../../include/mi355_sw.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <memory>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "sw_exact_kernel.h"
#include "sw_score_kernel.h"
#include "sw_wave_kernel.h"
#include "sw_strip_kernel.h"
#include "sw_batch_kernels.h"
#include "sw_solo_kernel.h"
#include "sw_long_kernel.h"

using namespace mi355sw;

#include "host_common.h"
#include "host_score.h"
#include "host_exact.h"
#include "host_wave.h"
#include "host_batch.h"
#include "host_saved.h"
#include "host_pipeline.h"
#include "host_solo.h"
#include "host_multi.h"   // mi355_sw_multi_*: its own extern "C" block

// ================================= C-ABI ======================================================
extern "C" {

const char *mi355_sw_build_info(void) { return "mi355_sw gfx950 hip; score kernel 16-lane R={2..32}, 8-lane R={7..32}, 64-lane R={16,32} (+strips, twin) x {f16 pairs, u8 as f16 pairs, i16 pairs, u8sat pairs, f32}; wave/strip/exact kernels"; }

void mi355_sw_default_params(mi355_sw_params *p) {
  if (!p) return;
  p->lut = nullptr; p->match = 3.0f; p->mismatch = -3.0f; p->gap = 2.0f; p->semantics = MI355_SW_F32;
}

int mi355_sw_create(mi355_sw_ctx **out, int device) {
  if (!out) return MI355_SW_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MI355_SW_ENODEV;
  if (hipSetDevice(device) != hipSuccess) return MI355_SW_ENODEV;
  mi355_sw_ctx *c = new (std::nothrow) mi355_sw_ctx();
  if (!c) return MI355_SW_ENOMEM;
  c->device = device;
  c->opts = options_from_env();
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      if (prop.multiProcessorCount > 0) c->dev.cus = prop.multiProcessorCount;
      if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) c->dev.lds = (size_t)prop.maxSharedMemoryPerMultiProcessor;
    }
  }
  if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return MI355_SW_ENODEV; }
  if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return MI355_SW_ENODEV; }
  for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return MI355_SW_ENODEV; }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<0, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<1, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  *out = c;
  return 0;
}

void mi355_sw_destroy(mi355_sw_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  DevBuf *bufs[] = {&c->qcnt, &c->sel2, &c->gcnt, &c->wlut, &c->ref.bytes, &c->ref.codes, &c->batch.bytes, &c->batch.lens, &c->keys, &c->ranges, &c->stab,
                    &c->batch.offs, &c->batch.sel, &c->colsave, &c->rowsave, &c->pieces, &c->ftab, &c->ftab_s, &c->htab, &c->htab8, &c->soloblk, &c->flags, &c->submax, &c->lut, &c->probs, &c->dirs, &c->outs_f, &c->outs_i, &c->cons, &c->walkp, &c->hmat, &c->brow, &c->wprobs, &c->scan, &c->batch.cum, &c->ckpt, &c->first};
  for (DevBuf *b : bufs) b->release();
  c->adhoc.release(); c->one.release();
  c->pin_probs.release(); c->pin_walk.release(); c->pin_out.release(); c->pin_solo_up.release(); c->pin_solo_down.release();
  for (PinBuf &b : c->pin_cons) b.release();
  for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->score_ev) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *mi355_sw_last_error(const mi355_sw_ctx *c) { return c ? c->err.c_str() : "null context"; }

int mi355_sw_set_option(mi355_sw_ctx *ctx, const char *key, const char *value) {
  if (!ctx || !key) return MI355_SW_EINVAL;
  if (option_set(ctx->opts, key, value)) return fail(ctx, MI355_SW_EINVAL, std::string("unknown option: ") + key);
  return 0;
}

const char *mi355_sw_option_names(void) { return option_names(); }

int mi355_sw_set_reference(mi355_sw_ctx *ctx, const char *y, size_t ny) {
  OptScope opt_scope_(ctx);
  if (!ctx || (!y && ny)) return MI355_SW_EINVAL;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return upload_reference(ctx, ctx->ref, y, ny);
}

int mi355_sw_batch_upload(mi355_sw_ctx *ctx, size_t n, const char *const *xs, const size_t *nxs) {
  OptScope opt_scope_(ctx);
  if (!ctx || !xs || !nxs) return MI355_SW_EINVAL;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return upload_queries(ctx, ctx->batch, n, xs, nxs);
}

int mi355_sw_batch_upload_packed(mi355_sw_ctx *ctx, size_t n, const char *buf, const int64_t *offsets) {
  OptScope opt_scope_(ctx);
  if (!ctx || (n && (!buf || !offsets))) return MI355_SW_EINVAL;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return upload_queries_packed(ctx, ctx->batch, n, buf, offsets);
}

int mi355_sw_batch_run(mi355_sw_ctx *ctx, const mi355_sw_params *params, int flags, mi355_sw_result *outs) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!outs) return fail(ctx, MI355_SW_EINVAL, "outs is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  if (ctx->batch.nq == 0) return 0;
  return align_range(ctx, ctx->ref, ctx->batch, Range{0, (int64_t)ctx->ref.n}, *params, flags, outs);
}

int mi355_sw_batch_run_view(mi355_sw_ctx *ctx, const mi355_sw_params *params, int flags, mi355_sw_batch_view *out) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!out) return fail(ctx, MI355_SW_EINVAL, "out is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  memset(out, 0, sizeof *out);
  if (ctx->batch.nq == 0) return 0;
  return align_range_view(ctx, ctx->ref, ctx->batch, Range{0, (int64_t)ctx->ref.n}, *params, flags, out);
}

int mi355_sw_align_batch(mi355_sw_ctx *ctx, size_t n, const char *const *xs, const size_t *nxs,
                         const mi355_sw_params *params, int flags, mi355_sw_result *outs) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  rc = mi355_sw_batch_upload(ctx, n, xs, nxs);
  if (rc) return rc;
  return mi355_sw_batch_run(ctx, params, flags, outs);
}

int mi355_sw_align(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                   const mi355_sw_params *params, mi355_sw_result *out) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!out || (!x && nx) || (!y && ny)) return fail(ctx, MI355_SW_EINVAL, "null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  const RefData *ref = nullptr;
  AdhocSpeculation sp;
  memset(out, 0, sizeof *out);
  rc = adhoc_begin(ctx, y, ny, &ref, sp);
  // one short read against a long reference: the single-alignment chain (host_solo.h); everything else, or whatever
  // that chain declines, takes the general pipeline
  auto work = [&]() -> int {
    int r = solo_align(ctx, *ref, x, nx, std::vector<Range>{Range{0, (int64_t)ny}}, *params, true, out);
    if (r <= 0) return r;
    reset_timings(ctx);
    r = upload_queries(ctx, ctx->one, 1, &x, &nx);
    if (!r) r = align_range(ctx, *ref, ctx->one, Range{0, (int64_t)ny}, *params, 0, out);
    return r;
  };
  if (!rc) rc = work();
  int rc2 = 0;
  if (!adhoc_confirm(ctx, y, ny, &ref, sp, rc2)) {             // the caller's buffer changed since the last call
    mi355_sw_free_result(out);
    reset_timings(ctx);
    rc = rc2;
    if (!rc) rc = work();
  }
  return rc;
}

int mi355_sw_argmax(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                    const mi355_sw_params *params, int64_t *index_x, int64_t *index_y, float *mx) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  mi355_sw_result r;
  memset(&r, 0, sizeof r);
  const RefData *ref = nullptr;
  AdhocSpeculation sp;
  rc = adhoc_begin(ctx, y, ny, &ref, sp);
  auto work = [&]() -> int {
    int r2 = solo_align(ctx, *ref, x, nx, std::vector<Range>{Range{0, (int64_t)ny}}, *params, false, &r);
    if (r2 <= 0) return r2;
    reset_timings(ctx);
    r2 = upload_queries(ctx, ctx->one, 1, &x, &nx);
    if (!r2) r2 = align_range(ctx, *ref, ctx->one, Range{0, (int64_t)ny}, *params, MI355_SW_SCORE_ONLY, &r);
    return r2;
  };
  if (!rc) rc = work();
  int rc2 = 0;
  if (!adhoc_confirm(ctx, y, ny, &ref, sp, rc2)) {
    mi355_sw_free_result(&r);
    reset_timings(ctx);
    rc = rc2;
    if (!rc) rc = work();
  }
  if (rc) return rc;
  if (index_x) *index_x = r.end_x;
  if (index_y) *index_y = r.end_y;
  if (mx) *mx = r.score;
  mi355_sw_free_result(&r);
  return 0;
}

int mi355_sw_make_string_range(int npiece, int64_t shortlen, int64_t longlen, float overlap_ratio,
                               int64_t *lefts, int64_t *rights) {
  // plocalaligner.cpp:44-67
  if (npiece < 1 || !lefts || !rights) return MI355_SW_EINVAL;
  const int64_t overlap = (int64_t)((float)shortlen * overlap_ratio);
  if (npiece == 1) { lefts[0] = 0; rights[0] = longlen; return 0; }
  const int64_t piecelen = (longlen + (int64_t)(npiece - 1) * overlap) / npiece;
  if (!(overlap <= piecelen)) return MI355_SW_ERANGE;
  int64_t left = 0, right = piecelen;
  int k = 0;
  lefts[k] = left; rights[k] = right; ++k;
  while (k < npiece - 1) {
    left = std::max<int64_t>(0, right - overlap);
    right = std::min(left + piecelen, longlen);
    lefts[k] = left; rights[k] = right; ++k;
  }
  if (!(right < longlen)) return MI355_SW_ERANGE;
  lefts[k] = std::max<int64_t>(0, right - overlap); rights[k] = longlen;
  return 0;
}

int mi355_sw_align_split(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                         const mi355_sw_params *params, int sm_semantics, int la_semantics,
                         int npiece, float overlap_ratio, mi355_sw_result *out, int *winning_piece) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!out || npiece < 1) return fail(ctx, MI355_SW_EINVAL, "bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  std::vector<int64_t> lefts(npiece), rights(npiece);
  rc = mi355_sw_make_string_range(npiece, (int64_t)nx, (int64_t)ny, overlap_ratio, lefts.data(), rights.data());
  if (rc) return fail(ctx, rc, "_make_string_range: the reference's asserts would fire for these arguments");
  const RefData *refp = nullptr;
  AdhocSpeculation sp;
  memset(out, 0, sizeof *out);
  rc = adhoc_begin(ctx, y, ny, &refp, sp);
  QueryBatch &q = ctx->one;
  int bp = 0;
  // default scoring in both roles (what sw_solve_small.cpp:82 / sw_solve_big.cpp:78 construct): the per-piece maxima and
  // the winner's re-alignment are the same sweep -> the single-alignment chain over all pieces (host_solo.h)
  const bool same_sweep = params->lut == nullptr && params->match == 3.0f && params->mismatch == -3.0f && params->gap == 2.0f &&
                          sm_semantics == la_semantics;
  auto work = [&]() -> int {
    mi355_sw_params ps = *params;
    ps.semantics = sm_semantics;
    std::vector<Range> ranges(npiece);
    for (int k = 0; k < npiece; ++k) ranges[k] = Range{lefts[k], rights[k]};
    if (same_sweep) {
      int r = solo_align(ctx, *refp, x, nx, ranges, ps, true, out, &bp);
      if (r < 0) return r;
      if (r == 0) {
        if (out->score > 0) { out->pos += (uint32_t)lefts[bp]; out->end_y += lefts[bp]; }
        else out->pos = (uint32_t)lefts[bp];
        out->timings_us[1] = out->timings_us[0];
        return 0;
      }
      reset_timings(ctx);
    }
    int r = upload_queries(ctx, q, 1, &x, &nx);
    if (r) return r;
    // All the reference does with the per-piece maxima is pick the first piece with the strictly greatest one
    // (plocalaligner.cpp:106,122-129), so the sweep is the winner-only one of mi355_sw_best_range: a lone long query on the sampled
    // maximum behind an optimistic warm-up margin, certified by the call itself (range_maxima; pieces that tie with the
    // winner come out exact, so the first-piece rule holds).
    std::vector<float> pmax(npiece, 0.0f);
    r = range_maxima(ctx, *refp, q, ranges, ps, pmax.data(), true);
    if (r) return r;
    float best = -1.0f;                                  // plocalaligner.cpp:106,122-129
    bp = 0;
    for (int k = 0; k < npiece; ++k) if (pmax[k] > best) { best = pmax[k]; bp = k; }
    mi355_sw_params pd;
    mi355_sw_default_params(&pd);                        // LAT(x, piece): default scoring (:135)
    pd.semantics = la_semantics;
    const double t_score = ctx->timings[0];
    // The reference sweeps the winning piece a second time (:132-136).  With default scoring in both roles that sweep is the one
    // just made: the winner is finished from its keys (argmax window + traceback only), as mi355_sw_align_scored_range does.
    const ScoredRanges &sc = ctx->scored;
    const bool from_keys = same_sweep && sc.valid && sc.ref == (const void *)refp && sc.batch == (const void *)&q &&
                           (size_t)bp < sc.ranges.size() && (!sc.sampled || sc.has_located[bp]);
    r = align_range(ctx, *refp, q, ranges[bp], pd, 0, out, from_keys ? &sc : nullptr, (size_t)bp);
    if (r) return r;
    if (out->score > 0) { out->pos += (uint32_t)lefts[bp]; out->end_y += lefts[bp]; }
    else out->pos = (uint32_t)lefts[bp];
    out->timings_us[0] = (float)t_score;
    out->timings_us[1] = (float)t_score;
    return 0;
  };
  if (!rc) rc = work();
  int rc2 = 0;
  if (!adhoc_confirm(ctx, y, ny, &refp, sp, rc2)) {             // the caller's buffer changed since the last call
    mi355_sw_free_result(out);
    reset_timings(ctx);
    rc = rc2;
    if (!rc) rc = work();
  }
  if (winning_piece) *winning_piece = bp;
  return rc;
}

int mi355_sw_score_ranges(mi355_sw_ctx *ctx, size_t nranges, const int64_t *lefts, const int64_t *rights,
                          const mi355_sw_params *params, float *maxima) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!lefts || !rights || !maxima) return fail(ctx, MI355_SW_EINVAL, "null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  const QueryBatch &q = ctx->batch;
  if (q.nq == 0 || nranges == 0) return 0;
  std::vector<Range> ranges(nranges);
  int64_t maxlen = 0;
  for (size_t k = 0; k < nranges; ++k) {
    if (lefts[k] < 0 || rights[k] < lefts[k] || rights[k] > (int64_t)ctx->ref.n) return fail(ctx, MI355_SW_EINVAL, "range outside the resident reference");
    ranges[k] = Range{lefts[k], rights[k]};
    maxlen = std::max(maxlen, rights[k] - lefts[k]);
  }
  (void)maxlen;
  return range_maxima(ctx, ctx->ref, q, ranges, *params, maxima);
}

int mi355_sw_best_range(mi355_sw_ctx *ctx, size_t nranges, const int64_t *lefts, const int64_t *rights,
                        const mi355_sw_params *params, float known_best, float *maxima, float *best, int64_t *best_range,
                        float *exact_above) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!lefts || !rights || !best || !best_range) return fail(ctx, MI355_SW_EINVAL, "null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  const QueryBatch &q = ctx->batch;
  if (q.nq == 0) return 0;
  for (size_t k = 0; k < q.nq; ++k) { best[k] = -1.0f; best_range[k] = -1; }
  if (nranges == 0) return 0;
  std::vector<Range> ranges(nranges);
  for (size_t k = 0; k < nranges; ++k) {
    if (lefts[k] < 0 || rights[k] < lefts[k] || rights[k] > (int64_t)ctx->ref.n) return fail(ctx, MI355_SW_EINVAL, "range outside the resident reference");
    ranges[k] = Range{lefts[k], rights[k]};
  }
  std::vector<float> own;
  if (!maxima) { own.resize(nranges * q.nq); maxima = own.data(); }
  rc = range_maxima(ctx, ctx->ref, q, ranges, *params, maxima, true, known_best > 0.0f ? known_best : 0.0f, exact_above);
  if (rc) return rc;
  for (size_t k = 0; k < q.nq; ++k)                               // plocalaligner.cpp:106,122-129: starts at -1, strict '>'
    for (size_t r = 0; r < nranges; ++r)
      if (maxima[r * q.nq + k] > best[k]) { best[k] = maxima[r * q.nq + k]; best_range[k] = (int64_t)r; }
  return 0;
}

int mi355_sw_align_scored_range(mi355_sw_ctx *ctx, size_t range_index, const mi355_sw_params *params, int flags,
                                mi355_sw_result *outs) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!outs) return fail(ctx, MI355_SW_EINVAL, "outs is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const ScoredRanges &sc = ctx->scored;
  if (!sc.valid || sc.ref != (const void *)&ctx->ref || sc.batch != (const void *)&ctx->batch || sc.ref_version != ctx->ref.version ||
      sc.batch_version != ctx->batch.version || range_index >= sc.ranges.size())
    return fail(ctx, MI355_SW_EINVAL, "mi355_sw_align_scored_range: no mi355_sw_score_ranges call on this reference and batch covers that range");
  reset_timings(ctx);
  if (ctx->batch.nq == 0) return 0;
  const Range rg = sc.ranges[range_index];
  // the sweep's keys serve when it ran under the same engine and scoring; otherwise the range is swept again
  // ... and, after a winner-only sweep (mi355_sw_best_range), for the ranges that were re-evaluated exactly
  const bool same = params->lut == nullptr && params->semantics == sc.params.semantics && params->match == sc.params.match &&
                    params->mismatch == sc.params.mismatch && params->gap == sc.params.gap &&
                    (!sc.sampled || sc.has_located[range_index]);
  return align_range(ctx, ctx->ref, ctx->batch, rg, *params, flags, outs, same ? &sc : nullptr, range_index);
}

int mi355_sw_fill_matrix(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                         const mi355_sw_params *params, float *H) {
  OptScope opt_scope_(ctx);
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!H) return fail(ctx, MI355_SW_EINVAL, "H is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t cells = (nx + 1) * (ny + 1);
  if (cells * 4 > (size_t)8 << 30) return fail(ctx, MI355_SW_ENOTSUP, "matrix larger than 8 GiB");
  if (nx == 0 || ny == 0) { memset(H, 0, cells * 4); return 0; }
  RefData ref;
  QueryBatch q;
  rc = upload_reference(ctx, ref, y, ny);
  if (!rc) rc = upload_queries(ctx, q, 1, &x, &nx);
  if (!rc && ctx->hmat.ensure(cells * 4)) rc = fail(ctx, MI355_SW_ENOMEM, "hipMalloc(matrix) failed");
  if (!rc) {
    if (hipMemsetAsync(ctx->hmat.p, 0, cells * 4, ctx->stream) != hipSuccess) rc = fail(ctx, MI355_SW_ENODEV, "hipMemsetAsync failed");
  }
  if (!rc) {
    std::vector<ExactJob> jobs(1);
    ExactJob &j = jobs[0];
    j.q = 0; j.ylo = 0; j.nw = (int32_t)ny; j.col_offset = 0; j.full_n = (int64_t)ny; j.own_lo = 1;
    j.quirk = (params->semantics == MI355_SW_U8SAT && nx == ny) ? 1 : 0;
    j.target = -1.0f; j.want_dirs = false;
    rc = run_exact(ctx, ref, q, *params, jobs, 0, 1, ctx->hmat.as<float>());
  }
  if (!rc && hipMemcpy(H, ctx->hmat.p, cells * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, MI355_SW_ENODEV, "hipMemcpy(matrix) failed");
  ref.release(); q.release();
  return rc;
}

void mi355_sw_true2raw(size_t nx, size_t ny, size_t ti, size_t tj, size_t *ri, size_t *rj) {
  // similaritymatrix.cpp:353-364 with len_x = |y|+1, len_y = |x|+1 (constructor swap, :274-285)
  const size_t len_x = ny + 1, len_y = nx + 1;
  const size_t nrows = std::min(len_x, len_y), ncols = std::max(len_x, len_y);
  if (ti + tj < nrows - 1) { *ri = ti; *rj = ti + tj; }
  else if (ti + tj > ncols - 1) { *ri = ti - ncols + len_y; *rj = ti + tj - (ncols - 1) - 1; }
  else { *ri = (len_x <= len_y) ? ti : len_y - 1 - tj; *rj = ti + tj; }
}

void mi355_sw_raw2true(size_t nx, size_t ny, size_t ri, size_t rj, size_t *ti, size_t *tj) {
  // similaritymatrix.cpp:330-346
  const size_t len_x = ny + 1, len_y = nx + 1;
  const size_t nrows = std::min(len_x, len_y);
  if (rj < nrows - 1) {
    if (ri <= rj) { *ti = ri; *tj = rj - ri; }
    else { *ti = len_x - nrows + ri; *tj = len_y - ri + rj; }
  } else if (len_x <= len_y) { *ti = ri; *tj = rj - ri; }
  else { *ti = rj - (nrows - 1) + ri; *tj = nrows - 1 - ri; }
}

int mi355_sw_last_timings(const mi355_sw_ctx *ctx, double out[6]) {
  if (!ctx || !out) return MI355_SW_EINVAL;
  for (int k = 0; k < 6; ++k) out[k] = ctx->timings[k];
  return 0;
}

int mi355_sw_last_counters(const mi355_sw_ctx *ctx, uint64_t out[4]) {
  if (!ctx || !out) return MI355_SW_EINVAL;
  out[0] = ctx->requeried; out[1] = ctx->whole_again; out[2] = ctx->candidates; out[3] = ctx->left_window;
  return 0;
}

int mi355_sw_last_counter(const mi355_sw_ctx *ctx, const char *name, uint64_t *out) {
  if (!ctx || !name || !out) return MI355_SW_EINVAL;
  const std::string k(name);
  if (k == "requeried") *out = ctx->requeried;
  else if (k == "whole_batch_again") *out = ctx->whole_again;
  else if (k == "candidates") *out = ctx->candidates;
  else if (k == "left_window") *out = ctx->left_window;
  else if (k == "beyond_f16") *out = ctx->beyond_f16;
  else if (k == "first_settled") *out = ctx->first_settled;
  else if (k == "wait_retries") *out = ctx->wait_retries;
  else if (k == "early_settled") *out = ctx->early_settled;
  else if (k == "saved_locates") *out = ctx->saved_locates;
  else if (k == "saved_traces") *out = ctx->saved_traces;
  else if (k == "saved_fallbacks") *out = ctx->saved_fallbacks;
  else return MI355_SW_EINVAL;
  return 0;
}

const char *mi355_sw_last_path(const mi355_sw_ctx *ctx) { return ctx ? ctx->path.c_str() : ""; }

int mi355_sw_last_kernel(const mi355_sw_ctx *ctx, mi355_sw_kernel_info *out) {
  if (!ctx || !out) return MI355_SW_EINVAL;
  *out = ctx->last_kernel;
  return 0;
}

void mi355_sw_free_result(mi355_sw_result *r) {
  if (!r) return;
  free(r->cons_x);                    // cons_y points into the same allocation
  r->cons_x = r->cons_y = nullptr; r->cons_len = 0;
}

void mi355_sw_free_results(mi355_sw_result *r, size_t n) {
  if (!r) return;
  for (size_t k = 0; k < n; ++k) mi355_sw_free_result(r + k);
}

}  // extern "C"

