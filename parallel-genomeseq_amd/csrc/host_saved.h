// host_saved.h — the finish of a lone long query from what its score sweep saved (sw_long_kernel.h colsave / rowsave).
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
//
// The sweep of a 10 kbp query passes every cell of the matrix; locate and traceback used to recompute windows of it behind
// a zero border and a warm-up margin of their own (10-25 k columns in front of a 2 k-column window, swept by ONE pipeline of
// wavefronts: 4.3 + 8.0 ms of config 5's 214 ms, and a per-job constant on the owner's rank of the reference-sharded form,
// src/aligner/plocalaligner.cpp:132-141).  With the H column in front of every sub-chunk and the bottom row of every strip in
// HBM, a BLOCK — one strip's rows x one sub-chunk's columns — is an independent exact problem with known left column and top
// row: the candidate window of the locate step is ceil(|x| / (64 R)) blocks, the decision window of the traceback one block
// per (strip, sub-chunk) it covers, all resident at once on different CUs, each as long as ONE block's sweep.
//
// Exactness (DESIGN.md §3.3 lemma L9, with L2-L4): a saved value is what the sweep computed, i.e. the cell of a window whose zero
// border is the sweep tile's (column T * chunk - warm; tile 0: the matrix border).  Every margin argument therefore holds
// with that border in place of the window's own: locate takes a candidate only when the tile's margin covers the
// score-aware margin of locate_saturated, the walk kernel checks every cell it visits against the tile border
// (WaveWalk::zchunk), and whatever fails these checks takes the zero-border path of host_pipeline.h / host_wave.h.
namespace {

// rows per lane of the strip-kernel instance whose strips tile a block of 64 * R_long rows exactly
inline int saved_block_R(int SR) { return SR % (64 * 5) == 0 ? 5 : 8; }

// index of `rg` among the ranges of the launch that saved, -1 when the saved state does not belong to (ref, q, p, rg, qid)
int saved_range_index(const mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p, int qid) {
  const LongSaved &ls = ctx->lsaved;
  if (!ls.valid || opt().no_long_save || ls.ref != (const void *)&ref || ls.batch != (const void *)&q || ls.ref_version != ref.version ||
      ls.batch_version != q.version || ls.qid != qid)
    return -1;
  if (p.lut != nullptr || p.semantics != ls.params.semantics || p.match != ls.params.match || p.mismatch != ls.params.mismatch || p.gap != ls.params.gap)
    return -1;
  if (64 * ls.R % (64 * saved_block_R(64 * ls.R)) != 0) return -1;
  for (size_t r = 0; r < ls.ranges.size(); ++r)
    if (ls.ranges[r].lo == rg.lo && ls.ranges[r].hi == rg.hi) return (int)r;
  return -1;
}

// Launches the blocks (StripProblems with init / top set by the caller) on sw_strip_kernel<BR>, one workgroup per block.
// maxmode: every block reports (maximum, first cell in storage order); else decisions.  best / cells: per block.
int run_saved_blocks(mi355_sw_ctx *ctx, const mi355_sw_params &p, int BR, int nwb, std::vector<StripProblem> &pr, bool maxmode,
                     std::vector<float> &best, std::vector<int64_t> &cells) {
  HostTrace trace_("run_saved_blocks");
  const size_t n = pr.size();
  if (n == 0) return 0;
  if (ctx->wprobs.ensure(n * sizeof(StripProblem)) || ctx->outs_i.ensure(n * 16) || ctx->outs_f.ensure(2 * n * 4))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(block scratch) failed");
  for (size_t k = 0; k < n; ++k) {
    pr[k].cell = ctx->outs_i.as<int64_t>() + 2 * k;
    pr[k].status = ctx->outs_f.as<int32_t>() + k;
    pr[k].best = ctx->outs_f.as<float>() + n + k;
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->outs_f.p, 0, 2 * n * 4, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->wprobs.p, pr.data(), n * sizeof(StripProblem), hipMemcpyHostToDevice, ctx->stream));
  WaveScoring sc;
  sc.match = p.match; sc.mismatch = p.mismatch; sc.gap = p.gap;
  const U8Params u = u8_params(p);
  sc.u8M = (float)u.M; sc.u8X = (float)u.X; sc.u8G = (float)u.G;
  const StripProblem *dp = ctx->wprobs.as<StripProblem>();
  const dim3 grid((unsigned)n), block((unsigned)(64 * nwb));
  if (BR == 5) launch_strip<5>(false, maxmode, grid, block, ctx->stream, dp, sc, nullptr, 0, 1, maxmode);
  else launch_strip<8>(false, maxmode, grid, block, ctx->stream, dp, sc, nullptr, 0, 1, maxmode);
  HIPCHK(ctx, hipGetLastError());
  std::vector<int32_t> st(n);
  best.assign(maxmode ? n : 0, 0.0f);
  cells.assign(maxmode ? 2 * n : 0, 0);
  HIPCHK(ctx, hipMemcpyAsync(st.data(), ctx->outs_f.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (maxmode) {
    HIPCHK(ctx, hipMemcpyAsync(best.data(), ctx->outs_f.as<float>() + n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(cells.data(), ctx->outs_i.p, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; ++k)
    if (st[k] != 0) return fail(ctx, MI355_SW_ENODEV, "internal: strip pipeline wait expired");
  return 0;
}

// The blocks of one column tile [c_lo, c_hi) (range-relative, 0-based) of range r for rows < row_end.
void saved_tile_blocks(const mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, int r, int qid, int64_t sgl,
                       int64_t c_lo, int64_t c_hi, int64_t row_end, std::vector<StripProblem> &pr) {
  const LongSaved &ls = ctx->lsaved;
  const int m = q.len[qid];
  const int SR = 64 * ls.R, BR = saved_block_R(SR);
  const int64_t T = sgl / ls.spt;
  const float scale = std::ldexp(1.0f, ls.fshift);
  const float *colp = sgl > 0 ? ctx->colsave.as<float>() + ((size_t)r * (size_t)ls.col_subs + (size_t)sgl) * (size_t)ls.col_rows : nullptr;
  for (int row0 = 0; row0 < m && row0 < row_end; row0 += SR) {
    StripProblem s = {};
    s.a = q.bytes.as<uint8_t>() + q.off[qid] + row0;
    s.na = std::min(SR, m - row0);
    s.b = ref.bytes.as<uint8_t>() + rg.lo + c_lo;
    s.nb = (int32_t)(c_hi - c_lo);
    s.nstrips = (s.na + 64 * BR - 1) / (64 * BR);
    s.nw = s.nstrips;
    s.target = -1.0f; s.own_lo = 0;
    s.col_offset = c_lo;
    s.full_n = rg.hi - rg.lo;
    s.init = colp ? colp + row0 : nullptr;
    const int rb = row0 / SR;
    s.top = rb > 0 ? ctx->rowsave.as<float>() + (((size_t)r * (size_t)ls.tiles_stride + (size_t)T) * (size_t)(ls.nstrips - 1) + (size_t)(rb - 1)) * (size_t)ls.row_stride +
                         (size_t)(c_lo - T * ls.chunk + ls.warm)
                   : nullptr;
    s.in_scale = scale;
    s.row0 = row0;
    s.na_full = m;
    pr.push_back(s);
  }
}

// locate_saturated's job for ONE lone long query from the saved state: the maximum over the listed sub-chunks of range r and
// its first cell in storage order.  `qlower`: the sweep's key (a lower bound of the maximum).  Sub-chunks whose tile margin
// does not cover the score-aware margin are returned in `rest` (the caller runs them on zero-border windows).
int locate_from_saved(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p, int r, int qid,
                      float qlower, const ScoreTable &table, const std::vector<uint32_t> &subs, Located &loc, bool &found,
                      std::vector<uint32_t> &rest) {
  HostTrace trace_("locate_from_saved");
  const LongSaved &ls = ctx->lsaved;
  const int m = q.len[qid];
  const int64_t n = rg.hi - rg.lo;
  const int SR = 64 * ls.R, BR = saved_block_R(SR);
  int64_t need = kColsMax;                                           // locate_saturated's score-aware margin
  const Margin mg = table.margin(m);
  if (mg.finite()) {
    need = mg.cols(m);
    if (qlower > 0) need = std::min<int64_t>(need, clamp_cols((double)m + std::ceil(std::max(0.0, mg.smax * (double)m - (double)qlower) / mg.g) + 2.0));
  }
  std::vector<StripProblem> pr;
  for (uint32_t sgl : subs) {
    const int64_t c_lo = std::max<int64_t>(0, (int64_t)sgl * ls.sub_len - 63), c_hi = std::min(((int64_t)sgl + 1) * ls.sub_len, n);
    if (c_lo >= c_hi) continue;
    const int64_t T = (int64_t)sgl / ls.spt, b = (int64_t)sgl % ls.spt;
    if (T > 0 && ls.warm + b * ls.sub_len - 63 < need) { rest.push_back(sgl); continue; }   // the tile's margin in front of this window
    saved_tile_blocks(ctx, ref, q, rg, r, qid, sgl, c_lo, c_hi, m, pr);
  }
  if (pr.empty()) return 0;
  std::vector<float> best;
  std::vector<int64_t> cells;
  int rc = run_saved_blocks(ctx, p, BR, SR / (64 * BR), pr, true, best, cells);
  if (rc) return rc;
  unsigned long long bkey = ~0ull;
  if (found) bkey = host_order_key(p.semantics, loc.ix, loc.iy, m, n);
  for (size_t k = 0; k < pr.size(); ++k) {
    if (!(best[k] > 0) || cells[2 * k] <= 0) continue;
    const unsigned long long kk = host_order_key(p.semantics, cells[2 * k], cells[2 * k + 1], m, n);
    if (!found || best[k] > loc.score || (best[k] == loc.score && kk < bkey)) { loc.score = best[k]; loc.ix = cells[2 * k]; loc.iy = cells[2 * k + 1]; bkey = kk; found = true; }
  }
  ctx->saved_locates += 1;
  path_note(ctx, "saved_locate[R=%d]", BR);
  return 0;
}

// Traceback of the located alignment of a lone long query from the saved state: decisions of every block the walk can reach
// (all in flight at once), then the walk.  Returns 0 (out filled), 1 (the walk left the window or an exact zone: the caller
// takes the zero-border path), < 0 error.
int trace_from_saved(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p, int r, int qid,
                     const Located &loc, const ScoreTable &table, TraceOut &out) {
  HostTrace trace_("trace_from_saved");
  const LongSaved &ls = ctx->lsaved;
  const int m = q.len[qid];
  const int64_t ix = loc.ix, iy = loc.iy;
  if (!(loc.score > 0) || ix < 1 || iy < 1 || ix > m) return 1;
  const Margin mg = table.margin(m);
  if (!mg.finite()) return 1;
  const int SR = 64 * ls.R, BR = saved_block_R(SR), nwb = SR / (64 * BR);
  const int64_t budget = (int64_t)m / 8 + 64;
  const int64_t wl0 = std::max<int64_t>(0, iy - (ix + budget));
  const int64_t sgl_first = (wl0 + 63) / ls.sub_len, sgl_last = (iy - 1 + 63) / ls.sub_len;
  const int64_t wstart = std::max<int64_t>(0, sgl_first * ls.sub_len - 63);
  const int64_t nb = iy - wstart;
  const int nrb = (m + SR - 1) / SR;
  const int LT = nrb * (SR / BR);                                      // lanes of a decision row: the full problem's
  const size_t dbytes = (size_t)nb * (size_t)LT * 4 + 64;
  if (dbytes > kDirsBudget || ctx->dirs.ensure(dbytes)) return 1;
  std::vector<StripProblem> pr;
  for (int64_t sgl = sgl_first; sgl <= sgl_last; ++sgl) {
    const int64_t c_lo = std::max<int64_t>(0, sgl * ls.sub_len - 63), c_hi = std::min((sgl + 1) * ls.sub_len - 63, iy);
    if (c_lo >= c_hi) continue;
    const size_t first = pr.size();
    saved_tile_blocks(ctx, ref, q, rg, r, qid, sgl, c_lo, c_hi, ix, pr);
    for (size_t k = first; k < pr.size(); ++k) {
      pr[k].dirs = ctx->dirs.as<uint32_t>() + (size_t)(c_lo - wstart) * (size_t)LT + (size_t)(pr[k].row0 / BR);
      pr[k].lt = LT;
    }
  }
  std::vector<float> best;
  std::vector<int64_t> cells;
  int rc = run_saved_blocks(ctx, p, BR, nwb, pr, false, best, cells);
  if (rc) return rc;
  // the walk: one wavefront, looking ahead along the diagonal (sw_wave_walk_long_kernel)
  const int cap = m + (int)std::min<int64_t>(nb, INT32_MAX / 4) + 2;
  if (ctx->walkp.ensure(sizeof(WaveWalk) + 64) || ctx->pin_walk.ensure(sizeof(WaveWalk)) || ctx->pin_out.ensure(64) || ctx->cons.ensure(2 * (size_t)cap + 16))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(walk scratch) failed");
  WaveWalk &w = *ctx->pin_walk.as<WaveWalk>();
  int64_t *wout = reinterpret_cast<int64_t *>(ctx->walkp.as<uint8_t>() + sizeof(WaveWalk));
  int64_t *woffs = wout + 3;
  w.x = q.bytes.as<uint8_t>() + q.off[qid];
  w.y = ref.bytes.as<uint8_t>() + rg.lo;
  w.dirs = ctx->dirs.as<uint32_t>();
  w.na = m; w.nb = (int32_t)nb; w.orient = 0; w.R = BR; w.lanes = LT; w.skew = 0; w.row0 = 0;
  w.need_slope = (float)mg.slope();
  w.b_offset = wstart;
  w.start_i = ix; w.start_j = iy;
  w.exact_from = 0;
  w.cap = cap;
  w.out = wout;
  w.zchunk = ls.chunk; w.zwarm = ls.warm;
  int64_t *wo = ctx->pin_out.as<int64_t>();
  wo[3] = 0;
  HIPCHK(ctx, hipMemcpyAsync(ctx->walkp.p, &w, sizeof(WaveWalk), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(woffs, wo + 3, 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(sw_wave_walk_long_kernel, dim3(1), dim3(64), 0, ctx->stream, (const WaveWalk *)ctx->walkp.as<WaveWalk>(), 1,
                     ctx->cons.as<char>(), (const int64_t *)woffs);
  HIPCHK(ctx, hipGetLastError());
  if (ctx->pin_cons.size() <= ctx->cons_used) ctx->pin_cons.resize(ctx->cons_used + 1);
  PinBuf &cons = ctx->pin_cons[ctx->cons_used];
  if (cons.ensure(2 * (size_t)cap + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipHostMalloc(consensus) failed");
  HIPCHK(ctx, hipMemcpyAsync(wo, wout, 24, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (wo[2] == 2) return fail(ctx, MI355_SW_ENOTSUP, "consensus longer than |x| + |y|");
  if (wo[2] != 0) return 1;
  const size_t len = (size_t)wo[0];
  // only the bytes that exist cross PCIe: x at [0, len), y at [cap, cap + len)
  if (len) {
    HIPCHK(ctx, hipMemcpyAsync(cons.p, ctx->cons.p, len, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(cons.as<char>() + len, ctx->cons.as<char>() + cap, len, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->cons_used++;
  out.len = len;
  out.cx = cons.as<char>();
  out.cy = cons.as<char>() + len;
  out.pos = (uint32_t)wo[1];
  ctx->saved_traces += 1;
  path_note(ctx, "saved_trace[R=%d]", BR);
  return 0;
}

}  // namespace
