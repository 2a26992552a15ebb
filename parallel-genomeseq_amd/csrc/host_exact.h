// host_exact.h — sw_exact_kernel / sw_walk_kernel launches (LDS anti-diagonal engine: table scoring, whole uint8 problems, fill_matrix)
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
namespace {

// ---- exact kernel launches ------------------------------------------------------------------
struct ExactJob {
  int q;                 // query index in the batch
  int64_t ylo;           // window start (absolute reference index of local column 1)
  int32_t nw;
  int64_t col_offset;    // true (range-relative) column = col_offset + jl
  int64_t full_n;
  int32_t own_lo;
  int32_t quirk;
  float target;
  bool want_dirs;
  // results
  float best = -1;
  int64_t ci = 0, cj = 0;
  size_t dirs_off = 0;
};

// bytes of the diagonal-major decision array of an (m x nw) window (sw_exact_kernel.h)
size_t dirs_bytes(int64_t m, int64_t nw) { return (size_t)(m + nw + 1) * (size_t)std::max<int64_t>(1, std::min(m, nw)) + 16; }

size_t exact_lds_bytes(int m, int nw) { return (size_t)3 * (std::min(m, nw) + 2) * 4 + (size_t)m + 16; }

ExactScoring make_scoring(mi355_sw_ctx *ctx, const mi355_sw_params &p, bool &lut_uploaded, int &rc) {
  ExactScoring s;
  rc = 0;
  s.lut = nullptr;
  if (p.lut && p.semantics == MI355_SW_F32) {
    if (!lut_uploaded) {
      if (ctx->lut.ensure(65536 * 4)) { rc = MI355_SW_ENOMEM; return s; }
      if (hipMemcpyAsync(ctx->lut.p, p.lut, 65536 * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = MI355_SW_ENODEV; return s; }
      lut_uploaded = true;
    }
    s.lut = ctx->lut.as<float>();
  }
  s.match = p.match; s.mismatch = p.mismatch; s.gap = p.gap;
  const U8Params u = u8_params(p);
  s.u8M = u.M; s.u8X = u.X; s.u8G = u.G;
  return s;
}

// Runs jobs[lo,hi) in one launch.  Decisions (if wanted) land in ctx->dirs at job.dirs_off.
int run_exact(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const mi355_sw_params &p,
              std::vector<ExactJob> &jobs, size_t lo, size_t hi, float *hout /* device or null, single job */) {
  const size_t n = hi - lo;
  if (n == 0) return 0;
  size_t dirs_total = 0, lds = 0;
  for (size_t k = lo; k < hi; ++k) {
    ExactJob &j = jobs[k];
    lds = std::max(lds, exact_lds_bytes(q.len[j.q], j.nw));
    if (j.want_dirs) { j.dirs_off = dirs_total; dirs_total += dirs_bytes(q.len[j.q], j.nw); dirs_total = (dirs_total + 15) & ~(size_t)15; }
  }
  if (lds > kExactLdsMax) return fail(ctx, MI355_SW_ENOTSUP, "anti-diagonal longer than the exact kernel's LDS window");
  path_note(ctx, "exact[u8=%d,dirs=%d]", (int)(p.semantics == MI355_SW_U8SAT), (int)(dirs_total != 0));
  if (ctx->probs.ensure(n * sizeof(ExactProblem)) || ctx->outs_f.ensure(n * 4) || ctx->outs_i.ensure(n * 16) ||
      (dirs_total && ctx->dirs.ensure(dirs_total)))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(exact scratch) failed");
  // few problems with long diagonals: sixteen wavefronts per problem; the others one wavefront each
  std::vector<size_t> slot(n);                     // position of job lo + k in the problem array: wide ones first
  size_t nwide = 0, lds_narrow = 0;
  {
    std::vector<size_t> wide_k, narrow_k;
    for (size_t k = 0; k < n; ++k)
      (std::min<int>(q.len[jobs[lo + k].q], jobs[lo + k].nw) >= 1024 ? wide_k : narrow_k).push_back(k);
    if (wide_k.size() > 2048) { narrow_k.insert(narrow_k.end(), wide_k.begin(), wide_k.end()); wide_k.clear(); }
    nwide = wide_k.size();
    lds_narrow = 0;
    for (size_t k : narrow_k) lds_narrow = std::max(lds_narrow, exact_lds_bytes(q.len[jobs[lo + k].q], jobs[lo + k].nw));
    for (size_t t = 0; t < wide_k.size(); ++t) slot[wide_k[t]] = t;
    for (size_t t = 0; t < narrow_k.size(); ++t) slot[narrow_k[t]] = nwide + t;
  }
  std::vector<ExactProblem> pr(n);
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    ExactProblem &e = pr[slot[k]];
    e.x = q.bytes.as<uint8_t>() + q.off[j.q];
    e.y = ref.bytes.as<uint8_t>() + j.ylo;
    e.m = q.len[j.q];
    e.nw = j.nw;
    e.col_offset = j.col_offset;
    e.full_n = j.full_n;
    e.own_lo = j.own_lo;
    e.square_quirk = j.quirk;
    e.target = j.target;
    e.dirs = j.want_dirs ? ctx->dirs.as<uint8_t>() + j.dirs_off : nullptr;
    e.hout = hout;
    e.best = ctx->outs_f.as<float>() + slot[k];
    e.cell = ctx->outs_i.as<int64_t>() + 2 * slot[k];
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->probs.p, pr.data(), n * sizeof(ExactProblem), hipMemcpyHostToDevice, ctx->stream));
  bool lut_up = false;
  int rc = 0;
  const ExactScoring sc = make_scoring(ctx, p, lut_up, rc);
  if (rc) return fail(ctx, rc, "scoring table upload failed");
  const ExactProblem *dp = ctx->probs.as<ExactProblem>();
  const size_t nnarrow = n - nwide;
  if (p.semantics == MI355_SW_U8SAT) {
    if (nwide) hipLaunchKernelGGL((sw_exact_kernel<1, 1024>), dim3((unsigned)nwide), dim3(1024), lds, ctx->stream, dp, sc);
    if (nnarrow) hipLaunchKernelGGL((sw_exact_kernel<1, 64>), dim3((unsigned)nnarrow), dim3(64), lds_narrow, ctx->stream, dp + nwide, sc);
  } else {
    if (nwide) hipLaunchKernelGGL((sw_exact_kernel<0, 1024>), dim3((unsigned)nwide), dim3(1024), lds, ctx->stream, dp, sc);
    if (nnarrow) hipLaunchKernelGGL((sw_exact_kernel<0, 64>), dim3((unsigned)nnarrow), dim3(64), lds_narrow, ctx->stream, dp + nwide, sc);
  }
  HIPCHK(ctx, hipGetLastError());
  std::vector<float> bf(n);
  std::vector<int64_t> ci(2 * n);
  HIPCHK(ctx, hipMemcpyAsync(bf.data(), ctx->outs_f.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ci.data(), ctx->outs_i.p, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; ++k) { jobs[lo + k].best = bf[slot[k]]; jobs[lo + k].ci = ci[2 * slot[k]]; jobs[lo + k].cj = ci[2 * slot[k] + 1]; }
  return 0;
}

// Consensus of one alignment: views into a D2H buffer the context keeps until the next call (ctx->arenas) —
// with half a million alignments per call, a std::string pair each was a quarter of the host time.

// Walk over decisions of jobs[lo,hi) (all with want_dirs), starting at (start_i, local nw...).
int run_walk(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const std::vector<ExactJob> &jobs,
             size_t lo, size_t hi, const std::vector<std::pair<int32_t, int32_t>> &starts,
             const std::vector<int32_t> &exact_lo, std::vector<TraceOut> &outs, std::vector<int> &status,
             float need_slope = 0.0f) {
  const size_t n = hi - lo;
  if (n == 0) return 0;
  std::vector<WalkProblem> wp(n);
  std::vector<size_t> coff(n);
  size_t ctot = 0;
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    const int cap = q.len[j.q] + j.nw + 2;
    coff[k] = ctot;
    ctot += 2 * (size_t)cap;
  }
  if (ctx->cons.ensure(ctot + 16) || ctx->walkp.ensure(n * sizeof(WalkProblem) + n * 24))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(walk scratch) failed");
  int64_t *wout = reinterpret_cast<int64_t *>(ctx->walkp.as<uint8_t>() + n * sizeof(WalkProblem));
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    WalkProblem &w = wp[k];
    const int cap = q.len[j.q] + j.nw + 2;
    w.x = q.bytes.as<uint8_t>() + q.off[j.q];
    w.y = ref.bytes.as<uint8_t>() + j.ylo;
    w.dirs = ctx->dirs.as<uint8_t>() + j.dirs_off;
    w.m = q.len[j.q]; w.nw = j.nw;
    w.start_i = starts[k].first; w.start_jl = starts[k].second;
    w.exact_lo = exact_lo[k];
    w.need_slope = need_slope;
    w.col_offset = j.col_offset;
    w.cons_x = ctx->cons.as<char>() + coff[k];
    w.cons_y = w.cons_x + cap;
    w.cap = cap;
    w.out = wout + 3 * k;
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->walkp.p, wp.data(), n * sizeof(WalkProblem), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(sw_walk_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->walkp.as<WalkProblem>(), (int)n);
  HIPCHK(ctx, hipGetLastError());
  std::vector<int64_t> wo(3 * n);
  std::vector<char> cons(ctot);
  HIPCHK(ctx, hipMemcpyAsync(wo.data(), wout, n * 24, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(cons.data(), ctx->cons.p, ctot, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  outs.resize(n); status.resize(n);
  ctx->arenas.push_back(std::move(cons));
  const char *base = ctx->arenas.back().data();
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    const int cap = q.len[j.q] + j.nw + 2;
    status[k] = (int)wo[3 * k + 2];
    outs[k].len = (size_t)wo[3 * k];
    outs[k].cx = base + coff[k];
    outs[k].cy = base + coff[k] + cap;
    outs[k].pos = (uint32_t)wo[3 * k + 1];
  }
  return 0;
}

}  // namespace
