// host_solo.h — the latency path of mi355_sw_align / mi355_sw_argmax: one short query, one upload, two kernels, one
// synchronisation (sw_solo_kernel.h).  Part of the single translation unit mi355_sw.hip.
//
// Per call: a 512-byte block goes up (query bytes, its length / offset / id, the range, a zeroed key, the arrival
// counter, the storage-order minimum, a zeroed result header); sw_score_kernel sweeps the reference and leaves
// (maximum, first sub-chunk) in the key; sw_solo_kernel — launched right behind it, no host round trip — locates the
// first maximum, keeps the traceback decisions in LDS and walks them; the result block comes down.  Anything the
// kernel cannot take (status) sends the call down the general path of host_pipeline.h.
namespace {

constexpr int kSoloMaxRows = 320;                 // 64 lanes x R in {3, 5}
constexpr size_t kSoloLdsMax = 150 * 1024;
constexpr int kSoloMaxRanges = 64;                // pieces of one call (OMPParallelLocalAligner: sw_solve_small uses 17, sw_solve_big 2 x cores)
// the call block: header, query, range starts / ends, keys, result header, consensus strings
constexpr size_t kSoloQueryOff = 64, kSoloLoOff = 448, kSoloHiOff = kSoloLoOff + 8 * kSoloMaxRanges,
                 kSoloKeyOff = kSoloHiOff + 8 * kSoloMaxRanges, kSoloResultOff = kSoloKeyOff + 8 * kSoloMaxRanges,
                 kSoloConsOff = kSoloResultOff + 64;

struct SoloBlock {                                // first 64 bytes of the call block
  int64_t qoff;
  int32_t qlen, qsel;
  unsigned long long gmin;
  uint32_t done, pad[9];
};
static_assert(sizeof(SoloBlock) == 64, "layout of the call block");

// The query against every range of `ranges` (one range: SWAligner; the pieces of _make_string_range:
// OMPParallelLocalAligner with default scoring, where the per-piece maxima of plocalaligner.cpp:110-129 and the winner's
// re-alignment :132-141 are the same sweep): the FIRST range with the strictly greatest maximum is located and traced;
// pos / end_y are relative to that range, *piece says which.
// returns 1: not applicable / the kernel declined (continue on the general path); 0: *out filled; < 0: error
int solo_align(mi355_sw_ctx *ctx, const RefData &ref, const char *x, size_t nx, const std::vector<Range> &ranges,
               const mi355_sw_params &p, bool want_trace, mi355_sw_result *out, int *piece = nullptr) {
  if (opt().no_solo || nx < 1 || nx > (size_t)kSoloMaxRows || ranges.empty() || ranges.size() > (size_t)kSoloMaxRanges || !wave_scoring_ok(p)) return 1;
  int64_t n = 0, nmin = INT64_MAX;                 // longest / shortest range
  for (const Range &r : ranges) { n = std::max(n, r.hi - r.lo); nmin = std::min(nmin, r.hi - r.lo); }
  if (nmin < 1024) return 1;
  HostTrace trace_("solo_align");
  const bool u8 = p.semantics == MI355_SW_U8SAT;
  const ScoreTable table = plan_table(ref, p);
  if (!table.ok || !(table.smaxf > 0)) return 1;
  QueryBatch qv;                                  // one query; the device arrays are views into the call block
  qv.nq = 1; qv.len.assign(1, (int32_t)nx); qv.off.assign(1, 0); qv.order.assign(1, 0); qv.maxlen = (int)nx;
  std::vector<Bucket> buckets = make_buckets(ref, qv, table, p, n);
  if (buckets.size() != 1) return 1;
  Bucket &b = buckets[0];
  for (const Range &r : ranges) if (!bucket_fast_ok(ref, table, b, r.hi - r.lo, p)) return 1;
  const int keykind = b.sem == kSemF16 ? 2 : (b.sem == kSemF32 ? 4 : 0);
  if (keykind == 0) return 1;
  const Margin mg = table.margin((double)nx);
  if (!mg.finite()) return 1;
  // the largest window any candidate can need, and what it takes in LDS
  const int R = nx <= 192 ? 3 : 5;
  const int DB = R <= 4 ? 1 : 2;
  const int64_t sub_len = score_sub_len(p.semantics, b);
  const int64_t budget = (int64_t)nx / 8 + 64;
  const int64_t lane_need = clamp_cols((double)nx + std::ceil((double)nx * mg.slope()) + 3.0);
  const int64_t need_t = want_trace ? budget + std::min(b.warm, lane_need) : 0;
  const int64_t warm1 = std::min(b.warm, clamp_cols((double)nx + std::ceil(mg.smax * (double)nx / mg.g) + 3.0));
  const int64_t nbmax = sub_len + 63 + std::max(need_t, warm1);
  const int64_t cap = (int64_t)nx + nbmax + 2;
  const size_t lds = ((size_t)nbmax + 1) * 64 * DB + (((size_t)nbmax + 136 + 15) & ~(size_t)15) + 64 * (size_t)R + 2 * (size_t)cap + 64;
  if (lds > kSoloLdsMax) return 1;

  // ---- the call block ----
  static_assert(kSoloMaxRows + kSoloQueryOff <= kSoloLoOff, "layout of the call block");
  const size_t down_bytes = (kSoloConsOff - kSoloResultOff) + 2 * (size_t)cap;
  if (ctx->soloblk.ensure(kSoloConsOff + 2 * (size_t)cap + 64) || ctx->pin_solo_up.ensure(kSoloConsOff) ||
      ctx->pin_solo_down.ensure(down_bytes + 64))
    return fail(ctx, MI355_SW_ENOMEM, "allocation of the single-alignment block failed");
  uint8_t *up = ctx->pin_solo_up.as<uint8_t>();
  memset(up, 0, kSoloConsOff);
  SoloBlock *blk = reinterpret_cast<SoloBlock *>(up);
  blk->qoff = 0; blk->qlen = (int32_t)nx; blk->qsel = 0;
  blk->gmin = ~0ull; blk->done = 0;
  const size_t nr = ranges.size();
  for (size_t k = 0; k < nr; ++k) {
    reinterpret_cast<int64_t *>(up + kSoloLoOff)[k] = ranges[k].lo;
    reinterpret_cast<int64_t *>(up + kSoloHiOff)[k] = ranges[k].hi;
  }
  memcpy(up + kSoloQueryOff, x, nx);
  uint8_t *dev = ctx->soloblk.as<uint8_t>();
  HIPCHK(ctx, hipMemcpyAsync(dev, up, kSoloConsOff, hipMemcpyHostToDevice, ctx->stream));
  int rc = score_tables(ctx, (int)nx, n, table);
  if (rc) return rc;
  qv.bytes.alias(dev + kSoloQueryOff);
  qv.offs.alias(dev + offsetof(SoloBlock, qoff));
  qv.lens.alias(dev + offsetof(SoloBlock, qlen));
  qv.sel.alias(dev + offsetof(SoloBlock, qsel));
  ScoreIO io;
  io.range_lo = reinterpret_cast<const int64_t *>(dev + kSoloLoOff);
  io.range_hi = reinterpret_cast<const int64_t *>(dev + kSoloHiOff);
  io.keys = reinterpret_cast<unsigned long long *>(dev + kSoloKeyOff);
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  rc = score_launch(ctx, ref, qv, ranges, p, table, b, &io);
  if (rc) return rc;
  if (b.sub_len != sub_len) return fail(ctx, MI355_SW_ENODEV, "internal: sub-chunk granularity of the single-alignment path");

  SoloArgs a;
  a.key = io.keys;
  a.range_lo = io.range_lo; a.range_hi = io.range_hi; a.nranges = (int32_t)nr;
  a.gmin = reinterpret_cast<unsigned long long *>(dev + offsetof(SoloBlock, gmin));
  a.done = reinterpret_cast<unsigned int *>(dev + offsetof(SoloBlock, done));
  a.x = dev + kSoloQueryOff; a.m = (int32_t)nx;
  a.yref = ref.bytes.as<uint8_t>();
  a.sub_len = sub_len;
  a.keykind = keykind; a.fshift = ctx->fshift;
  a.mg_smax = (float)mg.smax; a.mg_g = (float)mg.g;
  a.warm = b.warm;
  a.budget = (int32_t)budget;
  a.want_trace = want_trace ? 1 : 0;
  a.ncand = u8 ? 5 : 1;
  a.cap = (int32_t)cap;
  a.lds_steps = (int32_t)nbmax;
  a.sc.match = p.match; a.sc.mismatch = p.mismatch; a.sc.gap = p.gap;
  const U8Params u = u8_params(p);
  a.sc.u8M = (float)u.M; a.sc.u8X = (float)u.X; a.sc.u8G = (float)u.G;
  a.out = reinterpret_cast<SoloResult *>(dev + kSoloResultOff);
#define SOLO_LAUNCH(r, eight)                                                                                              \
  do {                                                                                                                     \
    if (lds > 48 * 1024)                                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_solo_kernel<r, eight>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((sw_solo_kernel<r, eight>), dim3((unsigned)a.ncand), dim3(64), lds, ctx->stream, a);                \
  } while (0)
  if (R == 3) { if (u8) SOLO_LAUNCH(3, true); else SOLO_LAUNCH(3, false); }
  else { if (u8) SOLO_LAUNCH(5, true); else SOLO_LAUNCH(5, false); }
#undef SOLO_LAUNCH
  HIPCHK(ctx, hipGetLastError());
  path_note(ctx, "solo[R=%d,u8=%d,ranges=%zu]", R, (int)u8, nr);
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  uint8_t *down = ctx->pin_solo_down.as<uint8_t>();
  HIPCHK(ctx, hipMemcpyAsync(down, dev + kSoloResultOff, down_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  // device time of the score launch (iterate), and of the call
  for (size_t e = 0; e + 1 < ctx->score_ev_used; e += 2) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->score_ev[e], ctx->score_ev[e + 1]));
    ctx->timings[0] += (double)ms * 1000.0;
  }
  ctx->score_ev_used = 0;
  ctx->timings[3] += elapsed_us(ctx, ctx->ev[4], ctx->ev[5]);
  const SoloResult *r = reinterpret_cast<const SoloResult *>(down);
  if (!r->written || r->status == kSoloWindow || r->status == kSoloLds) return 1;       // the general path takes it
  if (r->status == kSoloExpired) return fail(ctx, MI355_SW_ENODEV, "internal: single-alignment kernel wait expired");
  if (r->status == kSoloLost) return fail(ctx, MI355_SW_ENODEV, "internal: maximum of the score pass not found again by the exact kernel");
  if (r->status != kSoloOk) return fail(ctx, MI355_SW_ENOTSUP, "consensus longer than |x| + window");
  TraceOut t;
  t.len = (size_t)r->len;
  t.cx = reinterpret_cast<const char *>(down) + (kSoloConsOff - kSoloResultOff);
  t.cy = t.cx + cap;
  t.pos = (uint32_t)r->pos;
  set_result(*out, r->score, r->ix, r->iy, (want_trace && r->score > 0) ? &t : nullptr);
  if (piece) *piece = r->piece;
  out->timings_us[0] = (float)(ctx->timings[0] > 0 ? ctx->timings[0] : ctx->timings[3]);
  out->timings_us[1] = 0;
  return 0;
}

}  // namespace
