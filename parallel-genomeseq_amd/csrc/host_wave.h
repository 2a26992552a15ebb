// host_wave.h — register-wavefront exact kernels: sw_wave_kernel (short sides), sw_strip_kernel (long queries), their walks
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
namespace {

// ---- wavefront exact kernel (sw_wave_kernel.h): small problems, identity scoring ---------------
constexpr int kWaveMaxLanesSide = 512;

// Scoring the wave kernel evaluates: no table, and penalties that strictly lower a path (so that padding cells
// can never reach the maximum).
bool wave_scoring_ok(const mi355_sw_params &p) {
  if (p.semantics == MI355_SW_U8SAT) { const U8Params u = u8_params(p); return u.M > 0; }
  return p.lut == nullptr && p.match > 0 && p.mismatch < 0 && p.gap > 0 && std::isfinite(p.match) &&
         std::isfinite(p.mismatch) && std::isfinite(p.gap);
}

// rows per lane of the wave kernel instance that covers `na` cells on the lane side, and its decision bytes
int wave_R(int na) { return na <= 160 ? 10 : (na <= 320 ? 20 : 32); }
// ... of the profile kernels (lanes = columns of the shared second sequence): nine where they cover it — |y| = 144 = 16 x 9 is the
// UniProt driver's query P02232 — and no column is padding
int wave_prof_R(int na) { return na <= 144 ? 9 : wave_R(na); }
size_t wave_dirs_bytes(int64_t nb, int R) { return (size_t)(nb + 16) * 16 * (size_t)((R + 15) / 16) * 4; }   // (+ the skew's rows)

struct WaveJob {
  int q;                  // query index
  int orient;             // 0: lanes = rows of x, stream = columns of y; 1: lanes = columns of y, stream = rows of x
  int64_t s_lo;           // stream window start (0-based, range-relative), nb positions
  int32_t nb;
  bool track, dirs;
  bool keyed = false;     // wave kernel, track: first cell equal to target in storage order (else: first maximum)
  bool maxmode = false;   // strip kernel, track: the maximum over the own positions and its first cell (kStripMax)
  float target = 0;       // strip kernel / keyed: only cells equal to target compete ...
  int32_t own_lo = 0;     // ... at stream positions >= own_lo (0-based)
  // results
  float best = 0;
  int64_t ci = 0, cj = 0;
  size_t dirs_off = 0;
};

template <int R, int ORIENT, bool U8>
void launch_wave_flags(bool track, bool dirs, unsigned blocks, hipStream_t st, const WaveProblem *pr, int n, const WaveScoring &sc) {
  if (track && dirs) hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, U8, true, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
  else if (track) hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, U8, true, false>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
  else hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, U8, false, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
}

template <int R>
void launch_wave_keyed(bool u8, unsigned blocks, hipStream_t st, const WaveProblem *pr, int n, const WaveScoring &sc) {
  if (u8) hipLaunchKernelGGL((sw_wave_kernel<R, 0, true, true, false, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
  else hipLaunchKernelGGL((sw_wave_kernel<R, 0, false, true, false, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
}

template <int R>
void launch_wave_R(int orient, bool u8, bool track, bool dirs, unsigned blocks, hipStream_t st, const WaveProblem *pr, int n, const WaveScoring &sc) {
  if (orient == 0) { if (u8) launch_wave_flags<R, 0, true>(track, dirs, blocks, st, pr, n, sc); else launch_wave_flags<R, 0, false>(track, dirs, blocks, st, pr, n, sc); }
  else { if (u8) launch_wave_flags<R, 1, true>(track, dirs, blocks, st, pr, n, sc); else launch_wave_flags<R, 1, false>(track, dirs, blocks, st, pr, n, sc); }
}

// sw_wave_prof_kernel for a launch whose lanes hold columns of the reference range (ORIENT 1), float engine, identity scoring:
// the query profile over the shared lane side + the three-op cell (sw_wave_kernel.h).  Returns 1 when it does not apply
// (the caller launches sw_wave_kernel), 0 when launched, < 0 on error.
bool wave_prof_ok(const RefData &ref, const mi355_sw_params &p, int R, int na, bool track) {
  if (opt().no_wave_prof || p.semantics != MI355_SW_F32 || !wave_scoring_ok(p) || ref.ncodes < 2 || ref.ncodes > 256) return false;
  if ((size_t)ref.ncodes * 16 * lane_stride(R) * 4 > 96 * 1024) return false;   // (alphabets of > 120 letters at R = 10)
  // The tracking key borrows the five lowest mantissa bits of a cell (sw_wave_kernel.h): every reachable value must be a
  // multiple of q = 2^e below 2^18 q — scores that are multiples of q, and match * (lane side + 1) < 2^18 q.
  if (track) {
    float q = 0.0f;
    for (int e = 10; e >= -10 && q == 0.0f; --e) {
      const float c = std::ldexp(1.0f, e);
      if (std::floor(p.match / c) == p.match / c && std::floor(p.mismatch / c) == p.mismatch / c && std::floor(p.gap / c) == p.gap / c) q = c;
    }
    if (q == 0.0f || (double)p.match * ((double)na + 1.0) >= 262144.0 * (double)q) return false;
  }
  return true;
}

// byte -> code and code -> byte tables of the resident reference, for the profile kernels
int wave_tables(mi355_sw_ctx *ctx, const RefData &ref) {
  if (ctx->wlut_ref != (const void *)&ref || ctx->wlut_version != ref.version) {
    ctx->h_wlut.assign(512, 0);
    for (int b = 0; b < 256; ++b) ctx->h_wlut[b] = (uint8_t)(ref.code_of[b] >= 0 ? ref.code_of[b] : ref.ncodes - 1);
    for (int c = 0; c + 1 < ref.ncodes; ++c) ctx->h_wlut[256 + c] = ref.byte_of[c];
    if (ctx->wlut.ensure(512)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(wave tables) failed");
    HIPCHK(ctx, hipMemcpyAsync(ctx->wlut.p, ctx->h_wlut.data(), 512, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                // (the staging vector may change with the next reference)
    ctx->wlut_ref = (const void *)&ref; ctx->wlut_version = ref.version;
  }
  return 0;
}

int launch_wave_prof(mi355_sw_ctx *ctx, const RefData &ref, const mi355_sw_params &p, int R, int na, bool track, bool dirs,
                     unsigned blocks, const WaveProblem *dp, int n) {
  if (!wave_prof_ok(ref, p, R, na, track)) return 1;
  const size_t lds = (size_t)ref.ncodes * 16 * lane_stride(R) * 4;
  { int rc_t = wave_tables(ctx, ref); if (rc_t) return rc_t; }
  WaveProfArgs sa;
  sa.lut = ctx->wlut.as<uint8_t>();
  sa.byte_of = ctx->wlut.as<uint8_t>() + 256;
  sa.ncodes = ref.ncodes;
  // cells hold H * 2^-k, 2^k above every value of the launch (at most match * min(|x|, |y|), the lane side is <= 512)
  const int k = std::max(1, std::min(100, std::ilogb((double)p.match * ((double)na + 1.0) + 1.0) + 2));
  sa.match_s = std::ldexp(p.match, -k); sa.mismatch_s = std::ldexp(p.mismatch, -k); sa.gap_s = std::ldexp(p.gap, -k);
  sa.unscale = std::ldexp(1.0f, k);
  {
    // states saved by sw_wave_prof16_kernel hold H / (q 2048), q = the power of two all three scores are multiples of
    float q = 1.0f;
    for (int e = 10; e >= -10; --e) {
      const float c = std::ldexp(1.0f, e);
      if (std::floor(p.match / c) == p.match / c && std::floor(p.mismatch / c) == p.mismatch / c && std::floor(p.gap / c) == p.gap / c) { q = c; break; }
    }
    sa.ck16_scale = std::ldexp(q, 11 - k);
  }
#define WAVE_PROF(r)                                                                                                                  \
  {                                                                                                                                   \
    if (lds > 48 * 1024) {                                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_wave_prof_kernel<r, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_wave_prof_kernel<r, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_wave_prof_kernel<r, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    }                                                                                                                                 \
    if (track && dirs) hipLaunchKernelGGL((sw_wave_prof_kernel<r, true, true>), dim3(blocks), dim3(256), lds, ctx->stream, dp, n, sa);  \
    else if (track) hipLaunchKernelGGL((sw_wave_prof_kernel<r, true, false>), dim3(blocks), dim3(256), lds, ctx->stream, dp, n, sa);   \
    else hipLaunchKernelGGL((sw_wave_prof_kernel<r, false, true>), dim3(blocks), dim3(256), lds, ctx->stream, dp, n, sa);              \
  }
  if (R == 9) WAVE_PROF(9) else if (R == 10) WAVE_PROF(10) else if (R == 20) WAVE_PROF(20) else WAVE_PROF(32)
#undef WAVE_PROF
  return 0;
}

// sw_wave_prof16_kernel (two problems per slot on packed float16 cells) for the TRACK pass of a device-built batch: returns 1 when
// it does not apply (the caller launches the float32 kernel), 0 when launched, < 0 on error.  DESIGN.md §3.3 lemma L13: every
// score is a multiple of q = 2^e and smaller than 2048 q in magnitude, match * (|y| + 1) < 2048 q (every cell value is then an
// integer multiple of q below 2048 q: exact in float16), ten columns per lane (the key's four bits), streams of at most 65 000
// rows (16-bit step counters).  A sequence whose maximum reaches 128 q comes back with best = -1: the caller leaves it to the
// float32 path.
int launch_wave_prof16(mi355_sw_ctx *ctx, const RefData &ref, const mi355_sw_params &p, int R, int na, int64_t max_stream,
                       const WaveProblem *dp, int n) {
  if (opt().no_wave_f16 || (R != 9 && R != 10) || max_stream > 65000 || !wave_prof_ok(ref, p, R, na, true)) return 1;
  float q = 0.0f;
  int e = 10;
  for (; e >= -10 && q == 0.0f; --e) {
    const float c = std::ldexp(1.0f, e);
    if (std::floor(p.match / c) == p.match / c && std::floor(p.mismatch / c) == p.mismatch / c && std::floor(p.gap / c) == p.gap / c) { q = c; break; }
  }
  if (q == 0.0f) return 1;
  const double lim = 2048.0 * (double)q;
  if ((double)p.match * ((double)na + 1.0) >= lim || std::fabs((double)p.mismatch) >= lim || (double)p.gap >= lim || !(p.gap > 0.0f)) return 1;
  const size_t lds = 2 * (size_t)ref.ncodes * 16 * lane_stride(R) * 4;
  if (lds > 60 * 1024) return 1;
  int rc = wave_tables(ctx, ref);
  if (rc) return rc;
  WaveProf16Args sa;
  sa.lut = ctx->wlut.as<uint8_t>();
  sa.byte_of = ctx->wlut.as<uint8_t>() + 256;
  sa.ncodes = ref.ncodes;
  const float unit = std::ldexp(q, 11);                              // cells hold H / (q 2048)
  sa.match_h = half_bits(p.match / unit); sa.mismatch_h = half_bits(p.mismatch / unit);
  sa.ngap2 = (uint32_t)half_bits(-p.gap / unit) * 0x00010001u;
  sa.unscale = unit;
  const unsigned blocks = (unsigned)((n + 31) / 32);
#define WAVE_PROF16(r)                                                                                                               \
  {                                                                                                                                   \
    if (lds > 48 * 1024)                                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_wave_prof16_kernel<r>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((sw_wave_prof16_kernel<r>), dim3(blocks), dim3(256), lds, ctx->stream, dp, n, sa);                             \
  }
  if (R == 9) WAVE_PROF16(9) else WAVE_PROF16(10)
#undef WAVE_PROF16
  return 0;
}

// One launch: all jobs share orientation and flags; lanes side <= 512.
int run_wave(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
             std::vector<WaveJob> &jobs) {
  HostTrace trace_("run_wave");
  const size_t n = jobs.size();
  if (n == 0) return 0;
  const int orient = jobs[0].orient;
  const bool track = jobs[0].track, dirs = jobs[0].dirs;
  const int64_t nref = rg.hi - rg.lo;
  size_t dirs_total = 0;
  int maxna = 0;
  for (WaveJob &j : jobs) maxna = std::max(maxna, orient == 0 ? q.len[j.q] : (int)nref);
  if (maxna > kWaveMaxLanesSide) return fail(ctx, MI355_SW_EINVAL, "internal: wave kernel side too long");
  const int R = wave_R(maxna);
  for (WaveJob &j : jobs)
    if (dirs) { j.dirs_off = dirs_total; dirs_total += wave_dirs_bytes(j.nb, R); }
  if (ctx->wprobs.ensure(n * sizeof(WaveProblem)) || ctx->outs_f.ensure(n * 4) || ctx->outs_i.ensure(n * 16) ||
      (dirs_total && ctx->dirs.ensure(dirs_total)))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(wave scratch) failed");
  if (ctx->pin_probs.ensure(n * sizeof(WaveProblem)) || ctx->pin_out.ensure(n * 20 + 64))
    return fail(ctx, MI355_SW_ENOMEM, "hipHostMalloc(wave staging) failed");
  WaveProblem *pr = ctx->pin_probs.as<WaveProblem>();
  parallel_for(n, [&](size_t k0, size_t k1) {
  for (size_t k = k0; k < k1; ++k) {
    const WaveJob &j = jobs[k];
    WaveProblem &w = pr[k];
    const uint8_t *xq = q.bytes.as<uint8_t>() + q.off[j.q];
    const uint8_t *yr = ref.bytes.as<uint8_t>() + rg.lo;
    if (orient == 0) { w.a = xq; w.na = q.len[j.q]; w.b = yr + j.s_lo; }
    else { w.a = yr; w.na = (int32_t)nref; w.b = xq + j.s_lo; }
    w.nb = j.nb;
    w.b_offset = j.s_lo;
    w.dirs = dirs ? reinterpret_cast<uint32_t *>(ctx->dirs.as<uint8_t>() + j.dirs_off) : nullptr;
    w.best = ctx->outs_f.as<float>() + k;
    w.cell = ctx->outs_i.as<int64_t>() + 2 * k;
    w.target = j.target; w.own_lo = j.own_lo; w.full_n = nref;
    w.ckpt = nullptr; w.k0 = 0; w.ck_half = 0; w.lanes_used = 0;
  }
  });
  const bool keyed = jobs[0].keyed;
  HIPCHK(ctx, hipMemcpyAsync(ctx->wprobs.p, pr, n * sizeof(WaveProblem), hipMemcpyHostToDevice, ctx->stream));
  WaveScoring sc;
  sc.match = p.match; sc.mismatch = p.mismatch; sc.gap = p.gap;
  const U8Params u = u8_params(p);
  sc.u8M = (float)u.M; sc.u8X = (float)u.X; sc.u8G = (float)u.G;
  const bool u8 = p.semantics == MI355_SW_U8SAT;
  const unsigned blocks = (unsigned)((n + 15) / 16);
  const WaveProblem *dp = ctx->wprobs.as<WaveProblem>();
  int prof_rc = 1;
  if (!keyed && orient == 1 && !u8) {
    prof_rc = launch_wave_prof(ctx, ref, p, R, (int)nref, track, dirs, blocks, dp, (int)n);
    if (prof_rc < 0) return prof_rc;
  }
  if (prof_rc == 0) {
  }
  else if (keyed) {
    if (R == 10) launch_wave_keyed<10>(u8, blocks, ctx->stream, dp, (int)n, sc);
    else if (R == 20) launch_wave_keyed<20>(u8, blocks, ctx->stream, dp, (int)n, sc);
    else launch_wave_keyed<32>(u8, blocks, ctx->stream, dp, (int)n, sc);
  }
  else if (R == 10) launch_wave_R<10>(orient, u8, track, dirs, blocks, ctx->stream, dp, (int)n, sc);
  else if (R == 20) launch_wave_R<20>(orient, u8, track, dirs, blocks, ctx->stream, dp, (int)n, sc);
  else launch_wave_R<32>(orient, u8, track, dirs, blocks, ctx->stream, dp, (int)n, sc);
  HIPCHK(ctx, hipGetLastError());
  path_note(ctx, "wave[orient=%d,R=%d,track=%d,dirs=%d,keyed=%d,prof=%d,u8=%d]", orient, R, (int)track, (int)dirs, (int)keyed, (int)(prof_rc == 0), (int)u8);
  if (track) {
    int64_t *ci = ctx->pin_out.as<int64_t>();
    float *bf = reinterpret_cast<float *>(ci + 2 * n);
    HIPCHK(ctx, hipMemcpyAsync(bf, ctx->outs_f.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ci, ctx->outs_i.p, n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    parallel_for(n, [&](size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; ++k) { jobs[k].best = bf[k]; jobs[k].ci = ci[2 * k]; jobs[k].cj = ci[2 * k + 1]; }
    });
  }
  return 0;
}

// Rows per lane of the strip kernel instance for a query of `na` rows, and its strips of 64*R rows (sixteen run
// concurrently; longer queries take several rounds).
// Short queries get a whole wavefront with few rows per lane (latency mode, below): the recurrence's dependent chain
// per step is R rows long, so 64 lanes x 3 rows sweep a 150 bp window three times faster than 16 lanes x 10 rows.
inline int strip_count_of(int na, int R) { return std::max(1, (na + 64 * R - 1) / (64 * R)); }
int strip_R(int na) { return na <= 192 ? 3 : (na <= 320 ? 5 : (na <= 512 ? 8 : (na <= 64 * kStripMaxWaves * 10 ? 10 : 16))); }
// A FEW long problems (a lone read of 513+ rows: locate and traceback of one alignment are latency-bound on ONE workgroup):
// the fewest rows per lane whose strips still run concurrently — sixteen wavefronts in the tracking modes, any number in
// the decision mode, whose strips are dealt to several workgroups (run_strip) — because a step's dependent chain is R
// rows long and more, shorter strips pipeline over more SIMDs.
constexpr size_t kFewJobs = 64;            // fewer problems than a quarter of the CUs
int strip_R_few(int na, size_t njobs, bool track) {
  if (njobs > kFewJobs || na <= 512 || na > 64 * kStripMaxWaves * 10) return strip_R(na);
  const long forced = opt().few_r;   // tuning aid
  if (forced == 3 || forced == 5 || forced == 8 || forced == 10 || forced == 16) return (int)forced;
  // (the strips of a few long problems are dealt to several workgroups, one wavefront per SIMD — run_strip — so their number
  // is no limit; a lone wavefront's step is a dependent chain R rows long, and at three rows it is shortest: config 5's locate
  // 5.5 -> 4.4 ms, traceback 9.8 -> 8.3 ms)
  if (!opt().no_strip_groups && njobs <= (track ? (size_t)32 : (size_t)8)) return 3;      // (run_strip's condition for several workgroups)
  static const int rs[] = {3, 5, 8, 10, 16};
  if (!track) return na <= 2560 ? 3 : 5;
  for (int r : rs) if (strip_count_of(na, r) <= kStripMaxWaves) return r;
  return 16;
}
constexpr size_t kLatencyJobs = 64;            // up to this many short problems per call run in latency mode
int strip_count(int na, int R) { return std::max(1, (na + 64 * R - 1) / (64 * R)); }
size_t strip_dirs_bytes(int64_t nb, int nstrips, int R) { return (size_t)nb * 64 * (size_t)nstrips * (size_t)((R + 15) / 16) * 4 + 64; }

template <int R>
void launch_strip(bool u8, bool track, dim3 grid, dim3 block, hipStream_t st, const StripProblem *dp, const WaveScoring &sc,
                  const float *gtab, int ncodes, int groups = 1, bool maxmode = false) {
  if (maxmode && u8) {                          // maximum + first cell in the uint8 engine's storage order (sampled sweep)
    hipLaunchKernelGGL((sw_strip_kernel<R, true, kStripMax>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0, groups);
    return;
  }
  if (maxmode) {                                // float engine (locate_saturated)
    if (gtab) hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripMax, true>), grid, block, (size_t)257 * ncodes * 4, st, dp, sc, gtab, ncodes, groups);
    else hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripMax>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0, groups);
    return;
  }
  if (gtab) {                                   // table scoring (float engine): tab[257][ncodes] in dynamic LDS
    const size_t lds = (size_t)257 * ncodes * 4;
    if (track) hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripTrack, true>), grid, block, lds, st, dp, sc, gtab, ncodes, groups);
    else hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripDirs, true>), grid, block, lds, st, dp, sc, gtab, ncodes, groups);
    return;
  }
  if (track) {
    if (u8) hipLaunchKernelGGL((sw_strip_kernel<R, true, kStripTrack>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0, groups);
    else hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripTrack>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0, groups);
  } else {
    if (u8) hipLaunchKernelGGL((sw_strip_kernel<R, true, kStripDirs>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0, groups);
    else hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripDirs>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0, groups);
  }
}

// Long queries (ORIENT 0 windows) on the pipelined strip kernel, one workgroup per job: traceback decisions
// (jobs[.].dirs) or the first cell equal to jobs[.].target in storage order (jobs[.].track -> ci, cj; ci = 0: none).
// The score table of general (non-identity) float scoring the strip kernel can hold in LDS next to its rings.
bool strip_table_ok(const RefData &ref, const mi355_sw_params &p) {
  return p.semantics == MI355_SW_F32 && (size_t)257 * ref.ncodes * 4 <= 96 * 1024;
}
// Which long queries the strip kernel takes: identity scoring in both engines, any table in the float engine.
bool strip_scoring_ok(const RefData &ref, const mi355_sw_params &p) {
  if (opt().no_strip) return false;
  return wave_scoring_ok(p) || strip_table_ok(ref, p);
}

int run_strip(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
              std::vector<WaveJob> &jobs, int R) {
  HostTrace trace_("run_strip");
  const bool use_table = !wave_scoring_ok(p);            // ctx->ftab holds plan_table()'s [256][ncodes] (score_begin)
  const size_t n = jobs.size();
  if (n == 0) return 0;
  const bool track = jobs[0].track;
  const bool maxmode = jobs[0].maxmode;
  size_t dirs_total = 0, gtotal = 0;
  int nwmax = 1, nsmax = 1;
  std::vector<size_t> goff(n, 0);
  for (size_t k = 0; k < n; ++k) nsmax = std::max(nsmax, strip_count(q.len[jobs[k].q], R));
  // the decisions of a FEW long alignments: the strips of each dealt to several workgroups, four wavefronts (one per
  // SIMD) each, instead of sixteen wavefronts on one CU (config 5: the sweep of the 26 k-column window)
  const bool no_groups = opt().no_strip_groups || tl_no_wait;
  const int spg = 4;
  // ... and the locate windows of a few long queries (first-cell / maximum tracking): each workgroup reports the best cell of
  // its strips, merged below
  // (every workgroup of a launch must be resident: the grid stays below the CU count)
  int groups = (!no_groups && nsmax > spg && n <= (track ? (size_t)32 : (size_t)8)) ? (nsmax + spg - 1) / spg : 1;
  if ((size_t)groups * n > (size_t)dev_cus()) groups = 1;
  const size_t og = track ? (size_t)groups : 1;                    // result slots per job
  for (size_t k = 0; k < n; ++k) {
    WaveJob &j = jobs[k];
    const int ns = strip_count(q.len[j.q], R);
    nwmax = std::max(nwmax, std::min(ns, kStripMaxWaves));
    if (!track) { j.dirs_off = dirs_total; dirs_total += strip_dirs_bytes(j.nb, ns, R); }
    if (groups > 1) { goff[k] = gtotal; gtotal += (((size_t)groups * ((size_t)j.nb + 192) + 1) & ~(size_t)1) + 2 * (size_t)groups; }   // rows (even count) + 64-bit counters
    else if (ns > kStripMaxWaves) { goff[k] = gtotal; gtotal += 2 * ((size_t)j.nb + 192); }
  }
  if (ctx->wprobs.ensure(n * sizeof(StripProblem)) || ctx->outs_i.ensure(n * og * 16) || ctx->outs_f.ensure((n + n * og) * 4) ||
      (dirs_total && ctx->dirs.ensure(dirs_total)) || (gtotal && ctx->brow.ensure(gtotal * 4)))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(strip scratch) failed");
  std::vector<StripProblem> pr(n);
  // test hook: drives the pipeline's bounded-wait expiry (the workgroup raises its status word and drains)
  const int32_t fault = opt().fault_inject == 1 ? 1 : 0;   // (mi355_sw_set_option only: never read from the environment)
  for (size_t k = 0; k < n; ++k) {
    const WaveJob &j = jobs[k];
    StripProblem &s = pr[k];
    s.a = q.bytes.as<uint8_t>() + q.off[j.q];
    s.na = q.len[j.q];
    s.b = (use_table ? ref.codes.as<uint8_t>() : ref.bytes.as<uint8_t>()) + rg.lo + j.s_lo;
    s.nb = j.nb;
    s.nstrips = strip_count(q.len[j.q], R);
    s.nw = std::min(s.nstrips, kStripMaxWaves);
    s.gbound = (groups > 1 || s.nstrips > kStripMaxWaves) ? ctx->brow.as<float>() + goff[k] : nullptr;
    s.gstride = (int64_t)j.nb + 192;
    s.spg = groups > 1 ? spg : 0;
    s.gcount = groups > 1 ? reinterpret_cast<long long *>(ctx->brow.as<float>() + goff[k] + (((size_t)groups * ((size_t)j.nb + 192) + 1) & ~(size_t)1))
                          : nullptr;                            // 8-byte aligned: every offset is an even number of floats
    s.dirs = track ? nullptr : reinterpret_cast<uint32_t *>(ctx->dirs.as<uint8_t>() + j.dirs_off);
    s.target = j.target;
    s.own_lo = j.own_lo;
    s.col_offset = j.s_lo;
    s.full_n = rg.hi - rg.lo;
    s.cell = ctx->outs_i.as<int64_t>() + 2 * k * og;
    s.status = ctx->outs_f.as<int32_t>() + k;
    s.best = ctx->outs_f.as<float>() + n + k * og;
    s.fault = fault;
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->outs_f.p, 0, (n + n * og) * 4, ctx->stream));
  if (og > 1) HIPCHK(ctx, hipMemsetAsync(ctx->outs_i.p, 0, n * og * 16, ctx->stream));       // (workgroups without strips report nothing)
  if (groups > 1) HIPCHK(ctx, hipMemsetAsync(ctx->brow.p, 0, gtotal * 4, ctx->stream));      // the progress counters start at 0
  HIPCHK(ctx, hipMemcpyAsync(ctx->wprobs.p, pr.data(), n * sizeof(StripProblem), hipMemcpyHostToDevice, ctx->stream));
  WaveScoring sc;
  sc.match = p.match; sc.mismatch = p.mismatch; sc.gap = p.gap;
  const U8Params u = u8_params(p);
  sc.u8M = (float)u.M; sc.u8X = (float)u.X; sc.u8G = (float)u.G;
  const bool u8 = p.semantics == MI355_SW_U8SAT;
  const StripProblem *dp = ctx->wprobs.as<StripProblem>();
  const dim3 grid((unsigned)(n * groups)), block((unsigned)(64 * (groups > 1 ? spg : nwmax)));
  const float *gtab = use_table ? ctx->ftab.as<float>() : nullptr;
  if (use_table && (size_t)257 * ref.ncodes * 4 > 48 * 1024) {
    const int lds = 257 * ref.ncodes * 4;
#define STRIP_LDS_ATTR(r)                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_strip_kernel<r, false, kStripTrack, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_strip_kernel<r, false, kStripDirs, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_strip_kernel<r, false, kStripMax, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    STRIP_LDS_ATTR(3) STRIP_LDS_ATTR(5) STRIP_LDS_ATTR(8) STRIP_LDS_ATTR(10) STRIP_LDS_ATTR(16)
#undef STRIP_LDS_ATTR
  }
  if (R == 3) launch_strip<3>(u8, track, grid, block, ctx->stream, dp, sc, gtab, ref.ncodes, groups, maxmode);
  else if (R == 5) launch_strip<5>(u8, track, grid, block, ctx->stream, dp, sc, gtab, ref.ncodes, groups, maxmode);
  else if (R == 8) launch_strip<8>(u8, track, grid, block, ctx->stream, dp, sc, gtab, ref.ncodes, groups, maxmode);
  else if (R == 10) launch_strip<10>(u8, track, grid, block, ctx->stream, dp, sc, gtab, ref.ncodes, groups, maxmode);
  else launch_strip<16>(u8, track, grid, block, ctx->stream, dp, sc, gtab, ref.ncodes, groups, maxmode);
  HIPCHK(ctx, hipGetLastError());
  path_note(ctx, "strip[R=%d,mode=%s,grouped=%d,lut=%d,u8=%d]", R, maxmode ? "max" : (track ? "track" : "dirs"), (int)(groups > 1), (int)use_table, (int)u8);
  std::vector<int32_t> st(n);
  std::vector<int64_t> ci(2 * n * og);
  std::vector<float> bv(maxmode ? n * og : 0);
  if (maxmode) HIPCHK(ctx, hipMemcpyAsync(bv.data(), ctx->outs_f.as<float>() + n, n * og * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(st.data(), ctx->outs_f.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (track) HIPCHK(ctx, hipMemcpyAsync(ci.data(), ctx->outs_i.p, n * og * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; ++k) {
    if (st[k] != 0) {
      // strips dealt to several workgroups that wait for each other: once more with every strip of a problem in ONE workgroup
      if (groups > 1 && !tl_no_wait) { tl_no_wait = true; ctx->wait_retries += 1; return run_strip(ctx, ref, q, rg, p, jobs, R); }
      return fail(ctx, MI355_SW_ENODEV, "internal: strip pipeline wait expired");
    }
    if (!track) continue;
    // the workgroups of a job: the greatest value (kStripMax), then the smallest storage-order key
    float best = 0.0f;
    int64_t bi = 0, bj = 0;
    unsigned long long bkey = ~0ull;
    for (size_t g = 0; g < og; ++g) {
      const int64_t i = ci[2 * (k * og + g)], j = ci[2 * (k * og + g) + 1];
      if (i <= 0) continue;
      const float v = maxmode ? bv[k * og + g] : jobs[k].target;
      const unsigned long long key = host_order_key(p.semantics, i, j, q.len[jobs[k].q], rg.hi - rg.lo);
      if (v > best || (v == best && key < bkey)) { best = v; bi = i; bj = j; bkey = key; }
    }
    jobs[k].ci = bi; jobs[k].cj = bj;
    jobs[k].best = maxmode ? best : (bi > 0 ? jobs[k].target : -1.0f);
  }
  return 0;
}

// Traceback of located alignments with the wave kernel: decisions over a window that ends at the argmax along
// the streamed side, grown on demand; then the greedy walk.  orient as WaveJob.
int wave_trace(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
               int orient, const std::vector<int> &qidx, const std::vector<Located> &loc, std::vector<TraceOut> &tout,
               bool strips = false /* long queries: pipelined strip kernel (orient 0 only) */,
               const ScoreTable *table = nullptr /* strips with table scoring: its smax / gap bound the margins */) {
  HostTrace trace_("wave_trace");
  const int64_t nref = rg.hi - rg.lo;
  tout.assign(qidx.size(), TraceOut());
  std::vector<size_t> todo;
  for (size_t k = 0; k < qidx.size(); ++k) if (loc[k].score > 0) todo.push_back(k);
  // exactness margin along the stream: a positive path ending at a stream index spans fewer than
  // na + smax*na/g stream positions (DESIGN.md §3.3 with the roles of the two sequences as given)
  int64_t maxna = 1;
  for (size_t k : todo) maxna = std::max<int64_t>(maxna, orient == 0 ? q.len[qidx[k]] : nref);
  Margin mg;
  if (p.semantics == MI355_SW_U8SAT) { const U8Params u = u8_params(p); mg = make_margin(u.M, u.G, true, (double)maxna); }
  else if (table != nullptr && table->ok) mg = table->margin((double)maxna);
  else {
    auto whole = [](float v) { return v == std::floor(v) && std::fabs(v) <= 1048576.0f; };
    mg = make_margin(p.match, p.gap, whole(p.match) && whole(p.mismatch) && whole(p.gap), (double)maxna);
  }
  // ... and a cell whose lane-side index is a (its path is confined to a rows / columns) is exact a + ceil(a*smax/g)
  // positions into the window: the window needs that margin at the argmax plus room for the walk's excursions
  // along the stream; the walk kernel checks every cell it visits
  const float slope = (float)mg.slope();
  auto lane_need = [&](int64_t a) { return mg.finite() ? clamp_cols((double)a + std::ceil((double)a * (double)slope) + 2.0) : kColsMax; };
  std::vector<int64_t> budget(qidx.size()), warm(qidx.size());
  for (size_t k : todo) {
    const int64_t na = orient == 0 ? q.len[qidx[k]] : nref;
    budget[k] = na / 8 + 64;
    warm[k] = mg.finite() ? mg.cols((double)na) : (int64_t)1 << 40;
  }
  while (!todo.empty()) {
    std::vector<size_t> next;
    size_t pos = 0;
    while (pos < todo.size()) {
      std::vector<WaveJob> jobs;
      std::vector<size_t> owner;
      size_t bytes = 0;
      while (pos < todo.size() && jobs.size() < 262144) {
        const size_t k = todo[pos];
        const int qi = qidx[k];
        const int64_t na = orient == 0 ? q.len[qi] : nref;
        const int64_t s_end = orient == 0 ? loc[k].iy : loc[k].ix;   // 1-based stream index of the argmax
        const int64_t a_end = orient == 0 ? loc[k].ix : loc[k].iy;   // lane-side index of the argmax
        const int64_t wl = std::max<int64_t>(0, s_end - (budget[k] + std::min(warm[k], lane_need(a_end))));
        const int64_t nb = s_end - wl;
        const int rfew = strip_R_few((int)na, 1, false);        // the instance with the most decision bytes this job can get
        const size_t need = strips ? strip_dirs_bytes(nb, strip_count((int)na, rfew), rfew)
                                   : wave_dirs_bytes(nb, 32);      // upper bound whatever instance the group gets
        if (need > kDirsBudget) return fail(ctx, MI355_SW_ENOTSUP, "traceback window exceeds the device scratch budget");
        if (!jobs.empty() && bytes + need > kDirsBudget) break;
        WaveJob j;
        j.q = qi; j.orient = orient; j.s_lo = wl; j.nb = (int32_t)nb; j.track = false; j.dirs = true;
        jobs.push_back(j); owner.push_back(k);
        bytes += need;
        ++pos;
      }
      int gmax = 0;
      for (const WaveJob &j : jobs) gmax = std::max(gmax, orient == 0 ? q.len[j.q] : (int)nref);
      int groupR = strips ? strip_R(gmax) : wave_R(gmax);           // the instance this group runs on
      // a few long alignments whose strips are dealt to several workgroups (run_strip): five rows per lane make twice
      // the strips, i.e. twice the workgroups, of ten (config 5: 32 strips on 8 CUs); three below 2560 rows
      if (strips && !opt().no_strip_groups) groupR = strip_R_few(gmax, jobs.size(), false);
      int rc = strips ? run_strip(ctx, ref, q, rg, p, jobs, groupR) : run_wave(ctx, ref, q, rg, p, jobs);
      if (rc) return rc;
      // walk: measure, lay out, write (only the bytes that exist are copied back)
      const size_t n = jobs.size();
      if (ctx->walkp.ensure(n * sizeof(WaveWalk) + n * 24 + n * 8 + 64))
        return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(walk scratch) failed");
      if (ctx->pin_walk.ensure(n * sizeof(WaveWalk)) || ctx->pin_out.ensure(n * 32 + 64))
        return fail(ctx, MI355_SW_ENOMEM, "hipHostMalloc(walk staging) failed");
      WaveWalk *wp = ctx->pin_walk.as<WaveWalk>();
      int64_t *wout = reinterpret_cast<int64_t *>(ctx->walkp.as<uint8_t>() + n * sizeof(WaveWalk));
      int64_t *woffs = wout + 3 * n;
      parallel_for(n, [&](size_t t0, size_t t1) {
      for (size_t t = t0; t < t1; ++t) {
        const WaveJob &j = jobs[t];
        const size_t k = owner[t];
        WaveWalk &w = wp[t];
        const int na = orient == 0 ? q.len[j.q] : (int)nref;
        w.x = q.bytes.as<uint8_t>() + q.off[j.q];
        w.y = ref.bytes.as<uint8_t>() + rg.lo;
        w.dirs = reinterpret_cast<const uint32_t *>(ctx->dirs.as<uint8_t>() + j.dirs_off);
        w.na = na;
        w.nb = j.nb;
        w.orient = orient;
        w.R = groupR;
        w.lanes = strips ? 64 * strip_count(na, groupR) : 16;
        w.skew = strips ? 0 : 1; w.row0 = 0;
        w.need_slope = slope;
        w.b_offset = j.s_lo;
        w.start_i = loc[k].ix; w.start_j = loc[k].iy;
        w.exact_from = j.s_lo == 0 ? 0 : j.s_lo + warm[k];
        w.cap = na + j.nb + 2;                                      // a walk inside the window emits <= na + nb pairs
        w.out = wout + 3 * t;
        w.zchunk = 0; w.zwarm = 0;
      }
      });
      HIPCHK(ctx, hipMemcpyAsync(ctx->walkp.p, wp, n * sizeof(WaveWalk), hipMemcpyHostToDevice, ctx->stream));
      const unsigned wblocks = (unsigned)((n + 63) / 64);
      int64_t *wo = ctx->pin_out.as<int64_t>();                     // [3n] walk outputs, then [n] offsets
      int64_t *offs = wo + 3 * n;
      // the consensus bytes land in a pinned buffer that stays alive until the next call (TraceOut points into it)
      if (ctx->pin_cons.size() <= ctx->cons_used) ctx->pin_cons.resize(ctx->cons_used + 1);
      PinBuf &cons = ctx->pin_cons[ctx->cons_used];
      size_t captot = 0;
      for (size_t t = 0; t < n; ++t) captot += 2 * (size_t)wp[t].cap;
      const bool one_pass = n <= 4096 && captot <= ((size_t)32 << 20);
      if (one_pass) {
        // few walks: each writes into a buffer of its own capacity (x at offs, y at offs + cap) in one pass
        size_t at = 0;
        for (size_t t = 0; t < n; ++t) { offs[t] = (int64_t)at; at += 2 * (size_t)wp[t].cap; }
        if (ctx->cons.ensure(captot + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(consensus) failed");
        HIPCHK(ctx, hipMemcpyAsync(woffs, offs, n * 8, hipMemcpyHostToDevice, ctx->stream));
        path_note(ctx, strips && orient == 0 ? "walk_long" : "walk_wave");
        if (strips && orient == 0)     // a few long walks: one wavefront each, looking ahead along the diagonal
          hipLaunchKernelGGL(sw_wave_walk_long_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, (const WaveWalk *)ctx->walkp.as<WaveWalk>(), (int)n,
                             ctx->cons.as<char>(), (const int64_t *)woffs);
        else
          hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkBoth>, dim3(wblocks), dim3(64), 0, ctx->stream, ctx->walkp.as<WaveWalk>(), (int)n,
                             ctx->cons.as<char>(), (const int64_t *)woffs);
        HIPCHK(ctx, hipGetLastError());
        if (cons.ensure(captot + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipHostMalloc(consensus) failed");
        HIPCHK(ctx, hipMemcpyAsync(wo, wout, n * 24, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(cons.p, ctx->cons.p, captot, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        path_note(ctx, "walk_wave");
        hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkMeasure>, dim3(wblocks), dim3(64), 0, ctx->stream, ctx->walkp.as<WaveWalk>(), (int)n,
                           (char *)nullptr, (const int64_t *)nullptr);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(wo, wout, n * 24, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        size_t ctot = 0;
        for (size_t t = 0; t < n; ++t) { offs[t] = (int64_t)ctot; if (wo[3 * t + 2] == 0) ctot += 2 * (size_t)wo[3 * t]; }
        if (ctx->cons.ensure(ctot + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(consensus) failed");
        HIPCHK(ctx, hipMemcpyAsync(woffs, offs, n * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkWrite>, dim3(wblocks), dim3(64), 0, ctx->stream, ctx->walkp.as<WaveWalk>(), (int)n,
                           ctx->cons.as<char>(), (const int64_t *)woffs);
        HIPCHK(ctx, hipGetLastError());
        if (cons.ensure(ctot + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipHostMalloc(consensus) failed");
        if (ctot) HIPCHK(ctx, hipMemcpyAsync(cons.p, ctx->cons.p, ctot, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      }
      ctx->cons_used++;                                             // the strings stay where the copy put them
      const char *base = cons.as<char>();
      parallel_for(n, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t) {
          if (wo[3 * t + 2] != 0) continue;
          const size_t k = owner[t];
          const size_t len = (size_t)wo[3 * t];
          tout[k].len = len;
          tout[k].cx = base + offs[t];
          tout[k].cy = base + offs[t] + (one_pass ? (size_t)wp[t].cap : len);
          tout[k].pos = (uint32_t)wo[3 * t + 1];
        }
      });
      for (size_t t = 0; t < n; ++t) {                              // the rare ones that need a wider window
        const int st = (int)wo[3 * t + 2];
        if (st == 1) { budget[owner[t]] *= 4; next.push_back(owner[t]); }
        else if (st != 0) return fail(ctx, MI355_SW_ENOTSUP, "consensus longer than |x| + |y|");
      }
    }
    todo.swap(next);
  }
  return 0;
}

}  // namespace
