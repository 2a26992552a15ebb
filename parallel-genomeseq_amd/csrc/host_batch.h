// host_batch.h — many small WHOLE problems on device-built job lists (sw_batch_kernels.h): the UniProt-shaped batch
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
//
// Float engine, identity scoring, problems whose short side fits the register wavefront (<= 512): sorted positions
// [first, first + count) of the batch against one range of the reference, all with the same orientation.
//   one pass of sw_wave_kernel<TRACK, DIRS>: first maximum in storage order + every cell's greedy decision (or, on
//   sw_wave_prof_kernel, a checkpointed first pass and decisions for a window in front of each argmax);
//   walks measured, laid out by a device scan, written; results copied down once.
// Nothing per-alignment is built on the host before the results arrive (src/mpi_sw_solve_uniprot.cpp:95-138: every
// database sequence x against the one query y — 561 356 of them in config 4).
namespace {

// exclusive scan of v[0..n) in place on the stream; the grand total lands in *total_dev (device)
int device_scan(mi355_sw_ctx *ctx, int64_t *v, int64_t n, int64_t *total_dev) {
  const int64_t per = (int64_t)kScanBlock * kScanItems;
  const int nb = (int)std::max<int64_t>(1, (n + per - 1) / per);
  if (ctx->scan.ensure((size_t)nb * 8 + 64)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(scan) failed");
  int64_t *partial = ctx->scan.as<int64_t>();
  hipLaunchKernelGGL(scan_partials, dim3(nb), dim3(kScanBlock), 0, ctx->stream, (const int64_t *)v, n, partial);
  hipLaunchKernelGGL(scan_of_partials, dim3(1), dim3(kScanBlock), 0, ctx->stream, partial, nb, total_dev);
  hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(kScanBlock), 0, ctx->stream, v, n, (const int64_t *)partial);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

template <int R, int ORIENT>
void launch_wave_batch(bool dirs, unsigned blocks, hipStream_t st, const WaveProblem *pr, int n, const WaveScoring &sc) {
  if (dirs) hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, false, true, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
  else hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, false, true, false>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
}

// loc / tout / handled are indexed by query id.  Long ranges are cut so that a launch's decisions fit the scratch
// budget; a single problem beyond it is left to the windowed path (handled stays 0).
int exact_full_device(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
                      size_t first, size_t count, int orient, bool want_trace, std::vector<Located> &loc,
                      std::vector<TraceOut> &tout, std::vector<char> &handled) {
  if (count == 0) return 0;
  const int64_t nref = rg.hi - rg.lo;
  const int maxna = orient == 0 ? q.len[q.order[first + count - 1]] : (int)nref;
  int R = wave_R(maxna);
  if (orient == 1 && wave_prof_ok(ref, p, wave_prof_R((int)nref), (int)nref, true)) R = wave_prof_R((int)nref);
  const int W = (R + 15) / 16;
  const int64_t stream_total = orient == 0 ? (int64_t)count * nref : q.cumlen[first + count] - q.cumlen[first];
  // lanes = columns of the shared second sequence: the profile kernel (three-op cell, first maximum per lane).  With traceback,
  // decisions are only made where the walk goes: a first pass keeps (maximum, cell) and saves the wavefront every kCkptEvery steps, a
  // second resumes every problem kWindowGuard + 1 .. + kCkptEvery rows in front of its argmax and stops at it (a database of mostly unrelated
  // sequences: walks of a dozen cells; a walk that leaves its window hands the problem to the host-driven path below).
  const bool prof = orient == 1 && wave_prof_ok(ref, p, R, (int)nref, true);
  const bool windows = prof && want_trace && !opt().no_wave_window;
  // Long streams in pieces (sw_batch_kernels.h; DESIGN.md §3.3 lemma L12): kPieceRows own rows behind a warm-up of the L1 margin along the stream (a path
  // that ends in row i of x spans fewer than |y| + ceil(smax |y| / g) rows) plus the reach of a decision window in front of an
  // own row (kWindowGuard + kCkptEvery), so that own rows and the checkpoints their windows resume from are exact.  Only where the
  // pass keeps no whole-problem decisions (score + argmax, or checkpointed windows).
  std::vector<int32_t> pc_seq, pc_start, pc_rows, pc_first;
  std::vector<int64_t> pc_before;
  int nlong = 0;
  int64_t piece_stream = 0, long_stream = 0;
  if (prof && (windows || !want_trace) && !opt().no_wave_pieces) {
    const Margin mg = make_margin(p.match, p.gap, true, (double)nref);   // (the profile kernel's scores are dyadic: exact arithmetic)
    const int64_t warm = mg.finite() ? (mg.cols((double)nref) + kWindowGuard + kCkptEvery + 63) / 64 * 64 : -1;
    const int64_t split_above = kPieceRows + warm;                   // (shorter sequences would not get shorter as pieces)
    if (warm > 0 && warm <= 4 * kPieceRows) {
      pc_first.push_back(0);
      for (size_t k = 0; k < count; ++k) {                           // launch order: longest first
        const int64_t m = q.len[q.order[first + count - 1 - k]];
        if (m <= split_above) break;
        for (int64_t own = 0; own < m; own += kPieceRows) {
          const int64_t start = std::max<int64_t>(0, own - warm), end = std::min<int64_t>(m, own + kPieceRows);
          pc_seq.push_back((int32_t)k); pc_start.push_back((int32_t)start); pc_rows.push_back((int32_t)(end - start));
          pc_before.push_back(piece_stream);
          piece_stream += end - start;
        }
        pc_first.push_back((int32_t)pc_seq.size());
        long_stream += m;
        ++nlong;
      }
    }
  }
  const size_t npieces = pc_seq.size();
  const size_t nprob = npieces + (count - (size_t)nlong);           // problems of the first pass
  const int64_t stream_all = stream_total - long_stream + piece_stream;
  const size_t dirs_total = !want_trace ? 0 : windows ? count * (size_t)kWindowRows * 16 * (size_t)W * 4
                                                      : (size_t)batch_dirs_offset(stream_total, (int64_t)count, W);
  const size_t ckpt_total = windows ? ((size_t)batch_ckpt_row(stream_all, (int64_t)nprob) + 1) * 16 * (size_t)(R + 1) * 4 : 0;
  if (dirs_total + ckpt_total > kDirsBudget || count > ((size_t)1 << 30)) {
    if (count == 1) return 0;
    const size_t half = count / 2;
    int rc = exact_full_device(ctx, ref, q, rg, p, first, half, orient, want_trace, loc, tout, handled);
    if (rc) return rc;
    return exact_full_device(ctx, ref, q, rg, p, first + half, count - half, orient, want_trace, loc, tout, handled);
  }
  HostTrace trace_("exact_full_device");
  const size_t n = count;
  // device scratch: problems, (best, cell), decisions, walk descriptors + (len, pos, status) + consensus offsets + total
  const size_t walk_bytes = n * sizeof(WaveWalk) + n * 24 + (n + 1) * 8 + 64;
  // problems of the first pass, then [n] decision windows; per problem (best, cell), then per sequence (best, cell, problem)
  const size_t pc_bytes = npieces ? (4 * npieces + (size_t)nlong + 1) * 4 + npieces * 8 + 64 : 0;
  if (ctx->wprobs.ensure((nprob + n) * sizeof(WaveProblem)) || ctx->outs_f.ensure((nprob + n) * 4 + n * 4 + 64) || ctx->outs_i.ensure((nprob + n) * 16) ||
      (dirs_total && ctx->dirs.ensure(dirs_total)) || (ckpt_total && ctx->ckpt.ensure(ckpt_total)) ||
      (want_trace && ctx->walkp.ensure(walk_bytes)) || (pc_bytes && ctx->pieces.ensure(pc_bytes)))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(batch scratch) failed");
  BatchWaveArgs a;
  a.qbytes = q.bytes.as<uint8_t>(); a.qoff = q.offs.as<int64_t>(); a.qlen = q.lens.as<int32_t>();
  a.qsel = q.sel.as<int32_t>(); a.qcum = q.cum.as<int64_t>();
  a.ref = ref.bytes.as<uint8_t>() + rg.lo; a.nref = nref;
  a.first = (int)first; a.count = (int)n; a.orient = orient; a.W = W;
  a.dirs = want_trace ? ctx->dirs.as<uint32_t>() : nullptr;
  a.probs = ctx->wprobs.as<WaveProblem>();
  a.best = ctx->outs_f.as<float>();
  a.cell = ctx->outs_i.as<int64_t>();
  a.ckpt = windows ? ctx->ckpt.as<float>() : nullptr;
  a.f16 = 0;
  a.R = R;
  a.nlong = nlong; a.npieces = (int)npieces; a.piece_rows = kPieceRows; a.piece_stream = piece_stream; a.long_stream = long_stream;
  a.pc_seq = a.pc_start = a.pc_rows = a.pc_first = nullptr; a.pc_before = nullptr;
  if (npieces) {
    // the piece table: a few thousand entries (the sequences beyond ~1.5 k residues), rebuilt and sent with every call
    uint8_t *dev = ctx->pieces.as<uint8_t>();
    const size_t o_before = 0, o_seq = npieces * 8, o_start = o_seq + npieces * 4, o_rows = o_start + npieces * 4, o_first = o_rows + npieces * 4;
    ctx->h_pieces.resize(pc_bytes);
    memcpy(ctx->h_pieces.data() + o_before, pc_before.data(), npieces * 8);
    memcpy(ctx->h_pieces.data() + o_seq, pc_seq.data(), npieces * 4);
    memcpy(ctx->h_pieces.data() + o_start, pc_start.data(), npieces * 4);
    memcpy(ctx->h_pieces.data() + o_rows, pc_rows.data(), npieces * 4);
    memcpy(ctx->h_pieces.data() + o_first, pc_first.data(), ((size_t)nlong + 1) * 4);
    HIPCHK(ctx, hipMemcpyAsync(dev, ctx->h_pieces.data(), o_first + ((size_t)nlong + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    a.pc_before = reinterpret_cast<const int64_t *>(dev + o_before);
    a.pc_seq = reinterpret_cast<const int32_t *>(dev + o_seq);
    a.pc_start = reinterpret_cast<const int32_t *>(dev + o_start);
    a.pc_rows = reinterpret_cast<const int32_t *>(dev + o_rows);
    a.pc_first = reinterpret_cast<const int32_t *>(dev + o_first);
  }
  // per sequence: results after the merge (whole sequences without pieces: the problem arrays themselves)
  a.sbest = npieces ? ctx->outs_f.as<float>() + nprob : a.best;
  a.scell = npieces ? ctx->outs_i.as<int64_t>() + 2 * nprob : a.cell;
  a.sprob = reinterpret_cast<int32_t *>(ctx->outs_f.as<float>() + nprob + n);
  a.probs2 = ctx->wprobs.as<WaveProblem>() + nprob;
  const unsigned sblocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(batch_wave_setup, dim3((unsigned)((nprob + 255) / 256)), dim3(256), 0, ctx->stream, a);
  WaveScoring sc;
  sc.match = p.match; sc.mismatch = p.mismatch; sc.gap = p.gap;
  sc.u8M = sc.u8X = sc.u8G = 0.0f;
  const unsigned blocks = (unsigned)((n + 15) / 16), pblocks = (unsigned)((nprob + 15) / 16);
  const WaveProblem *dp = ctx->wprobs.as<WaveProblem>();
  const WaveProblem *dp_walk = dp;                                   // the problems the walks read their decisions from
  // the first pass without decisions: two problems per slot on packed float16 cells where every value fits (sw_wave_kernel.h)
  int64_t max_stream = 0;
  for (size_t e = 0; e < npieces; ++e) max_stream = std::max<int64_t>(max_stream, pc_rows[e]);
  if (nprob > npieces) max_stream = std::max<int64_t>(max_stream, q.len[q.order[first + count - 1 - (size_t)nlong]]);
  int f16 = 0;
  if (windows) {
    int rc = launch_wave_prof16(ctx, ref, p, R, (int)nref, max_stream, dp, (int)nprob);
    if (rc < 0) return rc;
    f16 = rc == 0;
    a.f16 = f16;
    if (!f16) rc = launch_wave_prof(ctx, ref, p, R, (int)nref, true, false, pblocks, dp, (int)nprob);
    else rc = 0;
    if (rc) return rc < 0 ? rc : fail(ctx, MI355_SW_ENODEV, "internal: the profile kernel refused a launch it had accepted");
    hipLaunchKernelGGL(batch_seq_results, dim3(sblocks), dim3(256), 0, ctx->stream, a);   // (always: it also names every sequence's problem)
    hipLaunchKernelGGL(batch_window_setup, dim3(sblocks), dim3(256), 0, ctx->stream, a);
    rc = launch_wave_prof(ctx, ref, p, R, (int)nref, false, true, blocks, (const WaveProblem *)a.probs2, (int)n);
    if (rc) return rc < 0 ? rc : fail(ctx, MI355_SW_ENODEV, "internal: the profile kernel refused a launch it had accepted");
    dp_walk = a.probs2;
  } else {
    int prof_rc = prof && !want_trace ? launch_wave_prof16(ctx, ref, p, R, (int)nref, max_stream, dp, (int)nprob) : 1;
    if (prof_rc < 0) return prof_rc;
    f16 = prof_rc == 0;
    if (!f16) prof_rc = prof ? launch_wave_prof(ctx, ref, p, R, (int)nref, true, want_trace, pblocks, dp, (int)nprob) : 1;
    if (prof_rc < 0) return prof_rc;
    if (prof_rc == 0 && npieces) hipLaunchKernelGGL(batch_seq_results, dim3(sblocks), dim3(256), 0, ctx->stream, a);
#define BATCH_WAVE(r)                                                                                                   \
  if (orient == 0) launch_wave_batch<r, 0>(want_trace, blocks, ctx->stream, dp, (int)n, sc);                            \
  else launch_wave_batch<r, 1>(want_trace, blocks, ctx->stream, dp, (int)n, sc);
    if (prof_rc == 0) { }
    else if (R == 10) { BATCH_WAVE(10) } else if (R == 20) { BATCH_WAVE(20) } else { BATCH_WAVE(32) }
#undef BATCH_WAVE
  }
  HIPCHK(ctx, hipGetLastError());
  path_note(ctx, "devlist[orient=%d,R=%d,prof=%d,f16=%d,windows=%d,trace=%d,pieces=%d]", orient, R, (int)prof, f16, (int)windows, (int)want_trace, (int)(npieces != 0));

  // Results.  When this launch holds every non-empty query of the batch (the UniProt shape: one launch), a last kernel writes one
  // 40-byte record per alignment AT ITS QUERY ID and the host's loop over them reads and writes in order; else the per-launch-
  // position arrays come down and the loop scatters: [best n x 4][cell n x 16][wout n x 24][offs (n + 1) x 8].
  const size_t nq = q.nq;
  const bool by_id = !opt().no_devlist_by_id && first + n == nq && (first == 0 || q.len[q.order[first - 1]] == 0);
  path_note(ctx, "devlist_results[by_id=%d]", (int)by_id);
  const size_t o_best = 0, o_cell = (n * 4 + 15) & ~(size_t)15, o_wout = o_cell + n * 16, o_offs = o_wout + n * 24;
  if (ctx->pin_out.ensure(std::max(o_offs + (n + 1) * 8, nq * sizeof(BatchRec) + 16) + 64)) return fail(ctx, MI355_SW_ENOMEM, "hipHostMalloc(batch results) failed");
  if (by_id && ctx->recs.ensure(nq * sizeof(BatchRec))) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(batch records) failed");
  uint8_t *pin = ctx->pin_out.as<uint8_t>();
  float *h_best = reinterpret_cast<float *>(pin + o_best);
  int64_t *h_cell = reinterpret_cast<int64_t *>(pin + o_cell);
  int64_t *h_wout = reinterpret_cast<int64_t *>(pin + o_wout);
  int64_t *h_offs = reinterpret_cast<int64_t *>(pin + o_offs);
  int64_t *h_total = reinterpret_cast<int64_t *>(pin + nq * sizeof(BatchRec));   // by_id: behind the records
  const BatchRec *h_rec = reinterpret_cast<const BatchRec *>(pin);
  const BatchRecScore *h_rec_s = reinterpret_cast<const BatchRecScore *>(pin);
  if (!by_id) {
    HIPCHK(ctx, hipMemcpyAsync(h_best, a.sbest, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_cell, a.scell, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  }
  const char *cons_base = nullptr;
  bool passes_done = false;
  struct InFlight {                                                // (no exit leaves a download running into the staging buffer)
    hipStream_t s; bool armed = false;
    ~InFlight() { if (armed) (void)hipStreamSynchronize(s); }
  } records_in_flight{ctx->copy_stream};
  if (want_trace) {
    WaveWalk *walks = ctx->walkp.as<WaveWalk>();
    int64_t *wout = reinterpret_cast<int64_t *>(ctx->walkp.as<uint8_t>() + n * sizeof(WaveWalk));
    int64_t *offs = wout + 3 * n;                                  // [n] sizes -> offsets, then [1] total
    BatchWalkArgs b;
    b.probs = dp_walk; b.qbytes = a.qbytes; b.qoff = a.qoff; b.qsel = a.qsel; b.ref = a.ref;
    b.first = (int)first; b.count = (int)n; b.orient = orient; b.R = R;
    b.best = a.sbest; b.cell = a.scell; b.walks = walks; b.wout = wout;
    hipLaunchKernelGGL(batch_walk_setup, dim3(sblocks), dim3(256), 0, ctx->stream, b);
    const unsigned wblocks = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkMeasure>, dim3(wblocks), dim3(64), 0, ctx->stream, (const WaveWalk *)walks, (int)n,
                       (char *)nullptr, (const int64_t *)nullptr);
    hipLaunchKernelGGL(batch_walk_sizes, dim3(sblocks), dim3(256), 0, ctx->stream, (const int64_t *)wout, (int)n, offs);
    HIPCHK(ctx, hipGetLastError());
    int rc = device_scan(ctx, offs, (int64_t)n, offs + n);
    if (rc) return rc;
    if (by_id) {
      hipLaunchKernelGGL(batch_records_by_id, dim3(sblocks), dim3(256), 0, ctx->stream, a, (const int64_t *)wout, (const int64_t *)offs, ctx->recs.as<BatchRec>());
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(h_total, offs + n, 8, hipMemcpyDeviceToHost, ctx->stream));
      // the records (40 bytes per alignment: 22 MB for config 4) come down on the copy stream, beside the write pass of the walks
      // and the download of the strings — the wait below is only for the 8 bytes that size those
      HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev[6], 0));
      records_in_flight.armed = true;
      HIPCHK(ctx, hipMemcpyAsync(pin, ctx->recs.p, nq * sizeof(BatchRec), hipMemcpyDeviceToHost, ctx->copy_stream));
    } else {
      HIPCHK(ctx, hipMemcpyAsync(h_offs, offs, (n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(h_wout, wout, n * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (ctx->while_device_works) { ctx->while_device_works(); ctx->while_device_works = nullptr; }
    { HostTrace t_("  batch: passes + walk measure");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }            // the one mid-call round trip: how many bytes to expect
    const size_t ctot = (size_t)(by_id ? *h_total : h_offs[n]);
    if (ctx->pin_cons.size() <= ctx->cons_used) ctx->pin_cons.resize(ctx->cons_used + 1);
    PinBuf &cons = ctx->pin_cons[ctx->cons_used];
    if (ctx->cons.ensure(ctot + 16) || cons.ensure(ctot + 16)) return fail(ctx, MI355_SW_ENOMEM, "consensus buffers: allocation failed");
    hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkWrite>, dim3(wblocks), dim3(64), 0, ctx->stream, (const WaveWalk *)walks, (int)n,
                       ctx->cons.as<char>(), (const int64_t *)offs);
    HIPCHK(ctx, hipGetLastError());
    if (ctot) HIPCHK(ctx, hipMemcpyAsync(cons.p, ctx->cons.p, ctot, hipMemcpyDeviceToHost, ctx->stream));
    ctx->cons_used++;
    cons_base = cons.as<char>();
  } else if (by_id) {
    hipLaunchKernelGGL(batch_score_records_by_id, dim3(sblocks), dim3(256), 0, ctx->stream, a, ctx->recs.as<BatchRecScore>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));          // (the pass is done; the records are still to come down)
    passes_done = true;
    HIPCHK(ctx, hipMemcpyAsync(pin, ctx->recs.p, nq * sizeof(BatchRecScore), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (ctx->while_device_works) { ctx->while_device_works(); ctx->while_device_works = nullptr; }
  { HostTrace t_("  batch: to the last download");
  // the result loop below takes the pool: its workers are woken (they then spin for a millisecond) as close in front of it as a
  // wait allows — behind the pass when only the download of the records is left, else in front of the last wait
  if (passes_done) HIPCHK(ctx, hipEventSynchronize(ctx->ev[6]));
  if (n >= tl_pool_from) WorkerPool::get().nudge();
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (records_in_flight.armed) { records_in_flight.armed = false; HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream)); } }
  HostTrace t_results("  batch: results on the host");
  std::atomic<bool> bad{false};
  std::atomic<size_t> left{0}, beyond{0}, done{0};
  ViewStore *dv = by_id ? ctx->direct_view : nullptr;              // (mi355_sw_batch_run_view: finished alignments go straight into the view)
  // one alignment: the launch's (score, cell, walk) -> handled / loc / tout of query `id`
  auto take = [&](int id, float best, int64_t ix, int64_t iy, size_t len, uint32_t pos, int64_t status, int64_t off) -> bool {
    const bool hit = best > 0;
    if (best < 0) { beyond.fetch_add(1, std::memory_order_relaxed); return false; }      // the float16 pass left it undecided (handled stays 0)
    if (want_trace && hit && status != 0) {
      // a walk that left its window: the problem goes to the host-driven path (handled stays 0); on whole problems: a bug
      if (!windows || status != 1) bad.store(true, std::memory_order_relaxed);
      else left.fetch_add(1, std::memory_order_relaxed);
      return false;
    }
    handled[id] = 1;
    Located &L = loc[id];
    L.score = hit ? best : 0;
    L.ix = ix; L.iy = iy;
    if (!want_trace || !hit) return true;
    TraceOut &t = tout[id];
    t.len = len;
    t.cx = cons_base + off;
    t.cy = t.cx + t.len;
    t.pos = pos;
    return true;
  };
  if (by_id && dv) {
    // Straight into the caller's view (as align_range_view makes its entries from loc / tout).  The loop is bound by the host's
    // memory traffic at half a million alignments: the seven arrays are written once, in id order, so they leave as STREAMING
    // stores (no read-for-ownership of lines that are overwritten whole) — one array at a time out of a small tile, because
    // seven interleaved streams per thread overflow the core's write-combining buffers (measured: 0.3 ms, but 3.5 ms in one
    // call of three, against 0.95 ms with ordinary stores).  Entries of alignments that are NOT finished here (undecided by
    // the float16 pass, a walk that left its window, an empty query) are written as zeros and made by align_range_view.
    parallel_for(nq, [&](size_t k0, size_t k1) {
      constexpr size_t T = 256;
      float t_score[T]; int64_t t_ex[T], t_ey[T]; uint32_t t_pos[T], t_len[T]; const char *t_cx[T], *t_cy[T];
      size_t mine = 0;
      for (size_t id0 = k0; id0 < k1; id0 += T) {
        const size_t cnt = std::min(T, k1 - id0);
        for (size_t j = 0; j < cnt; ++j) {
          const size_t id = id0 + j;
          float best = 0; int64_t ix = 0, iy = 0, status = 0, off = 0; size_t len = 0; uint32_t pos = 0;
          bool ok = q.len[id] != 0;
          if (ok) {
            if (!want_trace) { const BatchRecScore &r = h_rec_s[id]; best = r.score; ix = r.ix; iy = r.iy; }
            else { const BatchRec &r = h_rec[id]; best = r.score; ix = r.ix; iy = r.iy; len = r.len; pos = r.pos; status = r.status; off = r.off; }
            const bool hit = best > 0;
            if (best < 0) { beyond.fetch_add(1, std::memory_order_relaxed); ok = false; }
            else if (want_trace && hit && status != 0) {
              if (!windows || status != 1) bad.store(true, std::memory_order_relaxed);
              else left.fetch_add(1, std::memory_order_relaxed);
              ok = false;
            }
          }
          const bool hit = ok && best > 0, tr = hit && want_trace;
          const size_t l = tr ? len : 0;
          if (ok) { handled[id] = 2; ++mine; }
          t_score[j] = hit ? best : 0.0f; t_ex[j] = hit ? ix : 0; t_ey[j] = hit ? iy : 0;
          t_pos[j] = tr ? pos : 0; t_len[j] = (uint32_t)l;
          t_cx[j] = l ? cons_base + off : nullptr; t_cy[j] = l ? cons_base + off + l : nullptr;
        }
        for (size_t j = 0; j < cnt; ++j) __builtin_nontemporal_store(t_score[j], &dv->score[id0 + j]);
        for (size_t j = 0; j < cnt; ++j) __builtin_nontemporal_store(t_ex[j], &dv->end_x[id0 + j]);
        for (size_t j = 0; j < cnt; ++j) __builtin_nontemporal_store(t_ey[j], &dv->end_y[id0 + j]);
        for (size_t j = 0; j < cnt; ++j) __builtin_nontemporal_store(t_pos[j], &dv->pos[id0 + j]);
        for (size_t j = 0; j < cnt; ++j) __builtin_nontemporal_store(t_len[j], &dv->cons_len[id0 + j]);
        for (size_t j = 0; j < cnt; ++j) __builtin_nontemporal_store(t_cx[j], &dv->cx[id0 + j]);
        for (size_t j = 0; j < cnt; ++j) __builtin_nontemporal_store(t_cy[j], &dv->cy[id0 + j]);
      }
      __builtin_ia32_sfence();                                     // (the streaming stores of this part, before it reports done)
      done.fetch_add(mine, std::memory_order_relaxed);
    });
  } else if (by_id) {
    parallel_for(nq, [&](size_t k0, size_t k1) {
      size_t mine = 0;
      for (size_t id = k0; id < k1; ++id) {
        if (q.len[id] == 0) continue;                              // (an empty query: not in the launch, its defaults stand)
        if (!want_trace) {
          const BatchRecScore &r = h_rec_s[id];
          mine += take((int)id, r.score, r.ix, r.iy, 0, 0, 0, 0) ? 1 : 0;
          continue;
        }
        const BatchRec &r = h_rec[id];
        mine += take((int)id, r.score, r.ix, r.iy, (size_t)r.len, r.pos, r.status, r.off) ? 1 : 0;
      }
      done.fetch_add(mine, std::memory_order_relaxed);
    });
  } else {
    parallel_for(n, [&](size_t k0, size_t k1) {
      size_t mine = 0;
      for (size_t k = k0; k < k1; ++k) {
        const int id = q.order[batch_sorted_pos((int)first, (int)n, (int)k)];
        const bool tr = want_trace && h_best[k] > 0;
        mine += take(id, h_best[k], h_cell[2 * k], h_cell[2 * k + 1], tr ? (size_t)h_wout[3 * k] : 0, tr ? (uint32_t)h_wout[3 * k + 1] : 0,
                     tr ? h_wout[3 * k + 2] : 0, tr ? h_offs[k] : 0) ? 1 : 0;
      }
      done.fetch_add(mine, std::memory_order_relaxed);
    });
  }
  if (dv) ctx->devlist_direct += done.load();
  if (bad.load()) return fail(ctx, MI355_SW_ENODEV, "internal: a walk over a whole-problem window failed");
  ctx->devlist_done += done.load();
  ctx->left_window += left.load();
  ctx->beyond_f16 += beyond.load();
  return 0;
}

}  // namespace
